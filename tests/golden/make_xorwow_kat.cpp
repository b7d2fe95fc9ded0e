// Generator of tests/golden/xorwow_rocrand.json: raw draws of rocRAND's XORWOW engine (rocrand_xorwow.h, host side of its __host__ __device__ functions)
// for a few seeds, subsequences and offsets -- the known answers tests/test_xorwow.py pins the oracle's own XORWOW (oracle/mon_oracle.c: seeding, next,
// 2^67 jump by matrix squaring) against.  rocRAND ships with the ROCm image; nothing of it is copied into the repository.
//   /opt/rocm/bin/hipcc -O1 -x hip --offload-arch=gfx950 tests/golden/make_xorwow_kat.cpp -o /tmp/make_xorwow_kat && /tmp/make_xorwow_kat > tests/golden/xorwow_rocrand.json
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_uniform.h>
#include <cstdio>
int main() {
    const unsigned long long seeds[] = { 0ull, 1ull, 0x123456789abcdefull };
    const unsigned long long subs[] = { 0ull, 1ull, 2ull, 7ull, 100ull, 4095ull, 16383ull };
    printf("{\"source\": \"rocrand_xorwow.h of ROCm 7.2 (host side), tests/golden/make_xorwow_kat.cpp\", \"cases\": [\n");
    bool first = true;
    for (unsigned long long seed : seeds) for (unsigned long long sub : subs) {
        rocrand_state_xorwow st; rocrand_init(seed, sub, 0ull, &st);
        printf("%s{\"seed\": %llu, \"subsequence\": %llu, \"draws\": [", first ? "" : ",\n", seed, sub); first = false;
        for (int i = 0; i < 8; ++i) printf("%s%u", i ? ", " : "", rocrand(&st));
        printf("], \"uniform\": [");
        rocrand_state_xorwow s2; rocrand_init(seed, sub, 0ull, &s2);
        for (int i = 0; i < 4; ++i) printf("%s%.9g", i ? ", " : "", (double)rocrand_uniform(&s2));
        printf("]}");
    }
    printf("\n]}\n");
    return 0;
}
