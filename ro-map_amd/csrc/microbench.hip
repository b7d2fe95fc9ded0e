// microbench.hip -- small device micro-benchmarks used to choose the gradient-scatter strategy
// (diagnostic entry point mon_microbench; not on the product path).
#include "device_common.h"
#include "model.h"

namespace mon {

void set_error(const char* fmt, ...);

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: packed-f16 atomic add, one shared table          mode 1: packed-f16 atomic add, one private table per XCD
// mode 2: fp32 atomic add, shared table                    mode 3: half2 gather (read only)
// mode 4: packed-f16 atomic add, shared table, 8 consecutive entries per lane (line-local bursts)
// pattern 0: uniform over n_entries; pattern 1: hash-grid like (16 equal-traffic levels: 4096, 32768, 14 x 65536 entries)
__global__ void __launch_bounds__(256) k_ub(int mode, int pattern, uint32_t n_entries, uint32_t ops_per_thread, uint32_t* __restrict__ table, float* __restrict__ sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    typedef __attribute__((address_space(1))) half2_t gh2;
    gh2* t16 = (gh2*)reinterpret_cast<half2_t*>(table);
    float acc = 0.f;
    for (uint32_t i = 0; i < ops_per_thread; ++i) {
        uint32_t r = mix32(gid * 0x9E3779B9u + i * 0x85EBCA6Bu + 12345u), idx;
        if (pattern == 0) idx = r % n_entries;
        else { const uint32_t lvl = i & 15u; const uint32_t size = lvl == 0 ? 4096u : (lvl == 1 ? 32768u : 65536u); const uint32_t off = lvl == 0 ? 0u : (lvl == 1 ? 4096u : 36864u + (lvl - 2u) * 65536u); idx = off + (r % size); }
        const half2_t v = { (half_t)1e-3f, (half_t)-1e-3f };
        if (mode == 0) __builtin_amdgcn_global_atomic_fadd_v2f16(t16 + idx, v);
        else if (mode == 1) __builtin_amdgcn_global_atomic_fadd_v2f16(t16 + (size_t)xcc * n_entries + idx, v);
        else if (mode == 2) atomicAdd(reinterpret_cast<float*>(table) + idx, 1e-3f);
        else if (mode == 3) { const half2_t g = reinterpret_cast<const half2_t*>(table)[idx]; acc += (float)g.x + (float)g.y; }
        else { const uint32_t b = (idx & ~7u) + ((idx + (i & 7u)) & 7u); __builtin_amdgcn_global_atomic_fadd_v2f16(t16 + b, v); }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int microbench(int device, int mode, int pattern, uint32_t n_entries, uint32_t n_ops, float* ms_out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || hipSetDevice(device) != hipSuccess) { set_error("microbench: no HIP device"); return MON_ERR_NO_DEVICE; }
    uint32_t* table = nullptr; float* sink = nullptr;
    const size_t bytes = (size_t)n_entries * 4 * 8;
    if (hipMalloc((void**)&table, bytes) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { set_error("microbench: hipMalloc failed"); return MON_ERR_HIP; }
    hipMemset(table, 0, bytes);
    const uint32_t ops_per_thread = 64, threads = n_ops / ops_per_thread, blocks = (threads + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_ub, dim3(blocks), dim3(256), 0, 0, mode, pattern, n_entries, ops_per_thread, table, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(table); hipFree(sink);
    *ms_out = best;
    return hipGetLastError() == hipSuccess ? MON_OK : MON_ERR_HIP;
}

}  // namespace mon
