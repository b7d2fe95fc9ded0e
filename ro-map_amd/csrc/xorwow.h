// xorwow.h -- the reference's sample stream for the "same inputs" mode (mon_config::rng_flags bits 0-1; default off: the counter RNG of device_common.h).
//
// NeRF_Model draws its three per-iteration arrays -- SampleXY[2R], RandColors[3R], RandDt[S R] -- with curandGenerateUniform from ONE host generator of the
// default kind (XORWOW) and the default seed (CORE/src/nerf_model.cu:1432,1434,1468; no seed call anywhere), and the render jitter from a NEW generator per
// Render
// (:1725-1728,1781).  For a comparison with the CUDA build on identical inputs the same numbers are produced here:
//   * Marsaglia's xorwow: x[0..4] + Weyl counter d;  t = x0 ^ (x0 >> 2); x0..x3 = x1..x4; x4 = (x4 ^ (x4 << 4)) ^ (t ^ (t << 1)); d += 362437; out = x4 + d;
// * host-API ordering (cuRAND documentation, CURAND_ORDERING_PSEUDO_DEFAULT): value n of a generate call = position (n mod LANES) 2^67 + floor(n / LANES) of
// the sequence, LANES = 4096; the lanes keep their states between calls. Lane states are computed on the host (2^67 jump = the xorshift part's 160 x 160 GF(2)
//     transition matrix squared 67 times) and advanced on the device by k_xorwow_fill, one thread per lane, in the reference's order of calls;
// * flavours: the seed scramble and the integer -> (0, 1] map differ between the two libraries.  rocRAND's (rocrand_xorwow.h:107-118, rocrand_uniform.h:67) is
//     what the tests pin bit for bit (engine: tests/test_xorwow.py against the header; host generator: tests/test_xorwow_gpu.py against librocrand on the GPU);
//     cuRAND's (curand_kernel.h _curand_init_scratch / _curand_uniform as published; the library is not in this image) differs in four constants and the
// addend 2^-33 -- named assumptions CURAND-A1 / A2, CURAND-A3: the n of the ordering rule counts values since the generator's creation, across
// calls (librocrand's behaviour, pinned on the GPU; base.json's call sizes are multiples of 4096, for which every reading of the rule agrees).
#pragma once
#include <stdint.h>
#include <vector>
#include "device_common.h"

namespace mon {

struct XorwowState { uint32_t x[5], d; };
enum : int { kXorwowCurand = 0, kXorwowRocrand = 1 };

__host__ __device__ inline uint32_t xorwow_next(XorwowState& s) {
    const uint32_t t = s.x[0] ^ (s.x[0] >> 2);
    s.x[0] = s.x[1]; s.x[1] = s.x[2]; s.x[2] = s.x[3]; s.x[3] = s.x[4];
    s.x[4] = (s.x[4] ^ (s.x[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u; return s.d + s.x[4];
}
__host__ __device__ inline float xorwow_uniform(uint32_t v, int flavour) {
    return flavour == kXorwowRocrand ? 2.3283064e-10f + ((float)v * 2.3283064e-10f) : (float)v * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
inline void xorwow_seed(XorwowState& s, uint64_t seed, int flavour) {
    const bool r = flavour == kXorwowRocrand;
    const uint32_t s0 = (uint32_t)seed ^ (r ? 0x2c7f967fu : 0xaad26b49u), s1 = (uint32_t)(seed >> 32) ^ (r ? 0xa03697cbu : 0xf7dcefddu);
    const uint32_t t0 = (r ? 1228688033u : 1099087573u) * s0, t1 = (r ? 2073658381u : 2591861531u) * s1;
    s.x[0] = 123456789u + t0; s.x[1] = 362436069u ^ t0; s.x[2] = 521288629u + t1; s.x[3] = 88675123u ^ t1; s.x[4] = 5783321u + t0; s.d = 6615241u + t1 + t0;
}
// lane k of a host generator = the seed state 2^67 k steps on
void xorwow_lane_states(uint64_t seed, int flavour, uint32_t lanes, std::vector<XorwowState>& out);

// mon_config::rng_flags decoding (include/mon_core.h)
inline int rng_stream_mode(uint32_t flags) { return (int)(flags & 3u); }
inline bool rng_tcnn_init_order(uint32_t flags) { return ((flags >> 4) & 1u) != 0u; }
inline uint32_t rng_xorwow_lanes(uint32_t flags) { return ((flags >> 16) ? (flags >> 16) : 4u) * 1024u; }

}  // namespace mon
