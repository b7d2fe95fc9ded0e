// encode_device.h -- ray compaction + sample positions of a training batch, shared by k_sample_points (kernels_encode.hip) and the position blocks of
// k_optimizer (kernels_optim.hip), which prepare the NEXT iteration's positions while the optimizer streams the parameter state.
#pragma once
#include "device_common.h"
#include "model.h"

namespace mon {

struct PointsLds { unsigned long long words[256]; uint32_t prefix[257]; uint32_t wave_live[4], wave_base[4]; };

// the candidates' ballot words and their exclusive prefix -> LDS (whole block; contains a barrier); returns the number of valid candidates
__device__ __forceinline__ uint32_t points_prefix(PointsLds& l, const unsigned long long* __restrict__ mask, uint32_t nwords) {
    if (threadIdx.x < 64u) {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nwords; base += 64u) {
            const uint32_t w = base + threadIdx.x;
            const unsigned long long wd = (w < nwords) ? mask[w] : 0ull;
            uint32_t c = (uint32_t)__popcll(wd), inc = c;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, sh, 64); if ((int)threadIdx.x >= sh) inc += o; }
            if (w < nwords) { l.words[w] = wd; l.prefix[w] = carry + inc - c; }
            carry += (uint32_t)__shfl((int)inc, 63, 64);
        }
        if (threadIdx.x == 0u) l.prefix[nwords] = carry;
    }
    __syncthreads();
    return l.prefix[nwords];
}

// sample s = ray * 32 + n of iteration `iter`: training ray j is valid candidate number (j mod n_valid) in candidate order (fill_rollover_rays,
// nerf_model.cu:280-294); position = GenerateInputPoints (:553-566) + WarpPoint (:140-150) -- the arithmetic of ray_sample in k_fused_train, which
// recomputes t for the composite and stores the same x for the gradient scatter
// LIVE (occupancy-grid skipping, LiveArgs in model.h; the WHOLE 256-thread block must call, s = 256 consecutive samples of one sample partition): the
// sample's cell is looked up in the bit grid, the ray's 32 live bits go into word 11 of its record and the block's live samples are appended to their
// partition's index list (one returning atomic per block)
template <bool LIVE = false>
__device__ __forceinline__ void points_sample(PointsLds& l, const BatchPtrs& b, const ObjectConst& oc, uint32_t iter, uint32_t nvalid, uint32_t nwords,
        uint32_t s, float4_t* __restrict__ x_all, const LiveArgs& lv = LiveArgs{}) {
    const uint32_t ray = s >> 5, n = s & 31u;
    const uint32_t kth = ray % nvalid;
    uint32_t lo = 0, hi = nwords - 1u;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (l.prefix[mid] <= kth) lo = mid; else hi = mid - 1u; }
    unsigned long long wd = l.words[lo]; uint32_t kk = kth - l.prefix[lo], pos = 0;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t c = (uint32_t)__popcll(wd & ((1ull << sh) - 1ull)); if (kk >= c) { kk -= c; wd >>= sh;
            pos += (uint32_t)sh; } }
    const uint32_t cand = (lo << 6) + pos;
    const float t0 = b.cand_t0[cand], t1 = b.cand_t1[cand];
    const float dtr = (t1 - t0) / 32.0f;
    const float t = fmaf(dtr, (float)n + batch_rand(oc, kStreamDt, iter, ray * 32u + n), t0);
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float p = fmaf(t, b.cand_d[3u * cand + d], b.cand_o[3u * cand + d]);
        x[d] = (p - oc.aabb.mn[d]) / (oc.aabb.mx[d] - oc.aabb.mn[d]); }
    x_all[s] = float4_t{ x[0], x[1], x[2], t };
    uint32_t live_word = 0u;
    if constexpr (LIVE) {
        // (the cell arithmetic of k_fused_train's ray_sample)
        const uint32_t cx = (uint32_t)min(max((int)(x[0] * (float)kOccRes), 0), kOccRes - 1), cy = (uint32_t)min(max((int)(x[1] * (float)kOccRes), 0), kOccRes - 1),
                cz = (uint32_t)min(max((int)(x[2] * (float)kOccRes), 0), kOccRes - 1);
        const bool live = ((lv.occ_bits[((cz * kOccRes + cy) * kOccRes + cx) >> 5] >> (cx & 31u)) & 1u) != 0u;
        const unsigned long long bal = __ballot(live);
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        live_word = (uint32_t)(bal >> (lane & 32u));                                  // the 32 samples of this thread's ray (a wave holds two rays)
        if (lane == 0u) l.wave_live[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0u) {
            const uint32_t c0 = l.wave_live[0], c1 = l.wave_live[1], c2 = l.wave_live[2], c3 = l.wave_live[3], tot = c0 + c1 + c2 + c3;
            const uint32_t part = (s >> 8) % lv.n_parts;
            const uint32_t base = tot ? atomicAdd(lv.cnt + ((size_t)(iter & 1u) * kLiveMaxParts + part) * kLiveCntStride, tot) : 0u;
            l.wave_base[0] = part * lv.spw + base; l.wave_base[1] = part * lv.spw + base + c0; l.wave_base[2] = part * lv.spw + base + c0 + c1;
            l.wave_base[3] = part * lv.spw + base + c0 + c1 + c2;
        }
        __syncthreads();
        if (live) lv.idx[l.wave_base[wave] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = s;
        __syncthreads();                                                              // (the next trip of the caller's loop rewrites the LDS words)
    }
    if (n == 0u && b.ray_rec) {          // the ray's record for k_fused_train<PRE> (the fields its load_record collects from the candidate arrays)
        float* r = b.ray_rec + 12u * (size_t)ray;
        reinterpret_cast<float4_t*>(r)[0] = float4_t{ __builtin_bit_cast(float, b.cand_rgba[cand]), t0, t1, b.cand_d[3u * cand] };
        reinterpret_cast<float4_t*>(r)[1] = float4_t{ b.cand_d[3u * cand + 1u], b.cand_d[3u * cand + 2u], b.cand_o[3u * cand], b.cand_o[3u * cand + 1u] };
        reinterpret_cast<float4_t*>(r)[2] = float4_t{ b.cand_o[3u * cand + 2u], b.cand_depth[cand], __builtin_bit_cast(float, cand),
                __builtin_bit_cast(float, live_word) };
    }
}

}  // namespace mon
