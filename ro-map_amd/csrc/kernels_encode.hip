// kernels_encode.hip -- hash-grid encode of a whole training batch from LDS-resident level tiles (gfx950).
//
// Why: inside k_fused_train the forward gathers are 16.8 M 4-byte reads of an L2-resident table per base.json step, and the chip serves
// ~190-210 G distinct 64-byte lines/s on that path whatever the occupancy or lane arrangement: 39.8 us of the kernel's 44.6
// (profiles/r02_fused_floor.md).  Random 4-byte LDS reads run at ~3.5 lanes/clk/CU = ~1.9 T reads/s chip-wide
// (profiles/r02_gatherbench.md, mode 17) -- five times the L2 request rate.  So the encode is turned inside out, the way k_grid_scatter
// already turns the backward: a workgroup owns one LEVEL (its table slice copied into the CU's 160 KB LDS with coalesced 16-byte loads) and walks
// a sixteenth of the batch's samples; the encoded features go to HBM once (4 B per sample and level, 8 MB per step) and k_fused_train<PRE>
// reads them back coalesced instead of gathering.
//
//   k_sample_points : ray compaction (fill_rollover_rays, nerf_model.cu:280-294) + GenerateInputPoints (:536-566) -> warped positions x[B] (float4)
//   k_encode_tiles  : tcnn kernel_grid forward (call site nerf_model.cu:1557) per (level, sample partition) workgroup -> E[L][B] half2
//
// Numerics are those of encode_interp in kernels_fused.hip, bit for bit: the 8-corner fp32 fmaf chain in corner order k = x + 2y + 4z, one
// rounding to fp16 per feature.  A level of up to 40 960 entries sits in LDS whole; a larger one (up to 65 536 entries) is walked in TWO passes
// over its even and its odd entries: the two x-corners of a (y, z) pair always differ in the lowest index bit (kernels_fused.hip scatter_item),
// so each pass reads exactly one corner of each of the four pairs -- no range test, no divergence -- and the thread keeps the four values of
// the first pass in registers until the second one completes the chain in its original order.
#include "device_common.h"
#include "model.h"
#include "batch_device.h"
#include "encode_device.h"
#include "tile_device.h"
#include "xorwow.h"
#ifdef MON_OVERLAP_PROBE
#include <hip/hip_ext.h>
#endif

namespace mon {

constexpr uint32_t kEncLdsBytes = 163840;                 // the CU's whole LDS
static_assert(kEncWholeMax == kEncLdsBytes / 4u, "model.h: a whole-level tile is the CU's LDS");
constexpr uint32_t kEncParityMax = 65536u;                // entries of a level walked as two parity tiles (cached indices are 16 bits)
constexpr uint32_t kEncWgPerLevel = 16;                   // sample partitions per level and chunk
constexpr uint32_t kEncSpt = 8;                           // samples per thread
constexpr uint32_t kEncThreads = kTileThreads;

bool encode_tiles_supported(const LevelTable& lt, const NetDims& nd) {
    if (nd.L < 1 || nd.L > kMaxLevels) return false;
    for (int l = 0; l < nd.L; ++l) { const uint32_t size = lt.offset[l + 1] - lt.offset[l]; if (size > kEncParityMax || (size & 7u)) return false; }
    return (nd.n_mlp & 7u) == 0u;         // the table starts 16-byte aligned behind the MLP matrices
}

// ------------------------------------------------------------------ sample positions
// One thread per sample.  Training ray j is valid candidate number (j mod n_valid) in candidate order; every block finds its rays' candidates
// from the candidates' ballot words (<= 256 words, prefix in LDS).  The position arithmetic is ray_sample's of k_fused_train, which recomputes
// t (it needs the distances for the composite) and stores the same x for the gradient scatter.
__global__ void __launch_bounds__(256) k_sample_points(BatchPtrs b, ObjectConst oc, DevState* __restrict__ st, float4_t* __restrict__ x_all, LiveArgs live) {
    __shared__ PointsLds lds;
    const uint32_t R = oc.R, nwords = R >> 6, iter = st->iter;
    const uint32_t nvalid = points_prefix(lds, b.mask, nwords);
    if (blockIdx.x == 0u && threadIdx.x == 0u) st->n_valid_pre = nvalid;
    if (nvalid == 0u) return;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= R * 32u) return;                                            // (uniform per block: the batch is a multiple of 256 samples)
    if (live.occ_bits) points_sample<true>(lds, b, oc, iter, nvalid, nwords, s, x_all, live);
    else points_sample<false>(lds, b, oc, iter, nvalid, nwords, s, x_all);
}
// the stand-alone position pass counts into the live-sample counters of its iteration like k_optimizer's position blocks do, but nothing has cleared them
// for it (k_encode_tiles clears the set of the iteration AFTER its own): one tiny launch in front
__global__ void __launch_bounds__(64) k_live_reset(const DevState* __restrict__ st, uint32_t* __restrict__ cnt) {
    if (threadIdx.x < kLiveMaxParts) cnt[((size_t)(st->iter & 1u) * kLiveMaxParts + threadIdx.x) * kLiveCntStride] = 0u;
}

// ------------------------------------------------------------------ level-tile encode
struct EncodeArgs {
    LevelFast lt; int L; uint32_t n_mlp;
    const uint16_t* half_tiles;    // the fp16 grid in tile order (ParamPtrs::half_tiles)
    const float4_t* x_all;         // [B] warped positions
    half2_t* e_soa;                // [L][B] encoded features
    uint32_t B, spw;               // samples of the batch, samples per workgroup (<= kEncThreads * kEncSpt)
    const DevState* st;
    // GenerateRays (nerf_model.cu:369-446) of the NEXT iteration rides on the workgroups of level 0 (the coarsest level's tile is a few KB and its walk the
    // shortest of the grid): workgroup p of level 0 generates candidates [256 p, 256 p + 256) into the OTHER candidate set; k_optimizer's position blocks
    // then turn them into next iteration's positions (gen_next = 0: nothing to prepare)
    uint32_t gen_next; BatchPtrs b_next; DatasetPtrs ds;  ObjectConst oc;
    LiveArgs live;                 // occupancy-grid skipping: walk each partition's list of live samples instead of all of its samples (idx == nullptr: all)
};

// the chain of encode_interp: corners in order k = x + 2y + 4z, c0[j] / c1[j] = x-corner 0 / 1 of pair j, weight ((wx * wy) * wz); the two x-corners of a pair
// and the two features of a corner are worked on as pairs (v_pk_mul_f32 / v_pk_fma_f32: the same IEEE operations, two per instruction)
__device__ __forceinline__ half2_t enc_chain(const uint32_t (&c0)[4], const uint32_t (&c1)[4], const float (&pos)[3]) {
    const float2_t wx = { 1.f - pos[0], pos[0] };
    const float wy[2] = { 1.f - pos[1], pos[1] }, wz[2] = { 1.f - pos[2], pos[2] };
    const float2_t wxy[2] = { wx * wy[0], wx * wy[1] };
    float2_t a = { 0.f, 0.f };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2_t w = wxy[j & 1] * wz[j >> 1];
        const half2_t v0 = __builtin_bit_cast(half2_t, c0[j]), v1 = __builtin_bit_cast(half2_t, c1[j]);
        a = __builtin_elementwise_fma(float2_t{ w.x, w.x }, float2_t{ (float)v0.x, (float)v0.y }, a);
        a = __builtin_elementwise_fma(float2_t{ w.y, w.y }, float2_t{ (float)v1.x, (float)v1.y }, a);
    }
    return half2_t{ (half_t)a.x, (half_t)a.y };
}

// A thread's kEncSpt sample slots and their positions, requested together (one load per loop trip put a global round trip in front of every sample).
// LIVE (occupancy-grid skipping): the slots come from the partition's list of live samples -- entry threadIdx.x + k * kEncThreads of `count` -- and a wave whose
// entries of round k all lie beyond the list skips that round altogether (wave-uniform: the unrolled rounds keep their static register arrays).
template <bool LIVE>
__device__ __forceinline__ void load_positions(float4_t (&xs)[kEncSpt], uint32_t (&slot)[kEncSpt], const EncodeArgs& a, uint32_t s_base, uint32_t s_end,
        uint32_t count) {
    if constexpr (LIVE) {
#pragma unroll
        for (uint32_t k = 0; k < kEncSpt; ++k) { const uint32_t j = threadIdx.x + k * kEncThreads; slot[k] = (j < count) ? a.live.idx[s_base + j] : 0xffffffffu; }
#pragma unroll
        for (uint32_t k = 0; k < kEncSpt; ++k) xs[k] = a.x_all[slot[k] != 0xffffffffu ? slot[k] : s_base];
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kEncSpt; ++k) { const uint32_t s = s_base + threadIdx.x + k * kEncThreads; slot[k] = s < s_end ? s : 0xffffffffu;
            xs[k] = a.x_all[min(s, s_end - 1u)]; }
    }
}
// round k has work for this wave (uniform)
template <bool LIVE> __device__ __forceinline__ bool round_on(uint32_t k, uint32_t count) {
    return !LIVE || k * kEncThreads + (threadIdx.x & ~63u) < count; }

template <bool HASHED, bool POW2, bool LIVE>
__device__ __forceinline__ void encode_whole(const uint32_t* tile, const EncodeArgs& a, uint32_t s_base, uint32_t s_end, uint32_t count,
        half2_t* __restrict__ out, float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask) {
    if (s_base + threadIdx.x >= s_end && !LIVE) return;
    float4_t xs[kEncSpt]; uint32_t slot[kEncSpt]; load_positions<LIVE>(xs, slot, a, s_base, s_end, count);
#pragma unroll
    for (uint32_t k = 0; k < kEncSpt; ++k) {
        if (!round_on<LIVE>(k, count)) continue;
        uint32_t i0[4], i1[4], c0[4], c1[4]; float pos[3];
        enc_indices<HASHED, POW2>(xs[k], scale, size, my, mz, mask, i0, i1, pos);
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0[j] = tile[i0[j]]; c1[j] = tile[i1[j]]; }
        const half2_t e = enc_chain(c0, c1, pos);
        if (slot[k] != 0xffffffffu) out[slot[k]] = e;
    }
}

template <bool HASHED, bool POW2, bool LIVE>
// (src: the level in the tile image, evens then odds)
__device__ __forceinline__ void encode_parity(uint32_t* tile, const uint4* __restrict__ src, const EncodeArgs& a, uint32_t s_base, uint32_t s_end, uint32_t count,
                                              half2_t* __restrict__ out, float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask) {
    // pass 0: the even entries.  Per pair the even one of (i0, i1) is read now; the odd one is kept for pass 1 as 16 bits whose (always set) lowest bit is
    // replaced by "the odd one is x-corner 0"
    // (the position inside the cell is kept too: 24 registers against a second load + 9 instructions per sample)
    uint32_t veven[kEncSpt][4], cache[kEncSpt][2]; float pos[kEncSpt][3];
    const bool walk = LIVE ? (threadIdx.x & ~63u) < count : s_base + threadIdx.x < s_end;
    float4_t xs[kEncSpt]; uint32_t slot[kEncSpt];
    if (walk) load_positions<LIVE>(xs, slot, a, s_base, s_end, count);          // (the copy of the even half is under way: k_encode_tiles requested it at its entry)
    __builtin_amdgcn_s_waitcnt(0x0f70);                                   // vmcnt(0): the LDS writes of the copy are counted there
    __syncthreads();
    if (walk) {
#pragma unroll
        for (uint32_t k = 0; k < kEncSpt; ++k) {
            if (!round_on<LIVE>(k, count)) continue;
            uint32_t i0[4], i1[4], c[4];
            enc_indices<HASHED, POW2>(xs[k], scale, size, my, mz, mask, i0, i1, pos[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool odd0 = (i0[j] & 1u) != 0u;
                const uint32_t ev = odd0 ? i1[j] : i0[j], od = odd0 ? i0[j] : i1[j];
                veven[k][j] = tile[ev >> 1];
                c[j] = (od & 0xfffeu) | (odd0 ? 1u : 0u);
            }
            cache[k][0] = c[0] | (c[1] << 16); cache[k][1] = c[2] | (c[3] << 16);
        }
    }
    __syncthreads();
    tile_copy(tile, src + size / 8u, size / 8u);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (walk) {
#pragma unroll
        for (uint32_t k = 0; k < kEncSpt; ++k) {
            if (!round_on<LIVE>(k, count)) continue;
            uint32_t c0[4], c1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = (j & 1) ? (cache[k][j >> 1] >> 16) : (cache[k][j >> 1] & 0xffffu);
                const uint32_t vodd = tile[c >> 1];
                const bool odd0 = (c & 1u) != 0u;
                c0[j] = odd0 ? vodd : veven[k][j]; c1[j] = odd0 ? veven[k][j] : vodd;
            }
            const half2_t e = enc_chain(c0, c1, pos[k]);
            if (slot[k] != 0xffffffffu) out[slot[k]] = e;
        }
    }
}

__global__ void __launch_bounds__(kEncThreads) k_encode_tiles(EncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem);
    const uint32_t level = blockIdx.x / kEncWgPerLevel, part = blockIdx.x - level * kEncWgPerLevel;
    if (a.gen_next && level == 0u && blockIdx.y == 0u && threadIdx.x < 256u)
        for (uint32_t c0 = part * 256u; c0 < a.oc.R; c0 += kEncWgPerLevel * 256u) gen_candidate(a.b_next, a.ds, a.oc, a.st->n_boxes, a.st->iter + 1u,
                c0 + threadIdx.x);
    const uint32_t w = blockIdx.y * kEncWgPerLevel + part;             // sample partition of the batch
    const uint32_t s_base = w * a.spw, s_end = min(s_base + a.spw, a.B);
    if (s_base >= a.B) return;
    const uint32_t off = a.lt.offset[level], size = a.lt.size[level], my = a.lt.my[level], mz = a.lt.mz[level], mask = a.lt.mask[level];
    const bool hashed = a.lt.hashed[level] != 0u, pow2 = mask != 0xffffffffu;
    const float scale = a.lt.scale[level];
    const uint4* src = reinterpret_cast<const uint4*>(a.half_tiles + 2u * (size_t)off);
    half2_t* out = a.e_soa + (size_t)level * a.B;
    // the first tile is requested BEFORE the state is looked at (everything above comes from the argument segment): the round trip for n_valid_pre runs under
    // the copy
    tile_copy(tile, src, size <= kEncWholeMax ? size / 4u : size / 8u);
    // occupancy-grid skipping: this partition's live-sample count -- both parity sets requested with the state, the iteration's one picked afterwards
    const bool live = a.live.idx != nullptr;
    uint32_t cnt0 = 0u, cnt1 = 0u;
    if (live) { cnt0 = a.live.cnt[(size_t)w * kLiveCntStride]; cnt1 = a.live.cnt[((size_t)kLiveMaxParts + w) * kLiveCntStride]; }
    const uint32_t iter = a.st->iter;
    if (live && blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x < kLiveMaxParts)      // (the set the next iteration's position pass counts in: nobody reads it now)
        a.live.cnt[((size_t)((iter + 1u) & 1u) * kLiveMaxParts + threadIdx.x) * kLiveCntStride] = 0u;
    if (a.st->n_valid_pre == 0u) return;                               // batch skipped (the position pass wrote the count)
    const uint32_t count = min((iter & 1u) ? cnt1 : cnt0, s_end - s_base);
#define MON_ENC_CALL(FN, H, P2, ...) do { if (live) FN<H, P2, true>(__VA_ARGS__); else FN<H, P2, false>(__VA_ARGS__); } while (0)
    if (size <= kEncWholeMax) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (hashed) { if (pow2) MON_ENC_CALL(encode_whole, true, true, tile, a, s_base, s_end, count, out, scale, size, my, mz, mask);
            else MON_ENC_CALL(encode_whole, true, false, tile, a, s_base, s_end, count, out, scale, size, my, mz, mask); }
        else MON_ENC_CALL(encode_whole, false, false, tile, a, s_base, s_end, count, out, scale, size, my, mz, mask);
    } else {
        if (hashed) { if (pow2) MON_ENC_CALL(encode_parity, true, true, tile, src, a, s_base, s_end, count, out, scale, size, my, mz, mask);
            else MON_ENC_CALL(encode_parity, true, false, tile, src, a, s_base, s_end, count, out, scale, size, my, mz, mask); }
        else MON_ENC_CALL(encode_parity, false, false, tile, src, a, s_base, s_end, count, out, scale, size, my, mz, mask);
    }
#undef MON_ENC_CALL
}

// the tile image from the fp16 working copy (object creation, set_params, backend switch: whenever the weights changed outside k_optimizer, which keeps it
// current itself)
__global__ void __launch_bounds__(256) k_build_tiles_image(LevelFast lt, int L, const uint32_t* __restrict__ grid /* half2 per entry */,
                                                           uint32_t* __restrict__ image) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= lt.offset[L]) return;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; ++l) lvl += (l < L && e >= lt.offset[l]) ? 1 : 0;
    image[tile_slot(lt.offset[lvl], lt.size[lvl], e - lt.offset[lvl])] = grid[e];
}
void launch_build_tiles_image(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* params, uint16_t* half_tiles) {
    const uint32_t n = lf.offset[nd.L];
    hipLaunchKernelGGL(k_build_tiles_image, dim3((n + 255u) / 256u), dim3(256), 0, s, lf, nd.L, reinterpret_cast<const uint32_t*>(params + nd.n_mlp),
            reinterpret_cast<uint32_t*>(half_tiles));
}

// ------------------------------------------------------------------ XORWOW sample stream (xorwow.h; mon_config::rng_flags, default off)
// One thread per lane of the host generator: up to three generate calls in sequence (an iteration's SampleXY, RandColors, RandDt; or one Render's RandDt),
// value j of a call from lane j mod LANES; the lane's state goes back to memory for the next iteration's calls.
// (start: generator offset mod lanes before the first call)
__global__ void __launch_bounds__(256) k_xorwow_fill(XorwowState* __restrict__ states, uint32_t lanes, int flavour, uint32_t start,
                                                     float* __restrict__ out0, uint32_t n0, float* __restrict__ out1, uint32_t n1, float* __restrict__ out2,
                                                             uint32_t n2) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= lanes) return;
    XorwowState st = states[k];
    // value j of a call sits at offset (values generated before it) + j of the generator's output and comes from lane offset mod LANES
    uint32_t s = start;
    for (uint32_t j = (k + lanes - s) % lanes; j < n0; j += lanes) out0[j] = xorwow_uniform(xorwow_next(st), flavour);
    s = (uint32_t)(((uint64_t)s + n0) % lanes);
    for (uint32_t j = (k + lanes - s) % lanes; j < n1; j += lanes) out1[j] = xorwow_uniform(xorwow_next(st), flavour);
    s = (uint32_t)(((uint64_t)s + n1) % lanes);
    for (uint32_t j = (k + lanes - s) % lanes; j < n2; j += lanes) out2[j] = xorwow_uniform(xorwow_next(st), flavour);
    states[k] = st;
}
void launch_xorwow_fill(hipStream_t s, void* lane_states, uint32_t lanes, int flavour, uint32_t start, float* out0, uint32_t n0, float* out1, uint32_t n1,
        float* out2, uint32_t n2) {
    hipLaunchKernelGGL(k_xorwow_fill, dim3((lanes + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<XorwowState*>(lane_states), lanes, flavour, start, out0,
            n0, out1, n1, out2, n2);
}

uint32_t encode_tiles_spw(uint32_t B) {
    const uint32_t per_chunk = kEncWgPerLevel * kEncThreads * kEncSpt, chunks = (B + per_chunk - 1u) / per_chunk;
    return (B + kEncWgPerLevel * chunks - 1u) / (kEncWgPerLevel * chunks);
}
void encode_tiles_setup_device() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, kEncLdsBytes); }

void launch_sample_points(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st, float* x_all, const LiveArgs& live) {
    const uint32_t B = oc.R * 32u;
    if (live.occ_bits) hipLaunchKernelGGL(k_live_reset, dim3(1), dim3(64), 0, s, st, live.cnt);
    hipLaunchKernelGGL(k_sample_points, dim3((B + 255u) / 256u), dim3(256), 0, s, b, oc, st, reinterpret_cast<float4_t*>(x_all), live);
}

void launch_encode_tiles(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* half_tiles, const float* x_all, uint16_t* e_soa, uint32_t B,
        const DevState* st,
                         const BatchPtrs* b_next, const DatasetPtrs& ds, const ObjectConst& oc, uint32_t lds_bytes, const LiveArgs& live) {
    const uint32_t per_chunk = kEncWgPerLevel * kEncThreads * kEncSpt, chunks = (B + per_chunk - 1u) / per_chunk;
    const uint32_t spw = encode_tiles_spw(B);
    EncodeArgs a{ lf, nd.L, nd.n_mlp, half_tiles, reinterpret_cast<const float4_t*>(x_all), reinterpret_cast<half2_t*>(e_soa), B, spw, st,
            b_next ? 1u : 0u, b_next ? *b_next : BatchPtrs{}, ds, oc, live };
#ifdef MON_OVERLAP_PROBE
    // (lds_bytes bit 0 = launch without the AQL barrier bit, hipExtAnyOrderLaunch: ignored on gfx950, HISTORY 7.9)
    if (lds_bytes & 1u) { hipExtLaunchKernelGGL(k_encode_tiles, dim3((uint32_t)nd.L * kEncWgPerLevel, chunks), dim3(kEncThreads), (lds_bytes & ~1u) ? (lds_bytes & ~1u)
            : kEncLdsBytes, s, nullptr, nullptr, hipExtAnyOrderLaunch, a); return; }
#endif
    hipLaunchKernelGGL(k_encode_tiles, dim3((uint32_t)nd.L * kEncWgPerLevel, chunks), dim3(kEncThreads), lds_bytes ? lds_bytes : kEncLdsBytes, s, a);
}

}  // namespace mon
