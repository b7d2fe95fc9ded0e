// fused_device.h -- device-side building blocks shared by the fused kernels of backend 1: the MFMA mapping (FusedShape, weight-fragment image), the
// hash-grid gather / interpolation helpers of the gather chain, the MLP forward of one 32-sample tile, DPP wave scans, and the shape dispatch macro.
// Users: kernels_fused.hip (k_fused_train), kernels_render.hip (k_fused_render, occupancy grid), kernels_scatter.hip (k_grid_scatter: the wave scans).
//
//
// One kernel does what Step_No_Compacted (CORE/src/nerf_model.cu:1552-1607) spreads over
// GenerateInputPoints + tcnn forward (2 kernels) + VolumeRender + memset + VolumeRenderGradient +
// SumLoss + tcnn backward (fused MLP backward, split-k GEMMs, grid scatter):
//   sample points -> hash-grid encode -> MLP (MFMA) -> composite (wave scans) -> dL/dO ->
//   MLP backward (MFMA) -> dW (MFMA, accumulated in registers) -> grid scatter (packed-f16 atomics)
// Nothing between the ray record and the gradient tables touches HBM: E, h, dh, dE stay in
// registers / LDS (the reference spills 8+16+8+16 MB per step at base.json sizes).
//
// Mapping (wave64, v_mfma_f32_32x32x16_f16, "samples on N, weights on M"):
//   * one wavefront = one ray = 32 samples; lane l: sample n = l & 31, half h = l >> 5;
//   * the two half-waves split the hash levels: half h owns levels [h*LPH, h*LPH+LPH), LPH = ceil(L/2);
//     its encoded features ARE its MFMA B-operand K-slots (k = 8h + j), so the encode feeds the MLP
//     with no cross-lane movement;
//   * every layer is computed transposed, Out^T[units x samples] = W[units x K] * In^T[K x samples];
//     the C/D fragment (lane = sample, registers = units rho(h,r) = (r&3) + 8(r>>2) + 4h) is directly
//     the next layer's B fragment; weight matrices are pre-permuted into A fragments in LDS once per
//     workgroup so that K-slot order matches;
//   * W0^T's rows are permuted so dE lands in the half-wave that owns the level (grid backward reuses
//     the lane's own sample position);
//   * composite / loss gradient: lanes 0-31 are the ray's samples in order; transmittance is an
//     exclusive multiplicative wave scan, colour/depth suffix sums are additive scans;
//   * weight gradients need samples on K: activations are transposed through a per-wave LDS scratch
//     ([unit][sample] fp16) and accumulated in MFMA accumulators across the wave's rays, then reduced
//     across the workgroup in LDS and written as one fp32 partial per workgroup (summed by the optimizer).
#pragma once
#include <atomic>
#include <cstdlib>
#include <mutex>
#include "device_common.h"
#include "model.h"
#include "frag_layout.h"
#include "batch_device.h"
#include "grid_walk.h"

namespace mon {

// Runs `setup` once per device and call site, and returns only after it has run: function attributes (the dynamic LDS size) are per device, objects of several
// devices and several host threads per device launch from one process, and a launch must never precede its kernel's attribute call (a flag set BEFORE the
// attribute call let a second thread's first launch slip past it and fail with the large LDS size).
template <class F> static void once_per_device(std::atomic<uint64_t>& done, std::mutex& mu, F&& setup) {
    int dev = 0; (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    std::lock_guard<std::mutex> l(mu);
    if (done.load(std::memory_order_relaxed) & bit) return;
    setup();
    done.fetch_or(bit, std::memory_order_release);
}

void set_error(const char* fmt, ...);

// ------------------------------------------------------------------ shared pieces
__device__ __forceinline__ int rho(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// hidden unit carried by K-slot (k-step s, half h, element j) of a W-wide activation in C/D layout
__device__ __forceinline__ int unit_of_slot(int s, int h, int j) { return 32 * (s >> 1) + rho(h, 8 * (s & 1) + j); }

#ifndef MON_ENCODE_BATCH
#define MON_ENCODE_BATCH 4
#endif
#ifndef MON_V_SBATCH
#define MON_V_SBATCH 4          // samples per thread and software-pipeline round of k_grid_scatter
#endif
constexpr int kEncodeBatch = MON_ENCODE_BATCH;
#ifndef MON_V_STAGGER
// odd waves of every workgroup start 16 x 1024 cycles late (measured on the gather chain: 51.0 -> 47.7 us dense, 45.8 -> 45.0 us late with 12; on round 2's
// final kernels 12 / 14 / 16 / 18 / 20 units: 46.8 / 46.5 / 46.4 / 46.7 / 47.8 us dense, 43.5 / 43.6 / 43.2 / 43.3 / 44.9 late; modes 0, 1, 3 were slower)
#define MON_V_STAGGER 0x20010
#endif
constexpr uint32_t kDefaultStagger = MON_V_STAGGER;

template <int EPAD, int W, int NH> struct FusedShape {
    static constexpr int MB = W / 32;            // 32-row M blocks of a hidden layer
    static constexpr int KS0 = EPAD / 16;        // k-steps over the encoded input
    static constexpr int KSW = W / 16;           // k-steps over a hidden activation
    static constexpr int LLV = EPAD / 4;         // max local levels per half-wave (2 features each, EPAD/2 features per half)
    // A-fragment table (units of 512 halves = 64 lanes x 8)
    static constexpr int F_W0 = 0;                                   // [MB][KS0]
    static constexpr int F_W1 = F_W0 + MB * KS0;                     // [MB][KSW]      (NH == 2)
    static constexpr int F_WO = F_W1 + (NH == 2 ? MB * KSW : 0);     // [KSW]
    static constexpr int F_WOT = F_WO + KSW;                         // [MB]
    static constexpr int F_W1T = F_WOT + MB;                         // [MB][KSW]      (NH == 2)
    static constexpr int F_W0T = F_W1T + (NH == 2 ? MB * KSW : 0);   // [KSW]
    static constexpr int N_FRAGS = F_W0T + KSW;
    static constexpr int FRAG_BYTES = N_FRAGS * 1024;
    static constexpr int LT_BYTES = 512 + 4096;                      // LevelLds (113 words) + ray-compaction table (256 ballot words, 257 prefixes)
    // per-wave transpose scratch, fp16 [row][32 samples]
    static constexpr int SCR_E = 0;                                  // EPAD rows
    static constexpr int SCR_HA = SCR_E + EPAD * 32;                 // W rows: last hidden layer / its gradient
    static constexpr int SCR_HB = SCR_HA + W * 32;                   // W rows: first hidden layer (NH == 2)
    static constexpr int SCR_DO = SCR_HB + (NH == 2 ? W * 32 : 0);   // 4 rows
    static constexpr int SCR_HALVES = SCR_DO + 4 * 32;
    static constexpr int SCR_BYTES = SCR_HALVES * 2;
    static constexpr int N_MLP = W * EPAD + (NH - 1) * W * W + kOutPad * W;
    static constexpr int OFF_W1 = W * EPAD;
    static constexpr int OFF_WO = W * EPAD + (NH - 1) * W * W;
    static constexpr int WAVES = 4;
    // dW partial row in accumulator layout (frag_layout.h acc_param): dW0 tiles, dW1 tiles, the 4 real columns of dWout, then the loss partial
    static constexpr int ACC_W1 = MB * 1024;
    static constexpr int ACC_WO = ACC_W1 + (NH == 2 ? MB * MB * 1024 : 0);
    static constexpr int ACC_COLS = ACC_WO + MB * 128;
    static constexpr int RED_BYTES = (ACC_COLS + 64) * 4 * WAVES;   // one private fp32 copy per wave
    static constexpr int SMEM_BYTES = FRAG_BYTES + LT_BYTES + ((WAVES * SCR_BYTES > RED_BYTES) ? WAVES * SCR_BYTES : RED_BYTES);
};

struct FusedArgs {
    LevelFast lt; NetDims nd; ObjectConst oc; BatchPtrs b;
    const uint16_t* params;     // fp16 parameter vector (MLP matrices then grid)
    uint16_t* ggrid;            // fp16 grid gradient table
    float* partials;            // [gridDim.x][ACC_COLS + 64] fp32: dW partial sums in accumulator layout, column ACC_COLS = loss partial
    DevState* st;
    half2_t* de_soa;            // [L][B] dL/dE of the levels scattered through LDS (k_grid_scatter), or nullptr
    float* x_soa;               // [B] float4 {x, y, z, 0}: warped sample positions for k_grid_scatter
    uint32_t lds_level_mask;    // bit l set: level l goes through k_grid_scatter instead of global atomics
    const uint16_t* frag_image; // A fragments in LDS layout (k_build_frag_image), N_FRAGS x 512 halves
    // training: 1 = keep zero-gradient samples (option keep_zero_samples, the exactness test's A/B).  Render launcher: 1 = build the fragment image first
    uint32_t keep_zero;
    // per 4 grid entries (= one 8-parameter optimizer chunk): set to 1 next to every global atomic, or nullptr (see ParamPtrs::touched)
    uint8_t* touched_grid;
    // > 0: while big_levels_binned(st, big_switch) holds, EVERY level's dE rows are stored (kernels_bigscatter.hip bins the large levels)
    uint32_t big_switch;
    uint32_t n_bins;            // ray bins of the compacted gradient rows (scatter_bins(R), host-chosen)
    // bits 0-15: start delay of the second wave group in units of 1024 cycles, bits 16-17: how the groups are formed (see k_fused_train)
    uint32_t stagger;
    // occupancy-grid skipping (mon_config::occupancy_skip, default off): kOccRes^3 bits, 1 = the cell may hold density; nullptr = evaluate every sample
    const uint32_t* occ_bits;
    const half2_t* e_soa;       // PRE variant: [L][B] encoded features written by k_encode_tiles (kernels_encode.hip); the kernel then issues no gathers at all
};

// A fragments: the weight matrices pre-permuted to K-slot order (see the header).  They depend only on the weights,
// so they are built ONCE per step by k_build_frag_image into a global image that every workgroup of the fused
// kernels copies into LDS with 16-byte loads (building them per workgroup cost ~28 dependent 2-byte loads per thread).
template <int EPAD, int W, int NH>
__device__ __forceinline__ half_t frag_element(const half_t* __restrict__ w, int L, int idx) {
    const int p = frag_source(FragDims{ EPAD, W, NH, L }, idx);      // frag_layout.h: the one table both directions come from
    return p < 0 ? (half_t)0.f : w[p];
}

template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_build_frag_image(const uint16_t* __restrict__ params, int L, uint16_t* __restrict__ image,
        const DevState* __restrict__ st) {
    using S = FusedShape<EPAD, W, NH>;
    if (st && st->n_valid == 0u) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < S::N_FRAGS * 512) reinterpret_cast<half_t*>(image)[idx] = frag_element<EPAD, W, NH>(reinterpret_cast<const half_t*>(params), L, idx);
}

// First kernel of a fused-backend iteration: the candidate rays (GenerateRays) and the weight-fragment image are
// independent, so they share one launch (blocks [0, cand_blocks) generate candidates, the rest build fragments).
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_candidates_and_frags(BatchPtrs b, DatasetPtrs ds, ObjectConst oc, const DevState* __restrict__ st,
        uint32_t cand_blocks,
                                                              const uint16_t* __restrict__ params, int L, uint16_t* __restrict__ image) {
    using S = FusedShape<EPAD, W, NH>;
    if (blockIdx.x < cand_blocks) { gen_candidate(b, ds, oc, st->n_boxes, st->iter, blockIdx.x * blockDim.x + threadIdx.x); return; }
    const int idx = (blockIdx.x - cand_blocks) * blockDim.x + threadIdx.x;
    if (idx < S::N_FRAGS * 512) reinterpret_cast<half_t*>(image)[idx] = frag_element<EPAD, W, NH>(reinterpret_cast<const half_t*>(params), L, idx);
}

// Workgroup prologue: fragment image + level constants -> LDS.
template <int EPAD, int W, int NH>
__device__ __forceinline__ void build_fragments(half_t* frags, LevelLds* llt, const FusedArgs& a, bool backward) {
    using S = FusedShape<EPAD, W, NH>;
    for (int i = threadIdx.x; i <= kMaxLevels; i += blockDim.x) {
        llt->offset[i] = a.lt.offset[i];
        if (i < kMaxLevels) { llt->scale[i] = a.lt.scale[i]; llt->size[i] = a.lt.size[i]; llt->my[i] = a.lt.my[i]; llt->mz[i] = a.lt.mz[i];
            llt->mask[i] = a.lt.mask[i]; llt->hashed[i] = a.lt.hashed[i]; }
    }
    const int total16 = (backward ? S::N_FRAGS : S::F_WOT) * 64;            // 16-byte pieces
    const uint4* src = reinterpret_cast<const uint4*>(a.frag_image); uint4* dst = reinterpret_cast<uint4*>(frags);
    for (int i = threadIdx.x; i < total16; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ half8_t lds_frag(const half_t* frags, int frag, int lane) {
    return *reinterpret_cast<const half8_t*>(frags + frag * 512 + lane * 8); }

// relu + round to fp16 of one 32x32 C/D fragment -> two B fragments (registers 0..7, 8..15)
__device__ __forceinline__ void relu_pack(const float16_t& acc, half8_t& lo, half8_t& hi) {
    // round first, clamp second -- rounding is monotone and 0 is exact, so h(max(a, 0)) == max(h(a), 0) -- as packed operations: 8 v_cvt_pk_f16_f32 +
    // 8 v_pk_max_f16 per fragment instead of 16 v_max_f32 + 8 conversions (a quarter of the inference kernel's VALU instructions were these)
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)acc[j]; hi[j] = (half_t)acc[8 + j]; }
    const half8_t zero = { (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f };
    lo = __builtin_elementwise_max(lo, zero); hi = __builtin_elementwise_max(hi, zero);
}
__device__ __forceinline__ void mask_pack(const float16_t& acc, const half8_t& flo, const half8_t& fhi, half8_t& lo, half8_t& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)(((float)flo[j] > 0.f) ? acc[j] : 0.f); hi[j] = (half_t)(((float)fhi[j] > 0.f) ? acc[8 + j] : 0.f); }
}
// store one packed C/D fragment pair transposed into the scratch: scr[unit][sample]
__device__ __forceinline__ void scratch_store_units(half_t* scr, int mb, int n, int h, const half8_t& lo, const half8_t& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { scr[(32 * mb + rho(h, j)) * 32 + n] = lo[j]; scr[(32 * mb + rho(h, 8 + j)) * 32 + n] = hi[j]; }
}

// Phase timing (tools/fused_timing.py builds a -DMON_FUSED_TIMING variant of the library): per-wave cycle totals per phase,
// every stamp drains the memory counters first so a phase owns the latency it waits for.  Compiles to nothing otherwise.
struct TimingCtx { float acc[16]; long long last; };
__device__ __forceinline__ void tstamp(TimingCtx* tc, int k) {
#ifdef MON_FUSED_TIMING
    if (tc) { __builtin_amdgcn_s_waitcnt(0); const long long t = clock64(); tc->acc[k] += (float)(t - tc->last); tc->last = t; }
#else
    (void)tc; (void)k;
#endif
}

// Forward pass of one 32-sample tile.  Leaves: ef (local encoded features), hp* (hidden activations as
// packed B fragments), out4 (raw network outputs of sample n, valid in half-wave 0).
template <int EPAD, int W, int NH>
struct TileState {
    using S = FusedShape<EPAD, W, NH>;
    half_t ef[EPAD / 2];
    half8_t h0[S::MB][2];
    half8_t h1[NH == 2 ? S::MB : 1][2];
    float out4[4];
};

// Per-level constants of the encode, one level per LANE: lane h * 32 + il holds level h * LPH + il, the level half-wave h owns in level pair il (a pair past
// the last level holds a 1-entry dummy of level 0).  The gather code fetches them with v_readlane at compile-time lane numbers: no scalar loads (and no lgkmcnt
// waits) inside the ray loop, and none of the 7 x 16 constants pinned in SGPRs (the kernel runs at the SGPR limit; as kernel arguments they were re-loaded from
// the argument segment for every level of every ray).
struct LevelRegs { float scale; uint32_t size, my, mz, mask, off4, hashed; };
// the same registers filled from the kernel ARGUMENTS (scalar loads + one select per field and level): nothing to wait for but the argument segment, no LDS
// copy, no barrier
__device__ __forceinline__ LevelRegs load_level_regs_uniform(const LevelFast& klt, int L, int lane) {
    const int LPH = (L + 1) >> 1;
    // the dummy level: always entry 0
    LevelRegs r; r.scale = klt.scale[0]; r.size = 1u; r.my = klt.my[0]; r.mz = klt.mz[0]; r.mask = 0u; r.off4 = 0u; r.hashed = 1u;
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) {
        const bool here = l < L && lane == ((l < LPH) ? l : 32 + l - LPH);
        r.scale = here ? klt.scale[l] : r.scale; r.size = here ? klt.size[l] : r.size; r.my = here ? klt.my[l] : r.my; r.mz = here ? klt.mz[l] : r.mz;
        r.mask = here ? klt.mask[l] : r.mask; r.off4 = here ? klt.offset[l] * 4u : r.off4; r.hashed = here ? klt.hashed[l] : r.hashed;
    }
    return r;
}
__device__ __forceinline__ uint32_t lane_u(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float lane_f(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }

// ---- hash-grid encode (tcnn kernel_grid; fp32 fmaf chain over the 8 corners, one rounding).  Half-wave h OWNS levels
//      h*LPH + il (their features are its K slots), but the GATHERS are issued level by level with all 64 lanes on one
//      level: lane (n, c) fetches the four (y, z) corners with x-corner c of sample n.  Measured on MI355X
//      (tools/run_gatherbench.py): a divergent gather costs ~2.4 clk per distinct 64-byte line per instruction and nothing
//      more for further lanes in the same line -- and corners x, x+1 share a line 15 times out of 16, on hashed levels too
//      (x ^ h keeps the upper bits).  So pairing them in one instruction halves the lines per level; a
//      v_permlane32_swap per value then hands each half the 8 corners of the level it owns, and the interpolation runs
//      the same chain in the same order as before (bit-identical results).
// All control flow around the loads is compile-time (pairs past the last level gather the dummy level: one line per instruction), so the compiler's vmcnt
// bookkeeping stays exact: a pair's interpolation waits for ITS eight loads only, and the next pair's loads are issued into the registers it frees
// (runtime guards around the gather groups made every first use wait for the whole batch).
template <int EPAD, int W, int NH> struct GatherWindow {
    static constexpr int LLV = FusedShape<EPAD, W, NH>::LLV;
    static constexpr int EB = (LLV < kEncodeBatch) ? LLV : kEncodeBatch;              // level pairs in flight
    // pair il lives in slot il % EB: lanes (n, c) hold x-corner c of the four (y, z) corners, ra = level il, rb = level LPH + il
    uint32_t ra[EB][4], rb[EB][4];
};

// the four gathers of one level (`slot` = the lane of `lr` that holds it: a compile-time number); `live` = false: this lane's sample sits in a cell the
// occupancy grid marks empty -- its gathers are not issued (an exec-masked load costs no L2 request; r[] was zeroed by the caller)
template <bool MASKED>
__device__ __forceinline__ void gather_level(uint32_t (&r)[4], const LevelRegs& lr, int slot, const __amdgpu_buffer_rsrc_t rsrc, const float x[3], int h,
        bool live) {
    const float scale = lane_f(lr.scale, slot);
    const uint32_t size = lane_u(lr.size, slot), my = lane_u(lr.my, slot), mz = lane_u(lr.mz, slot), mask = lane_u(lr.mask, slot), off4 = lane_u(lr.off4, slot);
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pg[d] = (uint32_t)(int32_t)floorf(fmaf(scale, x[d], 0.5f));
    const uint32_t ax = pg[0] + (uint32_t)h, y0 = pg[1] * my, z0 = pg[2] * mz;
    const uint32_t ay[2] = { y0, y0 + my }, az[2] = { z0, z0 + mz };
    if (MASKED && !live) return;
    // the hashed / dense choice is a scalar branch around index arithmetic ONLY: the four loads sit after the join (a load inside either arm made the
    // compiler drain vmcnt at the top of the other one -- every dense level waited for all gathers in flight)
    uint32_t idx[4];
    if (lane_u(lr.hashed, slot) != 0u) {                                            // hashed levels hold 2^T entries: the mask IS the modulo
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = (ax ^ ay[j & 1] ^ az[j >> 1]) & mask;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t i = (ax + ay[j & 1] + az[j >> 1]) & mask;
            // dense sizes are not powers of two: index < 2 * size, so % size is one subtract
            i -= (i >= size) ? size : 0u;
            // memory safety for positions far outside [0,1]^3 (never produced by the sampler)
            idx[j] = min(i, size - 1u);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (idx[j] << 2) + off4, 0, 0);
}
template <int EPAD, int W, int NH, bool MASKED>
__device__ __forceinline__ void encode_issue(GatherWindow<EPAD, W, NH>& g, int il, const LevelRegs& lr, const __amdgpu_buffer_rsrc_t rsrc, const float x[3],
        int h, bool live) {
    constexpr int EB = GatherWindow<EPAD, W, NH>::EB;
#pragma unroll
    for (int j = 0; j < 4; ++j) { g.ra[il % EB][j] = 0u; g.rb[il % EB][j] = 0u; }      // (dead unless MASKED)
    gather_level<MASKED>(g.ra[il % EB], lr, il, rsrc, x, h, live); gather_level<MASKED>(g.rb[il % EB], lr, 32 + il, rsrc, x, h, live);
}
// interpolation of level pair il (its eight loads must have been issued); returns the two features of the level this half-wave owns
template <int EPAD, int W, int NH>
__device__ __forceinline__ void encode_swap(const GatherWindow<EPAD, W, NH>& g, int il, uint32_t (&c0)[4], uint32_t (&c1)[4]) {
    constexpr int EB = GatherWindow<EPAD, W, NH>::EB;
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 4; ++j) { const u2v sw = __builtin_amdgcn_permlane32_swap(g.ra[il % EB][j], g.rb[il % EB][j], false, false); c0[j] = sw.x;
        c1[j] = sw.y; }
}
template <int EPAD, int W, int NH>
__device__ __forceinline__ void encode_interp(TileState<EPAD, W, NH>& ts, int il, const uint32_t (&c0)[4], const uint32_t (&c1)[4], const LevelRegs& lr,
        const float x[3], int h, int L) {
    const int LPH = (L + 1) >> 1;
    const float scale = h ? lane_f(lr.scale, 32 + il) : lane_f(lr.scale, il);
    float pos[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float q = fmaf(scale, x[d], 0.5f); pos[d] = q - floorf(q); }
    const float wx[2] = { 1.f - pos[0], pos[0] }, wy[2] = { 1.f - pos[1], pos[1] }, wz[2] = { 1.f - pos[2], pos[2] };
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const half2_t v = __builtin_bit_cast(half2_t, (k & 1) ? c1[k >> 1] : c0[k >> 1]);
        const float wgt = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
        a0 = fmaf(wgt, (float)v.x, a0); a1 = fmaf(wgt, (float)v.y, a1);
    }
    const bool real = il < LPH && h * LPH + il < L;                                 // (a select, not a branch)
    ts.ef[2 * il] = real ? (half_t)a0 : (half_t)0.f; ts.ef[2 * il + 1] = real ? (half_t)a1 : (half_t)0.f;
}
// the rest of a ray's encode once its first EB level pairs are in flight: a rolling window, pair il + EB is requested into the registers pair il frees
template <int EPAD, int W, int NH, bool MASKED>
__device__ __forceinline__ void encode_finish(TileState<EPAD, W, NH>& ts, GatherWindow<EPAD, W, NH>& g, const LevelRegs& lr, const __amdgpu_buffer_rsrc_t rsrc,
        const float x[3], int lane, int L, bool live) {
    using S = FusedShape<EPAD, W, NH>; constexpr int EB = GatherWindow<EPAD, W, NH>::EB; const int h = lane >> 5;
#pragma unroll
    for (int il = 0; il < S::LLV; ++il) {
        uint32_t c0[4], c1[4];
        encode_swap<EPAD, W, NH>(g, il, c0, c1);
        if (il + EB < S::LLV) encode_issue<EPAD, W, NH, MASKED>(g, il + EB, lr, rsrc, x, h, live);
        encode_interp<EPAD, W, NH>(ts, il, c0, c1, lr, x, h, L);
    }
}
template <int EPAD, int W, int NH, bool MASKED>
__device__ __forceinline__ void encode_begin(GatherWindow<EPAD, W, NH>& g, const LevelRegs& lr, const __amdgpu_buffer_rsrc_t rsrc, const float x[3], int lane,
        bool live) {
    constexpr int EB = GatherWindow<EPAD, W, NH>::EB;
#pragma unroll
    for (int il = 0; il < EB; ++il) encode_issue<EPAD, W, NH, MASKED>(g, il, lr, rsrc, x, lane >> 5, live);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const half2_t* table, uint32_t table_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<half2_t*>(table), 0, (int)table_bytes, 0x00020000); }

// MLP forward of one 32-sample tile from ts.ef: leaves the hidden activations as packed B fragments and out4 (raw network outputs of sample n, valid in
// half-wave 0)
template <int EPAD, int W, int NH>
__device__ __forceinline__ void mlp_forward(TileState<EPAD, W, NH>& ts, const half_t* frags, int lane) {
    using S = FusedShape<EPAD, W, NH>;
    // ---- layer 0
    float16_t acc[S::MB];
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) {
        acc[mb] = float16_t{ 0 };
#pragma unroll
        for (int s = 0; s < S::KS0; ++s) {
            half8_t bf;
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = ts.ef[8 * s + j];
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W0 + mb * S::KS0 + s, lane), bf, acc[mb], 0, 0, 0);
        }
        relu_pack(acc[mb], ts.h0[mb][0], ts.h0[mb][1]);
    }
    if constexpr (NH == 2) {
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb) {
            float16_t a1 = float16_t{ 0 };
#pragma unroll
            for (int s = 0; s < S::KSW; ++s) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W1 + mb * S::KSW + s, lane),
                    ts.h0[s >> 1][s & 1], a1, 0, 0, 0);
            relu_pack(a1, ts.h1[mb][0], ts.h1[mb][1]);
        }
    }
    // ---- output layer (rows 0..3 real)
    float16_t ao = float16_t{ 0 };
#pragma unroll
    for (int s = 0; s < S::KSW; ++s) {
        half8_t bf;
        if constexpr (NH == 2) bf = ts.h1[s >> 1][s & 1]; else bf = ts.h0[s >> 1][s & 1];
        ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_WO + s, lane), bf, ao, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) ts.out4[c] = (float)(half_t)ao[c];        // network output is fp16 (tcnn network_precision_t)
}

// Forward pass of one 32-sample tile in one go (render, occupancy grid): encode + MLP
template <int EPAD, int W, int NH>
__device__ __forceinline__ void tile_forward(TileState<EPAD, W, NH>& ts, const half_t* frags, const LevelRegs& lr, const half2_t* __restrict__ table,
        uint32_t table_bytes, int L, const float x[3], int lane) {
    const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(table, table_bytes);
    GatherWindow<EPAD, W, NH> g;
    encode_begin<EPAD, W, NH, false>(g, lr, rsrc, x, lane, true);
    encode_finish<EPAD, W, NH, false>(ts, g, lr, rsrc, x, lane, L, true);
    mlp_forward<EPAD, W, NH>(ts, frags, lane);
}

// Cross-lane helpers on DPP (VALU data path, a few cycles each) instead of __shfl_* (ds_bpermute through the LDS crossbar,
// ~100 cycles of dependent latency per step; the composite is a chain of ~30 of them per ray).
// dpp_ctrl: row_shr:n = 0x110+n (shift within a 16-lane row), row_bcast:15 = 0x142 (lane 15 of a row to the next row),
// row_bcast:31 = 0x143, wave_shr:1 = 0x138.  Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false); }
// 32-lane inclusive scans (each half-wave scans independently: rows 0-1 and rows 2-3)
__device__ __forceinline__ float scan_mul32(float v) {
    v *= dpp_f<0x111, 0xF>(1.f, v); v *= dpp_f<0x112, 0xF>(1.f, v); v *= dpp_f<0x114, 0xF>(1.f, v); v *= dpp_f<0x118, 0xF>(1.f, v);
    v *= dpp_f<0x142, 0xA>(1.f, v);
    return v;
}
__device__ __forceinline__ float scan_add32(float v) {
    v += dpp_f<0x111, 0xF>(0.f, v); v += dpp_f<0x112, 0xF>(0.f, v); v += dpp_f<0x114, 0xF>(0.f, v); v += dpp_f<0x118, 0xF>(0.f, v);
    v += dpp_f<0x142, 0xA>(0.f, v);
    return v;
}
__device__ __forceinline__ uint32_t scan_add64_u32(uint32_t v) {                     // whole-wave inclusive scan
    v += dpp_u<0x111, 0xF>(0u, v); v += dpp_u<0x112, 0xF>(0u, v); v += dpp_u<0x114, 0xF>(0u, v); v += dpp_u<0x118, 0xF>(0u, v);
    v += dpp_u<0x142, 0xA>(0u, v); v += dpp_u<0x143, 0xC>(0u, v);
    return v;
}
// value of the previous lane (lane 0 keeps `fill`; callers overwrite lane 32 themselves where the halves are independent)
__device__ __forceinline__ float lane_prev(float v, float fill) { return dpp_f<0x138, 0xF>(fill, v); }
__device__ __forceinline__ float lane_bcast(float v, int src_lane_uniform) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane_uniform)); }

#define MON_FUSED_DISPATCH(FN, ...)                                                            \
    do {                                                                                       \
        const int key = nd.Epad * 1000 + nd.W * 10 + nd.NH;                                    \
        switch (key) {                                                                         \
            case 16 * 1000 + 32 * 10 + 1: FN<16, 32, 1>(__VA_ARGS__); break;                   \
            case 16 * 1000 + 32 * 10 + 2: FN<16, 32, 2>(__VA_ARGS__); break;                   \
            case 16 * 1000 + 64 * 10 + 1: FN<16, 64, 1>(__VA_ARGS__); break;                   \
            case 16 * 1000 + 64 * 10 + 2: FN<16, 64, 2>(__VA_ARGS__); break;                   \
            case 32 * 1000 + 32 * 10 + 1: FN<32, 32, 1>(__VA_ARGS__); break;                   \
            case 32 * 1000 + 32 * 10 + 2: FN<32, 32, 2>(__VA_ARGS__); break;                   \
            case 32 * 1000 + 64 * 10 + 1: FN<32, 64, 1>(__VA_ARGS__); break;                   \
            case 32 * 1000 + 64 * 10 + 2: FN<32, 64, 2>(__VA_ARGS__); break;                   \
            case 16 * 1000 + 128 * 10 + 1: FN<16, 128, 1>(__VA_ARGS__); break;                 \
            case 32 * 1000 + 128 * 10 + 1: FN<32, 128, 1>(__VA_ARGS__); break;                 \
            default: break;                                                                    \
        }                                                                                      \
    } while (0)


}  // namespace mon
