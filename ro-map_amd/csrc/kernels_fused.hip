// kernels_fused.hip -- fused forward + backward of one object NeRF for gfx950 (backend 1).
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
#define MON_V_STAGGER 0x20010      // odd waves of every workgroup start 16 x 1024 cycles late (measured: 51.0 -> 47.7 us dense, 45.8 -> 45.0 us late with 12; on the final kernels 12 / 14 / 16 / 18 / 20 units: 46.8 / 46.5 / 46.4 / 46.7 / 47.8 us dense, 43.5 / 43.6 / 43.2 / 43.3 / 44.9 late; modes 0, 1, 3 were slower)
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
    uint32_t ablate;            // timing experiments only (option fused_ablate): 2 no dW, 4 no dE/x stores, 8 no rays (prologue + epilogue only), 16 keep zero-gradient samples, 32 no dW reduction, 64 encode only
    uint8_t* touched_grid;      // per 4 grid entries (= one 8-parameter optimizer chunk): set to 1 next to every global atomic, or nullptr (see ParamPtrs::touched)
    uint32_t big_switch;        // > 0: while big_levels_binned(st, big_switch) holds, EVERY level's dE rows are stored (kernels_bigscatter.hip bins the large levels)
    uint32_t n_bins;            // ray bins of the compacted gradient rows (scatter_bins(R), host-chosen)
    uint32_t stagger;           // bits 0-15: start delay of the second wave group in units of 1024 cycles, bits 16-17: how the groups are formed (see k_fused_train)
    const uint32_t* occ_bits;   // occupancy-grid skipping (mon_config::occupancy_skip, default off): kOccRes^3 bits, 1 = the cell may hold density; nullptr = evaluate every sample
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
__global__ void __launch_bounds__(256) k_build_frag_image(const uint16_t* __restrict__ params, int L, uint16_t* __restrict__ image, const DevState* __restrict__ st) {
    using S = FusedShape<EPAD, W, NH>;
    if (st && st->n_valid == 0u) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < S::N_FRAGS * 512) reinterpret_cast<half_t*>(image)[idx] = frag_element<EPAD, W, NH>(reinterpret_cast<const half_t*>(params), L, idx);
}

// First kernel of a fused-backend iteration: the candidate rays (GenerateRays) and the weight-fragment image are
// independent, so they share one launch (blocks [0, cand_blocks) generate candidates, the rest build fragments).
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_candidates_and_frags(BatchPtrs b, DatasetPtrs ds, ObjectConst oc, const DevState* __restrict__ st, uint32_t cand_blocks,
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
        if (i < kMaxLevels) { llt->scale[i] = a.lt.scale[i]; llt->size[i] = a.lt.size[i]; llt->my[i] = a.lt.my[i]; llt->mz[i] = a.lt.mz[i]; llt->mask[i] = a.lt.mask[i]; llt->hashed[i] = a.lt.hashed[i]; }
    }
    const int total16 = (backward ? S::N_FRAGS : S::F_WOT) * 64;            // 16-byte pieces
    const uint4* src = reinterpret_cast<const uint4*>(a.frag_image); uint4* dst = reinterpret_cast<uint4*>(frags);
    for (int i = threadIdx.x; i < total16; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ half8_t lds_frag(const half_t* frags, int frag, int lane) { return *reinterpret_cast<const half8_t*>(frags + frag * 512 + lane * 8); }

// relu + round to fp16 of one 32x32 C/D fragment -> two B fragments (registers 0..7, 8..15)
__device__ __forceinline__ void relu_pack(const float16_t& acc, half8_t& lo, half8_t& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)fmaxf(acc[j], 0.f); hi[j] = (half_t)fmaxf(acc[8 + j], 0.f); }
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

// Per-level constants of the encode, one level per LANE: lane h * 32 + il holds level h * LPH + il, the level half-wave h owns in level pair il (a pair past the
// last level holds a 1-entry dummy of level 0).  The gather code fetches them with v_readlane at compile-time lane numbers: no scalar loads (and no lgkmcnt waits) inside
// the ray loop, and none of the 7 x 16 constants pinned in SGPRs (the kernel runs at the SGPR limit; as kernel arguments they were re-loaded from the argument segment
// for every level of every ray).
struct LevelRegs { float scale; uint32_t size, my, mz, mask, off4, hashed; };
// the same registers filled from the kernel ARGUMENTS (scalar loads + one select per field and level): nothing to wait for but the argument segment, no LDS copy, no barrier
__device__ __forceinline__ LevelRegs load_level_regs_uniform(const LevelFast& klt, int L, int lane) {
    const int LPH = (L + 1) >> 1;
    LevelRegs r; r.scale = klt.scale[0]; r.size = 1u; r.my = klt.my[0]; r.mz = klt.mz[0]; r.mask = 0u; r.off4 = 0u; r.hashed = 1u;      // the dummy level: always entry 0
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
    uint32_t ra[EB][4], rb[EB][4];                                                    // pair il lives in slot il % EB: lanes (n, c) hold x-corner c of the four (y, z) corners, ra = level il, rb = level LPH + il
};

// the four gathers of one level (`slot` = the lane of `lr` that holds it: a compile-time number); `live` = false: this lane's sample sits in a cell the
// occupancy grid marks empty -- its gathers are not issued (an exec-masked load costs no L2 request; r[] was zeroed by the caller)
template <bool MASKED>
__device__ __forceinline__ void gather_level(uint32_t (&r)[4], const LevelRegs& lr, int slot, const __amdgpu_buffer_rsrc_t rsrc, const float x[3], int h, bool live) {
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
            i -= (i >= size) ? size : 0u;                                           // dense sizes are not powers of two: index < 2 * size, so % size is one subtract
            idx[j] = min(i, size - 1u);                                             // memory safety for positions far outside [0,1]^3 (never produced by the sampler)
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (idx[j] << 2) + off4, 0, 0);
}
template <int EPAD, int W, int NH, bool MASKED>
__device__ __forceinline__ void encode_issue(GatherWindow<EPAD, W, NH>& g, int il, const LevelRegs& lr, const __amdgpu_buffer_rsrc_t rsrc, const float x[3], int h, bool live) {
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
    for (int j = 0; j < 4; ++j) { const u2v sw = __builtin_amdgcn_permlane32_swap(g.ra[il % EB][j], g.rb[il % EB][j], false, false); c0[j] = sw.x; c1[j] = sw.y; }
}
template <int EPAD, int W, int NH>
__device__ __forceinline__ void encode_interp(TileState<EPAD, W, NH>& ts, int il, const uint32_t (&c0)[4], const uint32_t (&c1)[4], const LevelRegs& lr, const float x[3], int h, int L) {
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
__device__ __forceinline__ void encode_finish(TileState<EPAD, W, NH>& ts, GatherWindow<EPAD, W, NH>& g, const LevelRegs& lr, const __amdgpu_buffer_rsrc_t rsrc, const float x[3], int lane, int L, bool live) {
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
__device__ __forceinline__ void encode_begin(GatherWindow<EPAD, W, NH>& g, const LevelRegs& lr, const __amdgpu_buffer_rsrc_t rsrc, const float x[3], int lane, bool live) {
    constexpr int EB = GatherWindow<EPAD, W, NH>::EB;
#pragma unroll
    for (int il = 0; il < EB; ++il) encode_issue<EPAD, W, NH, MASKED>(g, il, lr, rsrc, x, lane >> 5, live);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const half2_t* table, uint32_t table_bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<half2_t*>(table), 0, (int)table_bytes, 0x00020000); }

// MLP forward of one 32-sample tile from ts.ef: leaves the hidden activations as packed B fragments and out4 (raw network outputs of sample n, valid in half-wave 0)
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
            for (int s = 0; s < S::KSW; ++s) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W1 + mb * S::KSW + s, lane), ts.h0[s >> 1][s & 1], a1, 0, 0, 0);
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
__device__ __forceinline__ void tile_forward(TileState<EPAD, W, NH>& ts, const half_t* frags, const LevelRegs& lr, const half2_t* __restrict__ table, uint32_t table_bytes, int L, const float x[3], int lane) {
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
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t src) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false); }
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
__device__ __forceinline__ float lane_bcast(float v, int src_lane_uniform) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane_uniform)); }

// ------------------------------------------------------------------ fused training kernel
template <int EPAD, int W, int NH, bool DUMP, bool ATOMIC_LEVELS, bool OCC = false, bool PRE = false /* the encode was done by k_encode_tiles: features are loaded, not gathered */>
__global__ void __launch_bounds__(256, ((NH == 2 && W == 64) ? 1 : 2)) k_fused_train(FusedArgs a) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    LevelLds* llt = reinterpret_cast<LevelLds*>(smem + S::FRAG_BYTES);
    unsigned long long* cwords = reinterpret_cast<unsigned long long*>(smem + S::FRAG_BYTES + 512);      // [256]
    uint32_t* cprefix = reinterpret_cast<uint32_t*>(smem + S::FRAG_BYTES + 512 + 2048);                   // [257] exclusive prefix, [nwords] = total
    unsigned char* dyn = smem + S::FRAG_BYTES + S::LT_BYTES;
#ifdef MON_FUSED_TIMING
    TimingCtx tcx; for (float& v : tcx.acc) v = 0.f; tcx.last = clock64(); TimingCtx* tc = &tcx;
    tcx.acc[13] = (float)(uint32_t)(wall_clock64() & 0xffffffull); tcx.acc[15] = (float)(uint32_t)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // start time (100 MHz ticks, low 24 bits), HW_ID[15:0]
#else
    TimingCtx* tc = nullptr;
#endif
    // ---- prologue.  Everything a wave needs before its first gather is requested up front, and only the weight fragments (first read by the MLP) wait for
    //      the workgroup barrier: the iteration counter, the candidates' ballot words, the fragment image.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int L = a.nd.L, LPH = (L + 1) >> 1;
    const uint32_t R = (a.ablate & 8u) ? 0u : a.oc.R, iter = a.st->iter;
    // ---- ray compaction (fill_rollover_rays :280-294 without a kernel of its own): training ray j is valid candidate number (j mod n_valid) in candidate
    //      order.  Up to 4096 candidates (64 ballot words) every WAVE keeps the words and their exclusive prefix in registers, one word per lane, and finds a
    //      ray's candidate with a ballot and two v_readlanes; larger batches go through a table in LDS built by wave 0.
    const uint32_t nwords = a.oc.R >> 6;                                           // <= 256 (fused_supported)
    const bool small = nwords <= 64u;                                              // uniform
    unsigned long long my_word = 0ull;
    const bool have_rec = PRE && a.b.ray_rec != nullptr;                           // (uniform) the position pass left the compacted rays' records: no ballot words, no scan, no select
    if (!have_rec && small && (uint32_t)lane < nwords) my_word = a.b.mask[lane];
    const LevelRegs lregs = load_level_regs_uniform(a.lt, L, lane); const uint32_t table_bytes = a.lt.offset[L] * 4u;      // (from the argument segment: it ends up in the buffer descriptor, which must be scalar)
    build_fragments<EPAD, W, NH>(frags, llt, a, true);
    half_t* scr = reinterpret_cast<half_t*>(dyn + wave * S::SCR_BYTES);
    for (int i = 2 * a.nd.L * 32 + lane; i < EPAD * 32; i += 64) scr[S::SCR_E + i] = (half_t)0.f;   // pad feature rows stay zero; every other row is rewritten per ray before it is read
    uint32_t my_excl = 0u, nvalid = 0u;
    if (have_rec) nvalid = a.st->n_valid_pre;
    else if (small) { const uint32_t c = __popcll(my_word), inc = scan_add64_u32(c); my_excl = inc - c; nvalid = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63); }
    else if (wave == 0) {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nwords; base += 64) {
            const unsigned long long wd = (base + lane < nwords) ? a.b.mask[base + lane] : 0ull;
            const uint32_t c = __popcll(wd); const uint32_t inc = scan_add64_u32(c);
            if (base + lane < nwords) { cwords[base + lane] = wd; cprefix[base + lane] = carry + inc - c; }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) cprefix[nwords] = carry;
    }
    // candidate number `kth` -> candidate index (wave-uniform)
    const auto select = [&](uint32_t kth) -> uint32_t {
        unsigned long long wd; uint32_t kk, lo;
        if (small) {
            lo = (uint32_t)__popcll(__ballot(my_excl <= kth)) - 1u;                  // the prefix is non-decreasing (lanes past the last word hold the total)
            wd = ((unsigned long long)lane_u((uint32_t)(my_word >> 32), (int)lo) << 32) | lane_u((uint32_t)my_word, (int)lo); kk = kth - lane_u(my_excl, (int)lo);
        } else {
            lo = 0; uint32_t hi = nwords - 1u;
            while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (cprefix[mid] <= kth) lo = mid; else hi = mid - 1u; }
            wd = cwords[lo]; kk = kth - cprefix[lo];
        }
        uint32_t pos = 0;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t c = __popcll(wd & ((1ull << sh) - 1ull)); if (kk >= c) { kk -= c; wd >>= sh; pos += sh; } }
        return (lo << 6) + pos;
    };
    // the candidate's record, one field per lane (rgba, t0, t1, d[3], o[3], depth): ONE load, requested a whole ray ahead of its use
    const char* rec_base; uint32_t rec_mul = 1u;
    {   const void* fb = lane == 0 ? (const void*)a.b.cand_rgba : lane == 1 ? (const void*)a.b.cand_t0 : lane == 2 ? (const void*)a.b.cand_t1
                       : lane < 6 ? (const void*)(a.b.cand_d + (lane - 3)) : lane < 9 ? (const void*)(a.b.cand_o + (lane - 6)) : (const void*)a.b.cand_depth;
        rec_base = reinterpret_cast<const char*>(fb); if (lane >= 3 && lane < 9) rec_mul = 3u; }
    const auto load_record = [&](uint32_t cand) -> uint32_t { uint32_t v = 0u; if (lane < 10) v = *reinterpret_cast<const uint32_t*>(rec_base + 4u * (size_t)(cand * rec_mul)); return v; };
    // (PRE with records: lane l < 10 takes field l of ray `r`'s 12-float record -- one coalesced 40-byte load, requested a ray ahead like the candidate record)
    const auto load_ray_rec = [&](uint32_t r) -> uint32_t { uint32_t v = 0u; if (lane < 10) v = reinterpret_cast<const uint32_t*>(a.b.ray_rec)[12u * (size_t)r + (uint32_t)lane]; return v; };
    const uint32_t ray0 = blockIdx.x * S::WAVES + wave;
    uint32_t cand = 0u, rec = 0u;
    if (have_rec) { if (nvalid != 0u && ray0 < R) rec = load_ray_rec(ray0); }
    else if (small && nvalid != 0u && ray0 < R) { cand = select(ray0 % nvalid); rec = load_record(cand); }
    __syncthreads();
    if (!have_rec && !small) { nvalid = cprefix[nwords]; if (nvalid != 0u && ray0 < R) { cand = select(ray0 % nvalid); rec = load_record(cand); } }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->n_valid = nvalid; a.st->loss_sum = 0.f; }
    if (nvalid == 0u) return;                                                        // batch skipped (uniform over the grid)

    const uint32_t lds_level_mask = (ATOMIC_LEVELS && a.big_switch != 0u && big_levels_binned(a.st->n_scatter_last, a.big_switch)) ? 0xffffffffu : a.lds_level_mask;   // wave-uniform
    const half2_t* table = reinterpret_cast<const half2_t*>(a.params + a.nd.n_mlp);
    typedef __attribute__((address_space(1))) half2_t gh2;
    gh2* gtable = (gh2*)reinterpret_cast<half2_t*>(a.ggrid);
    const float ls = a.oc.loss_scale / (float)R;
    const uint32_t n_bins = a.n_bins;                                                // ray bins of the compacted gradient rows

    float16_t dW0[S::MB], dWo[S::MB], dW1[NH == 2 ? S::MB : 1][NH == 2 ? S::MB : 1];
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) { dW0[mb] = float16_t{ 0 }; dWo[mb] = float16_t{ 0 }; }
    if constexpr (NH == 2) {
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < S::MB; ++nb) dW1[mb][nb] = float16_t{ 0 };
    }
    float loss_acc = 0.f;
    // ---- phase stagger.  All waves of a CU share one texture-address path, and a ray's 64 gather instructions keep it busy for ~1.5 us; the waves start together and
    //      their phases have equal lengths, so they would ALL gather, then ALL run the MLP / composite / backward with the address path idle.  Half of the waves therefore
    //      start late by about one gather phase: from then on one group computes while the other gathers.
    if (a.stagger & 0xffffu) {
        const uint32_t mode = (a.stagger >> 16) & 3u;
        const uint32_t slot = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) ;      // HW_ID.wave_id: this wave's slot on its SIMD
        const bool late = mode == 0u ? (slot & 1u) != 0u : mode == 1u ? blockIdx.x >= (gridDim.x >> 1) : mode == 2u ? (wave & 1) != 0 : wave != 0;
        const uint32_t units = (a.stagger & 0xffffu) * (mode == 3u ? (uint32_t)wave : 1u);        // mode 3: graduated, wave w waits w units
        if (late) for (uint32_t i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    tstamp(tc, 0);

    // ---- ray loop.  The candidate record of ray k + 1 is requested while ray k is processed.  (Also tried: requesting the first EB level pairs of ray k + 1
    //      before ray k's MLP / composite / backward -- no gain, 48.4 vs 48.6 us without the stagger and 9 spilled registers: the gather phase is bound by the
    //      chip-wide L2 request rate, not by its start-up latency; an encode-only variant of this kernel, `fused_ablate` 96, takes 39.5 us.)
    struct RaySample { float t, x[3]; uint32_t rgba; float tdp, t0, t1; bool live, any; };
    const auto ray_sample = [&](uint32_t recv, uint32_t ray) {                           // sample n of the ray described by record `recv` (GenerateInputPoints, nerf_model.cu:553-566)
        RaySample q; q.rgba = lane_u(recv, 0); q.tdp = __builtin_bit_cast(float, lane_u(recv, 9));
        const float t0 = __builtin_bit_cast(float, lane_u(recv, 1)), t1 = __builtin_bit_cast(float, lane_u(recv, 2));
        const float rd[3] = { __builtin_bit_cast(float, lane_u(recv, 3)), __builtin_bit_cast(float, lane_u(recv, 4)), __builtin_bit_cast(float, lane_u(recv, 5)) };
        const float ro[3] = { __builtin_bit_cast(float, lane_u(recv, 6)), __builtin_bit_cast(float, lane_u(recv, 7)), __builtin_bit_cast(float, lane_u(recv, 8)) };
        const float dtr = (t1 - t0) / 32.0f; q.t0 = t0; q.t1 = t1;
        q.t = fmaf(dtr, (float)n + batch_rand(a.oc, kStreamDt, iter, ray * 32u + (uint32_t)n), t0);
#pragma unroll
        for (int d = 0; d < 3; ++d) { const float p = fmaf(q.t, rd[d], ro[d]); q.x[d] = (p - a.oc.aabb.mn[d]) / (a.oc.aabb.mx[d] - a.oc.aabb.mn[d]); }
        // occupancy-grid skipping (default off): a sample whose cell the grid marks empty is not evaluated -- no gathers, alpha = 0, no gradient
        q.live = true;
        if constexpr (OCC) {
            const uint32_t cx = (uint32_t)min(max((int)(q.x[0] * (float)kOccRes), 0), kOccRes - 1), cy = (uint32_t)min(max((int)(q.x[1] * (float)kOccRes), 0), kOccRes - 1), cz = (uint32_t)min(max((int)(q.x[2] * (float)kOccRes), 0), kOccRes - 1);
            q.live = ((a.occ_bits[((cz * kOccRes + cy) * kOccRes + cx) >> 5] >> (cx & 31u)) & 1u) != 0u;
        }
        q.any = !OCC || __ballot(q.live) != 0ull;                                         // false: the whole ray crosses empty cells only, nothing to evaluate
        return q;
    };
    const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(table, table_bytes);
    const uint32_t ray_stride = gridDim.x * S::WAVES;
    // PRE: lane (n, h) reads the features of the levels half-wave h owns, one dword (half2) per level at [level][ray * 32 + n] -- 128 contiguous bytes per
    // half-wave and level; a ray's eight loads are requested one ray ahead, like its candidate record
    uint32_t epre[S::LLV];
    const auto load_encoded = [&](uint32_t r) {
#pragma unroll
        for (int il = 0; il < S::LLV; ++il) {
            const int level = h * LPH + il; epre[il] = 0u;
            if (il < LPH && level < L) epre[il] = reinterpret_cast<const uint32_t*>(a.e_soa)[(size_t)level * (R * 32u) + r * 32u + (uint32_t)n];
        }
    };
    if constexpr (PRE) { if (ray0 < R) load_encoded(ray0); }
    for (uint32_t ray = ray0; ray < R; ray += ray_stride) {
        const RaySample cur = ray_sample(rec, ray); const uint32_t cand_this = cand;
        const uint32_t kth = ray % nvalid, rgba = cur.rgba, s_idx = ray * 32u + (uint32_t)n;
        const float t = cur.t, tdp = cur.tdp, t0 = cur.t0, t1 = cur.t1; const bool live = cur.live, is_obj = (rgba >> 24) != 0u;
        const float x[3] = { cur.x[0], cur.x[1], cur.x[2] };
        tstamp(tc, 1);
        TileState<EPAD, W, NH> ts;
        if constexpr (PRE) {
#pragma unroll
            for (int il = 0; il < S::LLV; ++il) { const half2_t v = __builtin_bit_cast(half2_t, epre[il]); ts.ef[2 * il] = v.x; ts.ef[2 * il + 1] = v.y; }
        } else if (cur.any) { GatherWindow<EPAD, W, NH> gw; encode_begin<EPAD, W, NH, OCC>(gw, lregs, rsrc, x, lane, live); encode_finish<EPAD, W, NH, OCC>(ts, gw, lregs, rsrc, x, lane, L, live); }
        tstamp(tc, 2);
        // the next ray's candidate record is requested HERE, behind this ray's last gather: vmcnt retires in order, so a load issued before the gathers would
        // have to land before the first level pair can be consumed; now its latency runs under the MLP, composite and backward pass
        if (ray + ray_stride < R) { if (have_rec) rec = load_ray_rec(ray + ray_stride); else { cand = select((ray + ray_stride) % nvalid); rec = load_record(cand); } if constexpr (PRE) load_encoded(ray + ray_stride); }
        if (a.ablate & 64u) { float sacc = 0.f; for (int i = 0; i < EPAD / 2; ++i) sacc += (float)ts.ef[i]; loss_acc += sacc; continue; }      // timing experiments: the encode alone
        if (!OCC || __ballot(live) != 0ull) mlp_forward<EPAD, W, NH>(ts, frags, lane);
        else {
#pragma unroll
            for (int i = 0; i < EPAD / 2; ++i) ts.ef[i] = (half_t)0.f;
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) { ts.h0[mb][0] = half8_t{}; ts.h0[mb][1] = half8_t{}; if constexpr (NH == 2) { ts.h1[mb][0] = half8_t{}; ts.h1[mb][1] = half8_t{}; } }
#pragma unroll
            for (int c = 0; c < 4; ++c) ts.out4[c] = 0.f;
        }
        const bool do_dw = (a.ablate & 2u) == 0u;

        tstamp(tc, 4);
        // ---- composite (VolumeRender :762-813) as wave scans over lanes 0..31
        const float v0 = ts.out4[0], v1 = ts.out4[1], v2 = ts.out4[2], v3 = ts.out4[3];
        const float c0 = logistic_f(v0), c1 = logistic_f(v1), c2 = logistic_f(v2), sigma = __expf(v3);
        float tprev = lane_prev(t, 0.f); if (n == 0) tprev = 0.f;                         // :770 last_distance = 0
        const float dt = t - tprev;
        const float alpha = (!OCC || live) ? 1.f - __expf(-sigma * dt) : 0.f, om = 1.f - alpha;
        const float tincl = scan_mul32(om);                                               // T after this sample
        float T = lane_prev(tincl, 1.f); if (n == 0) T = 1.f;                             // T before this sample
        const bool active = T >= kTransmittanceEps;                                        // :774 early-out (T is non-increasing)
        const int nact = __popc((uint32_t)__ballot(active));                               // lanes 0..31 = the ray's samples
        const float Tfin = lane_bcast(tincl, nact - 1);                                    // broadcast from half-wave 0 (sample 0 is always active: nact >= 1)
        const float wgt = active ? alpha * T : 0.f;
        const float p0 = scan_add32(wgt * c0), p1 = scan_add32(wgt * c1), p2 = scan_add32(wgt * c2), pd = scan_add32(wgt * t);
        const float bg0 = batch_rand(a.oc, kStreamColor, iter, 3u * kth), bg1 = batch_rand(a.oc, kStreamColor, iter, 3u * kth + 1u), bg2 = batch_rand(a.oc, kStreamColor, iter, 3u * kth + 2u);   // :760, :438-441
        const float rgb0 = lane_bcast(p0, 31) + Tfin * bg0, rgb1 = lane_bcast(p1, 31) + Tfin * bg1, rgb2 = lane_bcast(p2, 31) + Tfin * bg2;
        const float dep = lane_bcast(pd, 31), mask = 1.f - Tfin;
        // ---- loss + dL/dO (VolumeRenderGradient_No_Compacted :853-953)
        const float tg0 = is_obj ? (float)(rgba & 0xffu) / 255.0f : bg0, tg1 = is_obj ? (float)((rgba >> 8) & 0xffu) / 255.0f : bg1, tg2 = is_obj ? (float)((rgba >> 16) & 0xffu) / 255.0f : bg2;
        const float e0 = rgb0 - tg0, e1 = rgb1 - tg1, e2 = rgb2 - tg2;
        const float g0 = 2.f * e0, g1 = 2.f * e1, g2 = 2.f * e2;
        float dl_dd = 0.f; if (tdp > 0.f) dl_dd = 0.5f * ((dep - tdp >= 0.f) ? 1.f : -1.f);
        const float mean_loss = (e0 * e0 + e1 * e1 + e2 * e2) / 3.f;
        const float loss = is_obj ? mean_loss + dl_dd * (dep - tdp) + (1.f - mask) : mean_loss + mask;
        half8_t bdo;
#pragma unroll
        for (int j = 0; j < 8; ++j) bdo[j] = (half_t)0.f;
        if (active && h == 0 && (!OCC || live)) {
            const float Tn = tincl;                                                       // T after the update (:912)
            const float s0 = rgb0 - p0, s1 = rgb1 - p1, s2 = rgb2 - p2;                   // suffix :915
            bdo[0] = (half_t)(ls * ((wgt * g0) * (c0 * (1.f - c0))));
            bdo[1] = (half_t)(ls * ((wgt * g1) * (c1 * (1.f - c1))));
            bdo[2] = (half_t)(ls * ((wgt * g2) * (c2 * (1.f - c2))));
            const float dsig = __expf(clamp_f(v3, -15.f, 15.f));
            const float depth_sup = dl_dd * (Tn * t - (dep - pd));
            const float dmask = 1.f - mask;
            float dl;
            if (is_obj) {
                const float dlm = 0.5f * (mask >= 1.f ? 1.f : -1.f);
                const float dot = g0 * (Tn * c0 - s0) + g1 * (Tn * c1 - s1) + g2 * (Tn * c2 - s2);
                dl = dsig * dt * (dot + depth_sup + dlm * dmask);
            } else {
                const float dlm = 0.5f * (mask >= 0.f ? 1.f : -1.f);
                dl = dsig * dt * dlm * dmask + dsig * 0.01f;
            }
            bdo[3] = (half_t)(ls * dl);
        }
        // ---- which samples carry a gradient at all.  dL/dO is fp16 with a loss scale of 128/R: once empty space is learnt
        //      (sigma = exp(-15), alpha * T -> 0) it underflows to exact zeros, 94-98 % of the samples after ~200 steps
        //      (tools/zero_grad_fraction.py).  Zero rows contribute exact zeros to dW and dE, so a ray without any is done
        //      here, and only the non-zero samples are handed to k_grid_scatter, compacted into ray bins (ray mod n_bins, up to 128: one returning atomic per ray on 16 counters cost 7 us of contention).  Inside a
        //      bin the order is whatever the atomics give -- irrelevant, the scatter's integer accumulation is exact -- while
        //      bin membership, and with it every fp16-rounded partial table, is a fixed function of the ray: the result stays
        //      deterministic.  The slot reservation is issued now and consumed after the MFMAs.
        const uint2 bdo_bits = __builtin_bit_cast(uint2, half4_t{ bdo[0], bdo[1], bdo[2], bdo[3] });
        const uint32_t nz32 = (a.ablate & 16u) ? 0xffffffffu : (uint32_t)__ballot(h == 0 && ((bdo_bits.x | bdo_bits.y) & 0x7fff7fffu) != 0u);   // ablate 16: no skipping (A/B check)
        const uint32_t nz_cnt = __popc(nz32);
        uint32_t slot_base = 0u;
        if (nz_cnt != 0u && lane == 0 && lds_level_mask) slot_base = atomicAdd(&a.st->n_scatter[scatter_counter(iter, ray & (n_bins - 1u))], nz_cnt);
        if (lane == 0) {
            loss_acc += loss;
            a.b.rgb_ray[3 * ray] = rgb0; a.b.rgb_ray[3 * ray + 1] = rgb1; a.b.rgb_ray[3 * ray + 2] = rgb2;
            a.b.depth_ray[ray] = dep; a.b.mask_ray[ray] = mask; a.b.loss_ray[ray] = loss;
        }
        if (DUMP) {
            if (lane == 0) {
                for (int d = 0; d < 3; ++d) { a.b.ray_o[3 * ray + d] = a.b.cand_o[3 * cand_this + d]; a.b.ray_d[3 * ray + d] = a.b.cand_d[3 * cand_this + d]; }
                a.b.ray_t0[ray] = t0; a.b.ray_t1[ray] = t1; a.b.ray_dn[ray] = a.b.cand_dn[cand_this]; a.b.ray_flag[ray] = is_obj ? 1 : 0; a.b.target_depth[ray] = tdp;
                a.b.bgcol[3 * ray] = bg0; a.b.bgcol[3 * ray + 1] = bg1; a.b.bgcol[3 * ray + 2] = bg2; a.b.target[3 * ray] = tg0; a.b.target[3 * ray + 1] = tg1; a.b.target[3 * ray + 2] = tg2;
            }
            if (h == 0) {
                a.b.pts[3 * s_idx] = x[0]; a.b.pts[3 * s_idx + 1] = x[1]; a.b.pts[3 * s_idx + 2] = x[2]; a.b.tdist[s_idx] = t;
                half4_t o4 = { (half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3 }; reinterpret_cast<half4_t*>(a.b.O)[s_idx] = o4;
                half4_t d4 = { bdo[0], bdo[1], bdo[2], bdo[3] }; reinterpret_cast<half4_t*>(a.b.dO)[s_idx] = d4;
            }
            half_t* Eo = reinterpret_cast<half_t*>(a.b.E) + (size_t)s_idx * EPAD;
#pragma unroll
            for (int il = 0; il < S::LLV; ++il) { const int level = h * LPH + il; if (il < LPH && level < L) { Eo[2 * level] = ts.ef[2 * il]; Eo[2 * level + 1] = ts.ef[2 * il + 1]; } }
            half_t* Ho = reinterpret_cast<half_t*>(a.b.Hid) + (size_t)s_idx * W * NH;
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    Ho[32 * mb + rho(h, j)] = ts.h0[mb][0][j]; Ho[32 * mb + rho(h, 8 + j)] = ts.h0[mb][1][j];
                    if constexpr (NH == 2) { Ho[W + 32 * mb + rho(h, j)] = ts.h1[mb][0][j]; Ho[W + 32 * mb + rho(h, 8 + j)] = ts.h1[mb][1][j]; }
                }
        }

        if (nz_cnt != 0u || DUMP) {
        // ---- transposes needed by the weight gradients
        if (do_dw) {
#pragma unroll
        for (int il = 0; il < S::LLV; ++il) {
            const int level = h * LPH + il;
            if (il < LPH && level < L) { scr[S::SCR_E + (2 * level) * 32 + n] = ts.ef[2 * il]; scr[S::SCR_E + (2 * level + 1) * 32 + n] = ts.ef[2 * il + 1]; }
        }
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb) {
            if constexpr (NH == 2) { scratch_store_units(scr + S::SCR_HB, mb, n, h, ts.h0[mb][0], ts.h0[mb][1]); scratch_store_units(scr + S::SCR_HA, mb, n, h, ts.h1[mb][0], ts.h1[mb][1]); }
            else scratch_store_units(scr + S::SCR_HA, mb, n, h, ts.h0[mb][0], ts.h0[mb][1]);
        }
        }

        if (h == 0 && do_dw) {
#pragma unroll
            for (int c = 0; c < 4; ++c) scr[S::SCR_DO + c * 32 + n] = bdo[c];
        }
        tstamp(tc, 5);
        // ---- backward: dWout += H_last^T-side outer products (K = samples, via the LDS transposes)
        const int m = n;     // A-fragment row / B-fragment column of this lane
        if (do_dw) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8_t bcol;
#pragma unroll
            for (int j = 0; j < 8; ++j) bcol[j] = (half_t)0.f;
            if (m < kOut) bcol = *reinterpret_cast<const half8_t*>(scr + S::SCR_DO + m * 32 + 16 * s + 8 * h);
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) {
                const half8_t arow = *reinterpret_cast<const half8_t*>(scr + S::SCR_HA + (32 * mb + m) * 32 + 16 * s + 8 * h);
                dWo[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow, bcol, dWo[mb], 0, 0, 0);
            }
        }
        }
        // ---- dH of the last hidden layer = relu' * (Wout^T dO)
        half8_t dhl[S::MB][2];
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb) {
            float16_t acc = float16_t{ 0 };
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_WOT + mb, lane), bdo, acc, 0, 0, 0);
            if constexpr (NH == 2) mask_pack(acc, ts.h1[mb][0], ts.h1[mb][1], dhl[mb][0], dhl[mb][1]);
            else mask_pack(acc, ts.h0[mb][0], ts.h0[mb][1], dhl[mb][0], dhl[mb][1]);
            if (do_dw) scratch_store_units(scr + S::SCR_HA, mb, n, h, dhl[mb][0], dhl[mb][1]);       // H_last no longer needed: reuse as dH_last
        }
        half8_t dh0[S::MB][2];
        if constexpr (NH == 2) {
            // dW1[u2][u1] += dH1[u2][n] * H0[n][u1]
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int nb = 0; nb < S::MB; ++nb) {
                    const half8_t bcol = *reinterpret_cast<const half8_t*>(scr + S::SCR_HB + (32 * nb + m) * 32 + 16 * s + 8 * h);
#pragma unroll
                    for (int mb = 0; mb < S::MB; ++mb) {
                        const half8_t arow = *reinterpret_cast<const half8_t*>(scr + S::SCR_HA + (32 * mb + m) * 32 + 16 * s + 8 * h);
                        dW1[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow, bcol, dW1[mb][nb], 0, 0, 0);
                    }
                }
            // dH0 = relu' * (W1^T dH1)
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) {
                float16_t acc = float16_t{ 0 };
#pragma unroll
                for (int s = 0; s < S::KSW; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W1T + mb * S::KSW + s, lane), dhl[s >> 1][s & 1], acc, 0, 0, 0);
                mask_pack(acc, ts.h0[mb][0], ts.h0[mb][1], dh0[mb][0], dh0[mb][1]);
                scratch_store_units(scr + S::SCR_HB, mb, n, h, dh0[mb][0], dh0[mb][1]);   // H0 no longer needed: reuse as dH0
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) { dh0[mb][0] = dhl[mb][0]; dh0[mb][1] = dhl[mb][1]; }
        }
        tstamp(tc, 6);
        // ---- dW0[u][f] += dH0[u][n] * E[n][f]
        if (do_dw) {
            const half_t* dh0_scr = scr + (NH == 2 ? S::SCR_HB : S::SCR_HA);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8_t bcol = *reinterpret_cast<const half8_t*>(scr + S::SCR_E + (m & (EPAD - 1)) * 32 + 16 * s + 8 * h);
#pragma unroll
                for (int mb = 0; mb < S::MB; ++mb) {
                    const half8_t arow = *reinterpret_cast<const half8_t*>(dh0_scr + (32 * mb + m) * 32 + 16 * s + 8 * h);
                    dW0[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow, bcol, dW0[mb], 0, 0, 0);
                }
            }
        }
        // ---- dE = W0^T dH0, rows permuted so register r of half h is local feature r of that half
        float16_t de = float16_t{ 0 };
#pragma unroll
        for (int s = 0; s < S::KSW; ++s) de = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W0T + s, lane), dh0[s >> 1][s & 1], de, 0, 0, 0);
        if (DUMP) {
            half_t* dEo = reinterpret_cast<half_t*>(a.b.dE) + (size_t)s_idx * EPAD;
            half_t* dHo = reinterpret_cast<half_t*>(a.b.dHid) + (size_t)s_idx * W * NH;
#pragma unroll
            for (int il = 0; il < S::LLV; ++il) { const int level = h * LPH + il; if (il < LPH && level < L) { dEo[2 * level] = (half_t)de[2 * il]; dEo[2 * level + 1] = (half_t)de[2 * il + 1]; } }
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dHo[(NH - 1) * W + 32 * mb + rho(h, j)] = dhl[mb][0][j]; dHo[(NH - 1) * W + 32 * mb + rho(h, 8 + j)] = dhl[mb][1][j];
                    if constexpr (NH == 2) { dHo[32 * mb + rho(h, j)] = dh0[mb][0][j]; dHo[32 * mb + rho(h, 8 + j)] = dh0[mb][1][j]; }
                }
        }
        // ---- grid backward (tcnn kernel_grid_backward).  Levels small enough for an LDS tile hand dL/dE to
        //      k_grid_scatter (global packed-f16 atomics sustain only ~21 Gop/s on gfx950, ~12x below the gather
        //      rate: profiles/); larger levels scatter here with 8 global_atomic_pk_add_f16 per level.
        tstamp(tc, 7);
        const uint32_t Btot = R * 32u;
        const bool mine = ((nz32 >> n) & 1u) != 0u;                                        // this lane's sample is one of the non-zero ones
        const uint32_t bin_cap = Btot / n_bins, in_bin = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_base) + __popc(nz32 & ((1u << n) - 1u));
        const uint32_t slot = (ray & (n_bins - 1u)) * bin_cap + in_bin;                    // a bin holds the samples of R / n_bins rays at most
        const bool do_store = (a.ablate & 4u) == 0u && mine && in_bin < bin_cap;
        if (lds_level_mask && h == 0 && do_store) reinterpret_cast<float4_t*>(a.x_soa)[slot] = float4_t{ x[0], x[1], x[2], 0.f };          // one 16-byte store per sample
#pragma unroll
        for (int il = 0; il < S::LLV; ++il) {
            const int level = h * LPH + il;
            if (il < LPH && level < L) {
                const half_t q0 = (half_t)de[2 * il], q1 = (half_t)de[2 * il + 1];
                if (!ATOMIC_LEVELS || ((lds_level_mask >> level) & 1u)) {
                    // (clamped to the fixed-point range of the exact LDS accumulation, LevelFast::fix_clamp: |dL/dE| * fix_scale stays inside int32)
                    if (do_store) a.de_soa[(size_t)level * Btot + slot] = half2_t{ (half_t)clamp_f((float)q0, -a.lt.fix_clamp, a.lt.fix_clamp), (half_t)clamp_f((float)q1, -a.lt.fix_clamp, a.lt.fix_clamp) };
                } else if constexpr (ATOMIC_LEVELS) {
                    const float gq0 = (float)q0, gq1 = (float)q1;
                    if (gq0 != 0.f || gq1 != 0.f) {
                        gh2* gl = gtable + llt->offset[level];
                        level_corners(*llt, level, x, [&](int, uint32_t idx, float wgt2) {
                            __builtin_amdgcn_global_atomic_fadd_v2f16(gl + idx, half2_t{ (half_t)(wgt2 * gq0), (half_t)(wgt2 * gq1) });
                            if (a.touched_grid) a.touched_grid[(llt->offset[level] + idx) >> 2] = 1;
                        });
                    }
                }
            }
        }
        }
        tstamp(tc, 8);
    }

    // ---- reduce the weight-gradient accumulators over the workgroup's waves, write one fp32 partial row -- in ACCUMULATOR layout (frag_layout.h acc_param):
    //      each wave stores its registers to a private LDS copy as 16-byte pieces, consecutive lanes at consecutive addresses (no bank conflicts, no
    //      transposition), then all threads sum the four copies with 16-byte reads and store the row coalesced.  The summing kernel maps columns to parameters.
    __syncthreads();
    if (a.ablate & 32u) return;                                                      // timing experiments: no dW reduction (wrong results)
    {
        float* cp = reinterpret_cast<float*>(dyn) + (size_t)wave * (S::ACC_COLS + 64);
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                *reinterpret_cast<float4_t*>(cp + ((mb * 4 + rq) * 64 + lane) * 4) = float4_t{ dW0[mb][4 * rq], dW0[mb][4 * rq + 1], dW0[mb][4 * rq + 2], dW0[mb][4 * rq + 3] };
                if (n < kOut) *reinterpret_cast<float4_t*>(cp + S::ACC_WO + ((mb * 4 + rq) * 8 + h * 4 + n) * 4) = float4_t{ dWo[mb][4 * rq], dWo[mb][4 * rq + 1], dWo[mb][4 * rq + 2], dWo[mb][4 * rq + 3] };
                if constexpr (NH == 2) {
#pragma unroll
                    for (int nb = 0; nb < S::MB; ++nb)
                        *reinterpret_cast<float4_t*>(cp + S::ACC_W1 + (((mb * S::MB + nb) * 4 + rq) * 64 + lane) * 4) = float4_t{ dW1[mb][nb][4 * rq], dW1[mb][nb][4 * rq + 1], dW1[mb][nb][4 * rq + 2], dW1[mb][nb][4 * rq + 3] };
                }
            }
        if (lane == 0) cp[S::ACC_COLS] = loss_acc;
    }
    __syncthreads();
    {
        const float* r0 = reinterpret_cast<const float*>(dyn);
        float* dst = a.partials + (size_t)blockIdx.x * (S::ACC_COLS + 64);
        constexpr int ST = S::ACC_COLS + 64;
        for (int i = threadIdx.x * 4; i < S::ACC_COLS; i += blockDim.x * 4) {
            const float4_t v0 = *reinterpret_cast<const float4_t*>(r0 + i), v1 = *reinterpret_cast<const float4_t*>(r0 + ST + i);
            const float4_t v2 = *reinterpret_cast<const float4_t*>(r0 + 2 * ST + i), v3 = *reinterpret_cast<const float4_t*>(r0 + 3 * ST + i);
            *reinterpret_cast<float4_t*>(dst + i) = (v0 + v1) + (v2 + v3);
        }
        if (threadIdx.x == 0) dst[S::ACC_COLS] = (r0[S::ACC_COLS] + r0[ST + S::ACC_COLS]) + (r0[2 * ST + S::ACC_COLS] + r0[3 * ST + S::ACC_COLS]);
    }
#ifdef MON_FUSED_TIMING
    tstamp(tc, 9);
    tcx.acc[14] = (float)(uint32_t)(wall_clock64() & 0xffffffull);
    if (lane == 0) for (int k = 0; k < 16; ++k) a.b.tdist[(blockIdx.x * S::WAVES + wave) * 16 + k] = tcx.acc[k];
#endif
}

// ------------------------------------------------------------------ LDS grid scatter
// Measured on MI355X (profiles/r01_microbench.md): global_atomic_pk_add_f16 sustains ~21 Gop/s chip-wide, LDS
// floating-point atomics (ds_pk_add_f16, ds_add_f32) ~0.35 op/clk/CU, LDS integer atomics (ds_add_u32) ~4 lanes/clk/CU.
// So the scatter accumulates in LDS in int32 FIXED POINT with scale 2^24: every fp16 value is an exact multiple of
// 2^-24, so each contribution h(w * dE) converts exactly, integer addition is exact and order-independent, and the tile sum
// equals the exact sum of tcnn's fp16 contributions -- deterministic, unlike atomicAdd(__half2).
// Range: |sum| < 2^31 / scale per entry, feature and sample partition = 128 in loss-scaled units for loss_scale <= 128;
// a larger loss scale coarsens the unit by the same factor (LevelFast::fix_scale, set by the host), which keeps the range at
// "un-scaled gradient below 1.0" -- tcnn's own fp16 atomics would be down to 3 significant digits there.
//
// A level's accumulators (entries x 2 features x 4 B: 512 KB at 65 536 entries) need several workgroups, and each of them walks every sample of
// its partition -- so what matters is how little a workgroup does per sample, and that every level's workgroups finish together (the kernel ends with
// the slowest).  Two costs set the pace (profiles/r02_*): VALU issue (~4 cycles per wave instruction) and the LDS atomic unit (~4 lanes per clock, more
// when lanes collide: the samples of a ray that share a coarse cell hit the same eight addresses).
//   * hashed / large levels: one workgroup = (FEATURE, PARITY of the entry index, 32 768-entry range of that parity half, sample partition), a 128 KB tile
//     of int32.  Both features share all index arithmetic, but the split halves the corner work per workgroup.  The two x-corners of a (y, z) pair always
//     differ in the lowest index bit (scatter_item), so the owner of the even (odd) entries takes exactly ONE corner of each of the four pairs: no in-tile
//     test, no divergent branch, all lanes busy (tiles by entry range: eight tests for four hits on average, inside a branch every wave took anyway).
//   * small levels (the dense coarse ones: the LDS atomic unit is their limit): BOTH features in one 64-bit accumulator per entry -- lo = feature 0,
//     hi = feature 1, added as one sign-extended 64-bit integer, so a corner costs one ds_add_u64 instead of two ds_add_u32 in two workgroups; the whole
//     level in one tile while it fits the CU's 160 KB (20 448 entries), else one tile per parity (40 896 entries).
// Every level gets 16 workgroups: parts_l x P_l sample partitions (parts = 1 / 2 for the 64-bit tiles, 4 x ceil(entries / 65 536) otherwise).  Tiles are
// written densely as fp16 to partial table p, plane (feature, parity) of that level ([P][2][2][entries / 2]); the optimizer sums the P_l partial tables.
// No global atomics, no memset: every tile is fully rewritten each step.
constexpr uint32_t kScatterTile = 32768;          // entries per int32 tile of a parity half (one feature) = 128 KB
constexpr uint32_t kScatterLdsBytes = 163840;     // the workgroup declares the CU's whole LDS
constexpr uint32_t kScatterTile64 = (kScatterLdsBytes - 256u) / 8u;      // entries per 64-bit tile (both features): 20 448
constexpr uint32_t kScatterWgPerLevel = 16;
enum : int { kTileParity = 0, kTileParityRanged = 1, kTileWhole64 = 2, kTileParity64 = 3 };
__host__ __device__ inline int scatter_tile_mode(uint32_t size) { return size <= kScatterTile64 ? kTileWhole64 : (size <= 2u * kScatterTile64 ? kTileParity64 : (size <= 2u * kScatterTile ? kTileParity : kTileParityRanged)); }
__host__ __device__ inline uint32_t scatter_parts(uint32_t size) { const int m = scatter_tile_mode(size); return m == kTileWhole64 ? 1u : (m == kTileParity64 ? 2u : 4u * ((size + 2u * kScatterTile - 1u) / (2u * kScatterTile))); }

struct ScatterItem { half2_t g; float4_t x; };

// fixed-point contribution of one corner and feature: tcnn's (T)(weight * grad), exact in 1 / fs units
__device__ __forceinline__ int contrib_fix(float w, float g, float fs) { return (int)((float)(half_t)(w * g) * fs); }

// sign-extended packing of two fixed-point contributions into one 64-bit addend: the 64-bit sum S of such addends decodes exactly as lo = (int32)S,
// hi = (S - lo) >> 32 while both sums stay inside int32 (they do: the same clamp as for the 32-bit tiles)
__device__ __forceinline__ unsigned long long pack_fix(int lo, int hi) { return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)(hi + (lo >> 31)) << 32); }

// One sample, one level.  The two x-corners of a (y, z) pair always have entry indices of different parity -- hashed: idx1 = idx0 ^ ((x ^ (x + 1)) & mask)
// and x ^ (x + 1) is odd; dense: idx1 = idx0 + 1 modulo an even size (the clamps below only act on positions far outside [0,1]^3, which the sampler never
// produces: they keep such a sample inside the table, where it lands is then as meaningless as the sample).
template <bool HASHED, bool POW2, int MODE, bool DEGEN /* the index ignores y and z: the four pairs of a sample are ONE entry */>
__device__ __forceinline__ void scatter_item(int* tab, const ScatterItem& it, bool valid, uint32_t feature, float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask,
                                             uint32_t parity, uint32_t base_half, uint32_t tile, float fs) {
    constexpr bool BOTH = MODE == kTileWhole64 || MODE == kTileParity64;
    const float g = (float)(feature ? it.g.y : it.g.x), g0 = (float)it.g.x, g1 = (float)it.g.y;          // (k_fused_train stores dL/dE already clamped to the fixed-point range)
    if (!valid || (BOTH ? (g0 == 0.f && g1 == 0.f) : g == 0.f)) return;
    float pos[3]; uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float q = fmaf(scale, it.x[d], 0.5f), fl = floorf(q); pg[d] = (uint32_t)(int32_t)fl; pos[d] = q - fl; }
    // hashed levels: only the index bits below the (power-of-two) table size matter, so the 24-bit multiply (full rate) serves: positions are < 2^24
    const uint32_t ax0 = pg[0], ax1 = pg[0] + 1u, y0 = (HASHED && POW2) ? __umul24(pg[1], my & 0xffffffu) : pg[1] * my, z0 = (HASHED && POW2) ? __umul24(pg[2], mz & 0xffffffu) : pg[2] * mz;
    const uint32_t ay[2] = { y0, y0 + my }, az[2] = { z0, z0 + mz };
    const float wx[2] = { 1.f - pos[0], pos[0] }, wy[2] = { 1.f - pos[1], pos[1] }, wz[2] = { 1.f - pos[2], pos[2] };
    unsigned long long* tab64 = reinterpret_cast<unsigned long long*>(tab);
    if (MODE == kTileWhole64) {                                   // the whole level is this workgroup's: eight corners, nothing to test, one 64-bit atomic each
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t t = HASHED ? (ay[(k >> 1) & 1] ^ az[k >> 2]) : (ay[(k >> 1) & 1] + az[k >> 2]);
            uint32_t idx = (HASHED ? ((k & 1 ? ax1 : ax0) ^ t) : ((k & 1 ? ax1 : ax0) + t)) & mask;
            if (!POW2) { idx -= (idx >= size) ? size : 0u; idx = min(idx, size - 1u); }
            const float w = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];      // ((wx * wy) * wz): the reference walk's product order
            atomicAdd(tab64 + idx, pack_fix(contrib_fix(w, g0, fs), contrib_fix(w, g1, fs)));
        }
        return;
    }
    const uint32_t dxm = (ax0 ^ ax1) & mask;                      // hashed power-of-two level: idx1 = idx0 ^ dxm (odd)
    int dsum = 0; uint32_t dlocal = 0;                            // (degenerate level, see below)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t idx0, idx1;
        if (HASHED && POW2) { idx0 = (ax0 ^ ay[j & 1] ^ az[j >> 1]) & mask; idx1 = idx0 ^ dxm; }
        else {
            const uint32_t t = HASHED ? (ay[j & 1] ^ az[j >> 1]) : (ay[j & 1] + az[j >> 1]);
            idx0 = (HASHED ? (ax0 ^ t) : (ax0 + t)) & mask; idx1 = (HASHED ? (ax1 ^ t) : (ax1 + t)) & mask;
            if (!POW2) { idx0 -= (idx0 >= size) ? size : 0u; idx0 = min(idx0, size - 1u); idx1 -= (idx1 >= size) ? size : 0u; idx1 = min(idx1, size - 1u); }
        }
        const bool second = ((idx0 ^ parity) & 1u) != 0u;        // which corner of the pair is this workgroup's
        const uint32_t idx = second ? idx1 : idx0;
        const float w = ((second ? wx[1] : wx[0]) * wy[j & 1]) * wz[j >> 1];
        const uint32_t local = (idx >> 1) - base_half;
        if (MODE == kTileParity64) atomicAdd(tab64 + local, pack_fix(contrib_fix(w, g0, fs), contrib_fix(w, g1, fs)));
        else if (DEGEN) { dsum += contrib_fix(w, g, fs); dlocal = local; }      // all four pairs are the SAME entry: one atomic for the (exact) sum
        else if (MODE == kTileParity || local < tile) atomicAdd(tab + local, contrib_fix(w, g, fs));
    }
    if (DEGEN) atomicAdd(tab + dlocal, dsum);
}

template <bool HASHED, bool POW2, int MODE, bool DEGEN = false>
__device__ __forceinline__ void scatter_samples(int* tab, const half2_t* __restrict__ de, const float4_t* __restrict__ x4, uint32_t cnt_lo, uint32_t cnt_hi /* lane b: run length of ray bin b / b + 64 */,
                                                uint32_t n_bins, uint32_t bin0, uint32_t bin_step, uint32_t bin_cap, uint32_t feature,
                                                float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask, uint32_t parity, uint32_t base_half, uint32_t tile, float fs) {
    // This workgroup's samples are the ray bins bin0, bin0 + bin_step, ... (< n_bins), each a compacted run of samples at b * bin_cap.  They are
    // walked in STEPS.  While the runs are long (every sample carries a gradient: 1024 per bin) a step is one bin and thread t takes offset
    // r * 1024 + t; once they are short (late training: a few dozen per bin) the workgroup's waves split into G groups of W2 = 1024 / G threads
    // and a step covers G bins at once.  Either way the bin is uniform per WAVE, so its run length and base come from scalar registers
    // (v_readlane with a scalar lane index) and a sample costs two vector instructions of bookkeeping.  NOTHING inside the loop may wait on an
    // LDS or scalar-memory read: both share the counter (lgkmcnt) the LDS atomics are counted on, in order, so such a wait drains every atomic
    // issued before it.
    // Software pipeline: the kBatch steps of round r + 1 are requested before round r's index math and LDS atomics run, so the global-load
    // latency hides behind arithmetic (all 16 waves of the workgroup start in phase; without the prefetch they also wait in phase).
    constexpr int kBatch = MON_V_SBATCH;
    const auto count_of = [&](uint32_t b) { return (uint32_t)((b < 64u) ? __builtin_amdgcn_readlane((int)cnt_lo, (int)b) : __builtin_amdgcn_readlane((int)cnt_hi, (int)(b - 64u))); };   // b uniform
    uint32_t nb = 0, width = 0;
    for (uint32_t b = bin0; b < n_bins; b += bin_step) { ++nb; width = max(width, count_of(b)); }
    if (width == 0u) return;
    uint32_t w2s = 6u; while ((1u << w2s) < blockDim.x && (1u << w2s) < width) ++w2s;   // threads per bin and step: W2 = 2^w2s = the run length rounded up to a power of two, one wave at least
    const uint32_t W2 = 1u << w2s, gs = 10u - w2s, G = 1u << gs;                         // (1024 threads: G = 1024 / W2 groups; powers of two throughout, no divisions)
    const uint32_t wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> w2s)), lane_o = threadIdx.x & (W2 - 1u);
    const uint32_t ksteps = (nb + G - 1u) >> gs, rounds = (width + W2 - 1u) >> w2s, n_steps = rounds * ksteps;
    uint32_t fks = 0, fo = lane_o, fs_left = n_steps;                                   // running state of the step the next fetch serves (all but fo uniform)
    const auto fetch = [&](ScatterItem& it, bool& valid) {
        const uint32_t k = (fks << gs) + wg, b = min(bin0 + k * bin_step, n_bins - 1u);
        const uint32_t cnt = (fs_left && k < nb) ? count_of(b) : 0u;
        valid = fo < cnt; const uint32_t sc = b * bin_cap + (valid ? fo : 0u);
        it.g = de[sc]; it.x = x4[sc];
        fs_left -= fs_left ? 1u : 0u;
        const bool wrap = fks + 1u == ksteps;                                    // (selects, not branches: the compiler turned conditional updates of the captured state into scratch memory)
        fo += wrap ? W2 : 0u; fks = wrap ? 0u : fks + 1u;
    };
    ScatterItem nxt[kBatch]; bool nv[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) fetch(nxt[u], nv[u]);
    for (uint32_t s0 = 0; s0 < n_steps; s0 += kBatch) {
        ScatterItem cur[kBatch]; bool cv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) { cur[u] = nxt[u]; cv[u] = nv[u]; }
        if (s0 + kBatch < n_steps) {
#pragma unroll
            for (int u = 0; u < kBatch; ++u) fetch(nxt[u], nv[u]);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) scatter_item<HASHED, POW2, MODE, DEGEN>(tab, cur[u], cv[u], feature, scale, size, my, mz, mask, parity, base_half, tile, fs);
    }
}

// The weight-gradient partial rows of k_fused_train (one per workgroup) are summed here as well: every scatter workgroup
// takes a few float4 column groups (128 row subsets x 8 groups per pass).  The loads are issued at kernel entry and the sums
// are finished (DPP + a small LDS exchange) after the tile has been written, so their latency hides behind the scatter itself
// (k_reduce_partials remains for networks whose levels all go through global atomics).
struct PartialsArgs { const float* partials; uint32_t n_partials, stride, n_cols; FragDims fd; float* gmlp; DevState* st; };      // rows in accumulator layout: n_cols = acc_cols(fd), loss partial behind them
constexpr uint32_t kPartialsMaxPasses = 2;          // column-group passes a workgroup may hold in registers (n_mlp + 1 <= 2 * 8 * 4 * gridDim.x)

// column groups (of 4 columns) a workgroup sums per pass: as few as cover all groups with the whole grid (1, 2, 4 or 8), so that every workgroup
// carries the same small share instead of the first third of the grid carrying everything
__device__ __forceinline__ uint32_t partials_groups(const PartialsArgs& pa) {
    const uint32_t n4 = (pa.n_cols + 1u + 3u) / 4u, need = (n4 + gridDim.x - 1u) / gridDim.x;
    return need <= 1u ? 1u : (need <= 2u ? 2u : (need <= 4u ? 4u : 8u));
}
// the column groups go to the LAST workgroups of the grid: the first ones hold the coarse dense levels, whose sample walk is the longest of the kernel (their samples
// collide in the LDS atomic unit), so the row sums ride on workgroups that have slack
__device__ __forceinline__ uint32_t partials_block() {
#ifdef MON_PARTIALS_FIRST            // (variant build for the A/B measurement)
    return blockIdx.x;
#else
    return gridDim.x - 1u - blockIdx.x;
#endif
}
__device__ __forceinline__ void partials_prefetch(const PartialsArgs& pa, float4_t (&acc)[kPartialsMaxPasses]) {
    // thread = (column group gs of G, row subset sub of 1024 / G)
    const uint32_t n4 = (pa.n_cols + 1u + 3u) / 4u, G = partials_groups(pa), subs = blockDim.x / G, gs = threadIdx.x / subs, sub = threadIdx.x - gs * subs;
#pragma unroll
    for (uint32_t ps = 0; ps < kPartialsMaxPasses; ++ps) {
        const uint32_t g = (partials_block() + ps * gridDim.x) * G + gs; acc[ps] = float4_t{ 0.f, 0.f, 0.f, 0.f };
        if (g < n4) for (uint32_t k0 = sub; k0 < pa.n_partials; k0 += 4u * subs) {            // four independent 16-byte loads per round (rows are padded to n_cols + 64 floats)
            float4_t v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) { const uint32_t k = k0 + subs * u; v[u] = (k < pa.n_partials) ? *reinterpret_cast<const float4_t*>(pa.partials + (size_t)k * pa.stride + 4u * g) : float4_t{ 0.f, 0.f, 0.f, 0.f }; }
            acc[ps] += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
}
__device__ __forceinline__ void partials_finish(const PartialsArgs& pa, const float4_t (&acc)[kPartialsMaxPasses], float* red) {
    // the 64 subsets of a wave are summed with DPP, the 16 / G waves of a column group through LDS
    const uint32_t n4 = (pa.n_cols + 1u + 3u) / 4u, G = partials_groups(pa), wave = threadIdx.x >> 6, wpg = (blockDim.x >> 6) / G;
#pragma unroll
    for (uint32_t ps = 0; ps < kPartialsMaxPasses; ++ps) {
        const uint32_t g0 = (partials_block() + ps * gridDim.x) * G;
        if (g0 >= n4) break;                                                               // uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[ps][c];
            v += dpp_f<0x111, 0xF>(0.f, v); v += dpp_f<0x112, 0xF>(0.f, v); v += dpp_f<0x114, 0xF>(0.f, v); v += dpp_f<0x118, 0xF>(0.f, v);
            v += dpp_f<0x142, 0xA>(0.f, v); v += dpp_f<0x143, 0xC>(0.f, v);
            if ((threadIdx.x & 63u) == 63u) red[wave * 4u + (uint32_t)c] = v;               // lane 63 holds the wave total
        }
        __syncthreads();
        if (threadIdx.x < 4u * G) {
            const uint32_t gi = threadIdx.x >> 2, gg = g0 + gi, c = threadIdx.x & 3u, pi = 4u * gg + c;
            float v = 0.f; for (uint32_t w = 0; w < wpg; ++w) v += red[(gi * wpg + w) * 4u + c];
            if (gg < n4) { if (pi < pa.n_cols) { const int prm = acc_param(pa.fd, (int)pi); if (prm >= 0) pa.gmlp[prm] = v; } else if (pi == pa.n_cols) pa.st->loss_sum = v; }
        }
        __syncthreads();
    }
}

#ifndef MON_HOUSEKEEPING_BLOCK
#define MON_HOUSEKEEPING_BLOCK (gridDim.x - 1u)
#endif
__global__ void __launch_bounds__(1024) k_grid_scatter(LevelFast lt, ScatterLevels sl, const half2_t* __restrict__ de_soa, const float4_t* __restrict__ x4,
                                                       uint32_t B, uint32_t n_bins, half_t* __restrict__ gpart, uint32_t n_entries, const DevState* __restrict__ st, DevState* st_rw, DevState* st_next, PartialsArgs pa, float* __restrict__ timing, uint32_t ablate /* timing experiments: 1 no tile write-out, 2 no dW row sums, 4 no sample walk */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t iter = st->iter;
    if (blockIdx.x == MON_HOUSEKEEPING_BLOCK && threadIdx.x < 64u) {      // (the last workgroup: the first ones hold the coarse dense levels, the kernel's critical path)
        // slot-counter housekeeping (also for a skipped batch): clear the counters k_fused_train of the NEXT iteration counts in -- they live in the other
        // DevState, which nobody reads during this iteration -- and note how many samples carried a gradient in this one (k_optimizer hands it to the next
        // iteration as n_scatter_last; the large-table path decides on it)
        uint32_t v = 0u;
        for (uint32_t b = threadIdx.x; b < n_bins; b += 64u) { v += st->n_scatter[scatter_counter(iter, b)]; st_next->n_scatter[scatter_counter(iter + 1u, b)] = 0u; }
        v = scan_add64_u32(v);
        if (threadIdx.x == 63u) st_rw->n_scatter_now = v;
    }
    if (st->n_valid == 0u) return;
#ifdef MON_SCATTER_TIMING
    long long tq[10]; int tn = 0;
#define MON_ST_STAMP() do { __builtin_amdgcn_s_waitcnt(0); tq[tn++] = clock64(); } while (0)
#else
#define MON_ST_STAMP() do { } while (0)
#endif
    MON_ST_STAMP();
    float4_t pacc[kPartialsMaxPasses];
    if (ablate & 2u) pa.partials = nullptr;
    bool pacc_loaded = false;
    int* tab = reinterpret_cast<int*>(smem);
    float* red = reinterpret_cast<float*>(smem + (size_t)kScatterLdsBytes - 256u);     // 256 B behind the largest tile
    // run lengths of the compacted ray bins, lane b of every wave holds bin b's and bin (b + 64)'s (read back with v_readlane: no memory access in the sample loop)
    const uint32_t bin_cap = B / n_bins, lb = threadIdx.x & 63u;
    // (both sets are requested and the iteration's one is picked afterwards: the address must not wait for the load of the iteration counter)
    const uint32_t c_lo0 = (lb < n_bins) ? st->n_scatter[scatter_counter(0u, lb)] : 0u, c_lo1 = (lb < n_bins) ? st->n_scatter[scatter_counter(1u, lb)] : 0u;
    const uint32_t c_hi0 = (lb + 64u < n_bins) ? st->n_scatter[scatter_counter(0u, lb + 64u)] : 0u, c_hi1 = (lb + 64u < n_bins) ? st->n_scatter[scatter_counter(1u, lb + 64u)] : 0u;
    const uint32_t cnt_lo = min((iter & 1u) ? c_lo1 : c_lo0, bin_cap), cnt_hi = min((iter & 1u) ? c_hi1 : c_hi0, bin_cap);
    const uint32_t slot = blockIdx.x / kScatterWgPerLevel, j = blockIdx.x - slot * kScatterWgPerLevel;
    const int level = sl.level[slot]; const uint32_t P = sl.P[level];
    const uint32_t part = j / P, p = j - part * P;
    const uint32_t off = lt.offset[level], size = lt.size[level], my = lt.my[level], mz = lt.mz[level], mask = lt.mask[level];
    const bool hashed = lt.hashed[level] != 0u, pow2 = mask != 0xffffffffu;
    const float scale = lt.scale[level], fs = lt.fix_scale;
    const int mode = scatter_tile_mode(size);                                         // uniform: which kind of tile this level's workgroups hold (see above)
    const bool both = mode == kTileWhole64 || mode == kTileParity64;
    const uint32_t feature = both ? 0u : (part & 1u), parity = mode == kTileWhole64 ? 0u : (both ? (part & 1u) : ((part >> 1) & 1u));
    const uint32_t half_size = size >> 1, base_half = mode == kTileParityRanged ? (part >> 2) * kScatterTile : 0u;      // (level sizes are multiples of 8) parity tiles: idx = 2 * (base_half + local) + parity
    const bool degenerate = hashed && pow2 && (my & mask) == 0u && (mz & mask) == 0u;  // the index ignores y and z (tcnn's stride wrap-around at res = 65 536): the four pairs of a sample are one entry
    MON_ST_STAMP();
    if (mode != kTileParityRanged || base_half < half_size) {                         // (levels whose part count does not divide 16 leave workgroups without a tile)
        const uint32_t tile = mode == kTileWhole64 ? size : min(mode == kTileParity64 ? kScatterTile64 : kScatterTile, half_size - base_half);      // entries
        typedef int int4v __attribute__((ext_vector_type(4)));
        {   // tiles are multiples of 4 entries (tcnn rounds level sizes up to 8): clear with 16-byte stores
            int4v* t4 = reinterpret_cast<int4v*>(tab); const uint32_t n16 = both ? tile / 2u : tile / 4u;
            for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) t4[i] = int4v{ 0, 0, 0, 0 };
        }
        MON_ST_STAMP();
        __syncthreads();
        MON_ST_STAMP();
        // sample partition p of this level = the ray bins b = p, p + P, ... (16 bins, compacted by k_fused_train: only samples with a non-zero gradient)
        const half2_t* de = de_soa + (size_t)level * B;
#define MON_SCATTER_CALL(H, PW, MD, ...) scatter_samples<H, PW, MD, ##__VA_ARGS__>(tab, de, x4, cnt_lo, cnt_hi, n_bins, p, P, bin_cap, feature, scale, size, my, mz, mask, parity, base_half, tile, fs)
#define MON_SCATTER_MODE(MD) do { if (hashed) { if (pow2) MON_SCATTER_CALL(true, true, MD); else MON_SCATTER_CALL(true, false, MD); } else { if (pow2) MON_SCATTER_CALL(false, true, MD); else MON_SCATTER_CALL(false, false, MD); } } while (0)
        if (ablate & 4u) { }
        else if (mode == kTileWhole64) MON_SCATTER_MODE(kTileWhole64);
        else if (mode == kTileParity64) MON_SCATTER_MODE(kTileParity64);
        else if (mode == kTileParity) { if (degenerate) MON_SCATTER_CALL(true, true, kTileParity, true); else MON_SCATTER_MODE(kTileParity); }
        else MON_SCATTER_MODE(kTileParityRanged);
#undef MON_SCATTER_MODE
#undef MON_SCATTER_CALL
        // the dW partial rows are requested HERE, behind the walk's last load: vmcnt retires in order, so anything loaded after them -- the bin counters, every
        // sample fetch -- would wait for these HBM round trips first (requested at kernel entry they cost 1.8 us); now they land while the tile is written out
        if (pa.partials) { partials_prefetch(pa, pacc); pacc_loaded = true; }
        MON_ST_STAMP();
        __syncthreads();
        MON_ST_STAMP();
        const int4v* t4 = reinterpret_cast<const int4v*>(tab);
        const float inv = 1.0f / fs;
        if (!(ablate & 1u)) {
        const size_t plane = n_entries >> 1;                                           // partial table p, plane (feature, parity): entry idx at [idx >> 1]
        half_t* pl = gpart + ((size_t)p * 4u) * plane + (off >> 1);
        const auto lo_hi = [&](int lo_bits, int hi_bits, float& f0, float& f1) { f0 = (float)lo_bits * inv; f1 = (float)(hi_bits - (lo_bits >> 31)) * inv; };      // undo pack_fix
        if (mode == kTileWhole64) {            // entries 2k, 2k + 1 interleaved, both features: 4 entries (32 B) per thread and pass -> 2 halves into each of the four planes
            for (uint32_t i = threadIdx.x; i < tile / 4u; i += blockDim.x) {
                const int4v a = t4[2u * i], c = t4[2u * i + 1u];                        // entries 4i, 4i+1 | 4i+2, 4i+3
                float e0f0, e0f1, e1f0, e1f1, e2f0, e2f1, e3f0, e3f1; lo_hi(a[0], a[1], e0f0, e0f1); lo_hi(a[2], a[3], e1f0, e1f1); lo_hi(c[0], c[1], e2f0, e2f1); lo_hi(c[2], c[3], e3f0, e3f1);
                *reinterpret_cast<half2_t*>(pl + 0u * plane + 2u * i) = half2_t{ (half_t)e0f0, (half_t)e2f0 };      // feature 0, even entries
                *reinterpret_cast<half2_t*>(pl + 1u * plane + 2u * i) = half2_t{ (half_t)e1f0, (half_t)e3f0 };      // feature 0, odd
                *reinterpret_cast<half2_t*>(pl + 2u * plane + 2u * i) = half2_t{ (half_t)e0f1, (half_t)e2f1 };      // feature 1, even
                *reinterpret_cast<half2_t*>(pl + 3u * plane + 2u * i) = half2_t{ (half_t)e1f1, (half_t)e3f1 };      // feature 1, odd
            }
        } else if (mode == kTileParity64) {    // one parity, both features: 4 entries (32 B) per thread and pass -> 4 halves into each of the two feature planes
            for (uint32_t i = threadIdx.x; i < tile / 4u; i += blockDim.x) {
                const int4v a = t4[2u * i], c = t4[2u * i + 1u];
                float f0[4], f1[4]; lo_hi(a[0], a[1], f0[0], f1[0]); lo_hi(a[2], a[3], f0[1], f1[1]); lo_hi(c[0], c[1], f0[2], f1[2]); lo_hi(c[2], c[3], f0[3], f1[3]);
                *reinterpret_cast<half4_t*>(pl + (0u + parity) * plane + 4u * i) = half4_t{ (half_t)f0[0], (half_t)f0[1], (half_t)f0[2], (half_t)f0[3] };
                *reinterpret_cast<half4_t*>(pl + (2u + parity) * plane + 4u * i) = half4_t{ (half_t)f1[0], (half_t)f1[1], (half_t)f1[2], (half_t)f1[3] };
            }
        } else {                               // int32 tile of one feature and parity: 8 entries per thread and pass, one 16-byte store of eight halves
            half_t* dst = pl + (feature * 2u + parity) * plane + base_half;
            for (uint32_t i = threadIdx.x; i < tile / 8u; i += blockDim.x) {
                const int4v a0 = t4[2u * i], a1 = t4[2u * i + 1u];
                half8_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (half_t)((float)a0[e] * inv); o[4 + e] = (half_t)((float)a1[e] * inv); }
                *reinterpret_cast<half8_t*>(dst + 8u * i) = o;
            }
            if ((tile & 4u) && threadIdx.x == 0u) {                                    // a parity half is a multiple of 4 entries, not always of 8
                const int4v a0 = t4[tile / 4u - 1u];
                *reinterpret_cast<half4_t*>(dst + (tile & ~7u)) = half4_t{ (half_t)((float)a0[0] * inv), (half_t)((float)a0[1] * inv), (half_t)((float)a0[2] * inv), (half_t)((float)a0[3] * inv) };
            }
        }
        }
    }
    MON_ST_STAMP();
    if (pa.partials && !pacc_loaded) partials_prefetch(pa, pacc);                     // (a workgroup without a tile)
    if (pa.partials) partials_finish(pa, pacc, red);
    MON_ST_STAMP();
#ifdef MON_SCATTER_TIMING
    if (timing && (threadIdx.x & 63u) == 0u) { float* o = timing + ((size_t)blockIdx.x * 16u + (threadIdx.x >> 6)) * 8u; for (int k = 0; k + 1 < tn && k < 6; ++k) o[k] = (float)(tq[k + 1] - tq[k]); o[6] = (float)(tq[0] & 0xffffff); o[7] = (float)level; }
#endif
}

// Host: which levels go through the LDS scatter, with how many sample partitions each.
uint32_t scatter_plan(const LevelTable& lt, const NetDims& nd, ScatterLevels& sl) {
    uint32_t mask = 0; sl.n_levels = 0; sl.max_P = 0;
    for (int l = 0; l < kMaxLevels; ++l) { sl.P[l] = 0; sl.level[l] = 0; sl.entry_offset[l] = lt.offset[l]; }
    sl.entry_offset[kMaxLevels] = lt.offset[kMaxLevels];
    // A level of up to 16 tiles fits the 16-workgroup plan, but with more than 4 tiles every workgroup walks all the samples of the batch
    // for its one tile.  When the table has levels that need the large-table path anyway (kernels_bigscatter.hip), levels of 5..16
    // tiles go there too.
    uint32_t max_parts = kScatterWgPerLevel;
    for (int l = 0; l < nd.L; ++l) if (scatter_parts(lt.offset[l + 1] - lt.offset[l]) > kScatterWgPerLevel) max_parts = 4;
    for (int l = 0; l < nd.L; ++l) {
        const uint32_t size = lt.offset[l + 1] - lt.offset[l];
        const uint32_t parts = scatter_parts(size);
        if (parts <= max_parts) {
            mask |= 1u << l; sl.level[sl.n_levels++] = (uint8_t)l;
            sl.P[l] = (uint8_t)(kScatterWgPerLevel / parts); if (sl.P[l] > sl.max_P) sl.max_P = sl.P[l];
        }
    }
    return mask;
}
uint32_t scatter_level_mask(const LevelTable& lt, const NetDims& nd) { ScatterLevels sl; return scatter_plan(lt, nd, sl); }

#ifdef MON_SCATTER_TIMING
static float* g_scatter_timing_buf = nullptr;
#endif
uint32_t fused_partial_cols(const NetDims& nd) { return (uint32_t)acc_cols(FragDims{ nd.Epad, nd.W, nd.NH, nd.L }); }
bool grid_scatter_sums_partials(const LevelTable& lt, const NetDims& nd) {
    // the scatter workgroups hold their share of the dW column groups in registers (kPartialsMaxPasses passes of 8 groups of 4 columns)
    ScatterLevels sl; if (!scatter_plan(lt, nd, sl)) return false;
    return fused_partial_cols(nd) + 1u <= kPartialsMaxPasses * 8u * 4u * sl.n_levels * kScatterWgPerLevel;
}
void launch_grid_scatter(hipStream_t s, const LevelTable& lt, const LevelFast& lf, const NetDims& nd, const uint16_t* de_soa, const float* x_soa, uint32_t B, uint32_t n_bins, uint16_t* gpart, uint32_t part_stride_entries, DevState* st,
                         const float* partials, uint32_t n_partials, float* gmlp, DevState* st_next) {
    ScatterLevels sl; if (!scatter_plan(lt, nd, sl)) return;
    const PartialsArgs pa{ partials, n_partials, fused_partial_cols(nd) + 64u, fused_partial_cols(nd), FragDims{ nd.Epad, nd.W, nd.NH, nd.L }, gmlp, st };
    constexpr uint32_t smem = kScatterLdsBytes;
    static std::atomic<uint64_t> attr_devices{ 0 }; static std::mutex attr_mu;
    once_per_device(attr_devices, attr_mu, [] { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
    float* timing = nullptr;
#ifdef MON_SCATTER_TIMING
    static float* g_timing = nullptr; if (!g_timing) hipMalloc((void**)&g_timing, 256 * 16 * 8 * 4); timing = g_timing; g_scatter_timing_buf = g_timing;
#endif
    hipLaunchKernelGGL(k_grid_scatter, dim3(sl.n_levels * kScatterWgPerLevel), dim3(1024), smem, s, lf, sl, reinterpret_cast<const half2_t*>(de_soa), reinterpret_cast<const float4_t*>(x_soa), B, n_bins,
                       reinterpret_cast<half_t*>(gpart), part_stride_entries, st, st, st_next, pa, timing, (uint32_t)options().scatter_ablate);
}
#ifdef MON_SCATTER_TIMING
extern "C" int mon_debug_scatter_timing(float* out) { hipDeviceSynchronize(); return g_scatter_timing_buf ? (int)hipMemcpy(out, g_scatter_timing_buf, 256 * 16 * 8 * 4, hipMemcpyDeviceToHost) : -1; }
#endif

// ------------------------------------------------------------------ fused render kernel
// One wavefront per pixel ray, 2S = 64 samples as two 32-sample tiles with a carried transmittance;
// rays that miss the box and tiles behind an opaque prefix are skipped (wave-uniform).
// GenerateRenderInputPoints :593-626 + inference + VolumeRender_Render :1134-1229.
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_fused_render(FusedArgs a, uint32_t n_rays, uint32_t idx_base, float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ mask) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    LevelLds* llt = reinterpret_cast<LevelLds*>(smem + S::FRAG_BYTES);
    build_fragments<EPAD, W, NH>(frags, llt, a, false);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31;
    const int L = a.nd.L; const uint32_t S2 = 2u * a.oc.S;      // 64
    const half2_t* table = reinterpret_cast<const half2_t*>(a.params + a.nd.n_mlp);
    const LevelRegs lregs = load_level_regs_uniform(a.lt, L, lane); const uint32_t table_bytes = a.lt.offset[L] * 4u;      // (from the argument segment: it ends up in the buffer descriptor, which must be scalar)
    for (uint32_t ray = blockIdx.x * S::WAVES + wave; ray < n_rays; ray += gridDim.x * S::WAVES) {
        float o0 = 1.f, o1 = 1.f, o2 = 1.f, od = 0.f, om_ = 0.f;
        if (a.b.ray_flag[ray]) {
            const float t0 = a.b.ray_t0[ray], t1 = a.b.ray_t1[ray], dtr = (t1 - t0) / (float)S2;
            float Tc = 1.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, tlast = 0.f;
            for (uint32_t tile = 0; tile < 2u; ++tile) {
                if (Tc < kTransmittanceEps) break;
                const uint32_t k = tile * 32u + (uint32_t)n;
                const float t = fmaf(dtr, (float)k + render_rand(a.oc, idx_base + ray * S2 + k), t0);
                float x[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { const float p = fmaf(t, a.b.ray_d[3 * ray + d], a.b.ray_o[3 * ray + d]); x[d] = (p - a.oc.aabb.mn[d]) / (a.oc.aabb.mx[d] - a.oc.aabb.mn[d]); }
                TileState<EPAD, W, NH> ts;
                tile_forward<EPAD, W, NH>(ts, frags, lregs, table, table_bytes, L, x, lane);
                const float c0 = logistic_f(ts.out4[0]), c1 = logistic_f(ts.out4[1]), c2 = logistic_f(ts.out4[2]), sigma = __expf(ts.out4[3]);
                float tprev = lane_prev(t, tlast); if (n == 0) tprev = tlast;
                const float alpha = 1.f - __expf(-sigma * (t - tprev)), omv = 1.f - alpha;
                const float tincl = scan_mul32(omv) * Tc;
                float T = lane_prev(tincl, Tc); if (n == 0) T = Tc;
                const bool active = T >= kTransmittanceEps;
                const int nact = __popc((uint32_t)__ballot(active));
                const float wgt = active ? alpha * T : 0.f;
                r0 += lane_bcast(scan_add32(wgt * c0), 31); r1 += lane_bcast(scan_add32(wgt * c1), 31); r2 += lane_bcast(scan_add32(wgt * c2), 31);
                dep += lane_bcast(scan_add32(wgt * t), 31);
                Tc = (nact > 0) ? lane_bcast(tincl, nact > 0 ? nact - 1 : 0) : Tc;      // all 64 lanes carry half-wave 0's state (uniform control flow)
                tlast = lane_bcast(t, 31);
            }
            if (1.f - Tc > 0.5f) { o0 = r0 + Tc; o1 = r1 + Tc; o2 = r2 + Tc; od = dep / a.b.ray_dn[ray]; om_ = 1.f; }      // :1213-1220
        }
        if (lane == 0) { rgb[3 * ray] = o0; rgb[3 * ray + 1] = o1; rgb[3 * ray + 2] = o2; depth[ray] = od; mask[ray] = om_; }
    }
}

// ------------------------------------------------------------------ occupancy grid (N1: forward-pass skipping, default off)
// BASELINE.json's north star names occupancy-grid skipping; the reference has none (it always takes 32 uniform samples inside the box,
// nerf_model.cu:536-566), so the feature is opt-in (mon_config::occupancy_skip) and the parity tests run without it.  A kOccRes^3 bit grid over
// the object's box is refreshed from the CURRENT training weights every kOccInterval iterations after a warm-up: one wavefront evaluates the
// network's raw density at the centres of 32 cells (the same tile_forward as training) and ballots "density above the threshold" into one
// word; a second pass dilates by one cell in every direction.  k_fused_train then skips the gathers of samples in empty cells.
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_occ_density(FusedArgs a, float raw_threshold, uint32_t* __restrict__ bits_out) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    LevelLds* llt = reinterpret_cast<LevelLds*>(smem + S::FRAG_BYTES);
    build_fragments<EPAD, W, NH>(frags, llt, a, false);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31;
    const half2_t* table = reinterpret_cast<const half2_t*>(a.params + a.nd.n_mlp);
    constexpr uint32_t n_words = kOccRes * kOccRes * kOccRes / 32;
    const LevelRegs lregs = load_level_regs_uniform(a.lt, a.nd.L, lane); const uint32_t table_bytes = a.lt.offset[a.nd.L] * 4u;
    for (uint32_t word = blockIdx.x * S::WAVES + wave; word < n_words; word += gridDim.x * S::WAVES) {
        const uint32_t cell = word * 32u + (uint32_t)n, cx = cell % kOccRes, cy = (cell / kOccRes) % kOccRes, cz = cell / (kOccRes * kOccRes);
        const float x[3] = { ((float)cx + 0.5f) / (float)kOccRes, ((float)cy + 0.5f) / (float)kOccRes, ((float)cz + 0.5f) / (float)kOccRes };
        TileState<EPAD, W, NH> ts;
        tile_forward<EPAD, W, NH>(ts, frags, lregs, table, table_bytes, a.nd.L, x, lane);
        const uint32_t occ = (uint32_t)__ballot(lane < 32 && ts.out4[3] > raw_threshold);      // raw channel 3 = log density (network_to_density = exp, nerf_model.cu:49)
        if (lane == 0) bits_out[word] = occ;
    }
}
// a cell stays live if it or any of its 26 neighbours is occupied (the network is only sampled at cell centres)
__global__ void __launch_bounds__(256) k_occ_dilate(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    constexpr int WPR = kOccRes / 32;                                                   // words per x row
    const uint32_t word = blockIdx.x * blockDim.x + threadIdx.x;
    if (word >= (uint32_t)(kOccRes * kOccRes * WPR)) return;
    const int wx = (int)(word % WPR), cy = (int)((word / WPR) % kOccRes), cz = (int)(word / (WPR * kOccRes));
    uint32_t acc = 0u;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
        const int y = cy + dy, z = cz + dz; if (y < 0 || y >= kOccRes || z < 0 || z >= kOccRes) continue;
        const uint32_t* row = in + ((size_t)z * kOccRes + y) * WPR;
        const uint32_t w = row[wx], wl = wx > 0 ? row[wx - 1] : 0u, wr = wx + 1 < WPR ? row[wx + 1] : 0u;
        acc |= w | (w << 1) | (w >> 1) | (wl >> 31) | (wr << 31);
    }
    out[word] = acc;
}
template <int EPAD, int W, int NH>
static void occ_update_t(hipStream_t s, const FusedArgs& a, float raw_threshold, uint32_t* tmp, uint32_t* bits) {
    using S = FusedShape<EPAD, W, NH>;
    constexpr uint32_t n_words = kOccRes * kOccRes * kOccRes / 32;
    hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::F_WOT * 512 + 255) / 256), dim3(256), 0, s, a.params, a.nd.L, const_cast<uint16_t*>(a.frag_image), (const DevState*)nullptr);
    hipLaunchKernelGGL((k_occ_density<EPAD, W, NH>), dim3(n_words / S::WAVES), dim3(256), S::FRAG_BYTES + S::LT_BYTES, s, a, raw_threshold, tmp);
    hipLaunchKernelGGL(k_occ_dilate, dim3((n_words + 255) / 256), dim3(256), 0, s, tmp, bits);
}
// ------------------------------------------------------------------ host side
bool fused_supported(const NetDims& nd, uint32_t S, uint32_t R) {
    return S == 32 && R <= 16384u && nd.L >= 1 && nd.L <= kMaxLevels && (nd.Epad == 16 || nd.Epad == 32) && (nd.W == 32 || nd.W == 64) && (nd.NH == 1 || nd.NH == 2);
}

uint32_t fused_train_grid(const NetDims&, uint32_t R) {
    const uint32_t want = (R + 3) / 4;            // one ray per wavefront when it fits
    uint32_t cap = options().fused_grid > 0 ? (uint32_t)options().fused_grid : kMaxFusedGrid;
    if (cap > kMaxFusedGrid) cap = kMaxFusedGrid;          // (the dW partial rows are allocated for kMaxFusedGrid workgroups)
    return want < cap ? want : cap;
}

template <int EPAD, int W, int NH>
static void fused_train_t(hipStream_t s, const FusedArgs& a, uint32_t grid, int dump) {
    using S = FusedShape<EPAD, W, NH>;
    static std::atomic<uint64_t> attr_devices{ 0 }; static std::mutex attr_mu;
    once_per_device(attr_devices, attr_mu, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
    });
    const bool all_lds = a.lds_level_mask != 0u && (a.lds_level_mask == ((a.nd.L >= 32) ? 0xffffffffu : ((1u << a.nd.L) - 1u)));
    if (a.e_soa && all_lds && !dump && !a.occ_bits) { hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a); return; }
    if (dump) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, true, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);          // (the debug dump evaluates every sample)
    else if (a.occ_bits) { if (all_lds) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
                           else hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, true, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a); }
    else if (all_lds) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
    else hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
}
template <int EPAD, int W, int NH>
static void fused_render_t(hipStream_t s, const FusedArgs& a, uint32_t n_rays, uint32_t idx_base, float* rgb, float* depth, float* mask) {
    using S = FusedShape<EPAD, W, NH>;
    const uint32_t smem = S::FRAG_BYTES + S::LT_BYTES;
    uint32_t grid = (n_rays + 3) / 4; if (grid > 2048u) grid = 2048u;
    if (a.ablate & 1u) hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::F_WOT * 512 + 255) / 256), dim3(256), 0, s, a.params, a.nd.L, const_cast<uint16_t*>(a.frag_image), (const DevState*)nullptr);   // first chunk of a render call
    hipLaunchKernelGGL((k_fused_render<EPAD, W, NH>), dim3(grid), dim3(256), smem, s, a, n_rays, idx_base, rgb, depth, mask);
}

template <int EPAD, int W, int NH>
static void candidates_frags_t(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st, const uint16_t* params, const NetDims& nd, uint16_t* image) {
    using S = FusedShape<EPAD, W, NH>;
    const uint32_t cand_blocks = (oc.R + 255) / 256, frag_blocks = (S::N_FRAGS * 512 + 255) / 256;
    hipLaunchKernelGGL((k_candidates_and_frags<EPAD, W, NH>), dim3(cand_blocks + frag_blocks), dim3(256), 0, s, b, ds, oc, st, cand_blocks, params, nd.L, image);
}

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
            default: break;                                                                    \
        }                                                                                      \
    } while (0)

void launch_fused_train(hipStream_t s, const LevelFast& lt, const NetDims& nd, const ParamPtrs& p, const BatchPtrs& b, const ObjectConst& oc, DevState* st, float* dw_partials, int debug_dump,
                        uint16_t* de_soa, float* x_soa, uint32_t lds_level_mask, uint16_t* frag_image, uint32_t big_switch, uint8_t* touched, const uint32_t* occ_bits, uint32_t n_bins, const uint16_t* e_soa) {
    const uint32_t ablate = (uint32_t)options().fused_ablate;
    uint32_t stagger = options().fused_stagger < 0 ? kDefaultStagger : (uint32_t)options().fused_stagger;
    if (oc.R < 2u * 4u * fused_train_grid(nd, oc.R)) stagger = 0u;      // a wave with one ray has no second phase to interleave: the delay would only be lost
    FusedArgs a{ lt, nd, oc, b, p.half, p.ggrid, dw_partials, st, reinterpret_cast<half2_t*>(de_soa), x_soa, lds_level_mask, frag_image, ablate, touched ? touched + (nd.n_mlp >> 3) : nullptr, big_switch, n_bins, stagger, occ_bits, reinterpret_cast<const half2_t*>(e_soa) };
    if (e_soa) a.stagger = 0u;          // (the stagger interleaves gather phases with compute phases; a pre-encoded batch has no gather phase)
    const uint32_t grid = fused_train_grid(nd, oc.R);
    MON_FUSED_DISPATCH(fused_train_t, s, a, grid, debug_dump);
}
template <int EPAD, int W, int NH>
static void build_frag_image_t(hipStream_t s, const uint16_t* params, const NetDims& nd, uint16_t* image) {
    using S = FusedShape<EPAD, W, NH>;
    hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::N_FRAGS * 512 + 255) / 256), dim3(256), 0, s, params, nd.L, image, (const DevState*)nullptr);
}
void launch_build_frag_image(hipStream_t s, const uint16_t* params, const NetDims& nd, uint16_t* image) { MON_FUSED_DISPATCH(build_frag_image_t, s, params, nd, image); }

void launch_candidates_and_frags(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st, const uint16_t* params, const NetDims& nd, uint16_t* frag_image) {
    MON_FUSED_DISPATCH(candidates_frags_t, s, b, ds, oc, st, params, nd, frag_image);
}
void launch_fused_render(hipStream_t s, const LevelFast& lt, const NetDims& nd, const uint16_t* params, const BatchPtrs& b, const ObjectConst& oc, uint32_t n_rays, uint32_t idx_base, float* rgb, float* depth, float* mask, uint16_t* frag_image, int build_image) {
    FusedArgs a{ lt, nd, oc, b, params, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, frag_image, build_image ? 1u : 0u };   // `ablate` bit 0 doubles as "build the fragment image first" on the host side of the render path
    MON_FUSED_DISPATCH(fused_render_t, s, a, n_rays, idx_base, rgb, depth, mask);
}

void launch_occupancy_update(hipStream_t s, const LevelFast& lt, const NetDims& nd, const uint16_t* params, const ObjectConst& oc, uint16_t* frag_image, float raw_threshold, uint32_t* tmp, uint32_t* bits) {
    FusedArgs a{}; a.lt = lt; a.nd = nd; a.oc = oc; a.params = params; a.frag_image = frag_image;
    MON_FUSED_DISPATCH(occ_update_t, s, a, raw_threshold, tmp, bits);
}


}  // namespace mon
