// kernels_scatter.hip -- k_grid_scatter: the grid backward as exact int32 fixed-point accumulation in LDS tiles (one workgroup per level, feature, parity and
// sample partition), the dW partial-row sums that ride on its workgroups, and the scatter plan.  Wave scans: fused_device.h.
#include "fused_device.h"

namespace mon {

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
#ifndef MON_V_SGROUP
#define MON_V_SGROUP 2
#endif
constexpr uint32_t kScatterTile = 32768;          // entries per int32 tile of a parity half (one feature) = 128 KB
constexpr uint32_t kScatterLdsBytes = 163840;     // the workgroup declares the CU's whole LDS
constexpr uint32_t kScatterTile64 = (kScatterLdsBytes - 256u) / 8u;      // entries per 64-bit tile (both features): 20 448
constexpr uint32_t kScatterWgPerLevel = 16;
enum : int { kTileParity = 0, kTileParityRanged = 1, kTileWhole64 = 2, kTileParity64 = 3 };
__host__ __device__ inline int scatter_tile_mode(uint32_t size) {
    return size <= kScatterTile64 ? kTileWhole64
        : (size <= 2u * kScatterTile64 ? kTileParity64 : (size <= 2u * kScatterTile ? kTileParity : kTileParityRanged)); }
__host__ __device__ inline uint32_t scatter_parts(uint32_t size) { const int m = scatter_tile_mode(size);
    return m == kTileWhole64 ? 1u : (m == kTileParity64 ? 2u : 4u * ((size + 2u * kScatterTile - 1u) / (2u * kScatterTile))); }

struct ScatterItem { half2_t g; float4_t x; };

// fixed-point contribution of one corner and feature: tcnn's (T)(weight * grad), exact in 1 / fs units
__device__ __forceinline__ int contrib_fix(float w, float g, float fs) { return (int)((float)(half_t)opaque_f32(w * g) * fs); }

// sign-extended packing of two fixed-point contributions into one 64-bit addend: the 64-bit sum S of such addends decodes exactly as lo = (int32)S,
// hi = (S - lo) >> 32 while both sums stay inside int32 (they do: the same clamp as for the 32-bit tiles)
__device__ __forceinline__ unsigned long long pack_fix(int lo, int hi) {
    return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)(hi + (lo >> 31)) << 32); }

// (written as instructions: from the C forms the compiler rebuilt a compare + select pair for each of the two)
// a ^ b ^ c
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r; asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int floor_to_int(float q) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(q)); return r; }      // (int)floorf(q)
__device__ __forceinline__ uint32_t sign_of_bit0(uint32_t h) { uint32_t m; asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m) : "v"(h)); return m; }
__device__ __forceinline__ uint32_t sign_of_bit1(uint32_t h) { uint32_t m; asm("v_bfe_i32 %0, %1, 1, 1" : "=v"(m) : "v"(h)); return m; }   // bit 1 set ? ~0 : 0
// (m & a) | (~m & b)
__device__ __forceinline__ uint32_t bit_select(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }

// One sample, one level.  The two x-corners of a (y, z) pair always have entry indices of different parity -- hashed: idx1 = idx0 ^ ((x ^ (x + 1)) & mask)
// and x ^ (x + 1) is odd; dense: idx1 = idx0 + 1 modulo an even size (the clamps below only act on positions far outside [0,1]^3, which the sampler never
// produces: they keep such a sample inside the table, where it lands is then as meaningless as the sample).
template <bool HASHED, bool POW2, int MODE, bool DEGEN /* the index ignores y and z: the four pairs of a sample are ONE entry */>
__device__ __forceinline__ void scatter_item(int* tab, const ScatterItem& it, bool valid, uint32_t feature, float scale, uint32_t size, uint32_t my,
        uint32_t mz, uint32_t mask,
                                             uint32_t parity, uint32_t base_half, uint32_t tile, float fs) {
    constexpr bool BOTH = MODE == kTileWhole64 || MODE == kTileParity64;
    // (k_fused_train stores dL/dE already clamped to the fixed-point range)
    const float g = (float)(feature ? it.g.y : it.g.x), g0 = (float)it.g.x, g1 = (float)it.g.y;
    if (!valid || (BOTH ? (g0 == 0.f && g1 == 0.f) : g == 0.f)) return;
    // The floating-point side works on PAIRS (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations, two per instruction -- the walk is bound by
    // VALU issue): x | y of the position, the weights of the pairs j = 0, 1 (they share wz[0]) and j = 2, 3 (wz[1]), and the products with the fixed-point
    // unit.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 qxy = __builtin_elementwise_fma(f2{ scale, scale }, f2{ it.x[0], it.x[1] }, f2{ 0.5f, 0.5f }); const float qz = fmaf(scale, it.x[2], 0.5f);
    // (q - floor(q) and (int)floor(q) as ONE instruction each: v_fract_f32 is exactly that difference for the non-negative q of a sample inside the box)
    const f2 pxy = { __builtin_amdgcn_fractf(qxy.x), __builtin_amdgcn_fractf(qxy.y) }, nxy = f2{ 1.f, 1.f } - pxy; const float pz = __builtin_amdgcn_fractf(qz);
    const uint32_t pg[3] = { (uint32_t)floor_to_int(qxy.x), (uint32_t)floor_to_int(qxy.y), (uint32_t)floor_to_int(qz) };
    // hashed levels: only the index bits below the (power-of-two) table size matter, so the 24-bit multiply (full rate) serves: positions are < 2^24
    const uint32_t ax0 = pg[0], ax1 = pg[0] + 1u, y0 = (HASHED && POW2) ? __umul24(pg[1], my & 0xffffffu) : pg[1] * my, z0 = (HASHED && POW2)
            ? __umul24(pg[2], mz & 0xffffffu) : pg[2] * mz;
    const uint32_t ay[2] = { y0, y0 + my }, az[2] = { z0, z0 + mz };
    const float wx[2] = { nxy.x, pxy.x }, wz[2] = { 1.f - pz, pz }; const f2 wy2 = { nxy.y, pxy.y };
    unsigned long long* tab64 = reinterpret_cast<unsigned long long*>(tab);
    // (the fp32 products pass through an opaque register pair: h(w * g) is the ROUNDED product rounded again, like tcnn's `(T)(weight * grad)` and the oracle
    // -- the compiler's own choice, v_fma_mixlo_f16, rounds the exact product once and differs in ~2^-13 of the contributions)
    // Per pair of corners: v_pk_mul_f32 (x g), v_cvt_pk_f16_f32 (both h()), then the widening back to fp32 and the multiplication by the fixed-point unit as
    // ONE v_fma_mix_f32 per value (fp16 source operand: exact) -- six instructions where conversions + a packed multiply took eight.
    const auto fix2 = [&](f2 w, float gg) -> f2 {
        f2 pr = w * gg; asm volatile("" : "+v"(pr));
        const half2_t hv = { (half_t)pr.x, (half_t)pr.y }; const uint32_t hb = __builtin_bit_cast(uint32_t, hv);
        f2 r;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(hb), "s"(fs));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(hb), "s"(fs));
        return r;
    };      // contrib_fix of two corners, before the conversion to int
    if (MODE == kTileWhole64) {                                   // the whole level is this workgroup's: eight corners, nothing to test, one 64-bit atomic each
#pragma unroll
        for (int j = 0; j < 4; ++j) {                             // the two x-corners of pair j = y + 2 z
            const uint32_t t = HASHED ? (ay[j & 1] ^ az[j >> 1]) : (ay[j & 1] + az[j >> 1]);
            uint32_t idx[2] = { (HASHED ? (ax0 ^ t) : (ax0 + t)) & mask, (HASHED ? (ax1 ^ t) : (ax1 + t)) & mask };
            if (!POW2) { idx[0] -= (idx[0] >= size) ? size : 0u; idx[0] = min(idx[0], size - 1u); idx[1] -= (idx[1] >= size) ? size : 0u;
                idx[1] = min(idx[1], size - 1u); }
            const f2 w = (f2{ wx[0], wx[1] } * ((j & 1) ? wy2.y : wy2.x)) * wz[j >> 1];      // ((wx * wy) * wz): the reference walk's product order
            const f2 c0 = fix2(w, g0), c1 = fix2(w, g1);
            atomicAdd(tab64 + idx[0], pack_fix((int)c0.x, (int)c1.x)); atomicAdd(tab64 + idx[1], pack_fix((int)c0.y, (int)c1.y));
        }
        return;
    }
    uint32_t local[4]; float ws[4];                               // per pair: this workgroup's corner -- its place in the tile and its x-weight
    // the 32-bit tiles of the hashed levels (13 of base.json's 16): `local` holds LDS BYTE addresses
    constexpr bool BYTES = HASHED && POW2 && !BOTH;
    if constexpr (BYTES) {
        // the walk of the branch below with every index term carried DOUBLED (h2 = h << 1: xor / and commute with the shift, and only product bits below the
        // table size matter), so that a corner's place in the int32 tile comes out as its byte address -- (h >> 1) << 2 = h2 & mask4 -- and the four address
        // shifts in front of the atomics disappear (the tile starts at LDS address 0: the kernel has no static LDS, checked at its entry)
        const uint32_t my2 = my << 1, mz2 = mz << 1, mask4 = (mask << 1) & ~3u;
        uint32_t y2 = __umul24(pg[1], my2 & 0xffffffu), z2 = __umul24(pg[2], mz2 & 0xffffffu);
        // (kept as products: y2 + my2 is then one add with a scalar operand; the compiler's v_mad_u32_u24 needs the addend moved into a vector register first)
        asm volatile("" : "+v"(y2), "+v"(z2));
        const uint32_t ay2[2] = { y2, y2 + my2 }, az2[2] = { z2, z2 + mz2 };
        // ((x ^ (x + 1)) << 1 = 2x ^ (2x + 2); its bit 1 falls to mask4)
        const uint32_t a2 = ax0 << 1, axp2 = a2 ^ (parity << 1), dx4 = (a2 ^ (a2 + 2u)) & mask4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DEGEN && j) { local[j] = local[0]; ws[j] = ws[0]; continue; }
            const uint32_t h2 = xor3(axp2, ay2[j & 1], az2[j >> 1]);
            const uint32_t m = sign_of_bit1(h2);                   // bit 0 of the hash: which x-corner is this workgroup's (all ones: the second)
            local[j] = ((h2 & mask4) ^ (dx4 & m)) - (MODE == kTileParityRanged ? base_half * 4u : 0u);
            ws[j] = __builtin_bit_cast(float, bit_select(m, __builtin_bit_cast(uint32_t, wx[1]), __builtin_bit_cast(uint32_t, wx[0])));
        }
    } else if (HASHED && POW2) {
        // idx0 = (x ^ y' ^ z') & mask, idx1 = idx0 ^ dxm with dxm odd: with the tile's parity folded into x, bit 0 of h says which x-corner is this workgroup's
        // (m = all ones: the second), and the place in the parity half is (idx >> 1) = bits 1.. of h, xor-ed with dxm >> 1 for the second corner
        const uint32_t axp = ax0 ^ parity, dxh = ((ax0 ^ ax1) & mask) >> 1, nb = (uint32_t)__popc(mask) - 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DEGEN && j) { local[j] = local[0]; ws[j] = ws[0]; continue; }      // (the index ignores y and z: one entry, one x-corner for all four pairs)
            const uint32_t h = xor3(axp, ay[j & 1], az[j >> 1]);
            const uint32_t m = sign_of_bit0(h);
            local[j] = (__builtin_amdgcn_ubfe(h, 1u, nb) ^ (dxh & m)) - (MODE == kTileParityRanged ? base_half : 0u);
            ws[j] = __builtin_bit_cast(float, bit_select(m, __builtin_bit_cast(uint32_t, wx[1]), __builtin_bit_cast(uint32_t, wx[0])));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = HASHED ? (ay[j & 1] ^ az[j >> 1]) : (ay[j & 1] + az[j >> 1]);
            uint32_t idx0 = (HASHED ? (ax0 ^ t) : (ax0 + t)) & mask, idx1 = (HASHED ? (ax1 ^ t) : (ax1 + t)) & mask;
            if (!POW2) { idx0 -= (idx0 >= size) ? size : 0u; idx0 = min(idx0, size - 1u); idx1 -= (idx1 >= size) ? size : 0u; idx1 = min(idx1, size - 1u); }
            const bool second = ((idx0 ^ parity) & 1u) != 0u;    // which corner of the pair is this workgroup's
            local[j] = ((second ? idx1 : idx0) >> 1) - (MODE == kTileParityRanged ? base_half : 0u);
            ws[j] = second ? wx[1] : wx[0];
        }
    }
    const f2 w01 = (f2{ ws[0], ws[1] } * wy2) * wz[0], w23 = (f2{ ws[2], ws[3] } * wy2) * wz[1];      // ((wx * wy) * wz): the reference walk's product order
    if (MODE == kTileParity64) {
        const f2 a0 = fix2(w01, g0), a1 = fix2(w01, g1), b0 = fix2(w23, g0), b1 = fix2(w23, g1);
        atomicAdd(tab64 + local[0], pack_fix((int)a0.x, (int)a1.x)); atomicAdd(tab64 + local[1], pack_fix((int)a0.y, (int)a1.y));
        atomicAdd(tab64 + local[2], pack_fix((int)b0.x, (int)b1.x)); atomicAdd(tab64 + local[3], pack_fix((int)b0.y, (int)b1.y));
        return;
    }
    const f2 a = fix2(w01, g), b = fix2(w23, g);
    const int c[4] = { (int)a.x, (int)a.y, (int)b.x, (int)b.y };
    typedef __attribute__((address_space(3))) int* lds_int;
    const auto add = [&](uint32_t where, int v) {
        if constexpr (BYTES) (void)__hip_atomic_fetch_add((lds_int)(uintptr_t)where, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAdd(tab + where, v);
    };
    if (DEGEN) { add(local[3], (c[0] + c[1]) + (c[2] + c[3])); return; }      // all four pairs are the SAME entry: one atomic for the (exact) sum
#pragma unroll
    for (int j = 0; j < 4; ++j) if (MODE == kTileParity || local[j] < tile * (BYTES ? 4u : 1u)) add(local[j], c[j]);
}

template <bool HASHED, bool POW2, int MODE, bool DEGEN = false>
__device__ __forceinline__ void scatter_samples(int* tab, const half2_t* __restrict__ de, const float4_t* __restrict__ x4,
                                                uint32_t cnt_lo, uint32_t cnt_hi /* lane b: run length of ray bin b / b + 64 */,
                                                uint32_t n_bins, uint32_t bin0, uint32_t bin_step, uint32_t bin_cap, uint32_t feature,
                                                float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask, uint32_t parity, uint32_t base_half,
                                                        uint32_t tile, float fs) {
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
    // b uniform
    const auto count_of = [&](uint32_t b) {
        return (uint32_t)((b < 64u) ? __builtin_amdgcn_readlane((int)cnt_lo, (int)b) : __builtin_amdgcn_readlane((int)cnt_hi, (int)(b - 64u))); };
    uint32_t nb = 0, width = 0;
    for (uint32_t b = bin0; b < n_bins; b += bin_step) { ++nb; width = max(width, count_of(b)); }
    if (width == 0u) return;
    // threads per bin and step: W2 = 2^w2s = the run length rounded up to a power of two, one wave at least
    uint32_t w2s = 6u; while ((1u << w2s) < blockDim.x && (1u << w2s) < width) ++w2s;
    // (1024 threads: G = 1024 / W2 groups; powers of two throughout, no divisions)
    const uint32_t W2 = 1u << w2s, gs = 10u - w2s, G = 1u << gs;
    // Dense levels: the rows of a run are a ray's samples in order, and neighbours along a ray sit in the same coarse cell -- the lanes of a wave would add
    // into the same few entries, and the LDS serialises same-address atomics (ds_add_u64: 19 cycles per wave instruction on random addresses, 46 when four
    // lanes share one: tools/ldsatomicbench.py).  There a wave takes GROUPS of kGroup consecutive rows (still 16 * kGroup contiguous bytes per group) from
    // runs W2 / 16 rows apart; hashed levels scramble the addresses themselves and keep the contiguous rows.
    constexpr uint32_t kGroupBits = MON_V_SGROUP;
    const uint32_t wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> w2s)), lane_c = threadIdx.x & (W2 - 1u);
    const uint32_t lane_o = HASHED ? lane_c
            : ((((lane_c & 63u) >> kGroupBits) << (w2s - 6u + kGroupBits)) | ((lane_c >> 6) << kGroupBits) | (lane_c & ((1u << kGroupBits) - 1u)));
    const uint32_t ksteps = (nb + G - 1u) >> gs, rounds = (width + W2 - 1u) >> w2s, n_steps = rounds * ksteps;
    uint32_t fks = 0, fo = lane_o, fs_left = n_steps;                                   // running state of the step the next fetch serves (all but fo uniform)
    const auto fetch = [&](ScatterItem& it, bool& valid) {
        const uint32_t k = (fks << gs) + wg, b = min(bin0 + k * bin_step, n_bins - 1u);
        const uint32_t cnt = (fs_left && k < nb) ? count_of(b) : 0u;
        valid = fo < cnt; const uint32_t sc = b * bin_cap + (valid ? fo : 0u);
        it.g = de[sc]; it.x = x4[sc];
        fs_left -= fs_left ? 1u : 0u;
        // (selects, not branches: the compiler turned conditional updates of the captured state into scratch memory)
        const bool wrap = fks + 1u == ksteps;
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
        for (int u = 0; u < kBatch; ++u) scatter_item<HASHED, POW2, MODE,
                DEGEN>(tab, cur[u], cv[u], feature, scale, size, my, mz, mask, parity, base_half, tile, fs);
    }
}

// The weight-gradient partial rows of k_fused_train (one per workgroup) are summed here as well: every scatter workgroup
// takes a few float4 column groups (128 row subsets x 8 groups per pass).  The loads are issued at kernel entry and the sums
// are finished (DPP + a small LDS exchange) after the tile has been written, so their latency hides behind the scatter itself
// (k_reduce_partials remains for networks whose levels all go through global atomics).
// rows in accumulator layout: n_cols = acc_cols(fd), loss partial behind them
struct PartialsArgs { const float* partials; uint32_t n_partials, stride, n_cols; FragDims fd; float* gmlp; DevState* st; };
constexpr uint32_t kPartialsMaxPasses = 2;          // column-group passes a workgroup may hold in registers (n_mlp + 1 <= 2 * 8 * 4 * gridDim.x)

// column groups (of 4 columns) a workgroup sums per pass: as few as cover all groups with the whole grid (1, 2, 4 or 8), so that every workgroup
// carries the same small share instead of the first third of the grid carrying everything
__device__ __forceinline__ uint32_t partials_groups(const PartialsArgs& pa) {
    const uint32_t n4 = (pa.n_cols + 1u + 3u) / 4u, need = (n4 + gridDim.x - 1u) / gridDim.x;
    return need <= 1u ? 1u : (need <= 2u ? 2u : (need <= 4u ? 4u : 8u));
}
// the column groups go to the LAST workgroups of the grid: the first ones hold the coarse dense levels, whose sample walk is the longest of the kernel (their
// samples collide in the LDS atomic unit), so the row sums ride on workgroups that have slack
// (on the FIRST workgroups instead: 42.6 against 41.5 us, round 2)
__device__ __forceinline__ uint32_t partials_block() { return gridDim.x - 1u - blockIdx.x; }
__device__ __forceinline__ void partials_prefetch(const PartialsArgs& pa, float4_t (&acc)[kPartialsMaxPasses]) {
    // thread = (column group gs of G, row subset sub of 1024 / G)
    const uint32_t n4 = (pa.n_cols + 1u + 3u) / 4u, G = partials_groups(pa), subs = blockDim.x / G, gs = threadIdx.x / subs, sub = threadIdx.x - gs * subs;
#pragma unroll
    for (uint32_t ps = 0; ps < kPartialsMaxPasses; ++ps) {
        const uint32_t g = (partials_block() + ps * gridDim.x) * G + gs; acc[ps] = float4_t{ 0.f, 0.f, 0.f, 0.f };
        // four independent 16-byte loads per round (rows are padded to n_cols + 64 floats)
        if (g < n4) for (uint32_t k0 = sub; k0 < pa.n_partials; k0 += 4u * subs) {
            float4_t v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) { const uint32_t k = k0 + subs * u;
                v[u] = (k < pa.n_partials) ? *reinterpret_cast<const float4_t*>(pa.partials + (size_t)k * pa.stride + 4u * g) : float4_t{ 0.f, 0.f, 0.f,
                    0.f }; }
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
            if (gg < n4) { if (pi < pa.n_cols) { const int prm = acc_param(pa.fd, (int)pi); if (prm >= 0) pa.gmlp[prm] = v;
                    } else if (pi == pa.n_cols) pa.st->loss_sum = v; }
        }
        __syncthreads();
    }
}

#ifndef MON_HOUSEKEEPING_BLOCK
#define MON_HOUSEKEEPING_BLOCK (gridDim.x - 1u)
#endif
__global__ void __launch_bounds__(1024) k_grid_scatter(LevelFast lt, ScatterLevels sl, const half2_t* __restrict__ de_soa, const float4_t* __restrict__ x4,
                                                       uint32_t B, uint32_t n_bins, half_t* __restrict__ gpart, uint32_t n_entries,
                                                       const DevState* __restrict__ st, DevState* st_rw, DevState* st_next, PartialsArgs pa,
                                                       float* __restrict__ timing) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t iter = st->iter;
    // (the last workgroup: the first ones hold the coarse dense levels, the kernel's critical path)
    if (blockIdx.x == MON_HOUSEKEEPING_BLOCK && threadIdx.x < 64u) {
        // slot-counter housekeeping (also for a skipped batch): clear the counters k_fused_train of the NEXT iteration counts in -- they live in the other
        // DevState, which nobody reads during this iteration -- and note how many samples carried a gradient in this one (k_optimizer hands it to the next
        // iteration as n_scatter_last; the large-table path decides on it)
        uint32_t v = 0u;
        for (uint32_t b = threadIdx.x; b < n_bins; b += 64u) { v += st->n_scatter[scatter_counter(iter, b)];
            st_next->n_scatter[scatter_counter(iter + 1u, b)] = 0u; }
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
    bool pacc_loaded = false;
    int* tab = reinterpret_cast<int*>(smem);
    // scatter_item addresses the hashed levels' tiles from LDS offset 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();
    float* red = reinterpret_cast<float*>(smem + (size_t)kScatterLdsBytes - 256u);     // 256 B behind the largest tile
    // run lengths of the compacted ray bins, lane b of every wave holds bin b's and bin (b + 64)'s (read back with v_readlane: no memory access in the sample
    // loop)
    const uint32_t bin_cap = B / n_bins, lb = threadIdx.x & 63u;
    // (both sets are requested and the iteration's one is picked afterwards: the address must not wait for the load of the iteration counter)
    const uint32_t c_lo0 = (lb < n_bins) ? st->n_scatter[scatter_counter(0u, lb)] : 0u, c_lo1 = (lb < n_bins) ? st->n_scatter[scatter_counter(1u, lb)] : 0u;
    const uint32_t c_hi0 = (lb + 64u < n_bins) ? st->n_scatter[scatter_counter(0u, lb + 64u)] : 0u, c_hi1 = (lb + 64u < n_bins)
            ? st->n_scatter[scatter_counter(1u, lb + 64u)] : 0u;
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
    // (level sizes are multiples of 8) parity tiles: idx = 2 * (base_half + local) + parity
    const uint32_t half_size = size >> 1, base_half = mode == kTileParityRanged ? (part >> 2) * kScatterTile : 0u;
    // the index ignores y and z (tcnn's stride wrap-around at res = 65 536): the four pairs of a sample are one entry
    const bool degenerate = pow2 && (my & mask) == 0u && (mz & mask) == 0u && size > 1u;
    MON_ST_STAMP();
    // (levels whose part count does not divide 16 leave workgroups without a tile)
    if (mode != kTileParityRanged || base_half < half_size) {
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
#define MON_SCATTER_CALL(H, PW, MD, ...) \
    scatter_samples<H, PW, MD, ##__VA_ARGS__>(tab, de, x4, cnt_lo, cnt_hi, n_bins, p, P, bin_cap, feature, scale, size, my, mz, mask, parity, base_half, tile, \
                                              fs)
#define MON_SCATTER_MODE(MD) do { \
        if (hashed) { if (pow2) MON_SCATTER_CALL(true, true, MD); else MON_SCATTER_CALL(true, false, MD); } \
        else { if (pow2) MON_SCATTER_CALL(false, true, MD); else MON_SCATTER_CALL(false, false, MD); } } while (0)
        if (mode == kTileWhole64) MON_SCATTER_MODE(kTileWhole64);
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
        const size_t plane = n_entries >> 1;                                           // partial table p, plane (feature, parity): entry idx at [idx >> 1]
        half_t* pl = gpart + ((size_t)p * 4u) * plane + (off >> 1);
        // undo pack_fix
        const auto lo_hi = [&](int lo_bits, int hi_bits, float& f0, float& f1) { f0 = (float)lo_bits * inv; f1 = (float)(hi_bits - (lo_bits >> 31)) * inv; };
        // entries 2k, 2k + 1 interleaved, both features: 4 entries (32 B) per thread and pass -> 2 halves into each of the four planes
        if (mode == kTileWhole64) {
            for (uint32_t i = threadIdx.x; i < tile / 4u; i += blockDim.x) {
                const int4v a = t4[2u * i], c = t4[2u * i + 1u];                        // entries 4i, 4i+1 | 4i+2, 4i+3
                float e0f0, e0f1, e1f0, e1f1, e2f0, e2f1, e3f0, e3f1; lo_hi(a[0], a[1], e0f0, e0f1); lo_hi(a[2], a[3], e1f0, e1f1);
                lo_hi(c[0], c[1], e2f0, e2f1); lo_hi(c[2], c[3], e3f0, e3f1);
                *reinterpret_cast<half2_t*>(pl + 0u * plane + 2u * i) = half2_t{ (half_t)e0f0, (half_t)e2f0 };      // feature 0, even entries
                *reinterpret_cast<half2_t*>(pl + 1u * plane + 2u * i) = half2_t{ (half_t)e1f0, (half_t)e3f0 };      // feature 0, odd
                *reinterpret_cast<half2_t*>(pl + 2u * plane + 2u * i) = half2_t{ (half_t)e0f1, (half_t)e2f1 };      // feature 1, even
                *reinterpret_cast<half2_t*>(pl + 3u * plane + 2u * i) = half2_t{ (half_t)e1f1, (half_t)e3f1 };      // feature 1, odd
            }
        // one parity, both features: 4 entries (32 B) per thread and pass -> 4 halves into each of the two feature planes
        } else if (mode == kTileParity64) {
            for (uint32_t i = threadIdx.x; i < tile / 4u; i += blockDim.x) {
                const int4v a = t4[2u * i], c = t4[2u * i + 1u];
                float f0[4], f1[4]; lo_hi(a[0], a[1], f0[0], f1[0]); lo_hi(a[2], a[3], f0[1], f1[1]); lo_hi(c[0], c[1], f0[2], f1[2]);
                lo_hi(c[2], c[3], f0[3], f1[3]);
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
                *reinterpret_cast<half4_t*>(dst + (tile & ~7u)) = half4_t{ (half_t)((float)a0[0] * inv), (half_t)((float)a0[1] * inv),
                        (half_t)((float)a0[2] * inv), (half_t)((float)a0[3] * inv) };
            }
        }
    }
    MON_ST_STAMP();
    if (pa.partials && !pacc_loaded) partials_prefetch(pa, pacc);                     // (a workgroup without a tile)
    if (pa.partials) partials_finish(pa, pacc, red);
    MON_ST_STAMP();
#ifdef MON_SCATTER_TIMING
    if (timing && (threadIdx.x & 63u) == 0u) { float* o = timing + ((size_t)blockIdx.x * 16u + (threadIdx.x >> 6)) * 8u;
        for (int k = 0; k + 1 < tn && k < 6; ++k) o[k] = (float)(tq[k + 1] - tq[k]); o[6] = (float)(tq[0] & 0xffffff); o[7] = (float)level; }
#endif
}

// Layer-at-a-time backend (network shapes the fused kernel does not take: 16 neurons, 2 x 128, three / four hidden layers): its dL/dE rows [B][Epad] and positions
// [B][3] in the hand-over layout k_grid_scatter walks -- EVERY sample at its natural slot (bin = ray & (bins - 1), slot (ray / bins) * S + n, like
// k_fused_train with keep_zero_samples), every bin counter = its capacity -- so that the grid backward of those shapes is the same exact LDS accumulation
// instead of tcnn's 16.8 M global packed-f16 atomics (k_grid_backward: 902 of the 1100 us such a step took).
__global__ void __launch_bounds__(256) k_rows_to_bins(const half_t* __restrict__ dE, const float* __restrict__ pts, uint32_t Epad, int L, uint32_t R, uint32_t S,
                                                      uint32_t n_bins, float clampv, half2_t* __restrict__ de_soa, float4_t* __restrict__ x4, DevState* st) {
    if (st->n_valid == 0u) return;
    const uint32_t B = R * S, s = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0u && threadIdx.x < n_bins) st->n_scatter[scatter_counter(st->iter, threadIdx.x)] = B / n_bins;
    if (s >= B) return;
    const uint32_t ray = s / S, n = s - ray * S, bin = ray & (n_bins - 1u), slot = bin * (B / n_bins) + (ray / n_bins) * S + n;
    x4[slot] = float4_t{ pts[3 * (size_t)s], pts[3 * (size_t)s + 1], pts[3 * (size_t)s + 2], 0.f };
    const half2_t* row = reinterpret_cast<const half2_t*>(dE + (size_t)s * Epad);
    for (int l = 0; l < L; ++l) { const half2_t g = row[l];
        de_soa[(size_t)l * B + slot] = half2_t{ (half_t)clamp_f((float)g.x, -clampv, clampv), (half_t)clamp_f((float)g.y, -clampv, clampv) }; }
}
void launch_rows_to_bins(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* dE, const float* pts, uint32_t R, uint32_t S, uint32_t n_bins,
        uint16_t* de_soa, float* x_soa, DevState* st) {
    hipLaunchKernelGGL(k_rows_to_bins, dim3((R * S + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const half_t*>(dE), pts, (uint32_t)nd.Epad, nd.L, R, S, n_bins,
            lf.fix_clamp, reinterpret_cast<half2_t*>(de_soa), reinterpret_cast<float4_t*>(x_soa), st);
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
void launch_grid_scatter(hipStream_t s, const LevelTable& lt, const LevelFast& lf, const NetDims& nd, const uint16_t* de_soa, const float* x_soa, uint32_t B,
        uint32_t n_bins, uint16_t* gpart, uint32_t part_stride_entries, DevState* st,
                         const float* partials, uint32_t n_partials, float* gmlp, DevState* st_next) {
    ScatterLevels sl; if (!scatter_plan(lt, nd, sl)) return;
    const PartialsArgs pa{ partials, n_partials, fused_partial_cols(nd) + 64u, fused_partial_cols(nd), FragDims{ nd.Epad, nd.W, nd.NH, nd.L }, gmlp, st };
    constexpr uint32_t smem = kScatterLdsBytes;
    static std::atomic<uint64_t> attr_devices{ 0 }; static std::mutex attr_mu;
    once_per_device(attr_devices, attr_mu,
            [] { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
    float* timing = nullptr;
#ifdef MON_SCATTER_TIMING
    static float* g_timing = nullptr; if (!g_timing) hipMalloc((void**)&g_timing, 256 * 16 * 8 * 4); timing = g_timing; g_scatter_timing_buf = g_timing;
#endif
    hipLaunchKernelGGL(k_grid_scatter, dim3(sl.n_levels * kScatterWgPerLevel), dim3(1024), smem, s, lf, sl, reinterpret_cast<const half2_t*>(de_soa),
            reinterpret_cast<const float4_t*>(x_soa), B, n_bins,
                       reinterpret_cast<half_t*>(gpart), part_stride_entries, st, st, st_next, pa, timing);
}
#ifdef MON_SCATTER_TIMING
extern "C" int mon_debug_scatter_timing(float* out) { hipDeviceSynchronize();
    return g_scatter_timing_buf ? (int)hipMemcpy(out, g_scatter_timing_buf, 256 * 16 * 8 * 4, hipMemcpyDeviceToHost) : -1; }
#endif

}  // namespace mon
