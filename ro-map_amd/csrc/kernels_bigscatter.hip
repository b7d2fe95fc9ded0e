// kernels_bigscatter.hip -- grid backward for levels too large for k_grid_scatter's 16-workgroup plan (more than 2^18 entries, e.g. the
// T = 2^22 stress configuration), as long as many samples carry a gradient.
//
// tcnn scatters with atomicAdd(__half2) (call site CORE/src/nerf_model.cu:1604; 128 per sample); on MI355X global packed-f16 atomics
// sustain ~21 G/s (profiles/r01_microbench.md), i.e. ~0.65 ms for the 13 fine levels of a 131 072-sample batch.  Here the contributions
// are BINNED BY TILE first -- a two-pass counting sort over 8 192-entry tiles, all counters in LDS -- and every tile is then
// accumulated exactly in LDS in int32 fixed point like k_grid_scatter does, by one workgroup that owns those entries, and written to the
// fp16 gradient table with plain stores (the table the optimizer already reads and clears).  No global atomics on the data path; the
// result is the exact sum of tcnn's fp16 contributions rounded once, independent of execution order.
//   k_big_hist   (level, ray bin): per-tile contribution counts of the bin's gradient-carrying samples
//   k_big_scan   (level):          tile offsets + per-(bin, tile) write offsets
//   k_big_emit   (level, ray bin): records {index in level, h(w * dE) as half2} into the tile bins
//   k_big_accum  (level, tile):    LDS accumulation, non-zero entries -> gradient table
// Late in training (a few thousand samples) the per-tile fixed work outweighs ~10 us of atomics: every kernel here, and k_fused_train,
// reads the previous iteration's gradient-carrying sample count from DevState and takes the same wave-uniform decision
// (big_levels_binned, grid_walk.h) -- binned above `big_switch` samples, tcnn's global atomics inside k_fused_train below.  Once the
// host has seen the count well below the switch it stops launching these kernels (model.cpp).  k_big_accum also sets the lazy
// optimizer's chunk flags (ParamPtrs::touched) next to every entry it writes.
#include <atomic>
#include <mutex>
#include "model.h"
#include "grid_walk.h"

namespace mon {

// 64 KB of LDS per tile (two workgroups per CU); up to 2^24 entries per level
constexpr uint32_t kBigTile = 8192, kBigTileShift = 13, kBigBins = 16, kBigMaxTiles = 2048;

struct BigLevels { int n; int level[kMaxLevels]; uint32_t tiles[kMaxLevels]; uint32_t tile_base[kMaxLevels + 1]; };

// samples of bin group g (of kBigBins): the ray bins g, g + kBigBins, ... < n_bins, each a compacted run of st->n_scatter[b] samples at b * (B / n_bins)
template <class F>
__device__ __forceinline__ void for_bin_samples(const LevelFast& lf, int level, const half2_t* __restrict__ de, const float* __restrict__ x_soa, uint32_t B,
        uint32_t n_bins, uint32_t g,
                                                const DevState* __restrict__ st, F&& f) {
  const uint32_t cap = B / n_bins;
  for (uint32_t b = g; b < n_bins; b += kBigBins) {
    const uint32_t cnt = min(st->n_scatter[scatter_counter(st->iter, b)], cap), s0 = b * cap;
    for (uint32_t s = s0 + threadIdx.x; s < s0 + cnt; s += blockDim.x) {
        const half2_t g = de[s];
        float g0 = (float)g.x, g1 = (float)g.y;
        if (g0 == 0.f && g1 == 0.f) continue;
        g0 = clamp_f(g0, -lf.fix_clamp, lf.fix_clamp); g1 = clamp_f(g1, -lf.fix_clamp, lf.fix_clamp);
        const float4_t xv = reinterpret_cast<const float4_t*>(x_soa)[s]; const float x[3] = { xv[0], xv[1], xv[2] };
        level_corners(lf, level, x, [&](int, uint32_t idx, float w) {
            const half2_t c = { (half_t)(w * g0), (half_t)(w * g1) };                      // tcnn: (T)(weight * grad)
            const uint32_t bits = __builtin_bit_cast(uint32_t, c);
            if (bits & 0x7fff7fffu) f(idx, bits);
        });
    }
  }
}

__global__ void __launch_bounds__(1024) k_big_hist(LevelFast lf, BigLevels big, const half2_t* __restrict__ de_soa, const float* __restrict__ x_soa,
        uint32_t B, uint32_t n_bins,
                                                  const DevState* __restrict__ st, uint32_t big_switch, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[kBigMaxTiles];
    if (st->n_valid == 0u || !big_levels_binned(st->n_scatter_last, big_switch)) return;
    const uint32_t bl = blockIdx.x / kBigBins, b = blockIdx.x - bl * kBigBins; const int level = big.level[bl];
    for (uint32_t i = threadIdx.x; i < big.tiles[bl]; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    for_bin_samples(lf, level, de_soa + (size_t)level * B, x_soa, B, n_bins, b, st, [&](uint32_t idx, uint32_t) { atomicAdd(&h[idx >> kBigTileShift], 1u); });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < big.tiles[bl]; i += blockDim.x) hist[(size_t)blockIdx.x * kBigMaxTiles + i] = h[i];
}

// one workgroup per level: tile totals, exclusive scan over tiles, then per-bin write offsets
__global__ void __launch_bounds__(1024) k_big_scan(BigLevels big, const DevState* __restrict__ st, uint32_t big_switch, const uint32_t* __restrict__ hist,
        uint32_t* __restrict__ woff,
                                                   uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ tile_off) {
    __shared__ uint32_t part[1024];
    if (st->n_valid == 0u || !big_levels_binned(st->n_scatter_last, big_switch)) return;
    const uint32_t bl = blockIdx.x, t = threadIdx.x, nt = big.tiles[bl];
    uint32_t tot[2] = { 0u, 0u };                                                   // two consecutive tiles per thread
    for (uint32_t j = 0; j < 2u; ++j) if (2u * t + j < nt) for (uint32_t b = 0; b < kBigBins; ++b) tot[j] += hist[((size_t)bl * kBigBins + b) * kBigMaxTiles
            + 2u * t + j];
    part[t] = tot[0] + tot[1]; __syncthreads();
    // inclusive scan
    for (uint32_t d = 1; d < 1024u; d <<= 1) { const uint32_t v = (t >= d) ? part[t - d] : 0u; __syncthreads(); part[t] += v; __syncthreads(); }
    uint32_t off = part[t] - tot[0] - tot[1];
    for (uint32_t j = 0; j < 2u; ++j) {
        const uint32_t tile = 2u * t + j; if (tile >= nt) break;
        tile_cnt[(size_t)bl * kBigMaxTiles + tile] = tot[j]; tile_off[(size_t)bl * kBigMaxTiles + tile] = off;
        for (uint32_t b = 0; b < kBigBins; ++b) { woff[((size_t)bl * kBigBins + b) * kBigMaxTiles + tile] = off;
            off += hist[((size_t)bl * kBigBins + b) * kBigMaxTiles + tile]; }
    }
}

__global__ void __launch_bounds__(1024) k_big_emit(LevelFast lf, BigLevels big, const half2_t* __restrict__ de_soa, const float* __restrict__ x_soa,
        uint32_t B, uint32_t n_bins,
                                                  const DevState* __restrict__ st, uint32_t big_switch, const uint32_t* __restrict__ woff,
                                                          uint2* __restrict__ rec) {
    __shared__ uint32_t cur[kBigMaxTiles];
    if (st->n_valid == 0u || !big_levels_binned(st->n_scatter_last, big_switch)) return;
    const uint32_t bl = blockIdx.x / kBigBins, b = blockIdx.x - bl * kBigBins; const int level = big.level[bl];
    for (uint32_t i = threadIdx.x; i < big.tiles[bl]; i += blockDim.x) cur[i] = woff[(size_t)blockIdx.x * kBigMaxTiles + i];
    __syncthreads();
    uint2* out = rec + (size_t)bl * 8u * B;                                         // a level holds at most 8 contributions per sample
    for_bin_samples(lf, level, de_soa + (size_t)level * B, x_soa, B, n_bins, b, st, [&](uint32_t idx, uint32_t bits) {
        const uint32_t slot = atomicAdd(&cur[idx >> kBigTileShift], 1u);            // order inside a bin segment is arbitrary: the accumulation below is exact
        out[slot] = make_uint2(idx, bits);
    });
}

__global__ void __launch_bounds__(1024) k_big_accum(LevelFast lf, BigLevels big, const DevState* __restrict__ st, uint32_t big_switch,
        const uint32_t* __restrict__ tile_cnt,
                                                    const uint32_t* __restrict__ tile_off, const uint2* __restrict__ rec, uint32_t B,
                                                            uint32_t* __restrict__ ggrid_h2, uint8_t* __restrict__ touched_grid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (st->n_valid == 0u || !big_levels_binned(st->n_scatter_last, big_switch)) return;
    int bl = 0;
    while (bl + 1 < big.n && blockIdx.x >= big.tile_base[bl + 1]) ++bl;
    const uint32_t t = blockIdx.x - big.tile_base[bl]; const int level = big.level[bl];
    const uint32_t cnt = tile_cnt[(size_t)bl * kBigMaxTiles + t];
    if (cnt == 0u) return;                                                          // nothing landed in this tile: the table keeps its zeros
    int* tab = reinterpret_cast<int*>(smem);
    const uint32_t size = lf.size[level], base = t << kBigTileShift, tile = min(kBigTile, size - base);
    { typedef int int4v __attribute__((ext_vector_type(4))); int4v* t4 = reinterpret_cast<int4v*>(tab);
      for (uint32_t i = threadIdx.x; i < (tile + 1u) / 2u; i += blockDim.x) t4[i] = int4v{ 0, 0, 0, 0 }; }
    __syncthreads();
    const uint2* in = rec + (size_t)bl * 8u * B + tile_off[(size_t)bl * kBigMaxTiles + t];
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const uint2 r = in[i]; const uint32_t local = r.x - base; const half2_t c = __builtin_bit_cast(half2_t, r.y);
        // exact for loss_scale <= 128: every fp16 value is a multiple of 2^-24
        const int f0 = (int)((float)c.x * lf.fix_scale), f1 = (int)((float)c.y * lf.fix_scale);
        if (f0) atomicAdd(tab + 2u * local, f0);
        if (f1) atomicAdd(tab + 2u * local + 1u, f1);
    }
    __syncthreads();
    uint32_t* dst = ggrid_h2 + lf.offset[level] + base;
    for (uint32_t i = threadIdx.x; i < tile; i += blockDim.x) {
        const int a0 = tab[2u * i], a1 = tab[2u * i + 1u];
        if (a0 | a1) {
            dst[i] = __builtin_bit_cast(uint32_t, half2_t{ (half_t)((float)a0 * (1.0f / lf.fix_scale)), (half_t)((float)a1 * (1.0f / lf.fix_scale)) });
            if (touched_grid) touched_grid[(lf.offset[level] + base + i) >> 2] = 1;          // the optimizer's chunk flag (4 entries = 8 parameters)
        }
    }
}

// Host: the levels k_grid_scatter's plan leaves out.  Returns their count.
static int big_levels_plan(const LevelTable& lt, const NetDims& nd, uint32_t lds_mask, BigLevels& big) {
    big = BigLevels{}; uint32_t base = 0;
    for (int l = 0; l < nd.L; ++l) {
        if ((lds_mask >> l) & 1u) continue;
        const uint32_t size = lt.offset[l + 1] - lt.offset[l], tiles = (size + kBigTile - 1) / kBigTile;
        if (tiles > kBigMaxTiles) return -1;
        big.level[big.n] = l; big.tiles[big.n] = tiles; big.tile_base[big.n] = base; base += tiles; ++big.n;
    }
    big.tile_base[big.n] = base;
    return big.n;
}
// Bytes of workspace launch_big_scatter needs; 0 = no level qualifies (or one is larger than 2^24 entries: atomics stay).
// hist + woff + tile_cnt + tile_off, then the records
size_t big_scatter_workspace_bytes(const LevelTable& lt, const NetDims& nd, uint32_t lds_mask, uint32_t B) {
    BigLevels big; const int n_big = big_levels_plan(lt, nd, lds_mask, big); if (n_big <= 0) return 0;
    return (size_t)n_big * (2 * kBigBins + 2) * kBigMaxTiles * 4 + (size_t)n_big * 8u * B * 8u;
}
void launch_big_scatter(hipStream_t s, const LevelTable& lt, const LevelFast& lf, const NetDims& nd, uint32_t lds_mask, const uint16_t* de_soa,
        const float* x_soa, uint32_t B,
                        uint32_t n_bins, const DevState* st, uint32_t big_switch, void* workspace, uint16_t* ggrid, uint8_t* touched_grid) {
    BigLevels big; if (big_levels_plan(lt, nd, lds_mask, big) <= 0) return;
    uint32_t* hist = reinterpret_cast<uint32_t*>(workspace); uint32_t* woff = hist + (size_t)big.n * kBigBins * kBigMaxTiles;
    uint32_t* tcnt = woff + (size_t)big.n * kBigBins * kBigMaxTiles; uint32_t* toff = tcnt + (size_t)big.n * kBigMaxTiles;
    uint2* rec = reinterpret_cast<uint2*>(toff + (size_t)big.n * kBigMaxTiles);
    // once per device, and no launch before it has run (the flag is set AFTER the attribute call, under the lock: objects of one device launch from several
    // host threads)
    {
        static std::atomic<uint64_t> attr_devices{ 0 }; static std::mutex attr_mu;
        int dev = 0; (void)hipGetDevice(&dev); const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
            std::lock_guard<std::mutex> l(attr_mu);
            if (!(attr_devices.load(std::memory_order_relaxed) & bit)) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big_accum), hipFuncAttributeMaxDynamicSharedMemorySize, kBigTile * 8);
                attr_devices.fetch_or(bit, std::memory_order_release); }
        }
    }
    const half2_t* de = reinterpret_cast<const half2_t*>(de_soa);
    hipLaunchKernelGGL(k_big_hist, dim3(big.n * kBigBins), dim3(1024), 0, s, lf, big, de, x_soa, B, n_bins, st, big_switch, hist);
    hipLaunchKernelGGL(k_big_scan, dim3(big.n), dim3(1024), 0, s, big, st, big_switch, hist, woff, tcnt, toff);
    hipLaunchKernelGGL(k_big_emit, dim3(big.n * kBigBins), dim3(1024), 0, s, lf, big, de, x_soa, B, n_bins, st, big_switch, woff, rec);
    hipLaunchKernelGGL(k_big_accum, dim3(big.tile_base[big.n]), dim3(1024), kBigTile * 8, s, lf, big, st, big_switch, tcnt, toff, rec, B,
            reinterpret_cast<uint32_t*>(ggrid), touched_grid);
}

}  // namespace mon
