// grid_walk.h -- the 8-corner walk of one hash-grid level, shared by the fused kernels and the large-table scatter.
#pragma once
#include "device_common.h"

namespace mon {

// Large-table gradient path of the current iteration (kernels_bigscatter.hip): binned while many samples carry a gradient -- judged by
// the previous iteration's count, which k_optimizer's last block wrote -- and global atomics once only a few thousand do.  0 = the
// first iteration or a skipped batch: assume a full one.
__host__ __device__ inline bool big_levels_binned(uint32_t n_scatter_last, uint32_t big_switch) { return n_scatter_last == 0u || n_scatter_last > big_switch; }

typedef LevelFast LevelLds;      // the same constants, copied into LDS once per workgroup by the fused kernels

// Corner walk of one level for one position: calls f(k, index_within_level, weight) for the 8 corners.
// Same arithmetic as tcnn's grid_index / grid_hash (weights multiply in x, y, z order; see k_encode), restructured
// for instruction count -- this kernel is VALU-issue bound, not gather bound:
//   * tcnn's table sizes leave two cases only: dense (size = round_up(res^3, 8) >= res^3, linear index, can exceed
//     size only by the +1 boundary corner, so `% size` is one conditional subtract) and hashed (size = 2^T, `%` is a mask);
//   * the 8 corners share the per-axis terms (2 integer multiplies per level instead of 16), both index forms are
//     computed and selected per lane (the two half-waves may sit on a dense and a hashed level at the same time).
template <class F>
__device__ __forceinline__ void level_corners(const LevelLds& lt, int level, const float x[3], F&& f) {
    const float scale = lt.scale[level];
    const uint32_t size = lt.size[level], my = lt.my[level], mz = lt.mz[level], mask = lt.mask[level];
    const bool hashed = lt.hashed[level] != 0u;
    float pos[3]; uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float p = fmaf(scale, x[d], 0.5f), fl = floorf(p); pg[d] = (uint32_t)(int32_t)fl; pos[d] = p - fl; }
    const uint32_t ax[2] = { pg[0], pg[0] + 1u };
    const uint32_t y0 = pg[1] * my, z0 = pg[2] * mz;
    const uint32_t ay[2] = { y0, y0 + my }, az[2] = { z0, z0 + mz };
    const float wx[2] = { 1.f - pos[0], pos[0] }, wy[2] = { 1.f - pos[1], pos[1] }, wz[2] = { 1.f - pos[2], pos[2] };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ix = k & 1, iy = (k >> 1) & 1, iz = k >> 2;
        const uint32_t ih = ax[ix] ^ ay[iy] ^ az[iz], id = ax[ix] + ay[iy] + az[iz];
        uint32_t idx = (hashed ? ih : id) & mask;
        idx -= (idx >= size) ? size : 0u;                               // non-power-of-two (dense) sizes: index < 2*size, so % size is one subtract
        idx = min(idx, size - 1u);                                      // memory safety for positions far outside [0,1]^3 (never produced by the sampler)
        f(k, idx, (wx[ix] * wy[iy]) * wz[iz]);                          // same product order as the reference walk: ((1 * wx) * wy) * wz
    }
}

}  // namespace mon
