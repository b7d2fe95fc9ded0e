// tile_device.h -- device pieces shared by the kernels that read hash-grid levels from LDS-resident tiles: the training batch's
// k_encode_tiles (kernels_encode.hip) and the inference-side k_encode_feat (kernels_tilerender.hip).
#pragma once
#include "device_common.h"
#include "model.h"

namespace mon {

constexpr uint32_t kTileThreads = 1024;                    // threads of a tile workgroup (one per CU: the tile fills the LDS)

// x-corner-0 and x-corner-1 entry index of the four (y, z) pairs (j = y + 2z) and the position inside the cell; the arithmetic of gather_level / encode_interp.
// Hashed levels hold 2^T <= 65 536 entries: only index bits below the table size matter, so the products come from the full-rate 24-bit multiplier
// (v_mul_lo_u32 runs at a quarter of it), and the two x-corners differ by the xor with dxm = (x ^ (x + 1)) & mask.
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int floor_to_int(float q) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(q)); return r; }      // (int)floorf(q)
template <bool HASHED, bool POW2>
__device__ __forceinline__ void enc_indices(const float4_t& xv, float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask, uint32_t (&i0)[4],
        uint32_t (&i1)[4], float (&pos)[3]) {
    uint32_t pg[3];
    // (v_fract_f32 = q - floor(q) exactly for q >= 0; one instruction each instead of floor + subtract + convert)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float q = fmaf(scale, xv[d], 0.5f); pg[d] = (uint32_t)floor_to_int(q); pos[d] = __builtin_amdgcn_fractf(q); }
    const uint32_t y0 = (HASHED && POW2) ? __umul24(pg[1], my & 0xffffffu) : pg[1] * my, z0 = (HASHED && POW2) ? __umul24(pg[2], mz & 0xffffffu) : pg[2] * mz;
    const uint32_t ay[2] = { y0, y0 + my }, az[2] = { z0, z0 + mz };
    const uint32_t dxm = (pg[0] ^ (pg[0] + 1u)) & mask;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (HASHED) {
            const uint32_t t = ay[j & 1] ^ az[j >> 1];
            i0[j] = (pg[0] ^ t) & mask;
            if (POW2) i1[j] = i0[j] ^ dxm;
            else { i1[j] = ((pg[0] + 1u) ^ t) & mask; i0[j] -= (i0[j] >= size) ? size : 0u; i0[j] = min(i0[j], size - 1u);
                i1[j] -= (i1[j] >= size) ? size : 0u; i1[j] = min(i1[j], size - 1u); }
        } else {
            const uint32_t t = ay[j & 1] + az[j >> 1];
            i0[j] = (pg[0] + t) & mask; i1[j] = (pg[0] + 1u + t) & mask;
            i0[j] -= (i0[j] >= size) ? size : 0u; i0[j] = min(i0[j], size - 1u); i1[j] -= (i1[j] >= size) ? size : 0u; i1[j] = min(i1[j], size - 1u);
        }
    }
}
// Tile copy: global -> LDS without a register round trip (global_load_lds_dwordx4: every lane names its 16 source bytes, the wave's 1 KB lands at the
// wave-uniform LDS base in M0 + 16 * lane).  The source is the TILE IMAGE of the grid (ParamPtrs::half_tiles, tile_slot in model.h), in which a tile is one
// contiguous run; all of a thread's loads are in flight at once (a copy through registers with one load per loop trip left a workgroup with 16 KB
// outstanding, and the copy of a 256 KB level took longer than the walk it feeds).
__device__ __forceinline__ void tile_copy(uint32_t* tile, const uint4* __restrict__ src, uint32_t n16) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    unsigned char* base = reinterpret_cast<unsigned char*>(tile);
    const uint32_t wave0 = threadIdx.x & ~63u;
    for (uint32_t i0 = 0; i0 < n16; i0 += kTileThreads) {
        const uint32_t i = i0 + threadIdx.x;
        // (lanes past the tile's end stay out: a lane's LDS address is its position in the wave, active or not)
        if (i < n16)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + i), (lds_void*)(base + (size_t)(i0 + wave0) * 16u), 16, 0, 0);
    }
}

}  // namespace mon
