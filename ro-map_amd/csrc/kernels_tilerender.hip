// kernels_tilerender.hip -- the inference side on LDS-resident level tiles (gfx950): NeRF_Model::Render / RenderVideo
// (CORE/src/nerf_model.cu:1702-1830, 1832-1991), GetDensityOnGrid (:2007-2048) and the mesh's vertex colours (:2050-2069).
//
// Why: k_fused_render evaluates tcnn's grid forward as 128 four-byte gathers per sample from an L2-resident table, the access
// pattern round 2 measured at the chip's L1->L2 request floor (profiles/r02_fused_floor.md; the render window of
// profiles/r04_window_render.md "before": 26 M L2 requests and 135 us per 16 384-ray chunk).  Training left that path in round 3
// (k_encode_tiles: a workgroup owns a LEVEL whose table slice sits in the CU's LDS).  A render has 35 training batches' worth of
// samples per crop, so here a tile is loaded ONCE per workgroup and walked for many rounds:
//
//   k_build_feat_image  : the fp16 grid FEATURE-PLANAR, [level][feature][entry] -- one feature of a 65 536-entry level is a
//                         128 KB tile, so all eight corners of a sample are resident at once (k_encode_tiles needs two passes
//                         over parity tiles because a half2 level is 256 KB) and a workgroup can walk any number of samples
//   k_render_rays_jobs  : GenerateRenderRays / GenerateRenderVideoRays (:448-534) for the WHOLE crop; rays that hit the box are
//                         compacted into a job list (wave ballot + one atomic per wave), misses get their white pixel at once
//   k_render_points     : GenerateRenderInputPoints (:593-626): 2S = 64 jittered samples per job -> float4 {x, y, z, t}
//   k_encode_feat       : tcnn kernel_grid forward, one workgroup per (level, feature, sample partition): 8 x ds_read_u16, the
//                         feature's 8-corner fp32 fmaf chain in corner order (bit-identical to encode_interp / k_encode_tiles:
//                         the two features of a level are independent chains), one rounding -> E[level][feature][sample]
//   k_tile_render       : inference MLP (MFMA, fused_device.h) + VolumeRender_Render (:1134-1229), one wavefront per job, two
//                         32-sample tiles with a carried transmittance; a ray that is opaque after its first tile stops there
//   k_tile_points_mlp   : the MLP alone for point queries (density lattice, mesh vertices) -> raw fp16 outputs
#include "fused_device.h"
#include "tile_device.h"

namespace mon {

constexpr uint32_t kFeatLdsBytes = 163840;                       // the CU's whole LDS
constexpr uint32_t kFeatMaxEntries = kFeatLdsBytes / 2u;         // entries of one feature plane that fit
constexpr uint32_t kFeatSpt = 4;                                 // samples per thread and round
constexpr uint32_t kFeatRound = kTileThreads * kFeatSpt;         // samples a workgroup takes per round
constexpr uint32_t kFeatParts = 8;                               // workgroups per (level, feature) tile: one per XCD (blockIdx % 8)

bool tile_render_supported(const LevelTable& lt, const NetDims& nd) {
    if (nd.L < 1 || nd.L > kMaxLevels || (nd.n_mlp & 1u)) return false;
    for (int l = 0; l < nd.L; ++l) {
        const uint32_t size = lt.offset[l + 1] - lt.offset[l];
        if (size > kFeatMaxEntries || (size & 7u) || (lt.offset[l] & 7u)) return false;
    }
    return true;
}

// samples of a chunk whose jobs are counted on the device: jobs [job_base, job_base + jobs_cap) of `*count`, spj samples each
__device__ __forceinline__ uint32_t chunk_jobs(const uint32_t* __restrict__ count, uint32_t job_base, uint32_t jobs_cap) {
    const uint32_t c = *count;
    return c > job_base ? min(c - job_base, jobs_cap) : 0u;
}

// ------------------------------------------------------------------ feature-planar tile image
__global__ void __launch_bounds__(256) k_build_feat_image(LevelFast lt, int L, const uint32_t* __restrict__ grid /* half2 per entry */,
                                                          uint16_t* __restrict__ image, uint32_t* __restrict__ zero_counter) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0u && zero_counter) *zero_counter = 0u;
    if (e >= lt.offset[L]) return;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; ++l) lvl += (l < L && e >= lt.offset[l]) ? 1 : 0;
    const uint32_t off = lt.offset[lvl], size = lt.size[lvl], rel = e - off, v = grid[e];
    image[2u * (size_t)off + rel] = (uint16_t)(v & 0xffffu);
    image[2u * (size_t)off + size + rel] = (uint16_t)(v >> 16);
}

// ------------------------------------------------------------------ rays of a whole crop, hits compacted into jobs
// Job record: three float4 {o, t0} {d, t1} {d_norm, pixel index bits, 0, 0}.
__global__ void __launch_bounds__(1024) k_render_rays_jobs(Intrinsics K, ObjectConst oc, mon_frame_bbox box, Mat4 pose, int pose_is_Toc, uint32_t n_pix,
                                                           float4_t* __restrict__ rec, uint32_t* __restrict__ count, uint32_t* __restrict__ next_count,
                                                           float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ mask) {
    __shared__ uint32_t wave_hits[16], wg_base;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0u && next_count) *next_count = 0u;              // the counter the NEXT render call on this workspace will use
    const bool in = p < n_pix;
    float o[3] = { 0.f, 0.f, 0.f }, d[3] = { 0.f, 0.f, 1.f }, dn = 1.f, t0 = 0.f, t1 = 0.f;
    bool hit = false;
    if (in) {
        const int x = (int)box.x + (int)(p % box.w), y = (int)box.y + (int)(p / box.w);
        pixel_ray(K, (float)x, (float)y, pose.m, oc.Tow.m, pose_is_Toc != 0, o, d, dn);
        hit = ray_intersect(oc.aabb, o, d, t0, t1);
        if (!hit) { rgb[3 * (size_t)p] = 1.f; rgb[3 * (size_t)p + 1] = 1.f; rgb[3 * (size_t)p + 2] = 1.f; depth[p] = 0.f; mask[p] = 0.f; }      // :1221-1226
    }
    // slots: ballot per wave, the waves' counts summed in LDS, ONE returning atomic per workgroup (same-address atomics serialise in their L2 channel: one
    // per wave, 1120 of them for a 313 x 229 crop, cost 11 us of a 16 us kernel)
    const unsigned long long bal = __ballot(hit);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_hits[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0u;
        for (int w = 0; w < 16; ++w) { const uint32_t c = wave_hits[w]; wave_hits[w] = tot; tot += c; }      // -> exclusive prefix
        wg_base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    if (hit) {
        const uint32_t job = wg_base + wave_hits[wave] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        rec[3 * (size_t)job] = float4_t{ o[0], o[1], o[2], fmaxf(t0, 0.0f) };
        rec[3 * (size_t)job + 1] = float4_t{ d[0], d[1], d[2], t1 };
        rec[3 * (size_t)job + 2] = float4_t{ dn, __builtin_bit_cast(float, p), 0.f, 0.f };
    }
}

// ------------------------------------------------------------------ sample positions of a chunk of jobs (one thread per sample)
__global__ void __launch_bounds__(256) k_render_points(ObjectConst oc, const float4_t* __restrict__ rec, const uint32_t* __restrict__ count,
                                                       uint32_t job_base, uint32_t jobs_cap, float4_t* __restrict__ x) {
    const uint32_t S2 = 2u * oc.S;
    const uint32_t n = chunk_jobs(count, job_base, jobs_cap) * S2;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t j = s / S2, k = s - j * S2;
    const float4_t ra = rec[3 * (size_t)(job_base + j)], rb = rec[3 * (size_t)(job_base + j) + 1], rc = rec[3 * (size_t)(job_base + j) + 2];
    const float rc_y = rc.y; const uint32_t pix = __builtin_bit_cast(uint32_t, rc_y);
    const float t0 = ra.w, t1 = rb.w, dtr = (t1 - t0) / (float)S2;
    const float t = fmaf(dtr, (float)k + render_rand(oc, pix * S2 + k), t0);
    const float rd[3] = { rb.x, rb.y, rb.z }, ro[3] = { ra.x, ra.y, ra.z };
    float xw[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float q = fmaf(t, rd[a], ro[a]); xw[a] = (q - oc.aabb.mn[a]) / (oc.aabb.mx[a] - oc.aabb.mn[a]); }
    x[s] = float4_t{ xw[0], xw[1], xw[2], t };
}

// lattice points of the unit cube, x fastest (generate_grid_samples_nerf_uniform :296-309) / warped mesh vertices (:2050-2069), as float4
__global__ void __launch_bounds__(256) k_grid_points4(float4_t* __restrict__ x, int rx, int ry, int rz, uint32_t p0, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = p0 + i;
    const int cx = (int)(p % rx), cy = (int)((p / rx) % ry), cz = (int)(p / ((uint32_t)rx * ry));
    x[i] = float4_t{ (float)cx / (float)(rx - 1), (float)cy / (float)(ry - 1), (float)cz / (float)(rz - 1), 0.f };
}
__global__ void __launch_bounds__(256) k_mesh_warp4(const float* __restrict__ verts, float4_t* __restrict__ x, uint32_t v0, uint32_t n, Aabb box) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) w[a] = (verts[3 * (size_t)(v0 + i) + a] - box.mn[a]) / (box.mx[a] - box.mn[a]);
    x[i] = float4_t{ w[0], w[1], w[2], 0.f };
}

// ------------------------------------------------------------------ feature-tile encode
struct FeatArgs {
    LevelFast lt; int L;
    const uint16_t* image;        // feature-planar tile image (k_build_feat_image)
    const float4_t* x;            // [cap] positions of the chunk
    uint16_t* e;                  // [L][2][cap] encoded features (fp16)
    uint32_t cap;                 // samples per plane of `e`
    uint32_t n_host;              // samples to encode when `count` is null (point queries)
    const uint32_t* count; uint32_t job_base, jobs_cap, spj;      // else: jobs counted on the device, spj samples each
};

__device__ __forceinline__ void feat_load(float4_t (&xs)[kFeatSpt], const float4_t* __restrict__ x, uint32_t r, uint32_t n) {
#pragma unroll
    for (uint32_t k = 0; k < kFeatSpt; ++k) xs[k] = x[min(r * kFeatRound + k * kTileThreads + threadIdx.x, n - 1u)];
}

// LDS BYTE offsets of a sample's eight corners in a feature plane (2 bytes per entry) and its position inside the cell: enc_indices (tile_device.h) with every
// term carried doubled -- xor, and, add and the conditional subtract commute with the shift, and only product bits below the table size matter -- so the
// eight address shifts disappear.  size2 / my2 / mz2 / mask2 = 2 x the level's constants.
template <bool HASHED, bool POW2>
__device__ __forceinline__ void feat_offsets(const float4_t& xv, float scale, uint32_t size2, uint32_t my2, uint32_t mz2, uint32_t mask2,
                                             uint32_t (&o0)[4], uint32_t (&o1)[4], float (&pos)[3]) {
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float q = fmaf(scale, xv[d], 0.5f); pg[d] = (uint32_t)floor_to_int(q); pos[d] = __builtin_amdgcn_fractf(q); }
    const uint32_t y0 = (HASHED && POW2) ? __umul24(pg[1], my2 & 0xffffffu) : pg[1] * my2, z0 = (HASHED && POW2) ? __umul24(pg[2], mz2 & 0xffffffu)
            : pg[2] * mz2;
    const uint32_t ay[2] = { y0, y0 + my2 }, az[2] = { z0, z0 + mz2 };
    const uint32_t x2 = pg[0] << 1, dxm = (x2 ^ (x2 + 2u)) & mask2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (HASHED) {
            const uint32_t t = ay[j & 1] ^ az[j >> 1];
            o0[j] = (x2 ^ t) & mask2;
            if (POW2) o1[j] = o0[j] ^ dxm;
            else {
                o1[j] = ((x2 + 2u) ^ t) & mask2;
                o0[j] -= (o0[j] >= size2) ? size2 : 0u; o0[j] = min(o0[j], size2 - 2u); o1[j] -= (o1[j] >= size2) ? size2 : 0u; o1[j] = min(o1[j], size2 - 2u);
            }
        } else {
            const uint32_t t = ay[j & 1] + az[j >> 1];
            o0[j] = (x2 + t) & mask2; o1[j] = (x2 + 2u + t) & mask2;
            o0[j] -= (o0[j] >= size2) ? size2 : 0u; o0[j] = min(o0[j], size2 - 2u); o1[j] -= (o1[j] >= size2) ? size2 : 0u; o1[j] = min(o1[j], size2 - 2u);
        }
    }
}

// A sample in flight: its eight corner values (requested) and its position inside the cell.
struct FeatPend { uint16_t c0[4], c1[4]; float pos[3]; };
template <bool HASHED, bool POW2>
__device__ __forceinline__ void feat_issue(FeatPend& p, const unsigned char* tile, const float4_t& xv, float scale, uint32_t size2, uint32_t my2, uint32_t mz2,
        uint32_t mask2) {
    uint32_t o0[4], o1[4];
    feat_offsets<HASHED, POW2>(xv, scale, size2, my2, mz2, mask2, o0, o1, p.pos);
    // (the tile starts at LDS address 0 -- the kernel has no static LDS, checked at its entry -- so the byte offset IS the address: through `tile + offset`
    //  the compiler emits a v_add_u32 with a literal 0 per read, an eighth of the walk's instructions)
    typedef const __attribute__((address_space(3))) uint16_t* lds_u16;
    (void)tile;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p.c0[j] = *(lds_u16)(uintptr_t)o0[j]; p.c1[j] = *(lds_u16)(uintptr_t)o1[j]; }
}
// one feature's chain of encode_interp / enc_chain: corners in order k = x + 2y + 4z, weight ((wx * wy) * wz), one rounding
__device__ __forceinline__ uint16_t feat_finish(const FeatPend& p) {
    const float2_t wx = { 1.f - p.pos[0], p.pos[0] };
    const float wy[2] = { 1.f - p.pos[1], p.pos[1] }, wz[2] = { 1.f - p.pos[2], p.pos[2] };
    const float2_t wxy[2] = { wx * wy[0], wx * wy[1] };
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2_t w = wxy[j & 1] * wz[j >> 1];
        acc = fmaf(w.x, (float)__builtin_bit_cast(half_t, p.c0[j]), acc);
        acc = fmaf(w.y, (float)__builtin_bit_cast(half_t, p.c1[j]), acc);
    }
    // (the value passes through an opaque register: left alone, the compiler folds the last fma and the conversion into v_fma_mixlo_f16, which rounds the
    //  exact sum ONCE to fp16 -- one result in ~2^13 then differs from fmaf + conversion, the contract of encode_interp and the oracle)
    asm volatile("" : "+v"(acc));
    return __builtin_bit_cast(uint16_t, (half_t)acc);
}

// The walk of one workgroup: rounds part, part + P, ... of kFeatRound samples.  Inside a round a thread's samples form a two-deep software pipeline -- the
// corner reads of sample k + 1 are requested before sample k's chain consumes its own (16 LDS reads in flight per wave: the counter's range) -- and the next
// round's positions travel under the current round's arithmetic.
template <bool HASHED, bool POW2>
__device__ __forceinline__ void feat_walk(const unsigned char* tile, const FeatArgs& a, uint32_t n, uint32_t part, uint16_t* __restrict__ out,
                                          float scale, uint32_t size, uint32_t my, uint32_t mz, uint32_t mask) {
    const uint32_t rounds = (n + kFeatRound - 1u) / kFeatRound;
    const uint32_t size2 = size << 1, my2 = my << 1, mz2 = mz << 1, mask2 = mask << 1;
    float4_t cur[kFeatSpt], nxt[kFeatSpt];
    uint32_t r = part;
    if (r < rounds) feat_load(cur, a.x, r, n);                   // (requested behind the tile copy: both run under one wait)
    __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): the LDS writes of the copy are counted there
    __syncthreads();
    const auto round = [&](const float4_t (&xs)[kFeatSpt], uint32_t rr) {
        const uint32_t s0 = rr * kFeatRound + threadIdx.x;
        const bool full = (rr + 1u) * kFeatRound <= n;            // (uniform: only the last round of a chunk can be partial)
        FeatPend pend[2];
        feat_issue<HASHED, POW2>(pend[0], tile, xs[0], scale, size2, my2, mz2, mask2);
#pragma unroll
        for (uint32_t k = 1; k <= kFeatSpt; ++k) {
            if (k < kFeatSpt) feat_issue<HASHED, POW2>(pend[k & 1u], tile, xs[k], scale, size2, my2, mz2, mask2);
            const uint16_t e = feat_finish(pend[(k - 1u) & 1u]);
            const uint32_t s = s0 + (k - 1u) * kTileThreads;
            if (full || s < n) out[s] = e;
        }
    };
    for (; r < rounds; r += 2u * kFeatParts) {                    // two rounds per trip: the position buffers swap roles, no register copies
        const uint32_t r1 = r + kFeatParts, r2 = r + 2u * kFeatParts;
        if (r1 < rounds) feat_load(nxt, a.x, r1, n);
        round(cur, r);
        if (r1 >= rounds) break;
        if (r2 < rounds) feat_load(cur, a.x, r2, n);
        round(nxt, r1);
    }
}

__global__ void __launch_bounds__(kTileThreads) k_encode_feat(FeatArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    // feat_issue addresses the tile from LDS offset 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();
    const uint32_t tid = blockIdx.x / kFeatParts, part = blockIdx.x - tid * kFeatParts, level = tid >> 1, f = tid & 1u;
    const uint32_t n = a.count ? chunk_jobs(a.count, a.job_base, a.jobs_cap) * a.spj : a.n_host;
    if (n == 0u || part * kFeatRound >= n) return;               // (before the tile copy: an empty chunk, or a partition without a round)
    const uint32_t off = a.lt.offset[level], size = a.lt.size[level], my = a.lt.my[level], mz = a.lt.mz[level], mask = a.lt.mask[level];
    const bool hashed = a.lt.hashed[level] != 0u, pow2 = mask != 0xffffffffu;
    const float scale = a.lt.scale[level];
    const uint4* src = reinterpret_cast<const uint4*>(a.image + 2u * (size_t)off + (size_t)f * size);
    tile_copy(reinterpret_cast<uint32_t*>(tile), src, size / 8u);
    uint16_t* out = a.e + ((size_t)level * 2u + f) * a.cap;
    if (hashed) {
        if (pow2) feat_walk<true, true>(tile, a, n, part, out, scale, size, my, mz, mask);
        else feat_walk<true, false>(tile, a, n, part, out, scale, size, my, mz, mask);
    } else feat_walk<false, false>(tile, a, n, part, out, scale, size, my, mz, mask);
}

// ------------------------------------------------------------------ MLP + composite from the encoded features
struct TileMlpArgs {
    NetDims nd; ObjectConst oc;
    const uint16_t* frag_image;   // A fragments of the weights being rendered (k_build_frag_image)
    const float4_t* rec; const uint32_t* count; uint32_t job_base, jobs_cap;
    const float4_t* x;            // [cap] {x, y, z, t}
    const uint16_t* e; uint32_t cap;
    uint32_t n_points;            // k_tile_points_mlp: points of the chunk; k_tile_render: pixels of the crop
};

template <int EPAD, int W, int NH>
__device__ __forceinline__ void copy_forward_frags(half_t* frags, const uint16_t* __restrict__ frag_image) {
    using S = FusedShape<EPAD, W, NH>;
    const uint4* src = reinterpret_cast<const uint4*>(frag_image); uint4* dst = reinterpret_cast<uint4*>(frags);
    for (int i = threadIdx.x; i < S::F_WOT * 64; i += blockDim.x) dst[i] = src[i];
}
// lane (n, h): the features of the levels half-wave h owns (its MFMA B-operand K-slots), sample s of the chunk
template <int EPAD, int W, int NH>
__device__ __forceinline__ void load_features(TileState<EPAD, W, NH>& ts, const uint16_t* __restrict__ e, uint32_t cap, uint32_t s, int L, int h) {
    using S = FusedShape<EPAD, W, NH>;
    const int LPH = (L + 1) >> 1;
#pragma unroll
    for (int il = 0; il < S::LLV; ++il) {
        const int level = h * LPH + il;
        uint16_t f0 = 0, f1 = 0;
        if (il < LPH && level < L) { f0 = e[(size_t)(2 * level) * cap + s]; f1 = e[(size_t)(2 * level + 1) * cap + s]; }
        ts.ef[2 * il] = __builtin_bit_cast(half_t, f0); ts.ef[2 * il + 1] = __builtin_bit_cast(half_t, f1);
    }
}

// One 32-sample tile's operands as they come from memory: the sample distances and the encoded features of the levels this half-wave owns.
template <int EPAD> struct TileLoad { float t; uint16_t f[EPAD / 2]; };
// The features come through buffer loads: ONE vector offset per tile (the sample's place in the planes of the first level this half-wave owns) and a scalar
// offset per (local level, feature) -- address arithmetic per load was a fifth of the kernel's instructions.  A level slot this half-wave does not own gets a
// scalar offset past the buffer's end, and a level past the last one lies there by itself: out-of-range buffer loads return zero, the padding the MLP expects.
struct FeatBuf { __amdgpu_buffer_rsrc_t rsrc; uint32_t lane_base; uint32_t plane_bytes; };
__device__ __forceinline__ FeatBuf feat_buffer(const TileMlpArgs& a, int L, int h) {
    const int LPH = (L + 1) >> 1;
    FeatBuf fb; fb.plane_bytes = a.cap * 2u;
    fb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.e), 0, (int)((uint32_t)L * 2u * fb.plane_bytes), 0x00020000);
    fb.lane_base = (uint32_t)(h * LPH) * 2u * fb.plane_bytes;
    return fb;
}
template <int EPAD, int W, int NH>
__device__ __forceinline__ void tile_request(TileLoad<EPAD>& q, const TileMlpArgs& a, const FeatBuf& fb, uint32_t j, uint32_t tile, int L, int n) {
    using S = FusedShape<EPAD, W, NH>;
    const int LPH = (L + 1) >> 1;
    const uint32_t s = j * 64u + tile * 32u + (uint32_t)n;
    q.t = reinterpret_cast<const float*>(a.x)[4 * (size_t)s + 3];
    const uint32_t voff = fb.lane_base + s * 2u;
#pragma unroll
    for (int il = 0; il < S::LLV; ++il) {
        const uint32_t soff = il < LPH ? (uint32_t)(2 * il) * fb.plane_bytes : 0xfffffff0u;      // (uniform)
        q.f[2 * il] = __builtin_amdgcn_raw_buffer_load_b16(fb.rsrc, voff, soff, 0);
        q.f[2 * il + 1] = __builtin_amdgcn_raw_buffer_load_b16(fb.rsrc, voff, il < LPH ? soff + fb.plane_bytes : 0xfffffff0u, 0);
    }
}

// One wavefront per job, two 32-sample tiles with a carried transmittance.  A wave walks its tiles (job j tile 0, job j tile 1, job j + stride tile 0, ...)
// as a two-deep pipeline: the next tile's operands are requested before the current tile is evaluated (a second tile behind an opaque first one is
// requested in vain and skipped).
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_tile_render(TileMlpArgs a, float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ mask) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    const uint32_t njobs = chunk_jobs(a.count, a.job_base, a.jobs_cap);
    if (blockIdx.x * S::WAVES >= njobs) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int L = a.nd.L;
    const uint32_t stride = gridDim.x * S::WAVES;
    uint32_t j = blockIdx.x * S::WAVES + wave;
    TileLoad<EPAD> cur, nxt;
    const FeatBuf fb = feat_buffer(a, L, h);
    if (j < njobs) tile_request<EPAD, W, NH>(cur, a, fb, j, 0u, L, n);      // (ahead of the fragment copy: its round trip runs under it)
    copy_forward_frags<EPAD, W, NH>(frags, a.frag_image);
    __syncthreads();
    for (; j < njobs; j += stride) {
        const float4_t rc = a.rec[3 * (size_t)(a.job_base + j) + 2];
        float Tc = 1.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, tlast = 0.f;
#pragma unroll
        for (uint32_t tile = 0; tile < 2u; ++tile) {
            if (tile == 0u) tile_request<EPAD, W, NH>(nxt, a, fb, j, 1u, L, n);
            else if (j + stride < njobs) tile_request<EPAD, W, NH>(nxt, a, fb, j + stride, 0u, L, n);
            if (Tc >= kTransmittanceEps) {
                const float t = cur.t;
                TileState<EPAD, W, NH> ts;
#pragma unroll
                for (int i = 0; i < EPAD / 2; ++i) ts.ef[i] = __builtin_bit_cast(half_t, cur.f[i]);
                mlp_forward<EPAD, W, NH>(ts, frags, lane);
                // VolumeRender_Render :1134-1229 over lanes 0..31 (the arithmetic of k_fused_render)
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
            cur = nxt;
        }
        float o0 = 1.f, o1 = 1.f, o2 = 1.f, od = 0.f, om_ = 0.f;
        if (1.f - Tc > 0.5f) { o0 = r0 + Tc; o1 = r1 + Tc; o2 = r2 + Tc; od = dep / rc.x; om_ = 1.f; }      // :1213-1220
        const float rc_y = rc.y; const uint32_t pix = __builtin_bit_cast(uint32_t, rc_y);
        // (n_points = the crop's pixels: a record is never trusted with an address)
        if (lane == 0 && pix < a.n_points) { rgb[3 * (size_t)pix] = o0; rgb[3 * (size_t)pix + 1] = o1; rgb[3 * (size_t)pix + 2] = o2; depth[pix] = od;
            mask[pix] = om_; }
    }
}

// raw network outputs of a chunk of points, fp16 [n][4] (the layout extract_density / the mesh colours read)
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_tile_points_mlp(TileMlpArgs a, uint16_t* __restrict__ O) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    copy_forward_frags<EPAD, W, NH>(frags, a.frag_image);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = (a.n_points + 31u) / 32u;
    for (uint32_t w = blockIdx.x * S::WAVES + wave; w < n_tiles; w += gridDim.x * S::WAVES) {
        const uint32_t s = min(w * 32u + (uint32_t)n, a.n_points - 1u);
        TileState<EPAD, W, NH> ts;
        load_features<EPAD, W, NH>(ts, a.e, a.cap, s, a.nd.L, h);
        mlp_forward<EPAD, W, NH>(ts, frags, lane);
        if (h == 0 && w * 32u + (uint32_t)n < a.n_points)
            reinterpret_cast<half4_t*>(O)[s] = half4_t{ (half_t)ts.out4[0], (half_t)ts.out4[1], (half_t)ts.out4[2], (half_t)ts.out4[3] };
    }
}

// ------------------------------------------------------------------ launchers
static std::atomic<uint64_t> g_feat_attr{ 0 }; static std::mutex g_feat_attr_mu;
static void feat_setup_device() {
    once_per_device(g_feat_attr, g_feat_attr_mu, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_feat), hipFuncAttributeMaxDynamicSharedMemorySize, kFeatLdsBytes);
    });
}

void launch_build_feat_image(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* params, uint16_t* image, uint32_t* zero_counter) {
    const uint32_t n = lf.offset[nd.L];
    hipLaunchKernelGGL(k_build_feat_image, dim3((n + 255u) / 256u), dim3(256), 0, s, lf, nd.L, reinterpret_cast<const uint32_t*>(params + nd.n_mlp), image,
            zero_counter);
}
void launch_render_rays_jobs(hipStream_t s, const Intrinsics& K, const ObjectConst& oc, mon_frame_bbox box, const Mat4& pose, int pose_is_Toc, uint32_t n_pix,
                             float* rec, uint32_t* count, uint32_t* next_count, float* rgb, float* depth, float* mask) {
    hipLaunchKernelGGL(k_render_rays_jobs, dim3((n_pix + 1023u) / 1024u), dim3(1024), 0, s, K, oc, box, pose, pose_is_Toc, n_pix,
                       reinterpret_cast<float4_t*>(rec), count, next_count, rgb, depth, mask);
}
void launch_render_points(hipStream_t s, const ObjectConst& oc, const float* rec, const uint32_t* count, uint32_t job_base, uint32_t jobs_cap, float* x) {
    const uint32_t n = jobs_cap * 2u * oc.S;
    hipLaunchKernelGGL(k_render_points, dim3((n + 255u) / 256u), dim3(256), 0, s, oc, reinterpret_cast<const float4_t*>(rec), count, job_base, jobs_cap,
                       reinterpret_cast<float4_t*>(x));
}
void launch_grid_points4(hipStream_t s, float* x, int rx, int ry, int rz, uint32_t p0, uint32_t n) {
    hipLaunchKernelGGL(k_grid_points4, dim3((n + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<float4_t*>(x), rx, ry, rz, p0, n);
}
void launch_mesh_warp4(hipStream_t s, const float* verts, float* x, uint32_t v0, uint32_t n, const Aabb& box) {
    hipLaunchKernelGGL(k_mesh_warp4, dim3((n + 255u) / 256u), dim3(256), 0, s, verts, reinterpret_cast<float4_t*>(x), v0, n, box);
}
// n_host samples (count == nullptr) or the jobs [job_base, job_base + jobs_cap) of *count with spj samples each
void launch_encode_feat(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* image, const float* x, uint16_t* e, uint32_t cap,
                        uint32_t n_host, const uint32_t* count, uint32_t job_base, uint32_t jobs_cap, uint32_t spj) {
    feat_setup_device();
    FeatArgs a{ lf, nd.L, image, reinterpret_cast<const float4_t*>(x), e, cap, n_host, count, job_base, jobs_cap, spj };
    hipLaunchKernelGGL(k_encode_feat, dim3((uint32_t)nd.L * 2u * kFeatParts), dim3(kTileThreads), kFeatLdsBytes, s, a);
}

template <int EPAD, int W, int NH>
static void tile_render_t(hipStream_t s, const TileMlpArgs& a, float* rgb, float* depth, float* mask) {
    using S = FusedShape<EPAD, W, NH>;
    // five workgroups per CU: a wave takes several jobs and prefetches the next
    uint32_t grid = (a.jobs_cap + S::WAVES - 1u) / S::WAVES; if (grid > 1280u) grid = 1280u;
    hipLaunchKernelGGL((k_tile_render<EPAD, W, NH>), dim3(grid), dim3(256), S::F_WOT * 1024, s, a, rgb, depth, mask);
}
template <int EPAD, int W, int NH>
static void tile_points_mlp_t(hipStream_t s, const TileMlpArgs& a, uint16_t* O) {
    using S = FusedShape<EPAD, W, NH>;
    uint32_t grid = ((a.n_points + 31u) / 32u + S::WAVES - 1u) / S::WAVES; if (grid > 2048u) grid = 2048u;
    hipLaunchKernelGGL((k_tile_points_mlp<EPAD, W, NH>), dim3(grid ? grid : 1u), dim3(256), S::F_WOT * 1024, s, a, O);
}
template <int EPAD, int W, int NH>
static void frag_image_t(hipStream_t s, const uint16_t* params, int L, uint16_t* image) {
    using S = FusedShape<EPAD, W, NH>;
    hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::F_WOT * 512 + 255) / 256), dim3(256), 0, s, params, L, image, (const DevState*)nullptr);
}

void launch_tile_render(hipStream_t s, const NetDims& nd, const ObjectConst& oc, const uint16_t* frag_image, const float* rec, const uint32_t* count,
                        uint32_t job_base, uint32_t jobs_cap, const float* x, const uint16_t* e, uint32_t cap, uint32_t n_pix, float* rgb, float* depth,
                                float* mask) {
    TileMlpArgs a{ nd, oc, frag_image, reinterpret_cast<const float4_t*>(rec), count, job_base, jobs_cap, reinterpret_cast<const float4_t*>(x), e, cap, n_pix };
    MON_FUSED_DISPATCH(tile_render_t, s, a, rgb, depth, mask);
}
void launch_tile_points_mlp(hipStream_t s, const NetDims& nd, const uint16_t* frag_image, const uint16_t* e, uint32_t cap, uint32_t n_points, uint16_t* O) {
    TileMlpArgs a{ nd, ObjectConst{}, frag_image, nullptr, nullptr, 0u, 0u, nullptr, e, cap, n_points };
    MON_FUSED_DISPATCH(tile_points_mlp_t, s, a, O);
}
// the forward A fragments of `params` (the first F_WOT fragments of the training image's layout)
void launch_forward_frag_image(hipStream_t s, const NetDims& nd, const uint16_t* params, uint16_t* image) {
    MON_FUSED_DISPATCH(frag_image_t, s, params, nd.L, image);
}

}  // namespace mon
