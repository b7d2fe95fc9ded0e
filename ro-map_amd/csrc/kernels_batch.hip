// kernels_batch.hip -- batch generation and ray set-up (gfx950).
//
// Restates GenerateBatch (CORE/src/nerf_model.cu:1429-1502):
//   GenerateRays :369-446  -> k_gen_candidates   (validity as a wave ballot, no atomics)
//   fill_rollover_rays :280-294 + the atomicAdd compaction :419 -> k_build_rays
//       (order-stable select over the 64-bit ballot words; no host round trip, cf. the two
//        cudaStreamSynchronize calls at :1459,:1469)
//   GenerateInputPoints :536-566 -> k_gen_samples
// and the render-side GenerateRender(Video)Rays :448-534 / GenerateRenderInputPoints :593-626.
#include "device_common.h"
#include "model.h"
#include "batch_device.h"

namespace mon {

// One thread per candidate ray; body in batch_device.h.
__global__ void __launch_bounds__(256) k_gen_candidates(BatchPtrs b, DatasetPtrs ds, ObjectConst oc, const DevState* __restrict__ st) {
    gen_candidate(b, ds, oc, st->n_boxes, st->iter, blockIdx.x * blockDim.x + threadIdx.x);
}

// k-th set bit of a 64-bit word (k < popcount).
__device__ inline uint32_t select_bit(unsigned long long w, uint32_t k) {
    uint32_t pos = 0;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        const uint32_t c = __popcll(w & ((1ull << sh) - 1ull));
        if (k >= c) { k -= c; w >>= sh; pos += sh; }
    }
    return pos;
}

// One thread per training ray j: the ray is valid candidate number (j mod n_valid) in candidate
// order (rollover, :280-294).  Also resets the per-iteration accumulators in DevState.
__global__ void __launch_bounds__(256) k_build_rays(BatchPtrs b, ObjectConst oc, DevState* __restrict__ st) {
    __shared__ uint32_t prefix[65];
    __shared__ unsigned long long words[64];
    const uint32_t R = oc.R, nwords = R >> 6;     // R is a multiple of 64, nwords <= 64 per 4096 rays
    // R may exceed 4096: walk the mask in 64-word segments.
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    // total count
    uint32_t total = 0;
    for (uint32_t base = 0; base < nwords; base += 64) {
        if (threadIdx.x < 64) {
            const unsigned long long w = (base + threadIdx.x < nwords) ? b.mask[base + threadIdx.x] : 0ull;
            uint32_t c = __popcll(w);
            // wave-inclusive scan over 64 lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(c, off); if ((int)threadIdx.x >= off) c += v; }
            if (threadIdx.x == 63) prefix[64] = c;
        }
        __syncthreads();
        total += prefix[64];
        __syncthreads();
    }
    if (j == 0) { st->n_valid = total; st->n_valid_pre = total; st->loss_sum = 0.0f; }      // (n_valid_pre: where k_encode_tiles looks for a skipped batch)
    if (total == 0u) return;                  // uniform across the grid
    const bool active = j < R;
    const uint32_t iter = st->iter;
    const uint32_t k = (active ? j : 0u) % total;
    // locate candidate index of the k-th valid ray
    uint32_t seen = 0, cand = 0; bool found = false;
    for (uint32_t base = 0; base < nwords; base += 64) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned long long w = (base + threadIdx.x < nwords) ? b.mask[base + threadIdx.x] : 0ull;
            words[threadIdx.x] = w;
            uint32_t c = __popcll(w), inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off); if ((int)threadIdx.x >= off) inc += v; }
            prefix[threadIdx.x] = inc - c;            // exclusive
            if (threadIdx.x == 63) prefix[64] = inc;
        }
        __syncthreads();
        const uint32_t segtotal = prefix[64];
        if (!found && k < seen + segtotal) {
            const uint32_t kk = k - seen;
            uint32_t lo = 0, hi = 63;                 // largest wi with prefix[wi] <= kk
            while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (prefix[mid] <= kk) lo = mid; else hi = mid - 1; }
            cand = ((base + lo) << 6) + select_bit(words[lo], kk - prefix[lo]);
            found = true;
        }
        seen += segtotal;
    }
    if (!active) return;
    const uint32_t rgba = b.cand_rgba[cand];
    const bool is_obj = (rgba >> 24) != 0u;
    float bg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) bg[a] = batch_rand(oc, kStreamColor, iter, 3u * k + a);     // :760, :438-441
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        b.ray_o[3 * j + a] = b.cand_o[3 * cand + a];
        b.ray_d[3 * j + a] = b.cand_d[3 * cand + a];
        b.bgcol[3 * j + a] = bg[a];
        b.target[3 * j + a] = is_obj ? (float)((rgba >> (8 * a)) & 0xffu) / 255.0f : bg[a];           // nerf_data.cu:169
    }
    b.ray_dn[j] = b.cand_dn[cand]; b.ray_t0[j] = b.cand_t0[cand]; b.ray_t1[j] = b.cand_t1[cand];
    b.ray_flag[j] = is_obj ? 1 : 0;
    b.target_depth[j] = b.cand_depth[cand];
}

// One thread per sample (ray-major: sample s belongs to ray s / S).  :553-566
__global__ void __launch_bounds__(256) k_gen_samples(BatchPtrs b, ObjectConst oc, const DevState* __restrict__ st, uint32_t S, uint32_t n_samples,
                                                     uint32_t rng_stream, uint32_t idx_base, int render, float4_t* __restrict__ x4 /* also as k_encode_tiles' float4 */) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_samples) return;
    if (!render && st->n_valid == 0u) return;
    const uint32_t j = s / S, n = s - j * S;
    const uint32_t iter = render ? 0u : st->iter;
    // :599-602 (left uninitialised there)
    if (render && b.ray_flag[j] == 0) { b.pts[3 * s] = 0.f; b.pts[3 * s + 1] = 0.f; b.pts[3 * s + 2] = 0.f; b.tdist[s] = 0.f; return; }
    const float t0 = b.ray_t0[j], t1 = b.ray_t1[j];
    const float dt = (t1 - t0) / (float)S;
    const float t = fmaf(dt, (float)n + (render ? render_rand(oc, idx_base + s) : batch_rand(oc, rng_stream, iter, idx_base + s)), t0);
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float p = fmaf(t, b.ray_d[3 * j + a], b.ray_o[3 * j + a]);
        x[a] = (p - oc.aabb.mn[a]) / (oc.aabb.mx[a] - oc.aabb.mn[a]); b.pts[3 * s + a] = x[a];      // WarpPoint :140-144
    }
    if (x4) x4[s] = float4_t{ x[0], x[1], x[2], 0.f };
    b.tdist[s] = t;
}

// Render rays: one thread per pixel of the 2-D box chunk [pix0, pix0+n). :448-534
__global__ void __launch_bounds__(256) k_render_rays(BatchPtrs b, Intrinsics K, ObjectConst oc, mon_frame_bbox box, Mat4 pose, int pose_is_Toc,
                                                     uint32_t pix0, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pix0 + i;
    const int x = (int)box.x + (int)(p % box.w), y = (int)box.y + (int)(p / box.w);
    float o[3], d[3], dn, t0, t1;
    pixel_ray(K, (float)x, (float)y, pose.m, oc.Tow.m, pose_is_Toc != 0, o, d, dn);
    const bool hit = ray_intersect(oc.aabb, o, d, t0, t1);
    b.ray_flag[i] = hit ? 1 : 0;
    b.ray_dn[i] = dn;
    if (hit) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { b.ray_o[3 * i + a] = o[a]; b.ray_d[3 * i + a] = d[a]; }
        b.ray_t0[i] = fmaxf(t0, 0.0f); b.ray_t1[i] = t1;
    }
}

// Lattice positions of the unit cube, x fastest (generate_grid_samples_nerf_uniform :296-309).
__global__ void __launch_bounds__(256) k_grid_points(float* __restrict__ pts, int rx, int ry, int rz, uint32_t p0, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = p0 + i;
    const int x = (int)(p % rx), y = (int)((p / rx) % ry), z = (int)(p / ((uint32_t)rx * ry));
    pts[3 * i] = (float)x / (float)(rx - 1); pts[3 * i + 1] = (float)y / (float)(ry - 1); pts[3 * i + 2] = (float)z / (float)(rz - 1);
}

void launch_gen_candidates(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st) {
    hipLaunchKernelGGL(k_gen_candidates, dim3((oc.R + 255) / 256), dim3(256), 0, s, b, ds, oc, st);
}
void launch_build_rays(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st) {
    hipLaunchKernelGGL(k_build_rays, dim3((oc.R + 255) / 256), dim3(256), 0, s, b, oc, st);
}
void launch_gen_samples(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, const DevState* st, uint32_t S, uint32_t n_samples, uint32_t stream_id,
        uint32_t idx_base, int render, float* x4) {
    hipLaunchKernelGGL(k_gen_samples, dim3((n_samples + 255) / 256), dim3(256), 0, s, b, oc, st, S, n_samples, stream_id, idx_base, render,
            reinterpret_cast<float4_t*>(x4));
}
void launch_render_rays(hipStream_t s, const BatchPtrs& b, const Intrinsics& K, const ObjectConst& oc, mon_frame_bbox box, const Mat4& pose, int pose_is_Toc,
        uint32_t pix0, uint32_t n) {
    hipLaunchKernelGGL(k_render_rays, dim3((n + 255) / 256), dim3(256), 0, s, b, K, oc, box, pose, pose_is_Toc, pix0, n);
}
void launch_grid_points(hipStream_t s, float* pts, int rx, int ry, int rz, uint32_t p0, uint32_t n) {
    hipLaunchKernelGGL(k_grid_points, dim3((n + 255) / 256), dim3(256), 0, s, pts, rx, ry, rz, p0, n);
}

// One incoming frame: colour (3 or 4 bytes per pixel, in pinned host memory: the loads cross PCIe) + instance byte -> r | g << 8 | b << 16 | instance << 24.
// The host rewrites the same staging addresses for every frame, so they are read with SYSTEM-scope loads (sc0 sc1: past the GPU's caches): an ordinary load
// may legally be served from a line a cache kept from the previous frame's kernel, whatever coherence flag the pinned allocation carries.
template <class T> __device__ __forceinline__ T host_load(const T* p) {
#ifdef MON_PLAIN_HOST_LOADS          // (variant build for the upload test: ordinary loads, to show what it catches)
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
__global__ void __launch_bounds__(256) k_pack_frame(const uint8_t* rgb, int ch, int ri, int bi, const uint8_t* inst, uint32_t* __restrict__ dst, uint32_t px) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < px; i += gridDim.x * blockDim.x) {
        const uint8_t* c = rgb + (size_t)i * (uint32_t)ch;
        dst[i] = (uint32_t)host_load(c + ri) | ((uint32_t)host_load(c + 1) << 8) | ((uint32_t)host_load(c + bi) << 16) | ((uint32_t)host_load(inst + i) << 24);
    }
}
// n 4-byte words from rewritten pinned host memory to the device (a frame's depth image, its pose)
__global__ void __launch_bounds__(256) k_copy_from_host(const uint32_t* src, uint32_t* __restrict__ dst, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = host_load(src + i);
}
void launch_copy_from_host(hipStream_t s, const void* src, void* dst, uint32_t n_words) {
    const uint32_t blocks = (n_words + 255u) / 256u > 1024u ? 1024u : (n_words + 255u) / 256u;
    hipLaunchKernelGGL(k_copy_from_host, dim3(blocks ? blocks : 1u), dim3(256), 0, s, static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), n_words);
}
void launch_pack_frame(hipStream_t s, const uint8_t* rgb, int ch, int ri, int bi, const uint8_t* inst, uint32_t* dst, uint32_t px) {
    hipLaunchKernelGGL(k_pack_frame, dim3((px + 255u) / 256u > 1024u ? 1024u : (px + 255u) / 256u), dim3(256), 0, s, rgb, ch, ri, bi, inst, dst, px);
}

}  // namespace mon
