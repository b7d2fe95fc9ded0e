// device_common.h -- shared host/device definitions for the gfx950 Multi-Object-NeRF core.
// Arithmetic follows SURVEY.md 8(a) rows a5-a31; reference = CORE/src/nerf_model.cu (cited per function).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mon {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int kMaxLevels = 16;
constexpr int kOut = 4;        // rgb + density (nerf_model.cu:1318)
constexpr int kOutPad = 16;    // tcnn pads the output layer to 16 rows
constexpr float kTransmittanceEps = 1e-4f;   // nerf_model.cu:763
// Ray bins of the compacted gradient rows (k_fused_train -> k_grid_scatter): a power of two between 16 and 128, one bin per 32 rays where the batch allows
constexpr uint32_t kMaxScatterBins = 128, kScatterCounterStride = 16;      // (dwords between two bins' slot counters: one 64-byte line each)
// index into DevState::n_scatter
__host__ __device__ inline uint32_t scatter_counter(uint32_t iter, uint32_t bin) { return ((iter & 1u) * kMaxScatterBins + bin) * kScatterCounterStride; }
// measured (profiles/r02): with the counters on separate lines 16 bins cost k_fused_train nothing, and every further bin is another short run k_grid_scatter
// walks late in training
constexpr uint32_t kDefaultScatterBins = 16;
// most bins a batch of R rays supports (one per 32 rays; the count always divides R, a multiple of 64)
__host__ __device__ inline uint32_t scatter_bins_max(uint32_t R) {
    uint32_t nb = 16; while (nb < kMaxScatterBins && nb * 64u <= R && R % (nb * 2u) == 0u) nb *= 2u; return nb;
}
constexpr int kOccRes = 64;                  // occupancy grid (opt-in forward-pass skipping): cells per axis over the object's box, one bit each
constexpr int kOccWarmup = 256, kOccInterval = 32;   // iterations before the first refresh / between refreshes

// Per-level geometry of the multiresolution hash grid (tcnn grid.h; SURVEY TCNN-A1..A4).
struct LevelTable {
    uint32_t offset[kMaxLevels + 1];   // entry offsets, offset[L] = total entries
    uint32_t res[kMaxLevels];
    float scale[kMaxLevels];
};

// Per-level constants of the closed-form corner index used by the fused kernels.  Built on the host by replaying
// tcnn's grid_index stride loop in uint32 arithmetic (including its wrap-around: with base 16, scale 2, T=16 the
// level with res = 65536 gets strides {1, 65536, 0} and is NOT hashed -- the quirk is kept for parity):
//   index = hashed ? (x ^ y*my ^ z*mz) : (x + y*my + z*mz);  index &= mask;  if (index >= size) index -= size;
// mask = size-1 for power-of-two sizes (exact modulo), else ~0 with the conditional subtract (dense, index < 2*size).
struct LevelFast {
    float scale[kMaxLevels];
    uint32_t size[kMaxLevels], my[kMaxLevels], mz[kMaxLevels], mask[kMaxLevels], hashed[kMaxLevels], offset[kMaxLevels + 1];
    // fixed-point unit of the exact LDS gradient accumulation (k_grid_scatter, k_big_accum): 2^24 for loss_scale <= 128, coarser by the
    // same power of two for larger loss scales; every contribution is clamped to +-fix_clamp first (|contribution| * fix_scale < 2^31)
    float fix_scale, fix_clamp;
};

__host__ __device__ inline uint32_t fast_grid_index(const LevelFast& lf, int l, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t idx = (lf.hashed[l] ? (x ^ (y * lf.my[l]) ^ (z * lf.mz[l])) : (x + y * lf.my[l] + z * lf.mz[l])) & lf.mask[l];
    idx -= (idx >= lf.size[l]) ? lf.size[l] : 0u;
    return idx < lf.size[l] ? idx : lf.size[l] - 1u;
}

// Static shape of one object's network; passed by value to kernels.
struct NetDims {
    int L;          // hash levels
    int Epad;       // encoded width padded to 16
    int W;          // hidden width (32 / 64)
    int NH;         // hidden layers (1 / 2)
    uint32_t n_mlp; // MLP parameter count; grid params follow in the flat parameter vector
};

struct Aabb { float mn[3]; float mx[3]; };
struct Mat4 { float m[16]; };   // column-major: M(r,c) = m[c*4+r]
struct Intrinsics { float fx, fy, cx, cy; int H, W; };

// ---------------------------------------------------------------- counter RNG (bit-exact host/device)
// Replaces the three curandGenerateUniform streams of GenerateBatch (nerf_model.cu:1432,1434,1468).
enum { kStreamXY = 0, kStreamColor = 1, kStreamDt = 2, kStreamRender = 3 };
__host__ __device__ inline float rand01(uint64_t seed, uint32_t stream, uint32_t step, uint32_t idx) {
    uint64_t ctr = ((uint64_t)stream << 60) | ((uint64_t)step << 28) | (uint64_t)(idx & 0x0fffffffu);
    uint64_t z = ctr + seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- hash grid index (tcnn grid_index / grid_hash)
__host__ __device__ inline uint32_t grid_index(uint32_t size, uint32_t res, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t stride = 1, index = 0;
    if (stride <= size) { index += x * stride; stride *= res; }
    if (stride <= size) { index += y * stride; stride *= res; }
    if (stride <= size) { index += z * stride; stride *= res; }
    if (size < stride) index = x ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % size;
}

// ---------------------------------------------------------------- geometry
__host__ __device__ inline void rot3(const float* M, const float* v, float* o) {
    o[0] = fmaf(M[8], v[2], fmaf(M[4], v[1], M[0] * v[0]));
    o[1] = fmaf(M[9], v[2], fmaf(M[5], v[1], M[1] * v[0]));
    o[2] = fmaf(M[10], v[2], fmaf(M[6], v[1], M[2] * v[0]));
}
// Slab test, nerf_model.cu:87-138. Returns false on a miss.
__host__ __device__ inline bool ray_intersect(const Aabb& b, const float* o, const float* d, float& t0, float& t1) {
    float tmin = (b.mn[0] - o[0]) / d[0], tmax = (b.mx[0] - o[0]) / d[0], t;
    if (tmin > tmax) { t = tmin; tmin = tmax; tmax = t; }
    float tymin = (b.mn[1] - o[1]) / d[1], tymax = (b.mx[1] - o[1]) / d[1];
    if (tymin > tymax) { t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b.mn[2] - o[2]) / d[2], tzmax = (b.mx[2] - o[2]) / d[2];
    if (tzmin > tzmax) { t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    t0 = tmin; t1 = tmax;
    return true;
}
// Pixel -> ray in the object frame (nerf_model.cu:403-413 train, :467-477 render, :511-518 video).
__host__ __device__ inline void pixel_ray(const Intrinsics& K, float px, float py, const float* Twc, const float* Tow,
                                          bool pose_is_Toc, float* o, float* d, float& dnorm) {
    float dc[3] = { (px - K.cx) / K.fx, (py - K.cy) / K.fy, 1.0f };
    float n = sqrtf(fmaf(dc[2], dc[2], fmaf(dc[1], dc[1], dc[0] * dc[0])));
    float dn[3] = { dc[0] / n, dc[1] / n, dc[2] / n }, dw[3];
    rot3(Twc, dn, dw);
    if (!pose_is_Toc) {
        rot3(Tow, dw, d);
        float ow[3] = { Twc[12], Twc[13], Twc[14] }, t[3];
        rot3(Tow, ow, t);
        o[0] = t[0] + Tow[12]; o[1] = t[1] + Tow[13]; o[2] = t[2] + Tow[14];
    } else {
        d[0] = dw[0]; d[1] = dw[1]; d[2] = dw[2];
        o[0] = Twc[12]; o[1] = Twc[13]; o[2] = Twc[14];
    }
    dnorm = n;
}

// h(a * b) as the contract and the oracle define it: the fp32 product ROUNDED, then rounded to fp16 (tcnn's `(T)(weight * grad)`, the reference's
// `(network_precision_t)(loss_scale * ...)`).  Written plainly, the compiler folds the multiply and the conversion into v_fma_mixlo_f16, which rounds the exact
// product once -- one result in ~2^13 then differs by an fp16 ulp.  The product passes through an opaque register.
__device__ __forceinline__ float opaque_f32(float v) { asm volatile("" : "+v"(v)); return v; }

// ---------------------------------------------------------------- activations, nerf_model.cu:22-64
__device__ inline float logistic_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ inline float clamp_f(float x, float a, float b) { return fminf(fmaxf(x, a), b); }

}  // namespace mon
