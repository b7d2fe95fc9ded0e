// kernels_net.hip -- UNFUSED network path (backend 0): the straightforward, layer-at-a-time kernels.
// It exists as the on-device cross-check of the fused MFMA kernel (kernels_fused.hip) and as the
// general fallback for shapes the fused kernel does not cover.  It restates tiny-cuda-nn's
// kernel_grid / kernel_mlp_fused / kernel_grid_backward (absent submodule; call sites
// CORE/src/nerf_model.cu:1557,1604) with the rounding points fixed in DESIGN.md (numeric model).
// The MLP kernels are instantiated per (encoder padding, width, hidden layers); the instantiations are spread over three translation units so that the
// build compiles them side by side (this file: 32 / 64 neurons with one or two hidden layers, the shapes of BASELINE; kernels_net_wide.hip: 16 and 128
// neurons; kernels_net_deep.hip: three and four hidden layers) -- MON_NET_PART selects which dispatch table a unit carries.
#include "device_common.h"
#include "model.h"
#ifndef MON_NET_PART
#define MON_NET_PART 0
#endif

namespace mon {

#if MON_NET_PART == 0
// ------------------------------------------------------------------ hash-grid encode
// One thread per (sample, level).  Gathers 8 corners x half2, fp32 fmaf chain, one rounding to fp16.
__global__ void __launch_bounds__(256) k_encode(LevelTable lt, NetDims nd, const uint16_t* __restrict__ params, const float* __restrict__ pts,
                                                uint16_t* __restrict__ E, uint32_t n, const DevState* __restrict__ st) {
    if (st && st->n_valid == 0u) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t half_e = (uint32_t)nd.Epad / 2u;             // half2 slots per sample (levels + zero pad)
    const uint32_t s = t / half_e, l = t - s * half_e;
    if (s >= n) return;
    half2_t* out = reinterpret_cast<half2_t*>(E) + (size_t)s * half_e + l;
    if (l >= (uint32_t)nd.L) { *out = half2_t{ (half_t)0.f, (half_t)0.f }; return; }    // TCNN-A9 zero padding
    const half2_t* table = reinterpret_cast<const half2_t*>(params + nd.n_mlp);
    const float scale = lt.scale[l]; const uint32_t res = lt.res[l], off = lt.offset[l], size = lt.offset[l + 1] - off;
    float pos[3]; uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float p = fmaf(scale, pts[3 * (size_t)s + d], 0.5f), fl = floorf(p); pg[d] = (uint32_t)(int32_t)fl; pos[d] = p - fl; }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float w = 1.f; uint32_t q[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { if (k & (1 << d)) { w *= pos[d]; q[d] = pg[d] + 1u; } else { w *= 1.f - pos[d]; q[d] = pg[d]; } }
        const half2_t v = table[off + grid_index(size, res, q[0], q[1], q[2])];
        a0 = fmaf(w, (float)v.x, a0); a1 = fmaf(w, (float)v.y, a1);
    }
    *out = half2_t{ (half_t)a0, (half_t)a1 };
}

#endif  // MON_NET_PART == 0

// ------------------------------------------------------------------ MLP forward, one thread per sample
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_mlp_forward(const uint16_t* __restrict__ params, uint32_t n_mlp, const uint16_t* __restrict__ E,
                                                     uint16_t* __restrict__ Hid, uint16_t* __restrict__ O, uint32_t n, const DevState* __restrict__ st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* w = reinterpret_cast<half_t*>(smem);
    if (st && st->n_valid == 0u) return;
    for (uint32_t i = threadIdx.x; i < n_mlp; i += blockDim.x) w[i] = reinterpret_cast<const half_t*>(params)[i];
    __syncthreads();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    half_t in[W > EPAD ? W : EPAD];
    {
        const half8_t* src = reinterpret_cast<const half8_t*>(E + (size_t)s * EPAD);
#pragma unroll
        for (int k = 0; k < EPAD / 8; ++k) { const half8_t v = src[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) in[8 * k + j] = v[j]; }
    }
    const half_t* wl = w;
    half_t hid[W];
#pragma unroll
    for (int layer = 0; layer < NH; ++layer) {
        const int nin = (layer == 0) ? EPAD : W;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < nin; ++k) a = fmaf((float)wl[u * nin + k], (float)in[k], a);
            hid[u] = (half_t)fmaxf(a, 0.f);
        }
        if (Hid) {
            half8_t* dst = reinterpret_cast<half8_t*>(Hid + ((size_t)s * NH + layer) * W);
#pragma unroll
            for (int k = 0; k < W / 8; ++k) { half8_t v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hid[8 * k + j];
                dst[k] = v; }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) in[k] = hid[k];
        wl += W * nin;
    }
    half4_t o;
#pragma unroll
    for (int c = 0; c < kOut; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < W; ++k) a = fmaf((float)wl[c * W + k], (float)in[k], a);
        o[c] = (half_t)a;
    }
    *reinterpret_cast<half4_t*>(O + (size_t)s * kOut) = o;
}

// ------------------------------------------------------------------ MLP backward (dh, dE), one thread per sample
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_mlp_backward(const uint16_t* __restrict__ params, uint32_t n_mlp, const uint16_t* __restrict__ Hid,
                                                      const uint16_t* __restrict__ dO, uint16_t* __restrict__ dHid, uint16_t* __restrict__ dE,
                                                      uint32_t n, const DevState* __restrict__ st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* w = reinterpret_cast<half_t*>(smem);
    if (st->n_valid == 0u) return;
    for (uint32_t i = threadIdx.x; i < n_mlp; i += blockDim.x) w[i] = reinterpret_cast<const half_t*>(params)[i];
    __syncthreads();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const half_t* w0 = w; const half_t* wout = w + W * EPAD + (NH - 1) * W * W;
    const half4_t d_o = *reinterpret_cast<const half4_t*>(dO + (size_t)s * kOut);
    half_t dh[W];
    {
        const half_t* h = reinterpret_cast<const half_t*>(Hid) + ((size_t)s * NH + (NH - 1)) * W;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < kOut; ++c) a = fmaf((float)wout[c * W + u], (float)d_o[c], a);
            dh[u] = (half_t)(((float)h[u] > 0.f) ? a : 0.f);
        }
    }
#pragma unroll
    for (int layer = NH - 1; layer >= 0; --layer) {
        half8_t* dst = reinterpret_cast<half8_t*>(dHid + ((size_t)s * NH + layer) * W);
#pragma unroll
        for (int k = 0; k < W / 8; ++k) { half8_t v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dh[8 * k + j];
            dst[k] = v; }
        if (layer == 0) break;
        const half_t* wl = w + W * EPAD + (layer - 1) * W * W;       // maps layer-1 -> layer
        const half_t* hp = reinterpret_cast<const half_t*>(Hid) + ((size_t)s * NH + (layer - 1)) * W;
        float acc[W];
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = 0.f;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const float d = (float)dh[u];
#pragma unroll
            for (int k = 0; k < W; ++k) acc[k] = fmaf((float)wl[u * W + k], d, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) dh[k] = (half_t)(((float)hp[k] > 0.f) ? acc[k] : 0.f);
    }
    float acc[EPAD];
#pragma unroll
    for (int k = 0; k < EPAD; ++k) acc[k] = 0.f;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const float d = (float)dh[u];
#pragma unroll
        for (int k = 0; k < EPAD; ++k) acc[k] = fmaf((float)w0[u * EPAD + k], d, acc[k]);
    }
    half8_t* dst = reinterpret_cast<half8_t*>(dE + (size_t)s * EPAD);
#pragma unroll
    for (int k = 0; k < EPAD / 8; ++k) { half8_t v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)acc[8 * k + j];
        dst[k] = v; }
}

#if MON_NET_PART == 0
// ------------------------------------------------------------------ weight gradients
// G[row][col] += sum_s A[s][row] * Bm[s][col]   (A: lda halves per sample, Bm: ldb), fp32.
// One block per chunk of kChunk samples; chunk staged in LDS; one fp32 atomic per output per block.
constexpr int kChunk = 128, kMaxWidth = 128;
__global__ void __launch_bounds__(256) k_weight_grad(const uint16_t* __restrict__ A, int lda, int rows, const uint16_t* __restrict__ Bm, int ldb, int cols,
                                                     float* __restrict__ G, uint32_t n, const DevState* __restrict__ st) {
    __shared__ half_t sa[kChunk * kMaxWidth];      // rows, cols <= the widest layer (128 neurons)
    __shared__ half_t sb[kChunk * kMaxWidth];
    if (st->n_valid == 0u) return;
    const uint32_t s0 = blockIdx.x * kChunk;
    const uint32_t cnt = min((uint32_t)kChunk, n - s0);
    for (uint32_t i = threadIdx.x; i < cnt * (uint32_t)rows; i += blockDim.x) { const uint32_t s = i / rows, r = i - s * rows;
        sa[i] = reinterpret_cast<const half_t*>(A)[(size_t)(s0 + s) * lda + r]; }
    for (uint32_t i = threadIdx.x; i < cnt * (uint32_t)cols; i += blockDim.x) { const uint32_t s = i / cols, c = i - s * cols;
        sb[i] = reinterpret_cast<const half_t*>(Bm)[(size_t)(s0 + s) * ldb + c]; }
    __syncthreads();
    for (uint32_t o = threadIdx.x; o < (uint32_t)(rows * cols); o += blockDim.x) {
        const uint32_t r = o / cols, c = o - r * cols;
        float a = 0.f;
        for (uint32_t s = 0; s < cnt; ++s) a = fmaf((float)sa[s * rows + r], (float)sb[s * cols + c], a);
        if (a != 0.f) atomicAdd(&G[o], a);
    }
}

// ------------------------------------------------------------------ grid backward (tcnn kernel_grid_backward)
// One thread per (sample, level): 8 x global_atomic_pk_add_f16 of h(w * dE).
__global__ void __launch_bounds__(256) k_grid_backward(LevelTable lt, NetDims nd, const float* __restrict__ pts, const uint16_t* __restrict__ dE,
                                                       uint16_t* __restrict__ ggrid, uint32_t n, const DevState* __restrict__ st) {
    if (st->n_valid == 0u) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t / (uint32_t)nd.L, l = t - s * (uint32_t)nd.L;
    if (s >= n) return;
    const half2_t g = reinterpret_cast<const half2_t*>(dE + (size_t)s * nd.Epad)[l];
    const float g0 = (float)g.x, g1 = (float)g.y;
    if (g0 == 0.f && g1 == 0.f) return;
    const float scale = lt.scale[l]; const uint32_t res = lt.res[l], off = lt.offset[l], size = lt.offset[l + 1] - off;
    float pos[3]; uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float p = fmaf(scale, pts[3 * (size_t)s + d], 0.5f), fl = floorf(p); pg[d] = (uint32_t)(int32_t)fl; pos[d] = p - fl; }
    typedef __attribute__((address_space(1))) half2_t gh2;
    gh2* table = (gh2*)reinterpret_cast<half2_t*>(ggrid);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float w = 1.f; uint32_t q[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { if (k & (1 << d)) { w *= pos[d]; q[d] = pg[d] + 1u; } else { w *= 1.f - pos[d]; q[d] = pg[d]; } }
        const uint32_t idx = off + grid_index(size, res, q[0], q[1], q[2]);
        __builtin_amdgcn_global_atomic_fadd_v2f16(table + idx, half2_t{ (half_t)(w * g0), (half_t)(w * g1) });
    }
}

// ------------------------------------------------------------------ launchers
void launch_encode(hipStream_t s, const LevelTable& lt, const NetDims& nd, const uint16_t* params, const float* pts, uint16_t* E, uint32_t n,
        const DevState* st) {
    const uint64_t threads = (uint64_t)n * (nd.Epad / 2);
    hipLaunchKernelGGL(k_encode, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, lt, nd, params, pts, E, n, st);
}

#endif  // MON_NET_PART == 0

template <int EPAD, int W, int NH>
static void mlp_fwd_t(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n, const DevState* st) {
    hipLaunchKernelGGL((k_mlp_forward<EPAD, W, NH>), dim3((n + 255) / 256), dim3(256), nd.n_mlp * 2, s, params, nd.n_mlp, E, Hid, O, n, st);
}
template <int EPAD, int W, int NH>
static void mlp_bwd_t(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st) {
    hipLaunchKernelGGL((k_mlp_backward<EPAD, W, NH>), dim3((n + 255) / 256), dim3(256), nd.n_mlp * 2, s, params, nd.n_mlp, Hid, dO, dHid, dE, n, st);
}
#define MON_CASE(FN, E, W_, N, ...) case E * 1000 + W_ * 10 + N: FN<E, W_, N>(__VA_ARGS__); return true;
// (one table per translation unit; returns false when the shape is not in it)
#if MON_NET_PART == 0
#define MON_DISPATCH(FN, ...) do { switch (nd.Epad * 1000 + nd.W * 10 + nd.NH) { \
        MON_CASE(FN, 16, 32, 1, __VA_ARGS__) MON_CASE(FN, 16, 32, 2, __VA_ARGS__) MON_CASE(FN, 16, 64, 1, __VA_ARGS__) MON_CASE(FN, 16, 64, 2, __VA_ARGS__) MON_CASE(FN, 32, 32, 1, __VA_ARGS__) MON_CASE(FN, 32, 32, 2, __VA_ARGS__) MON_CASE(FN, 32, 64, 1, __VA_ARGS__) MON_CASE(FN, 32, 64, 2, __VA_ARGS__) \
        default: return false; } } while (0)
#elif MON_NET_PART == 1
/* tcnn FullyFusedMLP's other widths (base.json:30-36 is user-editable): 16 and 128 neurons */
#define MON_DISPATCH(FN, ...) do { switch (nd.Epad * 1000 + nd.W * 10 + nd.NH) { \
        MON_CASE(FN, 16, 16, 1, __VA_ARGS__) MON_CASE(FN, 16, 16, 2, __VA_ARGS__) MON_CASE(FN, 32, 16, 1, __VA_ARGS__) MON_CASE(FN, 32, 16, 2, __VA_ARGS__) MON_CASE(FN, 16, 128, 1, __VA_ARGS__) MON_CASE(FN, 16, 128, 2, __VA_ARGS__) MON_CASE(FN, 32, 128, 1, __VA_ARGS__) MON_CASE(FN, 32, 128, 2, __VA_ARGS__) \
        default: return false; } } while (0)
#else
/* three and four hidden layers (tcnn takes any count; base.json:35 has one) for the widths up to 64 */
#define MON_DISPATCH(FN, ...) do { switch (nd.Epad * 1000 + nd.W * 10 + nd.NH) { \
        MON_CASE(FN, 16, 16, 3, __VA_ARGS__) MON_CASE(FN, 16, 16, 4, __VA_ARGS__) MON_CASE(FN, 32, 16, 3, __VA_ARGS__) MON_CASE(FN, 32, 16, 4, __VA_ARGS__) MON_CASE(FN, 16, 32, 3, __VA_ARGS__) MON_CASE(FN, 16, 32, 4, __VA_ARGS__) MON_CASE(FN, 32, 32, 3, __VA_ARGS__) MON_CASE(FN, 32, 32, 4, __VA_ARGS__) \
        MON_CASE(FN, 16, 64, 3, __VA_ARGS__) MON_CASE(FN, 16, 64, 4, __VA_ARGS__) MON_CASE(FN, 32, 64, 3, __VA_ARGS__) MON_CASE(FN, 32, 64, 4, __VA_ARGS__) \
        default: return false; } } while (0)
#endif

#if MON_NET_PART == 0
bool mlp_forward_part1(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n, const DevState* st);
bool mlp_forward_part2(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n, const DevState* st);
bool mlp_backward_part1(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE, uint32_t n,
        const DevState* st);
bool mlp_backward_part2(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE, uint32_t n,
        const DevState* st);
static bool mlp_forward_part0(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n,
        const DevState* st) { MON_DISPATCH(mlp_fwd_t, s, nd, params, E, Hid, O, n, st); }
static bool mlp_backward_part0(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st) { MON_DISPATCH(mlp_bwd_t, s, nd, params, Hid, dO, dHid, dE, n, st); }
void launch_mlp_forward(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n,
        const DevState* st) {
    (void)(mlp_forward_part0(s, nd, params, E, Hid, O, n, st) || mlp_forward_part1(s, nd, params, E, Hid, O, n, st)
            || mlp_forward_part2(s, nd, params, E, Hid, O, n, st));          // (config.cpp admits exactly the shapes of the three tables)
}
void launch_mlp_backward(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st) {
    (void)(mlp_backward_part0(s, nd, params, Hid, dO, dHid, dE, n, st) || mlp_backward_part1(s, nd, params, Hid, dO, dHid, dE, n, st)
            || mlp_backward_part2(s, nd, params, Hid, dO, dHid, dE, n, st));
}
void launch_weight_grads(hipStream_t s, const NetDims& nd, const uint16_t* E, const uint16_t* Hid, const uint16_t* dHid, const uint16_t* dO, float* gmlp,
        uint32_t n, const DevState* st) {
    const dim3 grid((n + kChunk - 1) / kChunk), block(256);
    const int W = nd.W, NH = nd.NH, ld = NH * W;
    if (W > kMaxWidth || nd.Epad > kMaxWidth) return;          // (k_weight_grad stages kChunk x kMaxWidth halves per operand; config.cpp admits no wider network)
    // layer 0: dW0[u][k] = sum dh0[u] * E[k]
    hipLaunchKernelGGL(k_weight_grad, grid, block, 0, s, dHid, ld, W, E, nd.Epad, nd.Epad, gmlp, n, st);
    for (int layer = 1; layer < NH; ++layer)
        hipLaunchKernelGGL(k_weight_grad, grid, block, 0, s, dHid + layer * W, ld, W, Hid + (layer - 1) * W, ld, W, gmlp + W * nd.Epad + (layer - 1) * W * W,
                n, st);
    // output layer: rows 0..3 only (dO rows 4..15 are identically zero)
    hipLaunchKernelGGL(k_weight_grad, grid, block, 0, s, dO, kOut, kOut, Hid + (NH - 1) * W, ld, W, gmlp + W * nd.Epad + (NH - 1) * W * W, n, st);
}
void launch_grid_backward(hipStream_t s, const LevelTable& lt, const NetDims& nd, const float* pts, const uint16_t* dE, uint16_t* ggrid, uint32_t n,
        const DevState* st) {
    const uint64_t threads = (uint64_t)n * nd.L;
    hipLaunchKernelGGL(k_grid_backward, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, lt, nd, pts, dE, ggrid, n, st);
}

#elif MON_NET_PART == 1
bool mlp_forward_part1(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n, const DevState* st) {
    MON_DISPATCH(mlp_fwd_t, s, nd, params, E, Hid, O, n, st); }
bool mlp_backward_part1(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE, uint32_t n,
        const DevState* st) { MON_DISPATCH(mlp_bwd_t, s, nd, params, Hid, dO, dHid, dE, n, st); }
#else
bool mlp_forward_part2(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n, const DevState* st) {
    MON_DISPATCH(mlp_fwd_t, s, nd, params, E, Hid, O, n, st); }
bool mlp_backward_part2(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE, uint32_t n,
        const DevState* st) { MON_DISPATCH(mlp_bwd_t, s, nd, params, Hid, dO, dHid, dE, n, st); }
#endif

}  // namespace mon
