// kernels_layers.hip -- MFMA layer-at-a-time kernels for the network shapes the fused kernels do not take (round 6): tcnn FullyFusedMLP's 16 neurons, 2 x 128,
// three and four hidden layers (CORE/configs/base.json:30-36 is the user's to edit; call sites CORE/src/nerf_model.cu:1557 forward, :1604 backward).  Until
// round 5 those shapes ran kernels_net.hip's one-sample-per-thread MLP and its LDS-staged weight-gradient kernel: 0.24 / 0.74 / 1.28 ms per base.json-sized
// step for 16 neurons / 3 x 64 / 2 x 128, of which k_weight_grad alone was 29-54 % and the two MLP kernels 7-40 %.
//
// One launch per layer, one wave per 32 samples, v_mfma_f32_32x32x16_f16 with SAMPLES ON N:
//   forward   Out^T[units x 32]  = W[units x K]   . In^T[K x 32]        A = rows of W (16-byte LDS reads), B = a sample's K features (16-byte global loads)
//   backward  dIn^T[K x 32]      = W^T[K x units] . dAct^T[units x 32]  A = rows of W^T (transposed once per workgroup into LDS), masked by the ReLU of In
//   dW[units x K] = sum over samples dAct[s][unit] In[s][k]: SAMPLES ON K -- both operands then want eight consecutive samples of one feature in 16 bytes,
//     which the row-major activations do not have.  The producers therefore write every tensor a weight gradient reads a second time in "T layout"
//     XT[s / 8][feature][s % 8] (2-byte stores from the registers they hold anyway); k_weight_grad_mfma then feeds the MFMA straight from global memory with
//     one 16-byte load per operand -- no LDS, no transposition pass (the LDS-staged kernel was bound by its 2-byte LDS reads, HISTORY 0, r05 row 7).
// Numerics are kernels_net.hip's: fp16 operands, fp32 accumulation (in MFMA order instead of k-ascending fmaf order), one rounding to fp16 per activation.
#include "device_common.h"
#include "model.h"

namespace mon {

constexpr int kLayerMaxW = 128;
typedef float f16acc __attribute__((ext_vector_type(16)));

// unit / row that register r of the C/D fragment holds on half-wave h (v_mfma_f32_32x32x16_f16; checked by mon_selftest_mfma)
__device__ __forceinline__ int frag_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// XT[s / 8][f][s % 8]
__device__ __forceinline__ size_t t_index(uint32_t s, int f, int F) { return ((size_t)(s >> 3) * (size_t)F + (size_t)f) * 8u + (s & 7u); }

// ------------------------------------------------------------------ forward: Out[s][u] = act(sum_k W[u][k] In[s][k])
template <int KB /* K / 16 */, int MB /* ceil(units / 32) */>
__global__ void __launch_bounds__(256) k_layer_fwd(const half_t* __restrict__ Wg, int nout, const half_t* __restrict__ In, int ld_in, half_t* __restrict__ Out,
                                                   int ld_out, half_t* __restrict__ OutT, half_t* __restrict__ InT, int relu, uint32_t n,
                                                   const DevState* __restrict__ st) {
    __shared__ __attribute__((aligned(16))) half_t w[kLayerMaxW * kLayerMaxW];
    if (st && st->n_valid == 0u) return;
    constexpr int nin = 16 * KB;
    for (int i = threadIdx.x * 8; i < nout * nin; i += 256 * 8) *reinterpret_cast<half8_t*>(w + i) = *reinterpret_cast<const half8_t*>(Wg + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const uint32_t tiles = n >> 5;
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < tiles; t += gridDim.x * 4u) {
        const uint32_t s = t * 32u + (uint32_t)m;
        half8_t bf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) bf[kb] = *reinterpret_cast<const half8_t*>(In + (size_t)s * ld_in + 16 * kb + 8 * h);
        if (InT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) InT[t_index(s, 16 * kb + 8 * h + j, nin)] = bf[kb][j];
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f16acc acc = { 0 };
            const int row = 32 * mb + m;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                half8_t a = {};
                if (row < nout) a = *reinterpret_cast<const half8_t*>(w + row * nin + 16 * kb + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[kb], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int u0 = 32 * mb + 4 * h + 8 * q;
                if (u0 >= nout) continue;
                half4_t o;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float v = acc[4 * q + c]; o[c] = (half_t)(relu ? fmaxf(v, 0.f) : v); }
                *reinterpret_cast<half4_t*>(Out + (size_t)s * ld_out + u0) = o;
                if (OutT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) OutT[t_index(s, u0 + c, nout)] = o[c];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ backward: dIn[s][k] = relu'(InAct[s][k]) * sum_u W[u][k] dAct[s][u]
template <int KB /* ceil(units / 16): the K of this product */, int MB /* ceil(nin / 32) */>
__global__ void __launch_bounds__(256) k_layer_bwd(const half_t* __restrict__ Wg, int nout, int nin, const half_t* __restrict__ dAct, int ld_d, int d_valid,
                                                   const half_t* __restrict__ InAct, int ld_a, half_t* __restrict__ dIn, int ld_o, half_t* __restrict__ dInT,
                                                   half_t* __restrict__ dActT, uint32_t n, const DevState* __restrict__ st) {
    __shared__ __attribute__((aligned(16))) half_t wt[kLayerMaxW * kLayerMaxW];      // W^T [nin][16 KB], units beyond nout zero
    if (st->n_valid == 0u) return;
    constexpr int kp = 16 * KB;
    for (int i = threadIdx.x; i < nin * kp; i += 256) { const int k = i / kp, u = i - k * kp; wt[i] = u < nout ? Wg[u * nin + k] : (half_t)0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const uint32_t tiles = n >> 5;
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < tiles; t += gridDim.x * 4u) {
        const uint32_t s = t * 32u + (uint32_t)m;
        half8_t bf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int f0 = 16 * kb + 8 * h; bf[kb] = half8_t{};
            if (f0 + 8 <= d_valid) bf[kb] = *reinterpret_cast<const half8_t*>(dAct + (size_t)s * ld_d + f0);
            else if (f0 + 4 <= d_valid) { const half4_t v = *reinterpret_cast<const half4_t*>(dAct + (size_t)s * ld_d + f0);      // dL/dO: four values per sample
                bf[kb][0] = v[0]; bf[kb][1] = v[1]; bf[kb][2] = v[2]; bf[kb][3] = v[3]; }
        }
        if (dActT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int f = 16 * kb + 8 * h + j; if (f < d_valid) dActT[t_index(s, f, d_valid)] = bf[kb][j]; }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f16acc acc = { 0 };
            const int row = 32 * mb + m;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                half8_t a = {};
                if (row < nin) a = *reinterpret_cast<const half8_t*>(wt + row * kp + 16 * kb + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[kb], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 32 * mb + 4 * h + 8 * q;
                if (k0 >= nin) continue;
                half4_t act = { (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f };
                if (InAct) act = *reinterpret_cast<const half4_t*>(InAct + (size_t)s * ld_a + k0);
                half4_t o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (half_t)(((float)act[c] > 0.f) ? acc[4 * q + c] : 0.f);
                *reinterpret_cast<half4_t*>(dIn + (size_t)s * ld_o + k0) = o;
                if (dInT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) dInT[t_index(s, k0 + c, nin)] = o[c];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ weight gradients: G[u][k] = sum_s dAct[s][u] In[s][k], operands in T layout
// ONE launch for all layers: blockIdx.y = layer (a job: operands, shape, where its matrix sits in the parameter vector), blockIdx.x = a chunk of samples.  The
// (rows / 32) x (cols / 32) output blocks go round the four waves (up to four each); with fewer than four blocks the waves split the chunk's samples instead
// and meet in LDS.  A workgroup leaves its sums as one row of a PARTIALS buffer [chunks][n_mlp] (plain coalesced stores: one fp32 atomic per output and
// workgroup was 0.26-1 M atomics per layer at the chip's ~21 G/s); k_wgrad_reduce sums the rows.  K loop: the operands of kUnroll 16-sample steps are
// requested together (one step per trip was one memory round trip per MFMA: 49 us for a 64 x 64 layer on 64 workgroups).
struct WgradJob { const half_t* AT; const half_t* BT; int rows, cols; uint32_t out_off; };
struct WgradJobs { WgradJob j[5]; int n; };
constexpr int kWgUnroll = 4;
__global__ void __launch_bounds__(256) k_weight_grad_mfma(WgradJobs jobs, float* __restrict__ partials, uint32_t n_mlp, uint32_t n, uint32_t chunk,
                                                          const DevState* __restrict__ st) {
    __shared__ float red[2 * 1024];                         // the cross-wave sum of the (at most two) blocks of a job whose waves split the samples
    if (st->n_valid == 0u) return;
    const WgradJob jb = jobs.j[blockIdx.y];
    const half_t* __restrict__ AT = jb.AT; const half_t* __restrict__ BT = jb.BT; const int rows = jb.rows, cols = jb.cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const int MBt = (rows + 31) >> 5, NBt = (cols + 31) >> 5, TB = MBt * NBt;
    const int nsub = TB >= 4 ? 1 : 4 / TB, sub = TB >= 4 ? 0 : wave / TB, b0 = TB >= 4 ? wave : wave % TB;
    const uint32_t c0 = blockIdx.x * chunk, c1 = min(c0 + chunk, n), per = ((c1 - c0) / 16u + (uint32_t)nsub - 1u) / (uint32_t)nsub * 16u;
    const uint32_t s_lo = c0 + (uint32_t)sub * per, s_hi = min(s_lo + per, c1);
    if (nsub > 1) { for (int i = threadIdx.x; i < TB * 1024; i += 256) red[i] = 0.f; __syncthreads(); }
    f16acc acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f16acc{ 0 };
    for (uint32_t s16 = s_lo; s16 < s_hi; s16 += 16u * kWgUnroll) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int bi = b0 + 4 * b; if (bi >= TB) continue;                          // uniform
            const int mb = bi / NBt, nb = bi - mb * NBt, u = 32 * mb + m, k = 32 * nb + m;
            half8_t a[kWgUnroll], bb[kWgUnroll];
#pragma unroll
            for (int q = 0; q < kWgUnroll; ++q) {
                const uint32_t sq = s16 + 16u * (uint32_t)q; const size_t blk = (size_t)(sq >> 3) + (size_t)h; a[q] = half8_t{}; bb[q] = half8_t{};
                if (sq < s_hi && u < rows) a[q] = *reinterpret_cast<const half8_t*>(AT + (blk * (size_t)rows + (size_t)u) * 8u);
                if (sq < s_hi && k < cols) bb[q] = *reinterpret_cast<const half8_t*>(BT + (blk * (size_t)cols + (size_t)k) * 8u);
            }
#pragma unroll
            for (int q = 0; q < kWgUnroll; ++q) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], bb[q], acc[b], 0, 0, 0);
        }
    }
    float* out = partials + (size_t)blockIdx.x * n_mlp + jb.out_off;
    if (nsub > 1) {
        const int bi = b0;
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(&red[bi * 1024 + frag_row(r, h) * 32 + m], acc[0][r]);
        __syncthreads();
        for (int i = threadIdx.x; i < TB * 1024; i += 256) { const int bq = i >> 10, rr = (i >> 5) & 31, cc = i & 31, mb = bq / NBt, nb = bq - mb * NBt;
            const int u = 32 * mb + rr, k = 32 * nb + cc; if (u < rows && k < cols) out[u * cols + k] = red[i]; }
        return;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int bi = b0 + 4 * b; if (bi >= TB) continue;
        const int mb = bi / NBt, nb = bi - mb * NBt, k = 32 * nb + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int u = 32 * mb + frag_row(r, h); if (u < rows && k < cols) out[u * cols + k] = acc[b][r]; }
    }
}
// gmlp[p] = sum over the chunks' partial rows (gmlp holds zeros before: the optimizer clears what it read).  64 parameters x 16 row groups per workgroup: a
// thread sums every 16th row (four independent accumulators), the groups meet in LDS -- one thread per parameter walking all 256 rows took 22-34 us
__global__ void __launch_bounds__(1024) k_wgrad_reduce(const float* __restrict__ partials, uint32_t n_rows, uint32_t n_mlp, float* __restrict__ gmlp,
        const DevState* __restrict__ st) {
    __shared__ float red[16][64];
    if (st->n_valid == 0u) return;
    const uint32_t c = threadIdx.x & 63u, g = threadIdx.x >> 6, p = blockIdx.x * 64u + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (p < n_mlp) {
        uint32_t r = g;
        for (; r + 48u < n_rows; r += 64u) { a0 += partials[(size_t)r * n_mlp + p]; a1 += partials[(size_t)(r + 16u) * n_mlp + p];
            a2 += partials[(size_t)(r + 32u) * n_mlp + p]; a3 += partials[(size_t)(r + 48u) * n_mlp + p]; }
        for (; r < n_rows; r += 16u) a0 += partials[(size_t)r * n_mlp + p];
    }
    red[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0u && p < n_mlp) { float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][c];
        gmlp[p] = v; }
}

// ------------------------------------------------------------------ launchers
static uint32_t layer_grid(uint32_t n) { const uint32_t tiles = n >> 5, wgs = (tiles + 3u) / 4u; return wgs < 1u ? 1u : (wgs > 1024u ? 1024u : wgs); }

static bool launch_layer_fwd(hipStream_t s, const uint16_t* W, int nout, int nin, const uint16_t* In, int ld_in, uint16_t* Out, int ld_out, uint16_t* OutT,
        uint16_t* InT, int relu, uint32_t n, const DevState* st) {
    const int KB = nin / 16, MB = (nout + 31) / 32;
#define MON_FWD(K_, M_) if (KB == K_ && MB == M_) { hipLaunchKernelGGL((k_layer_fwd<K_, M_>), dim3(layer_grid(n)), dim3(256), 0, s, \
        reinterpret_cast<const half_t*>(W), nout, reinterpret_cast<const half_t*>(In), ld_in, reinterpret_cast<half_t*>(Out), ld_out, \
        reinterpret_cast<half_t*>(OutT), reinterpret_cast<half_t*>(InT), relu, n, st); return true; }
    MON_FWD(1, 1) MON_FWD(2, 1) MON_FWD(4, 1) MON_FWD(8, 1) MON_FWD(1, 2) MON_FWD(2, 2) MON_FWD(4, 2) MON_FWD(8, 2) MON_FWD(1, 4) MON_FWD(2, 4) MON_FWD(4, 4)
    MON_FWD(8, 4)
#undef MON_FWD
    return false;
}
static bool launch_layer_bwd(hipStream_t s, const uint16_t* W, int nout, int nin, const uint16_t* dAct, int ld_d, int d_valid, const uint16_t* InAct, int ld_a,
        uint16_t* dIn, int ld_o, uint16_t* dInT, uint16_t* dActT, uint32_t n, const DevState* st) {
    const int KB = (nout + 15) / 16, MB = (nin + 31) / 32;
#define MON_BWD(K_, M_) if (KB == K_ && MB == M_) { hipLaunchKernelGGL((k_layer_bwd<K_, M_>), dim3(layer_grid(n)), dim3(256), 0, s, \
        reinterpret_cast<const half_t*>(W), nout, nin, reinterpret_cast<const half_t*>(dAct), ld_d, d_valid, reinterpret_cast<const half_t*>(InAct), ld_a, \
        reinterpret_cast<half_t*>(dIn), ld_o, reinterpret_cast<half_t*>(dInT), reinterpret_cast<half_t*>(dActT), n, st); return true; }
    MON_BWD(1, 1) MON_BWD(2, 1) MON_BWD(4, 1) MON_BWD(8, 1) MON_BWD(1, 2) MON_BWD(2, 2) MON_BWD(4, 2) MON_BWD(8, 2) MON_BWD(1, 4) MON_BWD(2, 4) MON_BWD(4, 4)
    MON_BWD(8, 4)
#undef MON_BWD
    return false;
}
constexpr uint32_t kWgradChunk = 512;          // samples per workgroup of k_weight_grad_mfma: 256 workgroups per layer at base.json's batch
static uint32_t wgrad_rows(uint32_t n) { return (n + kWgradChunk - 1u) / kWgradChunk; }

// T-layout workspace of a batch of n samples: ET [Epad] | dOT [4] | HidT [NH][W] | dHidT [NH][W], each n halfs per feature
// followed by the weight-gradient partials, fp32 [chunks][n_mlp]
static size_t layers_t_halves(const NetDims& nd, uint32_t n) { return (((size_t)n * (size_t)(nd.Epad + kOut + 2 * nd.NH * nd.W)) + 7u) & ~(size_t)7u; }
size_t layers_workspace_halves(const NetDims& nd, uint32_t n) { return layers_t_halves(nd, n) + 2u * (size_t)wgrad_rows(n) * nd.n_mlp; }
struct LayerT { uint16_t *ET, *dOT, *HidT, *dHidT; float* partials; };
static LayerT layer_t(const NetDims& nd, uint16_t* ws, uint32_t n) {
    LayerT t; t.ET = ws; t.dOT = t.ET + (size_t)n * nd.Epad; t.HidT = t.dOT + (size_t)n * kOut; t.dHidT = t.HidT + (size_t)n * nd.NH * nd.W;
    t.partials = reinterpret_cast<float*>(ws + layers_t_halves(nd, n)); return t; }

// parameter offsets: W0 [W][Epad] | W_1 .. W_{NH-1} [W][W] | W_out [4][W]  (kernels_net.hip / frag_layout.h)
static size_t w_off(const NetDims& nd, int layer) { return layer == 0 ? 0 : (size_t)nd.W * nd.Epad + (size_t)(layer - 1) * nd.W * nd.W; }

bool launch_mlp_forward_layers(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n,
        const DevState* st, uint16_t* ws_T) {
    if (nd.W > kLayerMaxW || (n & 31u) || !Hid) return false;
    const int W = nd.W, NH = nd.NH, ld = NH * W; LayerT t{}; if (ws_T) t = layer_t(nd, ws_T, n);
    bool ok = launch_layer_fwd(s, params, W, nd.Epad, E, nd.Epad, Hid, ld, ws_T ? t.HidT : nullptr, ws_T ? t.ET : nullptr, 1, n, st);
    for (int l = 1; l < NH && ok; ++l)
        ok = launch_layer_fwd(s, params + w_off(nd, l), W, W, Hid + (size_t)(l - 1) * W, ld, Hid + (size_t)l * W, ld, ws_T ? t.HidT + (size_t)l * n * W : nullptr, nullptr, 1, n, st);
    if (ok) ok = launch_layer_fwd(s, params + w_off(nd, NH), kOut, W, Hid + (size_t)(NH - 1) * W, ld, O, kOut, nullptr, nullptr, 0, n, st);
    return ok;
}
bool launch_mlp_backward_layers(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st, uint16_t* ws_T) {
    if (nd.W > kLayerMaxW || (n & 31u) || !ws_T) return false;
    const int W = nd.W, NH = nd.NH, ld = NH * W; const LayerT t = layer_t(nd, ws_T, n);
    // dh_last = relu'(h_last) * W_out^T dO
    bool ok = launch_layer_bwd(s, params + w_off(nd, NH), kOut, W, dO, kOut, kOut, Hid + (size_t)(NH - 1) * W, ld, dHid + (size_t)(NH - 1) * W, ld,
            t.dHidT + (size_t)(NH - 1) * n * W, t.dOT, n, st);
    for (int l = NH - 1; l >= 1 && ok; --l)
        ok = launch_layer_bwd(s, params + w_off(nd, l), W, W, dHid + (size_t)l * W, ld, W, Hid + (size_t)(l - 1) * W, ld, dHid + (size_t)(l - 1) * W, ld,
                t.dHidT + (size_t)(l - 1) * n * W, nullptr, n, st);
    if (ok) ok = launch_layer_bwd(s, params, W, nd.Epad, dHid, ld, W, nullptr, 0, dE, nd.Epad, nullptr, nullptr, n, st);
    return ok;
}
void launch_weight_grads_layers(hipStream_t s, const NetDims& nd, float* gmlp, uint32_t n, const DevState* st, uint16_t* ws_T) {
    const int W = nd.W, NH = nd.NH; const LayerT t = layer_t(nd, ws_T, n);
    auto H = [](const uint16_t* p) { return reinterpret_cast<const half_t*>(p); };
    WgradJobs jobs{}; jobs.n = NH + 1;
    jobs.j[0] = WgradJob{ H(t.dHidT), H(t.ET), W, nd.Epad, 0u };
    for (int l = 1; l < NH; ++l) jobs.j[l] = WgradJob{ H(t.dHidT + (size_t)l * n * W), H(t.HidT + (size_t)(l - 1) * n * W), W, W, (uint32_t)w_off(nd, l) };
    jobs.j[NH] = WgradJob{ H(t.dOT), H(t.HidT + (size_t)(NH - 1) * n * W), kOut, W, (uint32_t)w_off(nd, NH) };
    const uint32_t rows = wgrad_rows(n);
    hipLaunchKernelGGL(k_weight_grad_mfma, dim3(rows, (uint32_t)(NH + 1)), dim3(256), 0, s, jobs, t.partials, (uint32_t)nd.n_mlp, n, kWgradChunk, st);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(((uint32_t)nd.n_mlp + 63u) / 64u), dim3(1024), 0, s, t.partials, rows, (uint32_t)nd.n_mlp, gmlp, st);
}

}  // namespace mon
