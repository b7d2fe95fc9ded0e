// kernels_layers.hip -- MFMA layer-at-a-time kernels for the network shapes the fused kernels do not take (round 6): tcnn FullyFusedMLP's 16 neurons, 2 x 128,
// three and four hidden layers (CORE/configs/base.json:30-36 is the user's to edit; call sites CORE/src/nerf_model.cu:1557 forward, :1604 backward).  Until
// round 5 those shapes ran kernels_net.hip's one-sample-per-thread MLP and its LDS-staged weight-gradient kernel: 0.24 / 0.74 / 1.28 ms per base.json-sized
// step for 16 neurons / 3 x 64 / 2 x 128, of which k_weight_grad alone was 29-54 % and the two MLP kernels 7-40 %.
//
// One launch for the whole forward pass and one for the backward pass, one wave per 32 samples, v_mfma_f32_32x32x16_f16 with SAMPLES ON N (a layer's C/D
// fragment is the next product's B fragment: activations stay in registers between layers):
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

// ------------------------------------------------------------------ the hand-over between two layers
// A layer's C/D fragment holds, on half-wave h, register r = the unit frag_row(r, h) of its 32-unit block.  Taken as they are, registers 0..7 / 8..15 of block mb
// are the B fragments of K blocks 2 mb / 2 mb + 1 of the NEXT product if that product's K slots are numbered to match: slot (kb, h, j) = unit
// 32 (kb >> 1) + 16 (kb & 1) + 8 (j >> 2) + 4 h + (j & 3).  The A fragment of such a K block is then two 8-byte pieces of a weight row -- units U0 .. U0 + 3 and
// U0 + 8 .. U0 + 11 with U0 = 32 (kb >> 1) + 16 (kb & 1) + 4 h -- so activations never move across lanes or through LDS between layers (the fused kernels' trick,
// fused_device.h, without a pre-permuted fragment image: the permutation is in the LDS read addresses).
__device__ __forceinline__ half8_t frag_a_perm(const half_t* row, int kb, int h) {
    const int u0 = 32 * (kb >> 1) + 16 * (kb & 1) + 4 * h;
    const half4_t lo = *reinterpret_cast<const half4_t*>(row + u0), hi = *reinterpret_cast<const half4_t*>(row + u0 + 8);
    return half8_t{ lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
}

// A 32-unit x 32-sample block (C/D fragment: lo = registers 0..7, hi = 8..15) into T layout: through a per-wave LDS tile [unit][sample] (row stride 40 halves:
// 16-byte aligned rows, the two half-waves' rows on different banks), read back as 16-byte pieces of eight samples -- lanes 0..31 take consecutive units of one
// sample block, so a store instruction writes 512 contiguous bytes (2-byte stores straight from the registers wrote eight 16-byte pieces per instruction and
// were a third of the two kernels' time).  DS operations of a wave execute in order: no barrier between the writes and the reads.
constexpr int kTRow = 40;
__device__ __forceinline__ void store_t_block(half_t* scr, const half8_t& lo, const half8_t& hi, half_t* __restrict__ XT, uint32_t s0, int ubase, int F, int lane) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[frag_row(r, h) * kTRow + n] = r < 8 ? lo[r] : hi[r - 8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = lane + 64 * i, unit = p & 31, sb = p >> 5;
        const half8_t v = *reinterpret_cast<const half8_t*>(scr + unit * kTRow + 8 * sb);
        if (ubase + unit < F) *reinterpret_cast<half8_t*>(XT + (((size_t)(s0 >> 3) + (size_t)sb) * (size_t)F + (size_t)(ubase + unit)) * 8u) = v;
    }
}

// ------------------------------------------------------------------ forward of the whole network: one launch, one wave per 32 samples
// O = W_out relu(W_{NH-1} ... relu(W_0 E)); every layer's activations also go out row-major (Hid: the backward pass masks with them) and in T layout (HidT, ET:
// the weight gradients).  Hid == nullptr: inference, nothing but O is written.
// e_soa != nullptr: the encoded features come from k_encode_tiles' [L][n] half2 layout (four 4-byte loads per 16-wide K block instead of one 16-byte load)
// and are ALSO written row-major to E_out (what the debug read-back and the tests see).
template <int EPAD, int W>
__global__ void __launch_bounds__(256) k_mlp_fwd_all(const half_t* __restrict__ params, int NH, const half_t* __restrict__ E, half_t* __restrict__ Hid,
                                                     half_t* __restrict__ O, half_t* __restrict__ ET, half_t* __restrict__ HidT, uint32_t n,
                                                     const DevState* __restrict__ st, const half2_t* __restrict__ e_soa, int L, half_t* __restrict__ E_out,
                                                     uint16_t* __restrict__ relu_bits /* whole steps: the ReLU masks as bits instead of the row-major activations */) {
    constexpr int KB0 = EPAD / 16, MB = (W + 31) / 32, KBW = W / 16, kMaxHid = W == 128 ? 1 : 3, kLds = W * EPAD + kMaxHid * W * W + kOut * W;
    __shared__ __attribute__((aligned(16))) half_t w[kLds];
    __shared__ __attribute__((aligned(16))) half_t tscr[4][32 * kTRow];
    if (st && st->n_valid == 0u) return;
    const int n_mlp = W * EPAD + (NH - 1) * W * W + kOut * W;
    for (int i = threadIdx.x * 8; i < n_mlp; i += 256 * 8) *reinterpret_cast<half8_t*>(w + i) = *reinterpret_cast<const half8_t*>(params + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5, ld = NH * W;
    const uint32_t tiles = n >> 5;
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < tiles; t += gridDim.x * 4u) {
        const uint32_t s = t * 32u + (uint32_t)m;
        half8_t bf[KB0];
        if (e_soa) {
#pragma unroll
            for (int kb = 0; kb < KB0; ++kb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int level = 8 * kb + 4 * h + i; half2_t v = { (half_t)0.f, (half_t)0.f };       // (zero padding beyond L levels)
                    if (level < L) v = e_soa[(size_t)level * n + s];
                    bf[kb][2 * i] = v.x; bf[kb][2 * i + 1] = v.y; }
                *reinterpret_cast<half8_t*>(E_out + (size_t)s * EPAD + 16 * kb + 8 * h) = bf[kb];
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < KB0; ++kb) bf[kb] = *reinterpret_cast<const half8_t*>(E + (size_t)s * EPAD + 16 * kb + 8 * h);
        }
        if (ET) {
#pragma unroll
            for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) ET[t_index(s, 16 * kb + 8 * h + j, EPAD)] = bf[kb][j];
        }
        half8_t hb[2 * MB];                                   // the current layer's activations as the next product's B fragments
        // relu + rounding of a block's accumulators; stores; the block's two B fragments
        const auto finish = [&](const f16acc& acc, int mb, int layer, half8_t& lo, half8_t& hi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const half_t v = (half_t)fmaxf(acc[r], 0.f); if (r < 8) lo[r] = v; else hi[r - 8] = v; }
            if (relu_bits) {          // bit r: the ROUNDED activation of register r is positive (what the backward pass masks with), one 16-bit word per lane
                uint32_t bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) bits |= ((float)(r < 8 ? lo[r] : hi[r - 8]) > 0.f) ? (1u << r) : 0u;
                relu_bits[((size_t)(layer * MB + mb) * tiles + t) * 64u + (uint32_t)lane] = (uint16_t)bits;
            }
            if (Hid) {
#pragma unroll
                for (int q = 0; q < 4 && !relu_bits; ++q) {
                    const int u0 = 32 * mb + 4 * h + 8 * q; if (u0 >= W) continue;
                    const half4_t o = q < 2 ? half4_t{ lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3] }
                                            : half4_t{ hi[4 * q - 8], hi[4 * q - 7], hi[4 * q - 6], hi[4 * q - 5] };
                    *reinterpret_cast<half4_t*>(Hid + (size_t)s * ld + (size_t)layer * W + u0) = o;
                }
                if (HidT) store_t_block(tscr[wave], lo, hi, HidT + (size_t)layer * n * W, t * 32u, 32 * mb, W, lane);
            }
        };
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {                     // layer 0: K = the encoded features in their natural order
            f16acc acc = { 0 }; const int row = 32 * mb + m;
#pragma unroll
            for (int kb = 0; kb < KB0; ++kb) { half8_t a = {}; if (row < W) a = *reinterpret_cast<const half8_t*>(w + row * EPAD + 16 * kb + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[kb], acc, 0, 0, 0); }
            finish(acc, mb, 0, hb[2 * mb], hb[2 * mb + 1]);
        }
        for (int l = 1; l < NH; ++l) {                        // hidden -> hidden
            const half_t* wl = w + W * EPAD + (l - 1) * W * W;
            half8_t nb[2 * MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f16acc acc = { 0 }; const int row = 32 * mb + m;
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb) { half8_t a = {}; if (row < W) a = frag_a_perm(wl + row * W, kb, h);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, hb[kb], acc, 0, 0, 0); }
                finish(acc, mb, l, nb[2 * mb], nb[2 * mb + 1]);
            }
#pragma unroll
            for (int k = 0; k < 2 * MB; ++k) hb[k] = nb[k];
        }
        {   // output layer: four rows, no activation
            const half_t* wo = w + W * EPAD + (NH - 1) * W * W;
            f16acc acc = { 0 };
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) { half8_t a = {}; if (m < kOut) a = frag_a_perm(wo + m * W, kb, h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, hb[kb], acc, 0, 0, 0); }
            if (h == 0) *reinterpret_cast<half4_t*>(O + (size_t)s * kOut) = half4_t{ (half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3] };
        }
    }
}

// ------------------------------------------------------------------ backward of the whole network (dh, dE): one launch
// dh_{NH-1} = relu'(h_{NH-1}) * W_out^T dO;  dh_{l-1} = relu'(h_{l-1}) * W_l^T dh_l;  dE = W_0^T dh_0.  The transposed matrices are built once per workgroup in LDS;
// every dh also goes out in T layout (dHidT, dOT) for the weight gradients.
template <int EPAD, int W>
__global__ void __launch_bounds__(256) k_mlp_bwd_all(const half_t* __restrict__ params, int NH, const half_t* __restrict__ Hid, const half_t* __restrict__ dO,
                                                     half_t* __restrict__ dHid, half_t* __restrict__ dE, half_t* __restrict__ dOT, half_t* __restrict__ dHidT,
                                                     uint32_t n, const DevState* __restrict__ st, int keep_rowmajor /* 0: nobody reads dHid row-major (whole steps) */,
                                                     const uint16_t* __restrict__ relu_bits /* the forward pass's mask bits; nullptr: masks from row-major Hid */,
                                                     BinsOut bins, int L) {
    constexpr int MB = (W + 31) / 32, KBW = W / 16, kMaxHid = W == 128 ? 1 : 3, kOffHid = W * 16, kOff0 = kOffHid + kMaxHid * W * W, kLds = kOff0 + EPAD * W;
    __shared__ __attribute__((aligned(16))) half_t wt[kLds];      // W_out^T [W][16] (columns 4..15 zero) | W_l^T [W][W], l = 1 .. NH-1 | W_0^T [EPAD][W]
    __shared__ __attribute__((aligned(16))) half_t tscr[4][32 * kTRow];
    if (st->n_valid == 0u) return;
    {
        const half_t* wo = params + W * EPAD + (NH - 1) * W * W;
        for (int i = threadIdx.x; i < W * 16; i += 256) { const int u = i >> 4, c = i & 15; wt[i] = c < kOut ? wo[c * W + u] : (half_t)0.f; }
        for (int l = 1; l < NH; ++l) { const half_t* wl = params + W * EPAD + (l - 1) * W * W; half_t* d = wt + kOffHid + (l - 1) * W * W;
            for (int i = threadIdx.x; i < W * W; i += 256) { const int k = i / W, u = i - k * W; d[i] = wl[u * W + k]; } }
        for (int i = threadIdx.x; i < EPAD * W; i += 256) { const int k = i / W, u = i - k * W; wt[kOff0 + i] = params[u * EPAD + k]; }
    }
    __syncthreads();
    if (bins.de_soa && blockIdx.x == 0u && threadIdx.x < bins.n_bins) bins.st->n_scatter[scatter_counter(bins.st->iter, threadIdx.x)] = n / bins.n_bins;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5, ld = NH * W;
    const uint32_t tiles = n >> 5;
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < tiles; t += gridDim.x * 4u) {
        const uint32_t s = t * 32u + (uint32_t)m;
        half8_t bo = {};
        if (h == 0) { const half4_t v = *reinterpret_cast<const half4_t*>(dO + (size_t)s * kOut); bo[0] = v[0]; bo[1] = v[1]; bo[2] = v[2]; bo[3] = v[3];
#pragma unroll
            for (int c = 0; c < kOut; ++c) dOT[t_index(s, c, kOut)] = v[c]; }
        half8_t db[2 * MB];
        // mask with the layer's activations, round, store row-major + T layout, hand the block over as two B fragments
        const auto finish = [&](const f16acc& acc, int mb, int layer, half8_t& lo, half8_t& hi) {
            uint32_t bits = 0u;
            if (relu_bits) bits = relu_bits[((size_t)(layer * MB + mb) * tiles + t) * 64u + (uint32_t)lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 32 * mb + 4 * h + 8 * q;
                half4_t o = { (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f };
                if (k0 < W) {
                    if (relu_bits) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = (half_t)(((bits >> (4 * q + c)) & 1u) ? acc[4 * q + c] : 0.f);
                    } else {
                        const half4_t act = *reinterpret_cast<const half4_t*>(Hid + (size_t)s * ld + (size_t)layer * W + k0);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = (half_t)(((float)act[c] > 0.f) ? acc[4 * q + c] : 0.f);
                    }
                    if (keep_rowmajor) *reinterpret_cast<half4_t*>(dHid + (size_t)s * ld + (size_t)layer * W + k0) = o;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { if (q < 2) lo[4 * q + c] = o[c]; else hi[4 * q - 8 + c] = o[c]; }
            }
            store_t_block(tscr[wave], lo, hi, dHidT + (size_t)layer * n * W, t * 32u, 32 * mb, W, lane);
        };
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {                     // dh of the last hidden layer: K = the four outputs (one K block, natural order)
            f16acc acc = { 0 }; const int row = 32 * mb + m;
            half8_t a = {}; if (row < W) a = *reinterpret_cast<const half8_t*>(wt + row * 16 + 8 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bo, acc, 0, 0, 0);
            finish(acc, mb, NH - 1, db[2 * mb], db[2 * mb + 1]);
        }
        for (int l = NH - 1; l >= 1; --l) {
            const half_t* wl = wt + kOffHid + (l - 1) * W * W;
            half8_t nb[2 * MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f16acc acc = { 0 }; const int row = 32 * mb + m;
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb) { half8_t a = {}; if (row < W) a = frag_a_perm(wl + row * W, kb, h);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, db[kb], acc, 0, 0, 0); }
                finish(acc, mb, l - 1, nb[2 * mb], nb[2 * mb + 1]);
            }
#pragma unroll
            for (int k = 0; k < 2 * MB; ++k) db[k] = nb[k];
        }
        {   // dE: EPAD <= 32 rows, no mask
            f16acc acc = { 0 };
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) { half8_t a = {}; if (m < EPAD) a = frag_a_perm(wt + kOff0 + m * W, kb, h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, db[kb], acc, 0, 0, 0); }
            if (bins.de_soa) {
                // k_rows_to_bins folded in (S = 32: tile t is ray t): the ray's bin is t & (bins - 1), the sample's slot (t / bins) * 32 + its number in the ray;
                // the half4 at features 4 h + 8 q holds levels 2 h + 4 q and 2 h + 4 q + 1
                const uint32_t slot = (t & (bins.n_bins - 1u)) * (n / bins.n_bins) + (t / bins.n_bins) * 32u + (uint32_t)m;
                half2_t* de = reinterpret_cast<half2_t*>(bins.de_soa);
                const auto cl = [&](float v) { return (half_t)clamp_f((float)(half_t)v, -bins.clampv, bins.clampv); };
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int l0 = 2 * h + 4 * q;
                    if (l0 < L) de[(size_t)l0 * n + slot] = half2_t{ cl(acc[4 * q]), cl(acc[4 * q + 1]) };
                    if (l0 + 1 < L) de[(size_t)(l0 + 1) * n + slot] = half2_t{ cl(acc[4 * q + 2]), cl(acc[4 * q + 3]) }; }
                if (h == 0) reinterpret_cast<float4_t*>(bins.x_soa)[slot] = float4_t{ bins.pts[3 * (size_t)s], bins.pts[3 * (size_t)s + 1], bins.pts[3 * (size_t)s + 2], 0.f };
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k0 = 4 * h + 8 * q; if (k0 >= EPAD) continue;
                *reinterpret_cast<half4_t*>(dE + (size_t)s * EPAD + k0) = half4_t{ (half_t)acc[4 * q], (half_t)acc[4 * q + 1], (half_t)acc[4 * q + 2],
                        (half_t)acc[4 * q + 3] }; }
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
    // the cross-wave sum of the (at most two) blocks of a job whose waves split the samples: a slice per sample share, summed after a barrier (LDS float
    // atomics run at ~0.35 per clock and CU: 4096 of them per workgroup were 5-6 us of the small shapes' launch)
    __shared__ float red[4][2 * 1024];
    if (st->n_valid == 0u) return;
    const WgradJob jb = jobs.j[blockIdx.y];
    const half_t* __restrict__ AT = jb.AT; const half_t* __restrict__ BT = jb.BT; const int rows = jb.rows, cols = jb.cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    const int MBt = (rows + 31) >> 5, NBt = (cols + 31) >> 5, TB = MBt * NBt;
    const int nsub = TB >= 4 ? 1 : 4 / TB, sub = TB >= 4 ? 0 : wave / TB, b0 = TB >= 4 ? wave : wave % TB;
    const uint32_t c0 = blockIdx.x * chunk, c1 = min(c0 + chunk, n), per = ((c1 - c0) / 16u + (uint32_t)nsub - 1u) / (uint32_t)nsub * 16u;
    const uint32_t s_lo = c0 + (uint32_t)sub * per, s_hi = min(s_lo + per, c1);
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
        for (int r = 0; r < 16; ++r) red[sub][bi * 1024 + frag_row(r, h) * 32 + m] = acc[0][r];
        __syncthreads();
        for (int i = threadIdx.x; i < TB * 1024; i += 256) { const int bq = i >> 10, rr = (i >> 5) & 31, cc = i & 31, mb = bq / NBt, nb = bq - mb * NBt;
            const int u = 32 * mb + rr, k = 32 * nb + cc;
            float v = red[0][i]; for (int q = 1; q < nsub; ++q) v += red[q][i];
            if (u < rows && k < cols) out[u * cols + k] = v; }
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

constexpr uint32_t kWgradChunk = 512;          // samples per workgroup of k_weight_grad_mfma: 256 workgroups per layer at base.json's batch
static uint32_t wgrad_rows(uint32_t n) { return (n + kWgradChunk - 1u) / kWgradChunk; }

// T-layout workspace of a batch of n samples: ET [Epad] | dOT [4] | HidT [NH][W] | dHidT [NH][W], each n halfs per feature
// followed by the weight-gradient partials, fp32 [chunks][n_mlp]
static size_t layers_t_halves(const NetDims& nd, uint32_t n) { return (((size_t)n * (size_t)(nd.Epad + kOut + 2 * nd.NH * nd.W)) + 7u) & ~(size_t)7u; }
// ... the weight-gradient partials, fp32 [chunks][n_mlp], and the ReLU mask bits, uint16 [layer][unit block][tile][64 lanes]
static size_t layers_part_halves(const NetDims& nd, uint32_t n) { return 2u * (size_t)wgrad_rows(n) * nd.n_mlp; }
static size_t layers_bits_halves(const NetDims& nd, uint32_t n) { return (size_t)nd.NH * (size_t)((nd.W + 31) / 32) * 2u * (size_t)n; }
size_t layers_workspace_halves(const NetDims& nd, uint32_t n) { return layers_t_halves(nd, n) + layers_part_halves(nd, n) + layers_bits_halves(nd, n); }
struct LayerT { uint16_t *ET, *dOT, *HidT, *dHidT; float* partials; uint16_t* bits; };
static LayerT layer_t(const NetDims& nd, uint16_t* ws, uint32_t n) {
    LayerT t; t.ET = ws; t.dOT = t.ET + (size_t)n * nd.Epad; t.HidT = t.dOT + (size_t)n * kOut; t.dHidT = t.HidT + (size_t)n * nd.NH * nd.W;
    t.partials = reinterpret_cast<float*>(ws + layers_t_halves(nd, n)); t.bits = ws + layers_t_halves(nd, n) + layers_part_halves(nd, n); return t; }

// parameter offsets: W0 [W][Epad] | W_1 .. W_{NH-1} [W][W] | W_out [4][W]  (kernels_net.hip / frag_layout.h)
static size_t w_off(const NetDims& nd, int layer) { return layer == 0 ? 0 : (size_t)nd.W * nd.Epad + (size_t)(layer - 1) * nd.W * nd.W; }

#define MON_LAYERS_DISPATCH(CALL) do { switch (nd.Epad * 1000 + nd.W) { \
        case 16016: { CALL(16, 16); return true; } case 16032: { CALL(16, 32); return true; } case 16064: { CALL(16, 64); return true; } \
        case 16128: { CALL(16, 128); return true; } case 32016: { CALL(32, 16); return true; } case 32032: { CALL(32, 32); return true; } \
        case 32064: { CALL(32, 64); return true; } case 32128: { CALL(32, 128); return true; } default: return false; } } while (0)
static bool layers_shape_ok(const NetDims& nd, uint32_t n) { return nd.W <= kLayerMaxW && (n & 31u) == 0u && nd.NH >= 1 && nd.NH <= (nd.W == 128 ? 2 : 4); }

bool launch_mlp_forward_layers(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n,
        const DevState* st, uint16_t* ws_T, const uint16_t* e_soa, uint16_t* E_out, bool keep_rowmajor) {
    if (!layers_shape_ok(nd, n)) return false;
    LayerT t{}; if (ws_T && Hid) t = layer_t(nd, ws_T, n);
    auto H = [](const uint16_t* p) { return reinterpret_cast<const half_t*>(p); }; auto Hm = [](uint16_t* p) { return reinterpret_cast<half_t*>(p); };
#define MON_FWD_ALL(E_, W_) hipLaunchKernelGGL((k_mlp_fwd_all<E_, W_>), dim3(layer_grid(n)), dim3(256), 0, s, H(params), nd.NH, H(E), Hm(Hid), Hm(O), Hm(t.ET), \
        Hm(t.HidT), n, st, reinterpret_cast<const half2_t*>(e_soa), nd.L, Hm(E_out), (ws_T && Hid && !keep_rowmajor) ? t.bits : nullptr)
    MON_LAYERS_DISPATCH(MON_FWD_ALL);
#undef MON_FWD_ALL
}
bool launch_mlp_backward_layers(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st, uint16_t* ws_T, bool keep_rowmajor, const BinsOut* bins) {
    if (!layers_shape_ok(nd, n) || !ws_T) return false;
    const BinsOut bo = bins ? *bins : BinsOut{};
    const LayerT t = layer_t(nd, ws_T, n);
    auto H = [](const uint16_t* p) { return reinterpret_cast<const half_t*>(p); }; auto Hm = [](uint16_t* p) { return reinterpret_cast<half_t*>(p); };
#define MON_BWD_ALL(E_, W_) hipLaunchKernelGGL((k_mlp_bwd_all<E_, W_>), dim3(layer_grid(n)), dim3(256), 0, s, H(params), nd.NH, H(Hid), H(dO), Hm(dHid), Hm(dE), \
        Hm(t.dOT), Hm(t.dHidT), n, st, keep_rowmajor ? 1 : 0, keep_rowmajor ? nullptr : t.bits, bo, nd.L)
    MON_LAYERS_DISPATCH(MON_BWD_ALL);
#undef MON_BWD_ALL
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
