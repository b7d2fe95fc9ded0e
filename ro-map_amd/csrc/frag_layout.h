// frag_layout.h -- where each MLP weight lives in the A-fragment image of the fused kernels (kernels_fused.hip header).
// One table, two directions: frag_source() says which parameter an image element holds (k_build_frag_image),
// frag_slots() says which image elements a parameter feeds (k_optimizer writes them as it updates the weight, so the
// steady-state training loop needs no fragment-building launch).  tests/test_abi.py checks that they are inverse.
#pragma once
#include "device_common.h"

namespace mon {

struct FragDims {
    int EPAD, W, NH, L;
    __host__ __device__ int MB() const { return W / 32; }
    __host__ __device__ int KS0() const { return EPAD / 16; }
    __host__ __device__ int KSW() const { return W / 16; }
    __host__ __device__ int LPH() const { return (L + 1) >> 1; }
    __host__ __device__ int F_W0() const { return 0; }
    __host__ __device__ int F_W1() const { return MB() * KS0(); }
    __host__ __device__ int F_WO() const { return F_W1() + (NH == 2 ? MB() * KSW() : 0); }
    __host__ __device__ int F_WOT() const { return F_WO() + KSW(); }
    __host__ __device__ int F_W1T() const { return F_WOT() + MB(); }
    __host__ __device__ int F_W0T() const { return F_W1T() + (NH == 2 ? MB() * KSW() : 0); }
    __host__ __device__ int N_FRAGS() const { return F_W0T() + KSW(); }
    __host__ __device__ int OFF_W1() const { return W * EPAD; }
    __host__ __device__ int OFF_WO() const { return W * EPAD + (NH - 1) * W * W; }
    __host__ __device__ int N_MLP() const { return OFF_WO() + kOutPad * W; }
};

// C/D row of register r in half h of a 32x32 MFMA tile, and the hidden unit carried by K-slot (k-step s, half h, element j)
__host__ __device__ inline int frag_rho(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__host__ __device__ inline int frag_unit_of_slot(int s, int h, int j) { return 32 * (s >> 1) + frag_rho(h, 8 * (s & 1) + j); }
// inverse of frag_unit_of_slot
__host__ __device__ inline void frag_slot_of_unit(int u, int& s, int& h, int& j) {
    const int v = u & 31, q = (v & 3) + 4 * (v >> 3);
    h = (v >> 2) & 1; s = 2 * (u >> 5) + (q >> 3); j = q & 7;
}

// image element idx = frag * 512 + lane * 8 + j  ->  index into the MLP parameter vector, or -1 for a structural zero
__host__ __device__ inline int frag_source(const FragDims& d, int idx) {
    const int frag = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7, m = lane & 31, h = lane >> 5, LPH = d.LPH();
    if (frag < d.F_W1()) {                                       // W0: rows = units, K slots = encoded features of the owning half
        const int mb = (frag - d.F_W0()) / d.KS0(), s = (frag - d.F_W0()) % d.KS0();
        const int il = 4 * s + (j >> 1), level = h * LPH + il;
        return (il < LPH && level < d.L) ? (32 * mb + m) * d.EPAD + 2 * level + (j & 1) : -1;
    }
    if (d.NH == 2 && frag < d.F_WO()) {                          // W1: rows = units of layer 1, K slots = units of layer 0
        const int mb = (frag - d.F_W1()) / d.KSW(), s = (frag - d.F_W1()) % d.KSW();
        return d.OFF_W1() + (32 * mb + m) * d.W + frag_unit_of_slot(s, h, j);
    }
    if (frag < d.F_WOT()) {                                      // Wout: 4 real rows of 32
        const int s = frag - d.F_WO();
        return (m < kOut) ? d.OFF_WO() + m * d.W + frag_unit_of_slot(s, h, j) : -1;
    }
    if (frag < d.F_W1T()) {                                      // Wout^T: rows = units, K slots 0..3 = output channels
        const int mb = frag - d.F_WOT(), c = 8 * h + j;
        return (c < kOut) ? d.OFF_WO() + c * d.W + 32 * mb + m : -1;
    }
    if (d.NH == 2 && frag < d.F_W0T()) {                         // W1^T: rows = units of layer 0, K slots = units of layer 1
        const int mb = (frag - d.F_W1T()) / d.KSW(), s = (frag - d.F_W1T()) % d.KSW();
        return d.OFF_W1() + frag_unit_of_slot(s, h, j) * d.W + 32 * mb + m;
    }
    {                                                            // W0^T: row m = (half hh, reg r) <-> local feature r of half hh
        const int s = frag - d.F_W0T();
        const int hh = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3), il = r >> 1, level = hh * LPH + il;
        return (il < LPH && level < d.L && r < d.EPAD / 2) ? frag_unit_of_slot(s, h, j) * d.EPAD + 2 * level + (r & 1) : -1;
    }
}

// parameter index p -> the image elements it feeds (at most 2: forward operand and transposed backward operand); returns the count
__host__ __device__ inline int frag_slots(const FragDims& d, int p, int out[2]) {
    const int LPH = d.LPH(); int n = 0, s, h, j;
    if (p < d.OFF_W1()) {                                        // W0[u][f]
        const int u = p / d.EPAD, f = p % d.EPAD, level = f >> 1, b = f & 1;
        if (level >= d.L) return 0;                              // pad feature: the image keeps its structural zero
        const int hh = level / LPH, il = level % LPH;
        out[n++] = (d.F_W0() + (u >> 5) * d.KS0() + (il >> 2)) * 512 + (hh * 32 + (u & 31)) * 8 + 2 * (il & 3) + b;
        const int r = 2 * il + b, m = (r & 3) + 4 * hh + 8 * (r >> 2);
        frag_slot_of_unit(u, s, h, j);
        out[n++] = (d.F_W0T() + s) * 512 + (h * 32 + m) * 8 + j;
        return n;
    }
    if (p < d.OFF_WO()) {                                        // W1[u2][u1] (NH == 2)
        const int q = p - d.OFF_W1(), u2 = q / d.W, u1 = q % d.W;
        frag_slot_of_unit(u1, s, h, j);
        out[n++] = (d.F_W1() + (u2 >> 5) * d.KSW() + s) * 512 + (h * 32 + (u2 & 31)) * 8 + j;
        frag_slot_of_unit(u2, s, h, j);
        out[n++] = (d.F_W1T() + (u1 >> 5) * d.KSW() + s) * 512 + (h * 32 + (u1 & 31)) * 8 + j;
        return n;
    }
    {                                                            // Wout[c][u], rows 4..15 are padding
        const int q = p - d.OFF_WO(), c = q / d.W, u = q % d.W;
        if (c >= kOut) return 0;
        frag_slot_of_unit(u, s, h, j);
        out[n++] = (d.F_WO() + s) * 512 + (h * 32 + c) * 8 + j;
        out[n++] = (d.F_WOT() + (u >> 5)) * 512 + (u & 31) * 8 + c;
        return n;
    }
}

// ---- dW partial rows of k_fused_train.  A workgroup writes its weight-gradient sums in ACCUMULATOR layout -- the order the MFMA C/D registers hold them --
// so that its epilogue is plain conflict-free 16-byte LDS stores and coalesced global stores; whoever sums the rows (k_grid_scatter, k_reduce_partials) maps a
// column to its parameter once, when it writes the total.  Column = ((tile * 4 + r / 4) * 64 + lane) * 4 + r % 4 for the 32x32 tiles of dW0 (tile = mb), then
// dW1 (tile = mb * MB + nb, NH == 2); dWout keeps its four real output columns only: ((mb * 4 + r / 4) * 8 + h * 4 + c) * 4 + r % 4.  Column acc_cols() = loss
// partial.
__host__ __device__ inline int acc_cols(const FragDims& d) { return d.MB() * 1024 + (d.NH == 2 ? d.MB() * d.MB() * 1024 : 0) + d.MB() * 128; }
__host__ __device__ inline int acc_off_w1(const FragDims& d) { return d.MB() * 1024; }
__host__ __device__ inline int acc_off_wo(const FragDims& d) { return d.MB() * 1024 + (d.NH == 2 ? d.MB() * d.MB() * 1024 : 0); }
// column -> index into the MLP parameter vector, or -1 (pad column of a narrow encoding)
__host__ __device__ inline int acc_param(const FragDims& d, int col) {
    if (col < acc_off_wo(d)) {
        const bool w1 = col >= acc_off_w1(d);
        const int q = (w1 ? col - acc_off_w1(d) : col) >> 2, r = 4 * ((q >> 6) & 3) + (col & 3), lane = q & 63, tile = q >> 8;
        const int n = lane & 31, h = lane >> 5;
        if (!w1) return n < d.EPAD ? (32 * tile + frag_rho(h, r)) * d.EPAD + n : -1;
        return d.OFF_W1() + (32 * (tile / d.MB()) + frag_rho(h, r)) * d.W + 32 * (tile % d.MB()) + n;
    }
    const int c0 = col - acc_off_wo(d), q = c0 >> 2, r = 4 * ((q >> 3) & 3) + (c0 & 3), h = (q >> 2) & 1, c = q & 3, mb = q >> 5;
    return d.OFF_WO() + c * d.W + 32 * mb + frag_rho(h, r);
}

}  // namespace mon
