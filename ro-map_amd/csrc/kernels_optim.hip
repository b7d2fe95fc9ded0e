// kernels_optim.hip -- tcnn Trainer::optimizer_step for Ema{ ExponentialDecay{ Adam } }
// (call site CORE/src/nerf_model.cu:1644,1681; hyper-parameters CORE/configs/base.json:5-22).
// One fused, grid-stride, 8-parameters-per-thread pass over the flat parameter vector: gradient read +
// reset (replaces the Overwrite memset of tcnn's backward), Adam on fp32 master weights, fp16 working
// copy, EMA shadow copy.  SURVEY TCNN-A6/A7/A8: grid entries with zero gradient are skipped by Adam (not
// by the EMA), L2 regularisation only on the MLP matrices, per-parameter step counters, debiased EMA.
// Untouched grid chunks cost 64 B per 8 parameters (fp16 grad, weight, EMA read + EMA write).
// The last block to finish advances the device-resident step / iteration counters and applies the
// exponential LR decay, so a whole training run needs no host synchronisation.
// Fused backend: the same launch also prepares the NEXT iteration -- the blocks that update the MLP matrices write the new
// fp16 weights straight into the MFMA A-fragment image (frag_layout.h), and `cand_blocks` extra blocks generate the next
// iteration's candidate rays (they depend on the iteration counter and the dataset only), so the steady-state loop is
// k_fused_train -> k_grid_scatter -> k_reduce_partials -> k_optimizer with no batch-generation launch.
// Variant builds for measurements (tools/variant_build.sh <tag> -D...; profiles/r03_scatter_levels.md): MON_OPT_ABLATE bits 1 no Adam arithmetic, 2 no
// partial-table reads, 4 no position blocks, 8 no tile-image stores.
#include <cstdlib>
#include "device_common.h"
#include "model.h"
#include "frag_layout.h"
#include "batch_device.h"
#include "encode_device.h"

namespace mon {

// Optimizer state is not read again before the next step.  Small tables (everything streamed once per step, working set inside the Infinity Cache):
// non-temporal stores, a wash against plain ones (round 2).  LARGE tables (T = 2^22: 2-3 GB of scattered 32-byte pieces per step, HBM-bound): plain stores --
// the L2 merges a chunk's pieces into whole lines before they leave; non-temporal ones cost 20 % of the kernel there (645-725 us against 535-550 us over steps
// 20..40, four runs each).
template <bool NT, class T> __device__ __forceinline__ void state_store(T v, T* p) {
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

__device__ __forceinline__ float adam_update(float g, float w, float& m1, float& m2, uint32_t& steps, float lr0, const OptimConst& oc, uint32_t step_cap) {
    const float gsq = g * g;
    m1 = oc.beta1 * m1 + (1.f - oc.beta1) * g;
    m2 = oc.beta2 * m2 + (1.f - oc.beta2) * gsq;
    // (16-bit counters saturate at 65535: both bias corrections are exactly 1.0f from far below that, ParamPtrs::steps16)
    const uint32_t cs = min(steps + 1u, step_cap); steps = cs;
    // beta^cs as exp2(cs * log2 beta): v_exp_f32-based, within ~3e-6 relative of powf for cs < 1e5
    const float lr = lr0 * sqrtf(1.f - exp2f((float)cs * oc.log2_beta2)) / (1.f - exp2f((float)cs * oc.log2_beta1));
    const float eff = lr / (sqrtf(m2) + oc.epsilon);
    return w - eff * m1;
}

// EMA of `k` steps during which the weight stayed w, in one go (debiased form): e(t+k) = (d^k e(t) (1 - d^t) + (1 - d^k) w) / (1 - d^(t+k)).
__device__ __forceinline__ void ema_catch_up(half8_t& e, const half8_t& w, uint32_t t, uint32_t k, float log2_d) {
    const float dk = exp2f((float)k * log2_d), a = dk * (1.f - exp2f((float)t * log2_d)), b = 1.f - dk, inv = 1.f / (1.f - exp2f((float)(t + k) * log2_d));
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (half_t)(((float)e[j] * a + (float)w[j] * b) * inv);
}

// LAZY (large tables, !DENSE): a grid chunk without a gradient is left alone entirely -- not even its EMA is touched; p.ema_step[chunk]
// remembers the optimizer step its EMA is current for, and the missing steps are applied in closed form when the chunk next receives a
// gradient or when the inference weights are needed (k_ema_finalize).  The weights and Adam state are exactly those of the eager
// schedule; the EMA differs from the step-by-step fp16 recurrence by rounding only.
// (LIVE: the position blocks also look the samples' occupancy cells up and compact the live ones -- a template parameter because the kernel's register count is the
// maximum over its paths: compiled into the default instantiation it cost a wave per SIMD, 96 -> 104 registers, and 1.4 us of the dense step)
template <bool DENSE, bool LAZY, bool ONE = false /* the grid covers every chunk with one thread: no second chunk's state to hold (44 registers less) */,
          bool LIVE = false>
__global__ void __launch_bounds__(256) k_optimizer(ParamPtrs p, OptimConst oc, const DevState* __restrict__ st, DevState* __restrict__ st_next, OptimNext nx,
        uint32_t lazy_below) {
    // block roles by VIRTUAL index: [0, extra) prepare the next iteration, the rest update parameters.  Physically the parameter blocks come first (they are
    // the ones that stream 80 MB and should be in flight from the first cycle), the short preparation blocks fill in behind them.
    const uint32_t n_extra = nx.cand_blocks + nx.pos_blocks, n_opt = gridDim.x - n_extra, vblock = blockIdx.x < n_opt ? blockIdx.x + n_extra
            : blockIdx.x - n_opt;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // (chunk records exist for the lazy optimizer of large tables only: compile-time null in the dense instantiations, whose record branches fold away)
    float* const prec = LAZY ? p.rec : nullptr;
    const uint32_t step_cap = (p.steps16 || prec) ? 65535u : 0xffffffffu;
    // where a chunk's optimizer state lives: the four SoA arrays, or (large tables, ParamPtrs::rec) ONE 128-byte record per chunk -- master | m1 | m2 | step
    // counters -- so that a touched chunk among untouched ones costs one full line instead of four half-used 64-byte sectors
    auto st_master = [&](uint32_t c) -> float* { return prec ? prec + 32u * (size_t)c : p.master + 8u * (size_t)c; };
    auto st_m1 = [&](uint32_t c) -> float* { return prec ? prec + 32u * (size_t)c + 8u : p.m1 + 8u * (size_t)c; };
    auto st_m2 = [&](uint32_t c) -> float* { return prec ? prec + 32u * (size_t)c + 16u : p.m2 + 8u * (size_t)c; };
    auto st_steps16 = [&](uint32_t c) -> uint16_t* { return prec ? reinterpret_cast<uint16_t*>(prec + 32u * (size_t)c + 24u) : p.steps16 + 8u * (size_t)c; };
    // lazy EMA: the optimizer step a chunk's EMA is current for -- in the chunk record's pad word (the same 128-byte line as the state: late in training a touched
    // chunk's own 64-byte line of a separate array was a seventh of what the kernel moved), or in the array
    auto ema_step_of = [&](uint32_t c) -> uint32_t* { return prec ? reinterpret_cast<uint32_t*>(prec + 32u * (size_t)c + 28u) : p.ema_step + c; };
    // a chunk's eight step counters: two 16-byte loads of uint32, or ONE of eight uint16 (4 B per parameter less to read and to write back)
    auto load_steps = [&](uint32_t i0, u32x4& s0, u32x4& s1) __attribute__((always_inline)) {
        if (p.steps16 || prec) { const u32x4 v = *reinterpret_cast<const u32x4*>(st_steps16(i0 >> 3));
            s0 = u32x4{ v[0] & 0xffffu, v[0] >> 16, v[1] & 0xffffu, v[1] >> 16 }; s1 = u32x4{ v[2] & 0xffffu, v[2] >> 16, v[3] & 0xffffu, v[3] >> 16 }; }
        else { s0 = *reinterpret_cast<const u32x4*>(p.steps + i0); s1 = *reinterpret_cast<const u32x4*>(p.steps + i0 + 4); }
    };
    // (plain vector types and no arrays: HIP's uint4 is a union, and either keeps the struct in scratch memory)
    struct Pre { float4_t w0, w1, a0, a1, b0, b1; u32x4 s0, s1; half8_t e; float4_t gm0, gm1; };
    auto issue = [&](uint32_t c, Pre& L) __attribute__((always_inline)) {
        const uint32_t i0 = c << 3;
        L.w0 = *reinterpret_cast<const float4_t*>(st_master(c)); L.w1 = *reinterpret_cast<const float4_t*>(st_master(c) + 4);
        L.a0 = *reinterpret_cast<const float4_t*>(st_m1(c)); L.a1 = *reinterpret_cast<const float4_t*>(st_m1(c) + 4);
        L.b0 = *reinterpret_cast<const float4_t*>(st_m2(c)); L.b1 = *reinterpret_cast<const float4_t*>(st_m2(c) + 4);
        load_steps(i0, L.s0, L.s1);
        L.e = *reinterpret_cast<const half8_t*>(p.ema + i0);
        if (i0 < oc.n_mlp) { L.gm0 = *reinterpret_cast<const float4_t*>(p.gmlp + i0); L.gm1 = *reinterpret_cast<const float4_t*>(p.gmlp + i0 + 4); }
    };
    // ONE chunk per thread: its state is requested HERE, before the kernel has seen its DevState -- the addresses come from the argument segment, and the round
    // trip for n_valid / step / lr would otherwise stand in front of the streams (eager path below; a skipped batch drops the values)
    Pre early; bool early_issued = false;
    if constexpr (ONE) {
        const uint32_t extra0 = nx.cand_blocks + nx.pos_blocks;
        if (vblock >= extra0) { const uint32_t ce = (vblock - extra0) * blockDim.x + threadIdx.x; if (ce < (oc.n_params >> 3)) { issue(ce, early);
                early_issued = true; } }
    }
    const uint32_t n_valid = st->n_valid, step = st->step;
    // DENSE tables: while most samples carry a gradient practically every chunk is updated and the optimizer state is requested together with the gradients
    // (one memory round trip); once few do (late training: k_grid_scatter left the count in n_scatter_now) most chunks only need their EMA advanced, and
    // the 112 B of Adam state per chunk are requested behind the gradient test instead
    // (ONE: the state is on its way already; the two orders measure the same late in training with one chunk per thread)
    const bool eager = DENSE && (ONE || !(lazy_below != 0u && st->n_scatter_now <= lazy_below));
    const uint32_t extra = nx.cand_blocks + nx.pos_blocks;          // (one or the other)
    const bool cand_block = vblock < extra;                     // GenerateRays of iteration iter + 1 / its sample positions
    if (vblock < nx.cand_blocks) gen_candidate(nx.b, nx.ds, nx.oc, st->n_boxes, st->iter + 1u, vblock * blockDim.x + threadIdx.x);
    // level-tile encode: the next iteration's candidates are complete (k_encode_tiles), sample their positions
    else if (cand_block) {
        __shared__ PointsLds plds;
        const uint32_t nwords = nx.oc.R >> 6, nv = points_prefix(plds, nx.b.mask, nwords);
        if (vblock == 0 && threadIdx.x == 0) st_next->n_valid_pre = nv;
#if defined(MON_OPT_ABLATE) && (MON_OPT_ABLATE & 4)
        if (false)
#else
        if (nv != 0u)
#endif
            for (uint32_t s = vblock * blockDim.x + threadIdx.x; s < nx.oc.R * 32u; s += nx.pos_blocks * blockDim.x) {
                if constexpr (LIVE) points_sample<true>(plds, nx.b, nx.oc, st->iter + 1u, nv, nwords, s, reinterpret_cast<float4_t*>(nx.x_all), nx.live);
                else points_sample<false>(plds, nx.b, nx.oc, st->iter + 1u, nv, nwords, s, reinterpret_cast<float4_t*>(nx.x_all));
            }
    }
    const uint32_t bid = vblock - extra, nblk = gridDim.x - extra;
    const float lr0 = st->lr;
    // EMA debias factors of this step (ema_step_half_precision; double-precision pow like tcnn's host code): left in the state by the previous step
    const uint32_t cur = step + 1u;
    const float d = oc.ema_decay;
    const float deb_old = st->ema_deb_old, deb_new = st->ema_deb_new;
    // ---- the state of the NEXT iteration.  Iteration i reads DevState i & 1, and everything that changes from one iteration to the next is known when this
    //      kernel starts, so one thread writes the other DevState right away.  (Advancing one shared state in place needed a "last block": a returning atomic
    //      on one address per block and the round trip behind the last of them -- 4.3 us of this kernel.)
    if (vblock == 0 && threadIdx.x == 0) {
        DevState& nxs = *st_next;
        // (n_valid / loss_sum: what the host reads after the call; k_fused_train overwrites them)
        nxs.iter = st->iter + 1u; nxs.n_valid = n_valid; nxs.loss_sum = st->loss_sum;
        // (the slot counters themselves are cleared and summed by k_grid_scatter)
        { const uint32_t tot = st->n_scatter_now; nxs.n_scatter_last = tot; nxs.n_scatter_total = st->n_scatter_total + tot; }
        uint32_t nstep = step;
        if (n_valid != 0u) {
            nstep = cur; nxs.skipped = st->skipped;
            nxs.lr = ((int)cur >= oc.decay_start && oc.decay_interval > 0 && ((int)cur - oc.decay_start) % oc.decay_interval == 0) ? lr0 * oc.decay_base : lr0;
        } else { nxs.skipped = st->skipped + 1u; nxs.lr = lr0; }
        nxs.step = nstep;
        // factors of step nstep + 1
        nxs.ema_deb_old = 1.f - (float)pow((double)d, (double)nstep); nxs.ema_deb_new = 1.f / (1.f - (float)pow((double)d, (double)(nstep + 1u)));
    }
    // gradient / loss_scale: a power-of-two scale (the reference's 128) divides exactly as a multiplication by its reciprocal (same
    // correctly rounded result, ~10 instructions less per parameter than an IEEE division); anything else keeps the division
    const bool pow2_scale = (__float_as_uint(oc.loss_scale) & 0x007fffffu) == 0u && oc.loss_scale > 0.f;
    const float inv_scale = 1.0f / oc.loss_scale;
    auto unscale = [&](float g) { return pow2_scale ? g * inv_scale : g / oc.loss_scale; };
    if (n_valid != 0u && !cand_block) {
        const uint32_t n_chunks = oc.n_params >> 3, c_stride = nblk * blockDim.x, c_first = bid * blockDim.x + threadIdx.x;
        // sparse (!DENSE) tables: a thread walks dozens of chunks, most of them untouched; the three always-needed loads of the NEXT chunk
        // (gradient, fp16 weight, EMA) are requested one iteration ahead so the walk is not one memory latency per chunk
        half8_t nx_g, nx_w, nx_e;
        auto prefetch = [&](uint32_t cn) {
            if (!DENSE && cn < n_chunks && (cn << 3) >= oc.n_mlp) {
                const uint32_t in0 = cn << 3;
                nx_g = *reinterpret_cast<const half8_t*>(p.ggrid + (in0 - oc.n_mlp));
                if (!LAZY) { nx_w = *reinterpret_cast<const half8_t*>(p.half + in0); nx_e = *reinterpret_cast<const half8_t*>(p.ema + in0); }
            }
        };
        // DENSE tables, eager state: the 160 B of optimizer state of a thread's SECOND chunk are requested before its first chunk is worked on (`Pre`), so that
        // their latency runs under that chunk's arithmetic (vmcnt retires in order: they have to be issued before the first chunk's stores, not after).
        // one 8-parameter chunk; `pre`: its always-needed loads (cur_*) were issued an iteration ago; `L`: state, EMA and MLP gradient were (eager dense path)
        auto update_chunk = [&](uint32_t c, bool pre, const half8_t& cur_g, const half8_t& cur_w, const half8_t& cur_e,
                const Pre* L = nullptr) __attribute__((always_inline)) {
            const uint32_t i0 = c << 3;
            const bool is_matrix = i0 < oc.n_mlp;                     // n_mlp is a multiple of 8: uniform per chunk
            float g[8]; bool any = false;
            uint32_t lvl_off = 0u, lvl_end = p.sl.entry_offset[1];    // the chunk's level: [lvl_off, lvl_end) in entries
            // DENSE (small tables: practically every entry has a gradient each step): the optimizer state is requested together
            // with the gradients -- one memory round trip instead of two; sparse tables keep the state loads behind the test.
            float4_t w0, w1, a0, a1, b0, b1; uint4 s0, s1;
            if (L) { w0 = L->w0; w1 = L->w1; a0 = L->a0; a1 = L->a1; b0 = L->b0; b1 = L->b1; s0 = uint4{ L->s0[0], L->s0[1], L->s0[2], L->s0[3] };
                s1 = uint4{ L->s1[0], L->s1[1], L->s1[2], L->s1[3] }; }
            else if (eager) {
                w0 = *reinterpret_cast<const float4_t*>(st_master(c)); w1 = *reinterpret_cast<const float4_t*>(st_master(c) + 4);
                a0 = *reinterpret_cast<const float4_t*>(st_m1(c)); a1 = *reinterpret_cast<const float4_t*>(st_m1(c) + 4);
                b0 = *reinterpret_cast<const float4_t*>(st_m2(c)); b1 = *reinterpret_cast<const float4_t*>(st_m2(c) + 4);
                { u32x4 t0, t1; load_steps(i0, t0, t1); s0 = uint4{ t0[0], t0[1], t0[2], t0[3] }; s1 = uint4{ t1[0], t1[1], t1[2], t1[3] }; }
            }
            const bool lazy_chunk = LAZY && !is_matrix;
            half8_t ema_in;
            if (L) ema_in = L->e;
            else if (!lazy_chunk) ema_in = pre ? cur_e : *reinterpret_cast<const half8_t*>(p.ema + i0);
            if (is_matrix) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t i = i0 + j;
                    const float gs = L ? (j < 4 ? L->gm0[j & 3] : L->gm1[j & 3]) : p.gmlp[i]; p.gmlp[i] = 0.f;
                    g[j] = unscale(gs);
                }
                any = true;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 0.f;
                if (!DENSE) {                                            // the global-atomic table (levels too large for an LDS tile); never written when DENSE
                    half8_t* gp = reinterpret_cast<half8_t*>(p.ggrid + (i0 - oc.n_mlp));
                    const half8_t gh = pre ? cur_g : *reinterpret_cast<const half8_t*>(p.ggrid + (i0 - oc.n_mlp));
                    bool anyg = false;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { g[j] = (float)gh[j]; anyg |= (float)gh[j] != 0.f; }
                    if (anyg) { half8_t z;
#pragma unroll
                        for (int j = 0; j < 8; ++j) z[j] = (half_t)0.f;
                        *gp = z; }
                }
                uint32_t n_part = 0;
                if (p.gpart) {                                       // which level is this chunk in -> how many partial tables it has
                    const uint32_t e0 = (i0 - oc.n_mlp) >> 1; int lvl = 0;
#pragma unroll
                    for (int l = 1; l < kMaxLevels; ++l) { const bool in = e0 >= p.sl.entry_offset[l]; lvl += in ? 1 : 0;
                        lvl_off = in ? p.sl.entry_offset[l] : lvl_off; lvl_end = in ? p.sl.entry_offset[l + 1] : lvl_end; }
                    n_part = p.sl.P[lvl];
                }
                // dense partial tables of k_grid_scatter (fused backend): [partition][feature][parity][entry / 2]; this chunk = entries e0 .. e0 + 3 (e0 a
                // multiple of 4), both features: per partition four 4-byte pieces -- plane (f, b) holds entries e0 + b and e0 + 2 + b next to each other
                const uint16_t* pp = p.gpart + ((i0 - oc.n_mlp) >> 2);
                const size_t plane = p.part_stride >> 2;                 // entries per (feature, parity) plane
                auto quad = [&](uint32_t q, float (&v)[8]) {             // partition q's eight values in parameter order (entry-major, feature-minor)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const half2_t h = *reinterpret_cast<const half2_t*>(pp + ((size_t)(q * 2u + (uint32_t)f) * 2u + (uint32_t)b) * plane);
                            v[2 * b + f] = (float)h.x; v[2 * (2 + b) + f] = (float)h.y;
                        }
                };
                uint32_t q = 0;
#if defined(MON_OPT_ABLATE) && (MON_OPT_ABLATE & 2)
                for (int j = 0; j < 8; ++j) g[j] = 1e-3f;
                n_part = 0;
#endif
                for (; q + 2 <= n_part; q += 2) {                       // 8 independent 4-byte loads in flight
                    float va[8], vb[8]; quad(q, va); quad(q + 1u, vb);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += va[j] + vb[j];
                }
                for (; q < n_part; ++q) {
                    float va[8]; quad(q, va);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += va[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { any |= g[j] != 0.f; g[j] = unscale(g[j]); }
            }
            if (lazy_chunk && !any) return;                                            // untouched: nothing to do now (see k_ema_finalize)
            // the fp16 working copy is h(master) by construction (creation, set_params, every update): where the master weights are loaded anyway it is not
            // read back
            half8_t wh;
            if (eager) {
                const float wm[8] = { w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3] };
#pragma unroll
                for (int j = 0; j < 8; ++j) wh[j] = (half_t)wm[j];
            } else wh = (pre && !LAZY) ? cur_w : *reinterpret_cast<const half8_t*>(p.half + i0);
            if (lazy_chunk) {                                                        // touched again: first the steps it sat out, with the weight it had
                ema_in = *reinterpret_cast<const half8_t*>(p.ema + i0);
                const uint32_t last = *ema_step_of(c), k = (cur - 1u) - last;
                if (k) ema_catch_up(ema_in, wh, last, k, oc.log2_decay);
                *ema_step_of(c) = cur;
            }
            if (any) {
                if (!eager) {
                    w0 = *reinterpret_cast<const float4_t*>(st_master(c)); w1 = *reinterpret_cast<const float4_t*>(st_master(c) + 4);
                    a0 = *reinterpret_cast<const float4_t*>(st_m1(c)); a1 = *reinterpret_cast<const float4_t*>(st_m1(c) + 4);
                    b0 = *reinterpret_cast<const float4_t*>(st_m2(c)); b1 = *reinterpret_cast<const float4_t*>(st_m2(c) + 4);
                    { u32x4 t0, t1; load_steps(i0, t0, t1); s0 = uint4{ t0[0], t0[1], t0[2], t0[3] }; s1 = uint4{ t1[0], t1[1], t1[2], t1[3] }; }
                }
                float w[8] = { w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3] };
                float m1[8] = { a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3] };
                float m2[8] = { b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3] };
                uint32_t sc[8] = { s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w };
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float gj = g[j];
                    if (is_matrix) gj += oc.l2_reg * w[j];                            // L2 only on matrix weights
                    else if (gj == 0.f) continue;                                     // untouched grid entry: skipped entirely
#if defined(MON_OPT_ABLATE) && (MON_OPT_ABLATE & 1)
                    m1[j] += gj; m2[j] += gj; sc[j] += 1u; w[j] -= 1e-6f * gj;
#else
                    w[j] = adam_update(gj, w[j], m1[j], m2[j], sc[j], lr0, oc, step_cap);
#endif
                    wh[j] = (half_t)w[j];
                }
                // optimizer state is not touched again before the next step: stream it past the caches
                state_store<!LAZY>(float4_t{ w[0], w[1], w[2], w[3] }, reinterpret_cast<float4_t*>(st_master(c)));
                state_store<!LAZY>(float4_t{ w[4], w[5], w[6], w[7] }, reinterpret_cast<float4_t*>(st_master(c) + 4));
                state_store<!LAZY>(float4_t{ m1[0], m1[1], m1[2], m1[3] }, reinterpret_cast<float4_t*>(st_m1(c)));
                state_store<!LAZY>(float4_t{ m1[4], m1[5], m1[6], m1[7] }, reinterpret_cast<float4_t*>(st_m1(c) + 4));
                state_store<!LAZY>(float4_t{ m2[0], m2[1], m2[2], m2[3] }, reinterpret_cast<float4_t*>(st_m2(c)));
                state_store<!LAZY>(float4_t{ m2[4], m2[5], m2[6], m2[7] }, reinterpret_cast<float4_t*>(st_m2(c) + 4));
                typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                if (p.steps16 || prec) state_store<!LAZY>(u4v{ sc[0] | (sc[1] << 16), sc[2] | (sc[3] << 16), sc[4] | (sc[5] << 16), sc[6] | (sc[7] << 16) },
                        reinterpret_cast<u4v*>(st_steps16(c)));
                else { state_store<!LAZY>(u4v{ sc[0], sc[1], sc[2], sc[3] }, reinterpret_cast<u4v*>(p.steps + i0));
                    state_store<!LAZY>(u4v{ sc[4], sc[5], sc[6], sc[7] }, reinterpret_cast<u4v*>(p.steps + i0 + 4)); }
                *reinterpret_cast<half8_t*>(p.half + i0) = wh;
#if defined(MON_OPT_ABLATE) && (MON_OPT_ABLATE & 8)
                if (false) {
#else
                if (p.half_tiles && !is_matrix) {
#endif                                     // the same four entries in tile order for k_encode_tiles (tile_slot): whole level = as they are, else evens | odds
                    const uint32_t e0 = (i0 - oc.n_mlp) >> 1, size = lvl_end - lvl_off, e_rel = e0 - lvl_off;
                    if (size <= kEncWholeMax) *reinterpret_cast<half8_t*>(p.half_tiles + 2u * (size_t)e0) = wh;
                    else {
                        const size_t s0 = lvl_off + (e_rel >> 1);
                        *reinterpret_cast<half4_t*>(p.half_tiles + 2u * s0) = half4_t{ wh[0], wh[1], wh[4], wh[5] };
                        *reinterpret_cast<half4_t*>(p.half_tiles + 2u * (s0 + (size >> 1))) = half4_t{ wh[2], wh[3], wh[6], wh[7] };
                    }
                }
                if (is_matrix && nx.frag_image) {                                     // next iteration's A fragments
#pragma unroll
                    for (int j = 0; j < 8; ++j) { int sl[2]; const int ns = frag_slots(nx.fd, (int)(i0 + j), sl);
                        for (int q = 0; q < ns; ++q) reinterpret_cast<half_t*>(nx.frag_image)[sl[q]] = wh[j]; }
                }
            }
            half8_t* ep = reinterpret_cast<half8_t*>(p.ema + i0);
            half8_t e = ema_in;
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (half_t)((((float)e[j] * d) * deb_old + (float)wh[j] * (1.f - d)) * deb_new);
            *ep = e;
        };
        if constexpr (LAZY) {
            // Few chunks carry a gradient, but nearly every wave has SOME lane that does in every iteration, so handling them in place makes
            // the whole wave sit through the touched path's three dependent memory round trips ~25 times.  Instead: scan (one prefetched
            // 16-byte load per chunk), queue the touched chunks per wave in LDS, then work the queue with full waves.
            constexpr uint32_t kQueueCap = 2048;
            __shared__ uint32_t queue[4][kQueueCap];
            const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u; uint32_t qn = 0;
            const half8_t none{};
            if (p.touched) {
                // Chunk flags (one byte per chunk, set next to every addition into ggrid): the scan reads 1/16 of what the gradient table
                // itself would cost.  Chunks below first_flag_chunk (MLP matrices, LDS-scattered levels) are few and always visited.
                const uint32_t w_end = (n_chunks + 3u) >> 2; uint32_t* flags = reinterpret_cast<uint32_t*>(p.touched);
                // a wave's flag words (6-7 trips at T = 2^22) are requested TOGETHER and ahead of everything else: one load per loop trip put a memory round
                // trip in front of every trip (late in training the kernel is a chain of such round trips, not bandwidth: 141 MB in 57 us)
                constexpr uint32_t kFlagPre = 8;
                const uint32_t wf0 = (p.first_flag_chunk >> 2) + c_first - lane; uint32_t fpre[kFlagPre];
#pragma unroll
                for (uint32_t k = 0; k < kFlagPre; ++k) { const uint32_t w = wf0 + k * c_stride + lane; fpre[k] = (w < w_end) ? flags[w] : 0u; }
                for (uint32_t c = c_first; c < p.first_flag_chunk; c += c_stride) update_chunk(c, false, none, none, none);
                uint32_t trip = 0;
                for (uint32_t w0 = wf0; w0 < w_end; w0 += c_stride, ++trip) {     // wave-uniform trip count
                    const uint32_t w = w0 + lane; uint32_t f = 0u;
                    if (trip < kFlagPre) {
                        // (static register indices: a select chain over the preloaded words)
#pragma unroll
                        for (uint32_t k = 0; k < kFlagPre; ++k) f = (trip == k) ? fpre[k] : f;
                    } else if (w < w_end) f = flags[w];
                    if (f) flags[w] = 0u;
                    // queue in ascending chunk order (lane-major: a lane's four chunks are neighbours, the next lane's follow), so that the
                    // lanes working the queue touch adjacent 32-byte pieces of the state arrays where adjacent chunks are both touched
                    uint32_t before = 0u, total = 0u;
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const unsigned long long tm = __ballot(((f >> (8u * j)) & 0xffu) != 0u);
                        before += (uint32_t)__popcll(tm & ((1ull << lane) - 1ull)); total += (uint32_t)__popcll(tm);
                    }
                    if (total == 0u) continue;
                    uint32_t pos = qn + before; qn += total;
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        if (((f >> (8u * j)) & 0xffu) == 0u) continue;
                        const uint32_t c = 4u * w + j;
                        if (pos < kQueueCap) queue[wave][pos] = c; else update_chunk(c, false, none, none, none);
                        ++pos;
                    }
                }
            } else {
            prefetch(c_first);
            for (uint32_t c0 = c_first - lane; c0 < n_chunks; c0 += c_stride) {          // wave-uniform trip count
                const uint32_t c = c0 + lane; const bool valid = c < n_chunks;
                const half8_t cur_g = nx_g; prefetch(c + c_stride);
                bool direct = false, touched = false;
                if (valid) {
                    const uint32_t i0 = c << 3;
                    if (i0 < oc.n_mlp) direct = true;
                    else {
                        const uint32_t e0 = (i0 - oc.n_mlp) >> 1; int lvl = 0;
#pragma unroll
                        for (int l = 1; l < kMaxLevels; ++l) lvl += (e0 >= p.sl.entry_offset[l]) ? 1 : 0;
                        if (p.gpart && p.sl.P[lvl]) direct = true;                        // LDS-scattered level: its gradient is in the partial tables
                        else { const uint4 gb = __builtin_bit_cast(uint4, cur_g); touched = ((gb.x | gb.y | gb.z | gb.w) & 0x7fff7fffu) != 0u; }
                    }
                }
                const unsigned long long tm = __ballot(touched);
                const uint32_t pos = qn + (uint32_t)__popcll(tm & ((1ull << lane) - 1ull));
                if (touched) { if (pos < kQueueCap) queue[wave][pos] = c; else direct = true; }
                qn += (uint32_t)__popcll(tm);
                if (direct) update_chunk(c, false, none, none, none);
            }
            }
            if (qn > kQueueCap) qn = kQueueCap;
            // A QUEUED chunk is a grid chunk outside the LDS-scattered levels whose flag (or gradient) says it was touched: its gradient, fp16 weights, EMA,
            // EMA step and optimizer state are requested in ONE go -- update_chunk asks for them in three dependent rounds (gradient -> weights + EMA -> state),
            // right for the in-place walk where most chunks stop after the first -- and the arithmetic below is update_chunk's lazy-chunk path, operation by
            // operation (the record / array CRC tests and the lazy-EMA test run through both).
            const auto update_queued = [&](uint32_t c) __attribute__((always_inline)) {
                const uint32_t i0 = c << 3;
                half8_t* gp = reinterpret_cast<half8_t*>(p.ggrid + (i0 - oc.n_mlp));
                // (the fp16 working copy is h(master) by construction -- creation, set_params, every update --: with the master weights on their way it is not
                // read back, one 64-byte line less per touched chunk)
                const half8_t gh = *gp; half8_t e = *reinterpret_cast<const half8_t*>(p.ema + i0);
                const uint32_t last = *ema_step_of(c);
                const float4_t w0 = *reinterpret_cast<const float4_t*>(st_master(c)), w1 = *reinterpret_cast<const float4_t*>(st_master(c) + 4);
                const float4_t a0 = *reinterpret_cast<const float4_t*>(st_m1(c)), a1 = *reinterpret_cast<const float4_t*>(st_m1(c) + 4);
                const float4_t b0 = *reinterpret_cast<const float4_t*>(st_m2(c)), b1 = *reinterpret_cast<const float4_t*>(st_m2(c) + 4);
                u32x4 t0, t1; load_steps(i0, t0, t1);
                float g[8]; bool any = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) { g[j] = (float)gh[j]; any |= g[j] != 0.f; }
                if (!any) return;                                                     // (flagged, but every contribution cancelled: untouched after all)
                { half8_t z;
#pragma unroll
                  for (int j = 0; j < 8; ++j) z[j] = (half_t)0.f;
                  *gp = z; }
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = unscale(g[j]);
                float w[8] = { w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3] };
                half8_t wh;
#pragma unroll
                for (int j = 0; j < 8; ++j) wh[j] = (half_t)w[j];
                const uint32_t k = (cur - 1u) - last;
                if (k) ema_catch_up(e, wh, last, k, oc.log2_decay);
                *ema_step_of(c) = cur;
                float m1[8] = { a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3] };
                float m2[8] = { b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3] };
                uint32_t sc[8] = { t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3] };
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (g[j] == 0.f) continue;                                        // untouched grid entry: skipped entirely
                    w[j] = adam_update(g[j], w[j], m1[j], m2[j], sc[j], lr0, oc, step_cap);
                    wh[j] = (half_t)w[j];
                }
                state_store<false>(float4_t{ w[0], w[1], w[2], w[3] }, reinterpret_cast<float4_t*>(st_master(c)));
                state_store<false>(float4_t{ w[4], w[5], w[6], w[7] }, reinterpret_cast<float4_t*>(st_master(c) + 4));
                state_store<false>(float4_t{ m1[0], m1[1], m1[2], m1[3] }, reinterpret_cast<float4_t*>(st_m1(c)));
                state_store<false>(float4_t{ m1[4], m1[5], m1[6], m1[7] }, reinterpret_cast<float4_t*>(st_m1(c) + 4));
                state_store<false>(float4_t{ m2[0], m2[1], m2[2], m2[3] }, reinterpret_cast<float4_t*>(st_m2(c)));
                state_store<false>(float4_t{ m2[4], m2[5], m2[6], m2[7] }, reinterpret_cast<float4_t*>(st_m2(c) + 4));
                typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                if (p.steps16 || prec) state_store<false>(u4v{ sc[0] | (sc[1] << 16), sc[2] | (sc[3] << 16), sc[4] | (sc[5] << 16), sc[6] | (sc[7] << 16) },
                        reinterpret_cast<u4v*>(st_steps16(c)));
                else { state_store<false>(u4v{ sc[0], sc[1], sc[2], sc[3] }, reinterpret_cast<u4v*>(p.steps + i0));
                    state_store<false>(u4v{ sc[4], sc[5], sc[6], sc[7] }, reinterpret_cast<u4v*>(p.steps + i0 + 4)); }
                *reinterpret_cast<half8_t*>(p.half + i0) = wh;
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (half_t)((((float)e[j] * d) * deb_old + (float)wh[j] * (1.f - d)) * deb_new);
                *reinterpret_cast<half8_t*>(p.ema + i0) = e;
            };
            for (uint32_t q0 = 0; q0 < qn; q0 += 64u) if (q0 + lane < qn) update_queued(queue[wave][q0 + lane]);
        } else if (eager) {
            // a thread has two chunks at these table sizes (the launch gives ~2 chunks per thread): the state of both is requested before the first is worked
            // on; straight-line code, no loop-carried buffers (those ended up in scratch memory)
            const half8_t none{};
            const uint32_t c0 = c_first, c1 = c_first + c_stride;
            if constexpr (ONE) {
                if (c0 < n_chunks) { if (!early_issued) issue(c0, early); update_chunk(c0, false, none, none, none, &early); }
            } else if (c0 < n_chunks) {
                Pre A, B2; const bool two = c1 < n_chunks;
                issue(c0, A); if (two) issue(c1, B2);
                update_chunk(c0, false, none, none, none, &A);
                if (two) update_chunk(c1, false, none, none, none, &B2);
                for (uint32_t c = c1 + c_stride; c < n_chunks; c += c_stride) update_chunk(c, false, none, none, none);
            }
        } else {
            prefetch(c_first);
            for (uint32_t c = c_first; c < n_chunks; c += c_stride) {
                const bool pre = !DENSE && (c << 3) >= oc.n_mlp;
                const half8_t cur_g = nx_g, cur_w = nx_w, cur_e = nx_e;
                prefetch(c + c_stride);
                update_chunk(c, pre, cur_g, cur_w, cur_e);
            }
        }
    }
}

// Fused backend: sums the per-workgroup fp32 weight-gradient partials (and loss partials) written by
// k_fused_train into gmlp / DevState::loss_sum.  One block = 16 parameters (4 float4 columns) x 64 row subsets,
// so every thread has only n_partials/64 independent 16-byte loads in flight (the kernel is latency bound).
__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ partials, uint32_t n_partials, uint32_t stride, uint32_t n_cols, FragDims fd,
                                                         float* __restrict__ gmlp, DevState* __restrict__ st) {
    if (st->n_valid == 0u) return;
    __shared__ float4_t red[256];
    const uint32_t c4 = threadIdx.x & 3u, sub = threadIdx.x >> 2;              // column group, row subset (0..63)
    const uint32_t p0 = blockIdx.x * 16u + c4 * 4u;
    float4_t acc = { 0.f, 0.f, 0.f, 0.f };
    if (p0 < n_cols + 4u) {                                                     // rows are padded to n_cols + 64 floats: in-bounds
        for (uint32_t k = sub; k < n_partials; k += 64u) acc += *reinterpret_cast<const float4_t*>(partials + (size_t)k * stride + p0);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off >= 4; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x < 4u) {
        const float4_t v = red[threadIdx.x];
        // rows are in accumulator layout
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t pi = p0 + j; if (pi < n_cols) { const int prm = acc_param(fd, (int)pi); if (prm >= 0) gmlp[prm] = v[j];
                } else if (pi == n_cols) st->loss_sum = v[j]; }
    }
}

void launch_reduce_partials(hipStream_t s, const float* partials, uint32_t n_partials, const NetDims& nd, float* gmlp, DevState* st) {
    const uint32_t n_cols = fused_partial_cols(nd);
    hipLaunchKernelGGL(k_reduce_partials, dim3((n_cols + 1 + 15) / 16), dim3(256), 0, s, partials, n_partials, n_cols + 64u, n_cols, FragDims{ nd.Epad, nd.W,
            nd.NH, nd.L }, gmlp, st);
}

// Brings every lazily maintained EMA chunk up to the last completed optimizer step (before render / mesh / parameter read-back).
__global__ void __launch_bounds__(256) k_ema_finalize(ParamPtrs p, OptimConst oc, const DevState* __restrict__ st) {
    const uint32_t done = st->step, n_chunks = oc.n_params >> 3;
    for (uint32_t c = (oc.n_mlp >> 3) + blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += gridDim.x * blockDim.x) {
        uint32_t* lp = p.rec ? reinterpret_cast<uint32_t*>(p.rec + 32u * (size_t)c + 28u) : p.ema_step + c;      // (records: in the pad word, k_optimizer)
        const uint32_t last = *lp;
        if (last >= done) continue;
        half8_t e = *reinterpret_cast<const half8_t*>(p.ema + (c << 3)); const half8_t w = *reinterpret_cast<const half8_t*>(p.half + (c << 3));
        ema_catch_up(e, w, last, done - last, oc.log2_decay);
        *reinterpret_cast<half8_t*>(p.ema + (c << 3)) = e; *lp = done;
    }
}
void launch_ema_finalize(hipStream_t s, const ParamPtrs& p, const OptimConst& oc, const DevState* st) {
    hipLaunchKernelGGL(k_ema_finalize, dim3(2048), dim3(256), 0, s, p, oc, st);
}

// fp16 working copy of the fp32 master weights, h(master): object creation and set_params (a scalar host loop without F16C costs ~13 ms for
// base.json's 1.9 M parameters -- time the SLAM thread spends inside CreateNeRF)
__global__ void __launch_bounds__(256) k_master_to_half(const float* __restrict__ master, uint16_t* __restrict__ half, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const half_t h = (half_t)master[i];
        half[i] = __builtin_bit_cast(uint16_t, h); }
}
// plain device copy of a parameter vector (the inference-side snapshots): a kernel of our own rather than hipMemcpyAsync, whose blit path brackets
// the copy with cache maintenance that the following training kernels pay for
__global__ void __launch_bounds__(256) k_copy_params(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16,
        const uint16_t* __restrict__ src_tail, uint16_t* __restrict__ dst_tail, uint32_t n_tail) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
void launch_copy_params(hipStream_t s, const uint16_t* src, uint16_t* dst, uint32_t n) {
    const uint32_t n16 = n / 8u, tail = n - n16 * 8u;
    const uint32_t blocks = n16 >= 512u * 256u ? 512u : (n16 + 255u) / 256u + (n16 == 0u ? 1u : 0u);
    hipLaunchKernelGGL(k_copy_params, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16,
            src + (size_t)n16 * 8u, dst + (size_t)n16 * 8u, tail);
}
// Record layout of the optimizer state (ParamPtrs::rec) <-> the flat arrays the boundary speaks (get / set_params, debug read-back): which = 0 master, 1 m1,
// 2 m2 (floats), 3 the step counters (uint16 widened to uint32 on the way out).
__global__ void __launch_bounds__(256) k_state_unpack(const float* __restrict__ rec, int which, uint32_t* __restrict__ dst, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* r = rec + 32u * (size_t)(i >> 3);
        dst[i] = which < 3 ? __builtin_bit_cast(uint32_t, r[8 * which + (i & 7u)]) : (uint32_t)reinterpret_cast<const uint16_t*>(r + 24)[i & 7u];
    }
}
__global__ void __launch_bounds__(256) k_state_pack_master(const float* __restrict__ master, float* __restrict__ rec, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) rec[32u * (size_t)(i >> 3) + (i & 7u)] = master[i];
}
void launch_state_unpack(hipStream_t s, const float* rec, int which, void* dst, uint32_t n) {
    hipLaunchKernelGGL(k_state_unpack, dim3(2048), dim3(256), 0, s, rec, which, static_cast<uint32_t*>(dst), n);
}
void launch_state_pack_master(hipStream_t s, const float* master, float* rec, uint32_t n) {
    hipLaunchKernelGGL(k_state_pack_master, dim3(2048), dim3(256), 0, s, master, rec, n);
}
void launch_master_to_half(hipStream_t s, const float* master, uint16_t* half, uint32_t n) {
    hipLaunchKernelGGL(k_master_to_half, dim3(1024), dim3(256), 0, s, master, half, n);
}

void launch_optimizer(hipStream_t s, const ParamPtrs& p, const OptimConst& oc, const DevState* st, DevState* st_next, const OptimNext& nx,
        uint32_t lazy_below) {
    const uint32_t chunks = oc.n_params >> 3;
    // measured: base.json (239 k chunks), parameter blocks ahead of the preparation blocks: 384 / 512 / 640 / 768 / 1024 blocks = 27.9 / 24.1 / 24.7 / 23.6 /
    // 22.9 us (one chunk per thread; with the preparation blocks FIRST 512 was the best: 28.1 / 23.7 / 26.2 us for 256 / 512 / 1024); T = 2^22 (13.2 M chunks)
    // 512 / 2048 / 8192 / 32768 blocks = 368 / 244 / 251 / 406 us
    uint32_t cap = chunks / (256u * 8u); if (cap < 1024u) cap = 1024u; if (cap > 2048u) cap = 2048u;
    uint32_t blocks = (chunks + 255) / 256; if (blocks > cap) blocks = cap; if (blocks < 1u) blocks = 1u;
    // dense = every level goes through the LDS scatter, i.e. tables of at most 2^18 entries that a 131 072-sample batch covers
    const dim3 grid(blocks + nx.cand_blocks + nx.pos_blocks); const bool live = nx.pos_blocks && nx.live.occ_bits;
#define MON_OPT_LAUNCH(D, L, O) do { if (live) hipLaunchKernelGGL((k_optimizer<D, L, O, true>), grid, dim3(256), 0, s, p, oc, st, st_next, nx, lazy_below); \
        else hipLaunchKernelGGL((k_optimizer<D, L, O, false>), grid, dim3(256), 0, s, p, oc, st, st_next, nx, lazy_below); } while (0)
    if (p.gpart && p.all_levels_dense) {
        if ((size_t)blocks * 256u >= chunks) MON_OPT_LAUNCH(true, false, true);
        else MON_OPT_LAUNCH(true, false, false);
    }
    else if (p.lazy) MON_OPT_LAUNCH(false, true, false);
    else MON_OPT_LAUNCH(false, false, false);
#undef MON_OPT_LAUNCH
}

}  // namespace mon
