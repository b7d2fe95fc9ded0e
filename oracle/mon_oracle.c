/*
 * mon_oracle.c -- CPU restatement of the RO-MAP Multi-Object-NeRF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ro-map_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it
 * (as the checker / the timed CPU baseline, never as the thing shipped).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 * and cannot be built here (CUDA-only; tiny-cuda-nn submodule absent; Eigen/OpenCV/GLEW
 * absent).  This file restates
 *   (a) the reference's own kernels, citing CORE/src/nerf_model.cu file:line
 *       (CORE = /root/reference/dependencies/Multi-Object-NeRF/Core), and
 *   (b) the published algorithm of the un-vendored dependency NVlabs/tiny-cuda-nn
 *       (v1.6 era, late 2022; pinned commit not recoverable) for the
 *       hash-grid encoding, fully fused MLP, Adam / ExponentialDecay / EMA optimizers.
 * It is pinned by this repo's own known-answer tests (tests/test_oracle_*.py): closed-form
 * composite, finite-difference / torch-autograd checks of the hand-derived gradient,
 * closed-form Adam/EMA step, hash-index KATs; since round 3 also the WHOLE backward chain against
 * torch autograd, the grid backward per entry against an fp64 NumPy scatter, and -- the one piece
 * with a second source in the image -- the XORWOW sample stream of the "same inputs" mode against
 * rocRAND (engine: tests/test_xorwow.py; host generator on the GPU: tests/test_xorwow_gpu.py).
 *
 * Numeric model (rounding points; "h()" = round-to-nearest-even to IEEE fp16):
 *   table/weights  fp16 working copy of fp32 master           (tcnn: half params + fp32 master)
 *   encode         fp32 fmaf chain over the 8 corners, h() once per feature
 *                  (tcnn accumulates the 8 products in half; <= 4 half-ulp apart)
 *   MLP            fp16 in, fp16 weights, fp32 fmaf chain over k, h() per activation
 *   composite      fp32                                       (nerf_model.cu:735-815)
 *   dL/dO          fp32 math, h() on store                    (nerf_model.cu:917-945)
 *   MLP backward   dh, dE: fp32 chain then h(); dW accumulated in fp32 (tcnn: fp16 GEMM out)
 *   grid backward  each contribution h(w * dE) as in tcnn; accumulated in fp32 here
 *                  (tcnn: atomicAdd(__half2), order-dependent) then h() before Adam
 *   Adam/EMA       fp32, as tcnn
 * All geometry uses explicit fmaf chains; build with -ffp-contract=off so that the HIP
 * path (same chains) is bit-comparable up to expf/powf implementation differences.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__F16C__)
#include <immintrin.h>
#endif

#define ORC_MAX_LEVELS 32
#define ORC_OUT 4          /* rgb + density (nerf_model.cu:1318: NetworkWithInputEncoding(3,4,..)) */
#define ORC_OUT_PAD 16     /* tcnn pads the MLP output to a multiple of 16 */

/* ---- configuration; layout mirrors include/mon_core.h:mon_config on purpose so one
 *      ctypes.Structure serves both (declared independently, not shared code). ---- */
typedef struct {
    int32_t n_levels;            /* base.json encoding.n_levels            */
    int32_t n_features;          /* must be 2 (base.json:25)               */
    int32_t log2_hashmap_size;   /* base.json:26                           */
    int32_t base_resolution;     /* base.json:27                           */
    float   per_level_scale;     /* tcnn default 2.0 (TCNN-A1; nerf_model.cu:1305-1313 only prints its own) */
    int32_t n_neurons;           /* base.json:34                           */
    int32_t n_hidden_layers;     /* base.json:35                           */
    int32_t rays_per_batch;      /* nerf_model.h:173 (4096)                */
    int32_t n_samples;           /* common.h:12 SampleNum 32; render uses 2x (nerf_model.h:175) */
    float   loss_scale;          /* nerf_model.h:166 (128)                 */
    float   learning_rate;       /* base.json:16                           */
    float   beta1, beta2, epsilon, l2_reg;   /* base.json:17-20            */
    float   ema_decay;           /* base.json:7                            */
    int32_t decay_start, decay_interval;     /* base.json:10-11            */
    float   decay_base;          /* base.json:12                           */
    uint32_t param_seed;         /* nerf_model.h:145 (1337)                */
    uint32_t rng_flags;          /* "same inputs" mode for a comparison with the CUDA build (0 = this repo's defaults):
                                  *   bits 0-1  sample stream: 0 counter RNG (below) | 1 XORWOW, cuRAND flavour | 2 XORWOW, rocRAND flavour
                                  *   bit  4    parameter init in tcnn's generate_random_uniform element order (pcg32 draws interleaved per thread)
                                  *   bits 16-31 XORWOW lanes (independent subsequences of the host generator) in units of 1024; 0 = 4 (cuRAND: 4096) */
    /* key of the counter RNG; the XORWOW stream uses the reference's seed (the generator's default, 0: nerf_model.cu never sets one) */
    uint64_t sample_seed;
    int32_t use_depth;           /* NeRF_Model::mbUseDepth                 */
    int32_t numerics_flags;      /* oracle-only, bit field (0 = the contract of DESIGN.md section 1):
                                  *   ORC_NUM_GRID_HALF  accumulate grid gradients sequentially in fp16 (tcnn: atomicAdd(__half2))
                                  *   ORC_NUM_TCNN_HALF  MODEL of tiny-cuda-nn's own accumulation (the submodule is absent; restated from
                                  *                      its published kernels): kernel_grid sums the 8 corner products in __half
                                  *                      (result += (T)(weight * value)); the fully fused MLP keeps __half WMMA
                                  *                      accumulators -- modelled as one fp16 rounding per 16-wide k block, the products
                                  *                      of a block summed in fp32 -- forward and backward; weight gradients leave the
                                  *                      split-k GEMM as fp16.  Call sites nerf_model.cu:1557,1604,1644. */
} orc_config;
#define ORC_NUM_GRID_HALF 1
#define ORC_NUM_TCNN_HALF 2
#define ORC_RNG_STREAM(c) ((c)->rng_flags & 3u)
#define ORC_RNG_TCNN_INIT(c) (((c)->rng_flags >> 4) & 1u)
#define ORC_RNG_LANES(c) ((((c)->rng_flags >> 16) ? ((c)->rng_flags >> 16) : 4u) * 1024u)

typedef struct { uint32_t FrameId, x, y, h, w; } orc_bbox;   /* common.h:18-23 (h before w) */

/* ------------------------------------------------------------------ fp16 */
static inline uint16_t f2h(float f) {
#if defined(__F16C__)
    return (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
#else
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          /* rounds to inf */
    if (x < 0x33000001u) return (uint16_t)sign;                       /* rounds to zero */
    int e = (int)(x >> 23) - 127; uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13; int he = (e < -14) ? 0 : e + 15;
    uint32_t hm = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    uint32_t r = (e < -14) ? hm : (((uint32_t)he << 10) + (hm - 0x400u));
    return (uint16_t)(sign | r);
#endif
}
static inline float h2f(uint16_t h) {
#if defined(__F16C__)
    return _cvtsh_ss(h);
#else
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) { if (!m) x = sign; else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
            } }
    else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
#endif
}
uint16_t orc_f2h(float f) { return f2h(f); }
float orc_h2f(uint16_t h) { return h2f(h); }

/* ------------------------------------------------------------------ RNG
 * Sampling stream: counter-based splitmix64 finaliser; u in [0,1).  Replaces the three
 * curandGenerateUniform calls (nerf_model.cu:1432,1434,1468; XORWOW, default seed).
 * streams: 0 SampleXY[2R], 1 RandColors[3R], 2 RandDt[S*R], 3 render RandDt. */
static inline float rand01(uint64_t seed, uint32_t stream, uint32_t step, uint32_t idx) {
    uint64_t ctr = ((uint64_t)stream << 60) | ((uint64_t)step << 28) | (uint64_t)(idx & 0x0fffffffu);
    uint64_t z = ctr + seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}
float orc_rand01(uint64_t seed, uint32_t stream, uint32_t step, uint32_t idx) { return rand01(seed, stream, step, idx); }

/* XORWOW (Marsaglia 2003, "Xorshift RNGs", the generator behind CURAND_RNG_PSEUDO_DEFAULT and ROCRAND_RNG_PSEUDO_XORWOW): the reference draws its three
 * per-iteration arrays from ONE host generator with the default seed (nerf_model.cu:1432,1434,1468; created :1392-1394, no seed call) and a fresh one per
 * Render (:1725-1728,1781).  Restated here from the published algorithm:
 *   state: five 32-bit xorshift words x[0..4] + a Weyl counter d;  next: t = x0 ^ (x0 >> 2); shift the words down; x4 = (x4 ^ (x4 << 4)) ^ (t ^ (t << 1));
 *          d += 362437; output = x4 + d.
 *   host-API ordering (cuRAND documentation, CURAND_ORDERING_PSEUDO_DEFAULT): the value at offset n of a generate call comes from position
 * (n mod LANES) * 2^67 + floor(n / LANES) of the sequence, LANES = 4096: lane k starts 2^67 * k steps ahead (the Weyl counter is unaffected: 2^67 = 0 mod
 * 2^32)
 *          and the lanes keep their states from one generate call to the next.
 *   the jump by 2^67: the xorshift part is linear over GF(2); its 160 x 160 transition matrix is squared 67 times here (no table).
 * What differs between the two libraries, and is therefore a named assumption for the CUDA side (cuRAND is not in this image): the seed scramble --
 * rocRAND (rocrand_xorwow.h:107-118, checked against that header in tests/test_xorwow.py): s0 = lo ^ 0x2c7f967f, s1 = hi ^ 0xa03697cb, t0 = 1228688033 s0,
 * t1 = 2073658381 s1;
 * cuRAND  (CURAND-A1, curand_kernel.h _curand_init_scratch, from the published header): s0 = lo ^ 0xaad26b49, s1 = hi ^ 0xf7dcefdd, t0 = 1099087573 s0, t1 =
 * 2591861531 s1;
 *   both then set x = {123456789 + t0, 362436069 ^ t0, 521288629 + t1, 88675123 ^ t1, 5783321 + t0}, d = 6615241 + t1 + t0 --
 * and the integer -> (0, 1] map: rocRAND 2^-32 + v 2^-32, cuRAND (CURAND-A2) v 2^-32 + 2^-33, in fp32.
 * CURAND-A3: the offset n of the ordering rule counts the values generated since the generator was created, ACROSS calls (what librocrand does with its 131
 * 072 lanes:
 * tests/test_xorwow_gpu.py); base.json's call sizes are multiples of 4096, for which every reading of the rule gives the same values. */
typedef struct { uint32_t x[5], d; } orc_xw;
static inline uint32_t xw_next(orc_xw* s) {
    const uint32_t t = s->x[0] ^ (s->x[0] >> 2);
    s->x[0] = s->x[1]; s->x[1] = s->x[2]; s->x[2] = s->x[3]; s->x[3] = s->x[4];
    s->x[4] = (s->x[4] ^ (s->x[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437u; return s->d + s->x[4];
}
static void xw_seed(orc_xw* s, uint64_t seed, int rocrand_flavour) {
    const uint32_t s0 = (uint32_t)seed ^ (rocrand_flavour ? 0x2c7f967fu : 0xaad26b49u),
            s1 = (uint32_t)(seed >> 32) ^ (rocrand_flavour ? 0xa03697cbu : 0xf7dcefddu);
    const uint32_t t0 = (rocrand_flavour ? 1228688033u : 1099087573u) * s0, t1 = (rocrand_flavour ? 2073658381u : 2591861531u) * s1;
    s->x[0] = 123456789u + t0; s->x[1] = 362436069u ^ t0; s->x[2] = 521288629u + t1; s->x[3] = 88675123u ^ t1; s->x[4] = 5783321u + t0;
    s->d = 6615241u + t1 + t0;
}
static inline float xw_uniform(uint32_t v, int rocrand_flavour) {
    return rocrand_flavour ? 2.3283064e-10f + ((float)v * 2.3283064e-10f) : (float)v * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }
/* 160 x 160 bit matrices as 160 columns of 5 words: (M v) = xor of the columns whose bit is set in v */
typedef struct { uint32_t col[160][5]; } xw_mat;
static void xw_matvec(const xw_mat* M, const uint32_t v[5], uint32_t out[5]) {
    uint32_t r[5] = { 0, 0, 0, 0, 0 };
    for (int b = 0; b < 160; ++b) if ((v[b >> 5] >> (b & 31)) & 1u) for (int k = 0; k < 5; ++k) r[k] ^= M->col[b][k];
    memcpy(out, r, 20);
}
static void xw_jump_2pow(xw_mat* M, int log2_steps) {          /* M = (one step of the xorshift part) ^ (2 ^ log2_steps) */
    for (int b = 0; b < 160; ++b) {
        orc_xw e; memset(&e, 0, sizeof e); e.x[b >> 5] = 1u << (b & 31); xw_next(&e); memcpy(M->col[b], e.x, 20);
    }
    xw_mat* T = (xw_mat*)malloc(sizeof(xw_mat));
    for (int q = 0; q < log2_steps; ++q) { for (int b = 0; b < 160; ++b) xw_matvec(M, M->col[b], T->col[b]); memcpy(M, T, sizeof(xw_mat)); }
    free(T);
}
/* the host generator: `lanes` states, lane k = seed state jumped k * 2^67 steps */
/* offset: values generated since creation (the n of the ordering rule runs across calls) */
typedef struct { orc_xw* lane; uint32_t lanes; int flavour; uint64_t offset; } orc_xwgen;
static void xwgen_init(orc_xwgen* g, uint64_t seed, int rocrand_flavour, uint32_t lanes) {
    static xw_mat* J = NULL;
    #pragma omp critical(orc_xw_jump)
    { if (!J) { J = (xw_mat*)malloc(sizeof(xw_mat)); xw_jump_2pow(J, 67); } }
    g->lanes = lanes; g->flavour = rocrand_flavour; g->offset = 0; g->lane = (orc_xw*)realloc(g->lane, sizeof(orc_xw) * lanes);
    xw_seed(&g->lane[0], seed, rocrand_flavour);
    for (uint32_t k = 1; k < lanes; ++k) { g->lane[k].d = g->lane[0].d; xw_matvec(J, g->lane[k - 1].x, g->lane[k].x); }
}
/* curandGenerateUniform(gen, out, n): value j of this call sits at offset g->offset + j of the generator's output */
static void xwgen_uniform(orc_xwgen* g, float* out, size_t n) {
    const uint32_t start = (uint32_t)(g->offset % g->lanes);
    for (uint32_t k = 0; k < g->lanes; ++k) for (size_t j = (k + g->lanes - start) % g->lanes; j < n; j += g->lanes) out[j] = xw_uniform(xw_next(&g->lane[k]),
            g->flavour);
    g->offset += n;
}
/* test hooks (tests/test_xorwow.py): raw draws of one lane, and one generate call of a fresh generator */
void orc_xorwow_lane_draws(uint64_t seed, int rocrand_flavour, uint32_t lane, uint32_t n, uint32_t* out) {
    orc_xwgen g; memset(&g, 0, sizeof g); xwgen_init(&g, seed, rocrand_flavour, lane + 1u);
    for (uint32_t i = 0; i < n; ++i) out[i] = xw_next(&g.lane[lane]);
    free(g.lane);
}
void orc_xorwow_generate(uint64_t seed, int rocrand_flavour, uint32_t lanes, uint32_t n_first, uint32_t n_second, float* out_second) {
    /* a fresh generator, one call of n_first values (discarded), then a call of n_second values -> out_second (states carry over) */
    orc_xwgen g; memset(&g, 0, sizeof g); xwgen_init(&g, seed, rocrand_flavour, lanes);
    if (n_first) { float* tmp = (float*)malloc(sizeof(float) * n_first); xwgen_uniform(&g, tmp, n_first); free(tmp); }
    xwgen_uniform(&g, out_second, n_second); free(g.lane);
}
void orc_xorwow_generate_calls(uint64_t seed, int rocrand_flavour, uint32_t lanes, uint32_t n_calls, const uint32_t* sizes, float* out) {
    /* a fresh generator and n_calls generate calls in sequence; outputs back to back */
    orc_xwgen g; memset(&g, 0, sizeof g); xwgen_init(&g, seed, rocrand_flavour, lanes);
    for (uint32_t c = 0; c < n_calls; ++c) { xwgen_uniform(&g, out, sizes[c]); out += sizes[c]; }
    free(g.lane);
}

/* Parameter-init stream: pcg32 as used by tcnn::default_rng_t (public-domain PCG, Jakob's
 * pcg32.h): seed(initstate, initseq=1).  Element k of the parameter vector takes the k-th
 * draw (tcnn's generate_random_uniform interleaves draws per thread; not reproduced). */
typedef struct { uint64_t state, inc; } pcg32;
static uint32_t pcg_next(pcg32* r) {
    uint64_t old = r->state; r->state = old * 0x5851f42d4c957f2dull + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
static void pcg_seed(pcg32* r, uint64_t initstate, uint64_t initseq) {
    r->state = 0; r->inc = (initseq << 1u) | 1u; pcg_next(r); r->state += initstate; pcg_next(r);
}
static float pcg_float(pcg32* r) { uint32_t u = (pcg_next(r) >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f; }

/* ------------------------------------------------------------------ grid geometry (tcnn grid.h) */
static inline uint32_t next_multiple(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

/* TCNN-A1/A2/A4: scale_l = 2^(l*log2 b) * Nmin - 1 ; res = ceil(scale)+1 ;
 * entries = min(round_up(res^3, 8), 2^T). offsets has n_levels+1 entries. Returns E_pad. */
int orc_level_table(const orc_config* c, uint32_t* offsets, float* scales, uint32_t* res) {
    uint32_t off = 0; float l2 = log2f(c->per_level_scale);
    for (int l = 0; l < c->n_levels; ++l) {
        float s = exp2f((float)l * l2) * (float)c->base_resolution - 1.0f;
        uint32_t r = (uint32_t)ceilf(s) + 1u;
        uint64_t dense = (uint64_t)r * r * r; uint32_t maxp = 0xffffffffu / 2;
        uint32_t n = dense > maxp ? maxp : (uint32_t)dense;
        n = next_multiple(n, 8u);
        uint32_t cap = 1u << c->log2_hashmap_size; if (n > cap) n = cap;
        offsets[l] = off; scales[l] = s; res[l] = r; off += n;
    }
    offsets[c->n_levels] = off;
    return (int)next_multiple((uint32_t)(c->n_levels * c->n_features), 16u);
}

/* TCNN-A3: dense index with overflow guard, else coherent-prime hash; then % size. */
static inline uint32_t grid_index(uint32_t size, uint32_t res, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t stride = 1, index = 0; const uint32_t p[3] = { x, y, z };
    for (int d = 0; d < 3 && stride <= size; ++d) { index += p[d] * stride; stride *= res; }
    if (size < stride) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % size;
}
uint32_t orc_grid_index(uint32_t size, uint32_t res, uint32_t x, uint32_t y, uint32_t z) { return grid_index(size, res, x, y, z); }

/* ------------------------------------------------------------------ model */
typedef struct {
    orc_config cfg;
    int L, W, NH, Epad, R, S;
    uint32_t off[ORC_MAX_LEVELS + 1], res[ORC_MAX_LEVELS]; float scale[ORC_MAX_LEVELS];
    uint32_t n_mlp, n_grid, n_params;
    /* parameters, tcnn layout: [MLP matrices row-major (out x in) | grid entries x F] */
    float* master; uint16_t* half; uint16_t* ema; float* m1; float* m2; uint32_t* steps;
    float* gmlp;            /* fp32 dW (n_mlp) */
    float* ggrid;           /* fp32 accumulated grid gradient (n_grid) */
    float* ggrid_abs;       /* sum |contribution| per entry, for tolerance bounds */
    uint16_t* ggrid_h;      /* h() of the above = what Adam consumes */
    uint32_t step;          /* optimizer steps taken */
    uint32_t iter;          /* batches generated: RNG counter, advances even when a batch is skipped */
    float lr; int has_ema;
    /* dataset (host copies) */
    int H, Wimg, n_frames, use_depth; float fx, fy, cx, cy;
    const uint8_t* rgba;    /* [n][H][W][4]: r,g,b,instance (caller owned) */
    const float* depth;     /* [n][H][W] metres or NULL */
    const float* poses;     /* [n][16] Twc column-major */
    /* object */
    float Tow[16]; float amin[3], amax[3]; uint8_t inst;
    orc_bbox* boxes; size_t n_boxes;
    /* last batch (kept for the tests) */
    uint32_t n_valid; uint8_t* valid; uint32_t* sel;
    float *ray_o, *ray_d, *ray_dn, *ray_tmin, *ray_tmax, *target, *target_depth, *bgcol; uint8_t* ray_flag;
    float *pts, *tdist; uint16_t *E, *Hid, *O, *dO, *dHid, *dE;
    float *rgb_ray, *depth_ray, *mask_ray, *loss_ray; float loss;
    /* XORWOW sample stream (cfg.rng_flags): the training generator and this iteration's three arrays SampleXY[2R] | RandColors[3R] | RandDt[S R] */
    /* NeRF_Model::Step schedule (forward_backward_compacted) instead of Step_No_Compacted; samples in the compacted batch */
    int step_variant; uint32_t n_compacted;
    orc_xwgen xw; float* xw_buf; uint32_t xw_iter;     /* xw_iter: the iteration xw_buf holds (UINT32_MAX: none) */
} orc_model;

static uint32_t mlp_params(int W, int NH, int Epad) { return (uint32_t)(W * Epad + (NH - 1) * W * W + ORC_OUT_PAD * W); }

orc_model* orc_create(const orc_config* c) {
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->cfg = *c; m->L = c->n_levels; m->W = c->n_neurons; m->NH = c->n_hidden_layers; m->R = c->rays_per_batch; m->S = c->n_samples;
    m->Epad = orc_level_table(c, m->off, m->scale, m->res);
    m->n_mlp = mlp_params(m->W, m->NH, m->Epad); m->n_grid = m->off[m->L] * 2u; m->n_params = m->n_mlp + m->n_grid;
    size_t n = m->n_params;
    m->master = (float*)calloc(n, 4); m->half = (uint16_t*)calloc(n, 2); m->ema = (uint16_t*)calloc(n, 2);
    m->m1 = (float*)calloc(n, 4); m->m2 = (float*)calloc(n, 4); m->steps = (uint32_t*)calloc(n, 4);
    m->gmlp = (float*)calloc(m->n_mlp, 4); m->ggrid = (float*)calloc(m->n_grid, 4); m->ggrid_abs = (float*)calloc(m->n_grid, 4);
    m->ggrid_h = (uint16_t*)calloc(m->n_grid, 2);
    m->lr = c->learning_rate;
    /* TCNN-A5: MLP Xavier-uniform per matrix, grid U(-1e-4,1e-4); network params first. */
    pcg32 rng; pcg_seed(&rng, c->param_seed, 1u);
    uint32_t k = 0;
    if (!ORC_RNG_TCNN_INIT(c)) {
    for (int layer = 0; layer <= m->NH; ++layer) {
        int rows = (layer == m->NH) ? ORC_OUT_PAD : m->W, cols = (layer == 0) ? m->Epad : m->W;
        float sc = sqrtf(6.0f / (float)(rows + cols));
        for (int i = 0; i < rows * cols; ++i, ++k) m->master[k] = pcg_float(&rng) * (2.0f * sc) - sc;
    }
    for (; k < m->n_params; ++k) m->master[k] = pcg_float(&rng) * 2e-4f - 1e-4f;
    } else {
        /* TCNN-A5b (rng_flags bit 4): tiny-cuda-nn's generate_random_uniform (common_device / random.h as published): one launch per tensor -- every MLP
         * matrix, then the grid -- of ceil(n / (128 * 4)) blocks of 128 threads; thread i advances the generator by 4 i and writes its draws j = 0..3 to
         * element i + n_threads * j; the host generator then advances by n.  So element e of a tensor takes draw 4 (e mod n_threads) + floor(e / n_threads) of the
         * tensor's stretch of the pcg32 sequence, value = draw * (hi - lo) + lo. */
        uint64_t base = 0;
        for (int layer = 0; layer <= m->NH + 1; ++layer) {
            size_t n; float lo, hi;
            if (layer <= m->NH) { int rows = (layer == m->NH) ? ORC_OUT_PAD : m->W, cols = (layer == 0) ? m->Epad : m->W;
                float sc = sqrtf(6.0f / (float)(rows + cols)); n = (size_t)rows * cols; lo = -sc; hi = sc; }
            else { n = m->n_grid; lo = -1e-4f; hi = 1e-4f; }
            const size_t n_threads = ((n + 511) / 512) * 128;
            float* draws = (float*)malloc(sizeof(float) * n_threads * 4);
            pcg32 r2; pcg_seed(&r2, c->param_seed, 1u); for (uint64_t a = 0; a < base; ++a) pcg_next(&r2);
            for (size_t a = 0; a < n_threads * 4; ++a) draws[a] = pcg_float(&r2);
            for (size_t e = 0; e < n; ++e) m->master[k + e] = draws[4 * (e % n_threads) + e / n_threads] * (hi - lo) + lo;
            free(draws); k += (uint32_t)n; base += n;
        }
    }
    for (k = 0; k < m->n_params; ++k) m->half[k] = f2h(m->master[k]);
    size_t R = (size_t)m->R, B = R * (size_t)m->S;
    m->valid = (uint8_t*)calloc(R, 1); m->sel = (uint32_t*)calloc(R, 4);
    m->ray_o = (float*)calloc(R * 3, 4); m->ray_d = (float*)calloc(R * 3, 4); m->ray_dn = (float*)calloc(R, 4);
    m->ray_tmin = (float*)calloc(R, 4); m->ray_tmax = (float*)calloc(R, 4); m->target = (float*)calloc(R * 3, 4);
    m->target_depth = (float*)calloc(R, 4); m->bgcol = (float*)calloc(R * 3, 4); m->ray_flag = (uint8_t*)calloc(R, 1);
    m->pts = (float*)calloc(B * 3, 4); m->tdist = (float*)calloc(B, 4);
    m->E = (uint16_t*)calloc(B * m->Epad, 2); m->Hid = (uint16_t*)calloc(B * m->W * m->NH, 2); m->O = (uint16_t*)calloc(B * ORC_OUT, 2);
    m->dO = (uint16_t*)calloc(B * ORC_OUT, 2); m->dHid = (uint16_t*)calloc(B * m->W * m->NH, 2); m->dE = (uint16_t*)calloc(B * m->Epad, 2);
    m->rgb_ray = (float*)calloc(R * 3, 4); m->depth_ray = (float*)calloc(R, 4); m->mask_ray = (float*)calloc(R, 4); m->loss_ray = (float*)calloc(R, 4);
    m->xw_iter = 0xffffffffu;
    if (ORC_RNG_STREAM(c)) {
        xwgen_init(&m->xw, 0ull /* the generator's default seed: nerf_model.cu never sets one */, ORC_RNG_STREAM(c) == 2u, ORC_RNG_LANES(c));
        m->xw_buf = (float*)calloc((5 + (size_t)m->S) * R, 4);
    }
    return m;
}
void orc_destroy(orc_model* m) {
    if (!m) return;
    void* p[] = { m->master, m->half, m->ema, m->m1, m->m2, m->steps, m->gmlp, m->ggrid, m->ggrid_abs, m->ggrid_h, m->boxes, m->valid, m->sel,
        m->ray_o, m->ray_d, m->ray_dn, m->ray_tmin, m->ray_tmax, m->target, m->target_depth, m->bgcol, m->ray_flag, m->pts, m->tdist,
        m->E, m->Hid, m->O, m->dO, m->dHid, m->dE, m->rgb_ray, m->depth_ray, m->mask_ray, m->loss_ray };
    for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i) free(p[i]);
    free(m->xw.lane); free(m->xw_buf);
    free(m);
}
uint32_t orc_n_params(const orc_model* m) { return m->n_params; }
uint32_t orc_n_mlp_params(const orc_model* m) { return m->n_mlp; }
uint32_t orc_step(const orc_model* m) { return m->step; }
uint32_t orc_n_valid(const orc_model* m) { return m->n_valid; }
float orc_loss(const orc_model* m) { return m->loss; }
int orc_epad(const orc_model* m) { return m->Epad; }

/* which: 0 master f32, 1 half u16, 2 ema u16, 3 m1, 4 m2, 5 steps, 6 gmlp f32, 7 ggrid f32, 8 ggrid_abs f32, 9 ggrid_h u16,
 * 10 pts, 11 tdist, 12 E, 13 Hid, 14 O, 15 dO, 16 dHid, 17 dE, 18 rgb_ray, 19 depth_ray, 20 mask_ray, 21 loss_ray,
 * 22 ray_o, 23 ray_d, 24 ray_tmin, 25 ray_tmax, 26 target, 27 target_depth, 28 bgcol, 29 ray_flag, 30 sel, 31 ray_dn */
const void* orc_buffer(const orc_model* m, int which) {
    const void* t[] = { m->master, m->half, m->ema, m->m1, m->m2, m->steps, m->gmlp, m->ggrid, m->ggrid_abs, m->ggrid_h,
        m->pts, m->tdist, m->E, m->Hid, m->O, m->dO, m->dHid, m->dE, m->rgb_ray, m->depth_ray, m->mask_ray, m->loss_ray,
        m->ray_o, m->ray_d, m->ray_tmin, m->ray_tmax, m->target, m->target_depth, m->bgcol, m->ray_flag, m->sel, m->ray_dn };
    if (which < 0 || which >= (int)(sizeof(t) / sizeof(t[0]))) return NULL;
    return t[which];
}
void orc_set_params(orc_model* m, const float* master) {
    memcpy(m->master, master, (size_t)m->n_params * 4);
    for (uint32_t k = 0; k < m->n_params; ++k) m->half[k] = f2h(m->master[k]);
}
/* test hook: load inference (EMA) weights, fp16 bit patterns */
void orc_set_ema(orc_model* m, const uint16_t* ema) { memcpy(m->ema, ema, (size_t)m->n_params * 2); m->has_ema = 1; }
void orc_set_dataset(orc_model* m, int H, int W, int n_frames, float fx, float fy, float cx, float cy,
                     const uint8_t* rgba, const float* depth, const float* poses) {
    m->H = H; m->Wimg = W; m->n_frames = n_frames; m->fx = fx; m->fy = fy; m->cx = cx; m->cy = cy;
    m->rgba = rgba; m->depth = depth; m->poses = poses; m->use_depth = (depth != NULL) && m->cfg.use_depth;
}
void orc_set_object(orc_model* m, const float* Tow16, const float* amin, const float* amax, int instance_id) {
    memcpy(m->Tow, Tow16, 64); memcpy(m->amin, amin, 12); memcpy(m->amax, amax, 12); m->inst = (uint8_t)instance_id;
}
void orc_add_boxes(orc_model* m, const orc_bbox* b, size_t n) {   /* nerf_model.cu:1609-1628 */
    m->boxes = (orc_bbox*)realloc(m->boxes, (m->n_boxes + n) * sizeof(orc_bbox));
    memcpy(m->boxes + m->n_boxes, b, n * sizeof(orc_bbox)); m->n_boxes += n;
}

/* ------------------------------------------------------------------ geometry */
/* column-major 4x4: M(r,c) = m[c*4+r].  rot(M) * v with fmaf chains. */
static inline void rot3(const float* M, const float* v, float* o) {
    for (int r = 0; r < 3; ++r) o[r] = fmaf(M[8 + r], v[2], fmaf(M[4 + r], v[1], M[r] * v[0]));
}
/* nerf_model.cu:87-138 slab test; returns 0 on miss. No special-casing of dir==0 (inf arithmetic). */
static int ray_intersect(const float* bmin, const float* bmax, const float* o, const float* d, float* t0, float* t1) {
    float tmin = (bmin[0] - o[0]) / d[0], tmax = (bmax[0] - o[0]) / d[0], t;
    if (tmin > tmax) { t = tmin; tmin = tmax; tmax = t; }
    float tymin = (bmin[1] - o[1]) / d[1], tymax = (bmax[1] - o[1]) / d[1];
    if (tymin > tymax) { t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (bmin[2] - o[2]) / d[2], tzmax = (bmax[2] - o[2]) / d[2];
    if (tzmin > tzmax) { t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *t0 = tmin; *t1 = tmax; return 1;
}
/* pixel -> object-frame ray.  nerf_model.cu:403-413 (train), :467-477 (render), :511-518 (video: Tow==NULL). */
static void pixel_ray(const orc_model* m, float px, float py, const float* Twc, const float* Tow, float* o, float* d, float* dn) {
    float dc[3] = { (px - m->cx) / m->fx, (py - m->cy) / m->fy, 1.0f };
    float n = sqrtf(fmaf(dc[2], dc[2], fmaf(dc[1], dc[1], dc[0] * dc[0])));
    float dnrm[3] = { dc[0] / n, dc[1] / n, dc[2] / n }, dw[3];
    rot3(Twc, dnrm, dw);
    if (Tow) { rot3(Tow, dw, d); float ow[3] = { Twc[12], Twc[13], Twc[14] }, t[3]; rot3(Tow, ow, t); for (int r = 0; r < 3; ++r) o[r] = t[r] + Tow[12 + r]; }
    else { for (int r = 0; r < 3; ++r) { d[r] = dw[r]; o[r] = Twc[12 + r]; } }
    *dn = n;
}

/* ------------------------------------------------------------------ GenerateBatch
 * nerf_model.cu:1429-1502 = GenerateRays (:369-446) + fill_rollover_rays (:280-294) +
 * GenerateInputPoints (:536-566).  The atomicAdd compaction order of the reference is
 * unspecified; this restatement (and the HIP path) use the stable candidate order. */
/* one uniform of iteration `step`: stream 0 SampleXY[2R], 1 RandColors[3R], 2 RandDt[S R].  XORWOW mode: the three arrays of an iteration are generated
 * once, in the reference's order (nerf_model.cu:1432,1434,1468), by the model's one host generator; iterations that were skipped over (advance_iter) consume their draws
 * too. */
static void xw_ensure(orc_model* m) {
    const size_t R = (size_t)m->R, S = (size_t)m->S;
    while (m->xw_iter == 0xffffffffu || m->xw_iter < m->iter) {
        xwgen_uniform(&m->xw, m->xw_buf, 2 * R); xwgen_uniform(&m->xw, m->xw_buf + 2 * R, 3 * R); xwgen_uniform(&m->xw, m->xw_buf + 5 * R, S * R);
        m->xw_iter = (m->xw_iter == 0xffffffffu) ? 0u : m->xw_iter + 1u;
    }
}
static inline float batch_rand(const orc_model* m, uint32_t stream, uint32_t step, uint32_t idx) {
    if (!ORC_RNG_STREAM(&m->cfg)) return rand01(m->cfg.sample_seed, stream, step, idx);
    return m->xw_buf[(stream == 0 ? 0u : stream == 1 ? 2u * (uint32_t)m->R : 5u * (uint32_t)m->R) + idx];
}
static void generate_batch(orc_model* m) {
    const orc_config* c = &m->cfg; const int R = m->R, S = m->S; const uint32_t step = m->iter;
    if (ORC_RNG_STREAM(c)) xw_ensure(m);
    float *co = (float*)malloc((size_t)R * 3 * 4), *cd = (float*)malloc((size_t)R * 3 * 4), *cdn = (float*)malloc((size_t)R * 4),
          *ct0 = (float*)malloc((size_t)R * 4), *ct1 = (float*)malloc((size_t)R * 4), *ctg = (float*)malloc((size_t)R * 3 * 4),
                  *ctd = (float*)malloc((size_t)R * 4);
    uint8_t* cfl = (uint8_t*)malloc((size_t)R);
    uint32_t nv = 0;
    for (int i = 0; i < R; ++i) {
        m->valid[i] = 0;
        const orc_bbox* b = &m->boxes[(size_t)i % m->n_boxes];
        float u0 = batch_rand(m, 0, step, 2u * i), u1 = batch_rand(m, 0, step, 2u * i + 1u);
        uint32_t x = b->x + (uint32_t)(u0 * (float)(int)b->w), y = b->y + (uint32_t)(u1 * (float)(int)b->h);
        /* guard (XORWOW uniforms reach 1.0: a box that touches the image border would read past the row, as the reference does) */
        if (x > (uint32_t)m->Wimg - 1u) x = (uint32_t)m->Wimg - 1u;
        if (y > (uint32_t)m->H - 1u) y = (uint32_t)m->H - 1u;
        size_t pix = ((size_t)b->FrameId * m->H + y) * m->Wimg + x;
        uint8_t inst = m->rgba[pix * 4 + 3];
        if (inst != 0 && inst != m->inst) continue;                            /* occlusion :398-401 */
        float o[3], d[3], dn, t0, t1;
        pixel_ray(m, (float)x, (float)y, m->poses + (size_t)b->FrameId * 16, m->Tow, o, d, &dn);
        if (!ray_intersect(m->amin, m->amax, o, d, &t0, &t1)) continue;
        m->valid[i] = 1; m->sel[nv++] = (uint32_t)i;
        memcpy(co + 3 * i, o, 12); memcpy(cd + 3 * i, d, 12); cdn[i] = dn; ct0[i] = fmaxf(t0, 0.0f); ct1[i] = t1;
        if (inst != 0) {
            for (int k = 0; k < 3; ++k) ctg[3 * i + k] = (float)m->rgba[pix * 4 + k] / 255.0f;   /* nerf_data.cu:169 convertTo(1/255) */
            ctd[i] = m->use_depth ? m->depth[pix] * dn : 0.0f; cfl[i] = 1;
        } else { ctd[i] = 0.0f; cfl[i] = 0; }
    }
    m->n_valid = nv;
    if (nv == 0) goto done;     /* reference: i % 0 UB (:287,:760); here the step is skipped */
    for (int j = 0; j < R; ++j) {
        uint32_t k = (uint32_t)j % nv, i = m->sel[k];
        for (int a = 0; a < 3; ++a) m->bgcol[3 * j + a] = batch_rand(m, 1, step, 3u * k + a);     /* :760 RandomColor[(i % n)*3] */
        memcpy(m->ray_o + 3 * j, co + 3 * i, 12); memcpy(m->ray_d + 3 * j, cd + 3 * i, 12);
        m->ray_dn[j] = cdn[i]; m->ray_tmin[j] = ct0[i]; m->ray_tmax[j] = ct1[i]; m->ray_flag[j] = cfl[i]; m->target_depth[j] = ctd[i];
        for (int a = 0; a < 3; ++a) m->target[3 * j + a] = cfl[i] ? ctg[3 * i + a] : m->bgcol[3 * j + a];     /* :438-441 */
        float dt = (ct1[i] - ct0[i]) / (float)S;
        for (int n = 0; n < S; ++n) {                                           /* :553-566 */
            float t = fmaf(dt, (float)n + batch_rand(m, 2, step, (uint32_t)(j * S + n)), ct0[i]);
            size_t s = (size_t)j * S + n;
            for (int a = 0; a < 3; ++a) { float p = fmaf(t, cd[3 * i + a], co[3 * i + a]); m->pts[3 * s + a] = (p - m->amin[a]) / (m->amax[a] - m->amin[a]); }
            m->tdist[s] = t;
        }
    }
done:
    free(co); free(cd); free(cdn); free(ct0); free(ct1); free(ctg); free(ctd); free(cfl);
}

/* ------------------------------------------------------------------ hash-grid encode (tcnn kernel_grid) */
typedef struct { uint32_t idx[8]; float w[8]; } corners;
static inline void level_corners(const orc_model* m, int l, const float* x, corners* c) {
    float pos[3]; uint32_t pg[3];
    for (int d = 0; d < 3; ++d) { float p = fmaf(m->scale[l], x[d], 0.5f), fl = floorf(p); pg[d] = (uint32_t)(int32_t)fl; pos[d] = p - fl; }
    uint32_t size = m->off[l + 1] - m->off[l];
    for (int k = 0; k < 8; ++k) {
        float w = 1.0f; uint32_t q[3];
        for (int d = 0; d < 3; ++d) { if (k & (1 << d)) { w *= pos[d]; q[d] = pg[d] + 1u; } else { w *= 1.0f - pos[d]; q[d] = pg[d]; } }
        c->w[k] = w; c->idx[k] = m->off[l] + grid_index(size, m->res[l], q[0], q[1], q[2]);
    }
}
static void encode_one(const orc_model* m, const uint16_t* table /* grid part, [entry][2] */, const float* x, uint16_t* E) {
    const int tcnn = m->cfg.numerics_flags & ORC_NUM_TCNN_HALF;
    for (int l = 0; l < m->L; ++l) {
        corners c; level_corners(m, l, x, &c); float a0 = 0.0f, a1 = 0.0f;
        if (tcnn) {          /* tcnn kernel_grid: ((T*)&result)[f] += (T)(weight * (float)val[f]) with T = __half */
            uint16_t r0 = 0, r1 = 0;
            for (int k = 0; k < 8; ++k) { r0 = f2h(h2f(r0) + h2f(f2h(c.w[k] * h2f(table[2 * c.idx[k]]))));
                r1 = f2h(h2f(r1) + h2f(f2h(c.w[k] * h2f(table[2 * c.idx[k] + 1])))); }
            E[2 * l] = r0; E[2 * l + 1] = r1; continue;
        }
        for (int k = 0; k < 8; ++k) { a0 = fmaf(c.w[k], h2f(table[2 * c.idx[k]]), a0); a1 = fmaf(c.w[k], h2f(table[2 * c.idx[k] + 1]), a1); }
        E[2 * l] = f2h(a0); E[2 * l + 1] = f2h(a1);
    }
    for (int k = 2 * m->L; k < m->Epad; ++k) E[k] = 0;       /* TCNN-A9 zero padding */
}
/* dot product of two fp16 vectors (strides sa, sb).  Contract (tcnn = 0): one fp32 fmaf chain.  ORC_NUM_TCNN_HALF: the __half WMMA
 * accumulator of tcnn's fully fused MLP -- every 16-wide k block adds its (fp32-summed) products into an fp16 running value. */
static inline float dot_h(const uint16_t* a, int sa, const uint16_t* b, int sb, int n, int tcnn) {
    if (!tcnn) { float acc = 0.0f; for (int k = 0; k < n; ++k) acc = fmaf(h2f(a[k * sa]), h2f(b[k * sb]), acc); return acc; }
    uint16_t acc = 0;
    for (int k0 = 0; k0 < n; k0 += 16) {
        float blk = 0.0f; const int k1 = k0 + 16 < n ? k0 + 16 : n;
        for (int k = k0; k < k1; ++k) blk = fmaf(h2f(a[k * sa]), h2f(b[k * sb]), blk);
        acc = f2h(h2f(acc) + blk);
    }
    return h2f(acc);
}
/* fully-fused MLP forward (tcnn; no biases, ReLU hidden, linear output; TCNN-A10) */
static void mlp_forward_one(const orc_model* m, const uint16_t* w, const uint16_t* E, uint16_t* hid /* NH*W */, uint16_t* out /* 4 */) {
    const int W = m->W, tcnn = m->cfg.numerics_flags & ORC_NUM_TCNN_HALF; const uint16_t* in = E; int nin = m->Epad;
    for (int layer = 0; layer < m->NH; ++layer) {
        for (int u = 0; u < W; ++u) { float a = dot_h(w + u * nin, 1, in, 1, nin, tcnn); hid[layer * W + u] = f2h(a > 0.0f ? a : 0.0f); }
        w += W * nin; in = hid + layer * W; nin = W;
    }
    for (int o = 0; o < ORC_OUT; ++o) out[o] = f2h(dot_h(w + o * W, 1, in, 1, W, tcnn));
}
/* stand-alone stage entry points (tests): params = fp16 parameter vector [n_params] */
void orc_encode(const orc_model* m, const uint16_t* params, const float* x, size_t n, uint16_t* E) {
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) encode_one(m, params + m->n_mlp, x + 3 * i, E + (size_t)i * m->Epad);
}
void orc_mlp_forward(const orc_model* m, const uint16_t* params, const uint16_t* E, size_t n, uint16_t* hid, uint16_t* out) {
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) mlp_forward_one(m, params, E + (size_t)i * m->Epad, hid + (size_t)i * m->W * m->NH, out + (size_t)i * ORC_OUT);
}

/* ------------------------------------------------------------------ activations nerf_model.cu:22-64 */
static inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float clampf(float x, float a, float b) { return x < a ? a : (x > b ? b : x); }

/* VolumeRender nerf_model.cu:735-815 (one ray).  out4: fp16 [S][4]; t: [S]. */
static void composite_ray(const uint16_t* out4, const float* t, int S, const float* bg, float* rgb, float* depth, float* mask) {
    float T = 1.0f, r[3] = { 0, 0, 0 }, dep = 0.0f, last = 0.0f;          /* :770 last_distance = 0 (first dt measured from the origin) */
    for (int n = 0; n < S; ++n) {
        if (T < 1e-4f) break;
        float c0 = logistic(h2f(out4[4 * n])), c1 = logistic(h2f(out4[4 * n + 1])), c2 = logistic(h2f(out4[4 * n + 2]));
        float cur = t[n], dt = cur - last, sigma = expf(h2f(out4[4 * n + 3]));        /* :49 unclamped */
        float alpha = 1.0f - expf(-sigma * dt), w = alpha * T;
        r[0] += w * c0; r[1] += w * c1; r[2] += w * c2; dep += w * cur; T *= (1.0f - alpha); last = cur;
    }
    rgb[0] = r[0] + T * bg[0]; rgb[1] = r[1] + T * bg[1]; rgb[2] = r[2] + T * bg[2]; *depth = dep; *mask = 1.0f - T;
}
/* VolumeRenderGradient_No_Compacted nerf_model.cu:817-954 (one ray).  dO pre-zeroed (:1578). */
static float gradient_ray(const uint16_t* out4, const float* t, int S, int nRays, float loss_scale, int is_obj,
                          const float* target, float target_depth, const float* rgb_ray, float depth_ray, float mask_ray, uint16_t* dO) {
    float g[3], lsum = 0.0f;
    for (int k = 0; k < 3; ++k) { float d = rgb_ray[k] - target[k]; lsum += d * d; g[k] = 2.0f * d; }       /* :78-84 */
    float mean_loss = lsum / 3.0f, dl_dd = 0.0f;
    if (target_depth > 0.0f) dl_dd = 0.5f * ((depth_ray - target_depth >= 0.0f) ? 1.0f : -1.0f);          /* :869-871 */
    float loss = is_obj ? mean_loss + dl_dd * (depth_ray - target_depth) + (1.0f - mask_ray) : mean_loss + mask_ray;  /* :877-880 */
    float ls = loss_scale / (float)nRays, T = 1.0f, r2[3] = { 0, 0, 0 }, d2 = 0.0f, last = 0.0f;
    for (int n = 0; n < S; ++n) {
        if (T < 1e-4f) break;
        float v[4]; for (int k = 0; k < 4; ++k) v[k] = h2f(out4[4 * n + k]);
        float c[3] = { logistic(v[0]), logistic(v[1]), logistic(v[2]) };
        float cur = t[n], dt = cur - last, sigma = expf(v[3]);
        float alpha = 1.0f - expf(-sigma * dt), w = alpha * T;
        for (int k = 0; k < 3; ++k) r2[k] += w * c[k];
        d2 += w * cur; T *= (1.0f - alpha);
        float suf[3] = { rgb_ray[0] - r2[0], rgb_ray[1] - r2[1], rgb_ray[2] - r2[2] };
        for (int k = 0; k < 3; ++k) dO[4 * n + k] = f2h(ls * ((w * g[k]) * (c[k] * (1.0f - c[k]))));        /* :916-920 */
        float dsig = expf(clampf(v[3], -15.0f, 15.0f));                                                   /* :60 */
        float depth_sup = dl_dd * (T * cur - (depth_ray - d2));                                            /* :924-925 */
        float dmask = 1.0f - mask_ray, dl;
        if (is_obj) {
            float dlm = 0.5f * (mask_ray >= 1.0f ? 1.0f : -1.0f);                                          /* :928 */
            float dot = g[0] * (T * c[0] - suf[0]) + g[1] * (T * c[1] - suf[1]) + g[2] * (T * c[2] - suf[2]);
            dl = dsig * dt * (dot + depth_sup + dlm * dmask);                                              /* :933-934 */
        } else {
            float dlm = 0.5f * (mask_ray >= 0.0f ? 1.0f : -1.0f);                                          /* :938 */
            dl = dsig * dt * dlm * dmask + dsig * 0.01f;                                                   /* :940 */
        }
        dO[4 * n + 3] = f2h(ls * dl); last = cur;
    }
    return loss;
}
/* stage entry points for KATs */
void orc_composite(const uint16_t* out4, const float* t, int S, const float* bg, float* rgb, float* depth, float* mask) {
    composite_ray(out4, t, S, bg, rgb, depth, mask); }
float orc_gradient(const uint16_t* out4, const float* t, int S, int nRays, float loss_scale, int is_obj, const float* target, float target_depth,
                   const float* rgb_ray, float depth_ray, float mask_ray, uint16_t* dO) {
    memset(dO, 0, (size_t)S * 4 * 2);
    return gradient_ray(out4, t, S, nRays, loss_scale, is_obj, target, target_depth, rgb_ray, depth_ray, mask_ray, dO);
}

static int g_parallel_scatter = 0;
void orc_set_parallel_scatter(int on) { g_parallel_scatter = on; }

/* ------------------------------------------------------------------ Step_No_Compacted nerf_model.cu:1552-1607 */
static void network_forward(orc_model* m) {          /* encode + MLP of every sample of m->pts (tcnn forward / inference, call sites :1557, :1509) */
    const int R = m->R, S = m->S, W = m->W, NH = m->NH, Ep = m->Epad; const size_t B = (size_t)R * S;
    const uint16_t* wt = m->half; const uint16_t* table = m->half + m->n_mlp;
    #pragma omp parallel for schedule(static)
    for (long s = 0; s < (long)B; ++s) {
        encode_one(m, table, m->pts + 3 * s, m->E + (size_t)s * Ep);
        mlp_forward_one(m, wt, m->E + (size_t)s * Ep, m->Hid + (size_t)s * W * NH, m->O + (size_t)s * 4);
    }
}
static void network_backward(orc_model* m);
static void forward_backward_compacted(orc_model* m);
static void forward_backward(orc_model* m) {
    if (m->step_variant) { forward_backward_compacted(m); return; }
    const int R = m->R, S = m->S; const size_t B = (size_t)R * S;
    network_forward(m);
    memset(m->dO, 0, B * 4 * 2);                                                 /* :1578 */
    #pragma omp parallel for schedule(static)
    for (long j = 0; j < R; ++j) {
        const uint16_t* o4 = m->O + (size_t)j * S * 4; const float* t = m->tdist + (size_t)j * S;
        composite_ray(o4, t, S, m->bgcol + 3 * j, m->rgb_ray + 3 * j, m->depth_ray + j, m->mask_ray + j);
        m->loss_ray[j] = gradient_ray(o4, t, S, R, m->cfg.loss_scale, m->ray_flag[j], m->target + 3 * j, m->target_depth[j],
                                      m->rgb_ray + 3 * j, m->depth_ray[j], m->mask_ray[j], m->dO + (size_t)j * S * 4);
    }
    double ls = 0; for (int j = 0; j < R; ++j) ls += m->loss_ray[j];             /* SumLoss :1231-1253 + :1650-1658 */
    m->loss = (float)(ls / R);
    network_backward(m);
}

/* NeRF_Model::Step (nerf_model.cu:1504-1550, "unavailable, for reference only": never called by either driver) -- the schedule with per-ray SAMPLE
 * compaction: (1) inference of every sample with the training weights (:1509); (2) VolumeRenderGradient (:957-1132): per ray, composite until T < 1e-4
 * (numsteps
 * samples), L2 loss on the colour only (no depth / mask terms), background = ONE colour for all rays (the kernel's by-value copy of the pcg32 generator: every
 * thread draws the same three floats, :1038) -- here the iteration's first three RandColors; the numsteps positions and their dL/dO go to a compacted batch
 * (reference: slot = atomicAdd in arrival order; here in ray order); (3) fill_rollover_and_rescale (:269-279) + fill_rollover (:258-266): the compacted
 * batch of n
 * samples is repeated cyclically up to the full batch size B and ONLY THE COPIES' gradients are scaled by n / B (the originals keep theirs: `i < n * stride`
 * returns
 * early); (4) forward + backward of the full-size compacted batch (:1545-1548), then the optimizer step as usual.  m->pts / m->dO hold the compacted batch
 * afterwards. */
static void forward_backward_compacted(orc_model* m) {
    const int R = m->R, S = m->S; const size_t B = (size_t)R * S; const orc_config* c = &m->cfg;
    network_forward(m);
    uint32_t* steps = (uint32_t*)calloc((size_t)R + 1, 4);
    float bg[3]; for (int a = 0; a < 3; ++a) bg[a] = batch_rand(m, 1, m->iter, (uint32_t)a);
    const float ls = c->loss_scale / (float)R;
    #pragma omp parallel for schedule(static)
    for (long j = 0; j < R; ++j) {                                               /* first loop of the kernel: :996-1036 */
        const uint16_t* o4 = m->O + (size_t)j * S * 4; const float* t = m->tdist + (size_t)j * S;
        float T = 1.0f, r[3] = { 0, 0, 0 }, dep = 0.0f, last = 0.0f; int n = 0;
        for (; n < S; ++n) {
            if (T < 1e-4f) break;
            float c0 = logistic(h2f(o4[4 * n])), c1 = logistic(h2f(o4[4 * n + 1])), c2 = logistic(h2f(o4[4 * n + 2]));
            float dt = t[n] - last, sigma = expf(h2f(o4[4 * n + 3])), alpha = 1.0f - expf(-sigma * dt), w = alpha * T;
            /* (depth: |point - o| = t for a unit direction) */
            r[0] += w * c0; r[1] += w * c1; r[2] += w * c2; dep += w * t[n]; T *= (1.0f - alpha); last = t[n];
        }
        for (int a = 0; a < 3; ++a) m->rgb_ray[3 * j + a] = r[a] + T * bg[a];
        m->depth_ray[j] = dep; m->mask_ray[j] = 1.0f - T; steps[j + 1] = (uint32_t)n;
    }
    for (int j = 0; j < R; ++j) steps[j + 1] += steps[j];                        /* exclusive prefix: ray order instead of atomicAdd's arrival order */
    const uint32_t n_comp = steps[R];
    float* pc = (float*)calloc(B * 3, 4); uint16_t* dc = (uint16_t*)calloc(B * 4, 2);
    #pragma omp parallel for schedule(static)
    for (long j = 0; j < R; ++j) {                                               /* second loop: :1048-1131 */
        const uint16_t* o4 = m->O + (size_t)j * S * 4; const float* t = m->tdist + (size_t)j * S; const float* tg = m->target + 3 * j;
        const float* rr = m->rgb_ray + 3 * j; const uint32_t base = steps[j], ns = steps[j + 1] - steps[j];
        float g[3], e2 = 0.0f; for (int a = 0; a < 3; ++a) { float d = rr[a] - tg[a]; g[a] = 2.0f * d; e2 += d * d; }
        m->loss_ray[j] = e2 / 3.0f;
        float T = 1.0f, q[3] = { 0, 0, 0 }, last = 0.0f;
        for (uint32_t n = 0; n < ns; ++n) {
            if (T < 1e-4f) break;
            size_t s = (size_t)j * S + n, d = (size_t)base + n;
            for (int a = 0; a < 3; ++a) pc[3 * d + a] = m->pts[3 * s + a];
            float v3 = h2f(o4[4 * n + 3]), cc[3] = { logistic(h2f(o4[4 * n])), logistic(h2f(o4[4 * n + 1])), logistic(h2f(o4[4 * n + 2])) };
            float dt = t[n] - last, sigma = expf(v3), alpha = 1.0f - expf(-sigma * dt), w = alpha * T; last = t[n];
            for (int a = 0; a < 3; ++a) q[a] += w * cc[a];
            T *= (1.0f - alpha);
            float dot = 0.0f;
            for (int a = 0; a < 3; ++a) { dc[4 * d + a] = f2h(ls * ((w * g[a]) * (cc[a] * (1.0f - cc[a])))); dot += g[a] * (T * cc[a] - (rr[a] - q[a])); }
            float dsig = expf(fminf(fmaxf(v3, -15.0f), 15.0f));
            dc[4 * d + 3] = f2h(ls * (dsig * (dt * dot)));
        }
    }
    double lsum = 0; for (int j = 0; j < R; ++j) lsum += m->loss_ray[j];
    m->loss = (float)(lsum / R);
    if (n_comp > 0) for (size_t i = n_comp; i < B; ++i) {                         /* fill_rollover + fill_rollover_and_rescale: copies only, i >= n */
        size_t src = i % n_comp;
        for (int a = 0; a < 3; ++a) pc[3 * i + a] = pc[3 * src + a];
        for (int a = 0; a < 4; ++a) dc[4 * i + a] = f2h((h2f(dc[4 * src + a]) * (float)n_comp) / (float)B);
    }
    memcpy(m->pts, pc, B * 3 * 4); memcpy(m->dO, dc, B * 4 * 2); m->n_compacted = n_comp;
    free(pc); free(dc); free(steps);
    network_forward(m);                                                          /* :1545 forward of the compacted batch (keeps E / hidden activations) */
    network_backward(m);                                                         /* :1547 */
}

static void network_backward(orc_model* m) {
    const int R = m->R, S = m->S, W = m->W, NH = m->NH, Ep = m->Epad; const size_t B = (size_t)R * S;
    const uint16_t* wt = m->half;
    const int tcnn = m->cfg.numerics_flags & ORC_NUM_TCNN_HALF, grid_half = (m->cfg.numerics_flags & (ORC_NUM_GRID_HALF | ORC_NUM_TCNN_HALF)) != 0;
    /* tcnn backward (EGradientMode::Overwrite): dh, dE per sample */
    #pragma omp parallel for schedule(static)
    for (long s = 0; s < (long)B; ++s) {
        const uint16_t* dO = m->dO + (size_t)s * 4; const uint16_t* hid = m->Hid + (size_t)s * W * NH;
        uint16_t* dh = m->dHid + (size_t)s * W * NH; uint16_t* dE = m->dE + (size_t)s * Ep;
        const uint16_t* wout = wt + (size_t)W * Ep + (size_t)(NH - 1) * W * W;
        for (int u = 0; u < W; ++u) {
            float a = dot_h(wout + u, W, dO, 1, ORC_OUT, tcnn);      /* the 12 padded output rows carry no gradient */
            dh[(NH - 1) * W + u] = f2h(h2f(hid[(NH - 1) * W + u]) > 0.0f ? a : 0.0f);
        }
        for (int layer = NH - 1; layer >= 1; --layer) {
            const uint16_t* wl = wt + (size_t)W * Ep + (size_t)(layer - 1) * W * W;   /* maps layer-1 -> layer */
            for (int k = 0; k < W; ++k) {
                float a = dot_h(wl + k, W, dh + layer * W, 1, W, tcnn);
                dh[(layer - 1) * W + k] = f2h(h2f(hid[(layer - 1) * W + k]) > 0.0f ? a : 0.0f);
            }
        }
        for (int k = 0; k < Ep; ++k) dE[k] = f2h(dot_h(wt + k, Ep, dh, 1, W, tcnn));
    }
    /* weight gradients dW = sum_s d(out) x in^T, fp32, sample order */
    memset(m->gmlp, 0, (size_t)m->n_mlp * 4);
    {
        int nthreads = 1;
        #ifdef _OPENMP
        nthreads = omp_get_max_threads();
        #endif
        float* part = (float*)calloc((size_t)nthreads * m->n_mlp, 4);
        #pragma omp parallel
        {
            int tid = 0;
            #ifdef _OPENMP
            tid = omp_get_thread_num();
            #endif
            float* g = part + (size_t)tid * m->n_mlp;
            #pragma omp for schedule(static)
            for (long s = 0; s < (long)B; ++s) {
                const uint16_t* E = m->E + (size_t)s * Ep; const uint16_t* hid = m->Hid + (size_t)s * W * NH;
                const uint16_t* dh = m->dHid + (size_t)s * W * NH; const uint16_t* dO = m->dO + (size_t)s * 4;
                float* g0 = g;
                for (int u = 0; u < W; ++u) { float d = h2f(dh[u]); if (d != 0.0f) for (int k = 0; k < Ep; ++k) g0[u * Ep + k] += d * h2f(E[k]); }
                for (int layer = 1; layer < NH; ++layer) {
                    float* gl = g + (size_t)W * Ep + (size_t)(layer - 1) * W * W;
                    for (int u = 0; u < W; ++u) { float d = h2f(dh[layer * W + u]);
                        if (d != 0.0f) for (int k = 0; k < W; ++k) gl[u * W + k] += d * h2f(hid[(layer - 1) * W + k]); }
                }
                float* go = g + (size_t)W * Ep + (size_t)(NH - 1) * W * W;
                for (int c = 0; c < ORC_OUT; ++c) { float d = h2f(dO[c]);
                    if (d != 0.0f) for (int k = 0; k < W; ++k) go[c * W + k] += d * h2f(hid[(NH - 1) * W + k]); }
            }
        }
        for (int t = 0; t < nthreads; ++t) for (uint32_t k = 0; k < m->n_mlp; ++k) m->gmlp[k] += part[(size_t)t * m->n_mlp + k];
        free(part);
        /* tcnn: the weight-gradient GEMMs write network_precision_t (fp16) */
        if (tcnn) for (uint32_t k = 0; k < m->n_mlp; ++k) m->gmlp[k] = h2f(f2h(m->gmlp[k]));
    }
    /* grid backward (tcnn kernel_grid_backward): contribution = h(w * dE) per corner; serial for determinism */
    memset(m->ggrid, 0, (size_t)m->n_grid * 4); memset(m->ggrid_abs, 0, (size_t)m->n_grid * 4);
    if (grid_half) memset(m->ggrid_h, 0, (size_t)m->n_grid * 2);
    if (g_parallel_scatter && !grid_half) {
        /* CPU-baseline mode (bench.py): same contributions, accumulated with fp32 atomics in thread order */
        #pragma omp parallel for schedule(static)
        for (long s = 0; s < (long)B; ++s) {
            const uint16_t* dE = m->dE + (size_t)s * Ep;
            for (int l = 0; l < m->L; ++l) {
                float g0 = h2f(dE[2 * l]), g1 = h2f(dE[2 * l + 1]);
                if (g0 == 0.0f && g1 == 0.0f) continue;
                corners c; level_corners(m, l, m->pts + 3 * s, &c);
                for (int k = 0; k < 8; ++k) {
                    float c0 = h2f(f2h(c.w[k] * g0)), c1 = h2f(f2h(c.w[k] * g1)); size_t e = 2 * (size_t)c.idx[k];
                    #pragma omp atomic
                    m->ggrid[e] += c0;
                    #pragma omp atomic
                    m->ggrid[e + 1] += c1;
                }
            }
        }
    } else
    for (size_t s = 0; s < B; ++s) {
        const uint16_t* dE = m->dE + s * Ep;
        for (int l = 0; l < m->L; ++l) {
            float g0 = h2f(dE[2 * l]), g1 = h2f(dE[2 * l + 1]);
            if (g0 == 0.0f && g1 == 0.0f) continue;
            corners c; level_corners(m, l, m->pts + 3 * s, &c);
            for (int k = 0; k < 8; ++k) {
                float c0 = h2f(f2h(c.w[k] * g0)), c1 = h2f(f2h(c.w[k] * g1)); size_t e = 2 * (size_t)c.idx[k];
                m->ggrid[e] += c0; m->ggrid[e + 1] += c1; m->ggrid_abs[e] += fabsf(c0); m->ggrid_abs[e + 1] += fabsf(c1);
                if (grid_half) { m->ggrid_h[e] = f2h(h2f(m->ggrid_h[e]) + c0); m->ggrid_h[e + 1] = f2h(h2f(m->ggrid_h[e + 1]) + c1); }
            }
        }
    }
    if (!grid_half) for (uint32_t k = 0; k < m->n_grid; ++k) m->ggrid_h[k] = f2h(m->ggrid[k]);
}

/* ------------------------------------------------------------------ Trainer::optimizer_step (tcnn), nerf_model.cu:1644
 * Ema(0.95){ ExponentialDecay{ Adam } }  base.json:5-22.  TCNN-A6/A7/A8. */
static void adam_one(orc_model* m, uint32_t i, float gradient, int is_matrix) {
    const orc_config* c = &m->cfg;
    if (!is_matrix && gradient == 0.0f) return;                         /* grid entries with zero gradient are skipped entirely */
    float w = m->master[i];
    if (is_matrix) gradient += c->l2_reg * w;                           /* L2 only on matrix weights */
    float gsq = gradient * gradient;
    float fm = m->m1[i] = c->beta1 * m->m1[i] + (1.0f - c->beta1) * gradient;
    float sm = m->m2[i] = c->beta2 * m->m2[i] + (1.0f - c->beta2) * gsq;
    uint32_t cs = ++m->steps[i];                                        /* per-parameter step counter */
    float lr = m->lr * sqrtf(1.0f - powf(c->beta2, (float)cs)) / (1.0f - powf(c->beta1, (float)cs));
    float eff = lr / (sqrtf(sm) + c->epsilon);
    float nw = w - eff * fm;
    m->master[i] = nw; m->half[i] = f2h(nw);
}
/* grid_f32: the grid gradient as fp32 values (a hook for gradients that are fp32 sums of several fp16 tables); NULL = the fp16 table, as tcnn */
static void optimizer_step_from(orc_model* m, const float* grid_f32) {
    const orc_config* c = &m->cfg; const float inv_ls = c->loss_scale;
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)m->n_mlp; ++i) adam_one(m, (uint32_t)i, m->gmlp[i] / inv_ls, 1);
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)m->n_grid; ++i) adam_one(m, m->n_mlp + (uint32_t)i, (grid_f32 ? grid_f32[i] : h2f(m->ggrid_h[i])) / inv_ls, 0);
    uint32_t cur = m->step + 1;                                          /* nested optimizer's step() after increment */
    if ((int32_t)cur >= c->decay_start && c->decay_interval > 0 && ((int32_t)cur - c->decay_start) % c->decay_interval == 0) m->lr *= c->decay_base;
    float d = c->ema_decay;
    float deb_old = 1.0f - (float)pow((double)d, (double)(cur - 1)), deb_new = 1.0f / (1.0f - (float)pow((double)d, (double)cur));
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)m->n_params; ++i)
        m->ema[i] = f2h(((h2f(m->ema[i]) * d) * deb_old + h2f(m->half[i]) * (1.0f - d)) * deb_new);
    m->has_ema = 1;
}
static void optimizer_step(orc_model* m) { optimizer_step_from(m, NULL); }
/* closed-form KAT hooks: run the optimizer on externally supplied gradients */
void orc_optimizer_step_with(orc_model* m, const float* gmlp, const uint16_t* ggrid_h) {
    memcpy(m->gmlp, gmlp, (size_t)m->n_mlp * 4); memcpy(m->ggrid_h, ggrid_h, (size_t)m->n_grid * 2);
    optimizer_step(m); m->step++;
}
/* the same with the grid gradient in fp32 (tests/test_gpu_parity.py: the optimizer of the product driven step by step with the product's own gradients,
 * which on the LDS-scattered levels are fp32 sums of fp16 partial tables) */
void orc_optimizer_step_with_f32(orc_model* m, const float* gmlp, const float* ggrid_f32) {
    memcpy(m->gmlp, gmlp, (size_t)m->n_mlp * 4);
    optimizer_step_from(m, ggrid_f32); m->step++;
}

/* NeRF_Model::Train_Step body, nerf_model.cu:1635-1648 (one iteration). Returns n_valid. */
uint32_t orc_train_step(orc_model* m) {
    generate_batch(m);
    m->iter++;
    if (m->n_valid == 0) return 0;
    forward_backward(m);
    optimizer_step(m);
    m->step++;
    return m->n_valid;
}
float orc_train(orc_model* m, int iters) { for (int i = 0; i < iters; ++i) orc_train_step(m); return m->loss; }
/* forward+backward only on the current batch (tests compare gradients without touching params) */
void orc_set_step_variant(orc_model* m, int on) { m->step_variant = on; }
uint32_t orc_n_compacted(const orc_model* m) { return m->n_compacted; }
void orc_generate_batch(orc_model* m) { generate_batch(m); }   /* does not advance iter */
void orc_advance_iter(orc_model* m) { m->iter++; }
void orc_forward_backward(orc_model* m) { forward_backward(m); }

/* ------------------------------------------------------------------ Render nerf_model.cu:1702-1830 / RenderVideo :1832-1991
 * GenerateRender(Video)Rays :448-534, GenerateRenderInputPoints :593-626 (2S samples, jittered; the
 * reference re-creates the generator per call => identical jitter every call), inference with the
 * EMA weights (TCNN-A8; fp16 compute, output widened to fp32), VolumeRender_Render :1134-1229.
 * pose_is_Toc: 0 -> Twc with the object's Tow; 1 -> pose given directly in the object frame. */
void orc_render(const orc_model* m, orc_bbox box, const float* pose16, int pose_is_Toc, int use_ema, float* rgb, float* depth, float* mask) {
    const int S = 2 * m->S; const long n = (long)box.w * box.h;
    const uint16_t* prm = (use_ema && m->has_ema) ? m->ema : m->half;
    /* XORWOW mode: the reference creates a NEW generator for every Render (default seed) and draws the whole crop's RandDt in one call (:1725-1728,1781) */
    float* xwr = NULL;
    if (ORC_RNG_STREAM(&m->cfg)) { orc_xwgen g; memset(&g, 0, sizeof g); xwgen_init(&g, 0ull, ORC_RNG_STREAM(&m->cfg) == 2u, ORC_RNG_LANES(&m->cfg));
        xwr = (float*)malloc(sizeof(float) * (size_t)n * S); xwgen_uniform(&g, xwr, (size_t)n * S); free(g.lane); }
    #pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < n; ++i) {
        int x = (int)box.x + (int)(i % box.w), y = (int)box.y + (int)(i / box.w);
        float o[3], d[3], dn, t0, t1;
        pixel_ray(m, (float)x, (float)y, pose16, pose_is_Toc ? NULL : m->Tow, o, d, &dn);
        if (!ray_intersect(m->amin, m->amax, o, d, &t0, &t1)) { rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 1.0f; depth[i] = 0.0f; mask[i] = 0.0f;
            continue; }
        t0 = fmaxf(t0, 0.0f);
        float dt = (t1 - t0) / (float)S, T = 1.0f, r[3] = { 0, 0, 0 }, dep = 0.0f, last = 0.0f;
        uint16_t E[2 * ORC_MAX_LEVELS + 16], hid[256], out[4];
        for (int k = 0; k < S; ++k) {
            if (T < 1e-4f) break;
            float t = fmaf(dt, (float)k + (xwr ? xwr[(size_t)i * S + k] : rand01(m->cfg.sample_seed, 3, 0, (uint32_t)(i * S + k))), t0), p[3];
            for (int a = 0; a < 3; ++a) { float q = fmaf(t, d[a], o[a]); p[a] = (q - m->amin[a]) / (m->amax[a] - m->amin[a]); }
            encode_one(m, prm + m->n_mlp, p, E); mlp_forward_one(m, prm, E, hid, out);
            float c0 = logistic(h2f(out[0])), c1 = logistic(h2f(out[1])), c2 = logistic(h2f(out[2]));
            float ddt = t - last, sigma = expf(h2f(out[3])), alpha = 1.0f - expf(-sigma * ddt), w = alpha * T;
            r[0] += w * c0; r[1] += w * c1; r[2] += w * c2; dep += w * t; T *= (1.0f - alpha); last = t;
        }
        if (1.0f - T > 0.5f) { rgb[3 * i] = r[0] + T; rgb[3 * i + 1] = r[1] + T; rgb[3 * i + 2] = r[2] + T; depth[i] = dep / dn; mask[i] = 1.0f; }
        else { rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 1.0f; depth[i] = 0.0f; mask[i] = 0.0f; }
    }
    free(xwr);
}

/* GetDensityOnGrid nerf_model.cu:2007-2048: raw (pre-activation) channel 3 on a res^3 lattice of the unit cube
 * (generate_grid_samples_nerf_uniform :296-309, x fastest), inference weights. */
void orc_density_grid(const orc_model* m, int rx, int ry, int rz, int use_ema, float* out) {
    const uint16_t* prm = (use_ema && m->has_ema) ? m->ema : m->half;
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)rx * ry * rz; ++i) {
        int x = (int)(i % rx), y = (int)((i / rx) % ry), z = (int)(i / ((long)rx * ry));
        float p[3] = { (float)x / (float)(rx - 1), (float)y / (float)(ry - 1), (float)z / (float)(rz - 1) };
        uint16_t E[2 * ORC_MAX_LEVELS + 16], hid[256], o4[4];
        encode_one(m, prm + m->n_mlp, p, E); mlp_forward_one(m, prm, E, hid, o4); out[i] = h2f(o4[3]);
    }
}

/* compute_mesh_vertex_colors nerf_model.cu:2050-2069: WarpPoint (:140-144) -> inference weights -> logistic rgb (:328-339) */
void orc_mesh_colors(const orc_model* m, const float* verts, uint32_t n, int use_ema, float* colors) {
    const uint16_t* prm = (use_ema && m->has_ema) ? m->ema : m->half;
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        float p[3]; for (int a = 0; a < 3; ++a) p[a] = (verts[3 * i + a] - m->amin[a]) / (m->amax[a] - m->amin[a]);
        uint16_t E[2 * ORC_MAX_LEVELS + 16], hid[256], o4[4];
        encode_one(m, prm + m->n_mlp, p, E); mlp_forward_one(m, prm, E, hid, o4);
        for (int c = 0; c < 3; ++c) colors[3 * i + c] = logistic(h2f(o4[c]));
    }
}

/* inference throughput probe for the CPU baseline: encode + MLP + composite over given points */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
