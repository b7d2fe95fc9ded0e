// model.h -- host-side mirror of nerf::NeRF_Model / nerf::NeRF_Dataset for gfx950.
// Reference: CORE/include/nerf_model.h:92-184, CORE/include/nerf_data.h:19-71.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/mon_core.h"
#include "device_common.h"
#include "frag_layout.h"

namespace mon {

// ---- device-resident per-object state updated by kernels (no host round trips inside an iteration)
struct DevState {
    uint32_t step;       // optimizer steps taken (mnTrainingStep)
    uint32_t iter;       // batches generated (RNG counter); advances even when a batch is skipped
    uint32_t n_valid;    // rays of the current batch inside the 3-D box
    uint32_t n_boxes;    // mnBbox
    float lr;            // Adam learning rate after ExponentialDecay
    float loss_sum;      // sum of per-ray losses of the current batch (SumLoss, nerf_model.cu:1231-1253)
    // level-tile encode: n_valid of THIS iteration as counted by the position pass that ran ahead of it (k_sample_points / k_optimizer's position blocks)
    uint32_t n_valid_pre;
    uint32_t skipped;    // batches skipped because n_valid == 0
    // gradient-carrying samples of the CURRENT iteration (written by k_grid_scatter; the optimizer's last block makes it n_scatter_last, so every kernel of an
    // iteration sees the same previous count)
    uint32_t n_scatter_now;
    float ema_deb_even_old, ema_deb_even_new;   // EMA debias factors of the next EVEN optimizer step (see ema_deb_old)
    uint32_t reserved_bins[13];
    uint32_t n_scatter_last;  // their sum in the last completed iteration (reporting)
    uint32_t n_scatter_total; // running sum over all iterations, modulo 2^32 (reporting: differences over a measurement window)
    // EMA debias factors (1 - d^(t-1), 1 / (1 - d^t)) of the next ODD optimizer step t; step t's kernel reads its pair and one of its threads writes the
    float ema_deb_old, ema_deb_new;
                                      // other pair for step t + 1 at kernel ENTRY (two double-precision pows: at the end of the last block they were ~1 us of
                                      // serial tail per step)
    // fused backend: slot counters of the gradient rows k_fused_train hands to k_grid_scatter -- samples with a non-zero dL/dO, per ray bin (ray & (bins - 1));
    // a wave reserves its slots with one returning atomic per ray.  Counter of bin b at [set * kMaxScatterBins * stride + b * stride]: a 64-byte line each
    // (returning atomics on one line serialise in its L2 channel: 4096 of them on 16 adjacent counters cost 7 us of k_fused_train).  TWO sets, by iteration
    // parity: k_fused_train(i) counts in set i & 1, k_grid_scatter(i) reads it and clears the other one for iteration i + 1.
    uint32_t n_scatter[2 * kMaxScatterBins * kScatterCounterStride];
};

// ---- dataset pointers (HBM layout: one slab per kind, frame-major)
struct DatasetPtrs {
    const uint32_t* rgba;    // [frames][H*W]  r | g<<8 | b<<16 | instance<<24  (4 B/pixel; the reference keeps 13 B/pixel)
    const float* depth;      // [frames][H*W]  metres, or nullptr
    const float* poses;      // [frames][16]   Twc column-major
    Intrinsics K;
};

struct ObjectConst {
    Mat4 Tow; Aabb aabb;
    uint32_t instance_id; uint32_t R; uint32_t S; int use_depth;
    uint64_t sample_seed;
    float loss_scale;
    // "same inputs" mode (mon_config::rng_flags, xorwow.h): per iteration parity the three arrays SampleXY[2R] | RandColors[3R] | RandDt[S R] that
    // k_xorwow_fill wrote for that iteration; nullptr = the counter RNG.  xw_render: RandDt of the crop being rendered (index = sample index within the crop).
    const float* xw[2]; const float* xw_render;
};
// one uniform of training iteration `step`, stream kStreamXY / kStreamColor / kStreamDt (index semantics of the reference's arrays: nerf_model.cu:395-396,
// :760, :553)
__device__ __forceinline__ float batch_rand(const ObjectConst& oc, uint32_t stream, uint32_t step, uint32_t idx) {
    if (oc.xw[0]) return oc.xw[step & 1u][(stream == kStreamXY ? 0u : stream == kStreamColor ? 2u * oc.R : 5u * oc.R) + idx];
    return rand01(oc.sample_seed, stream, step, idx);
}
__device__ __forceinline__ float render_rand(const ObjectConst& oc, uint32_t idx) {
    return oc.xw_render ? oc.xw_render[idx] : rand01(oc.sample_seed, kStreamRender, 0u, idx); }

struct BatchPtrs {
    const mon_frame_bbox* boxes;
    // candidates (un-compacted), R entries
    float *cand_o, *cand_d, *cand_dn, *cand_t0, *cand_t1, *cand_depth; uint32_t* cand_rgba; unsigned long long* mask;
    // training / render rays
    float *ray_o, *ray_d, *ray_dn, *ray_t0, *ray_t1, *target, *target_depth, *bgcol; uint8_t* ray_flag;
    // samples
    float *pts, *tdist;
    // network activations (unfused backend only) -- fp16
    uint16_t *E, *Hid, *O, *dO, *dHid, *dE;
    // per-ray results
    float *rgb_ray, *depth_ray, *mask_ray, *loss_ray;
    // level-tile encode: the compacted batch's ray records, 12 floats per training ray {rgba bits, t0, t1, d[3], o[3], target depth, candidate index bits, 0},
    // written by the position pass (k_sample_points / k_optimizer's position blocks) so that k_fused_train<PRE> needs neither the ballot scan nor the candidate
    // select
    float* ray_rec;
};

// LDS scatter plan: levels handled by k_grid_scatter and their sample-partition counts (partial tables per level)
struct ScatterLevels { uint32_t entry_offset[kMaxLevels + 1]; uint8_t P[kMaxLevels]; uint8_t level[kMaxLevels]; uint32_t n_levels; uint32_t max_P; };

struct ParamPtrs {
    float* master; uint16_t* half; uint16_t* ema; float* m1; float* m2; uint32_t* steps;
    // the per-parameter step counters as SATURATING 16-bit values (steps == nullptr then): exact whenever beta^65535 < 2^-25 for both betas -- the bias
    // correction 1 - beta^t is then exactly 1.0f for every t the counter can no longer tell apart (beta <= 0.99973; base.json: 0.9 / 0.99)
    uint16_t* steps16;
    // large tables (lazy EMA): the optimizer state as ONE 128-byte record per 8-parameter chunk -- master[8] | m1[8] | m2[8] | 8 x uint16 step counters | pad (word 28: the lazy EMA's step, round 5) --
    // instead of the four arrays above (which are null then): late in training a few per cent of the chunks are touched, and a touched chunk among untouched
    // ones is then one full line, not four half-used 64-byte sectors.  nullptr = the arrays (small tables: every chunk is streamed anyway)
    float* rec;
    float* gmlp;        // fp32 dW [n_mlp]
    uint16_t* ggrid;    // fp16 grid gradient [n_grid], accumulated with global_atomic_pk_add_f16
    // fused backend: dense fp16 partial gradient tables from k_grid_scatter (stride in halves); nullptr = unused
    const uint16_t* gpart; uint32_t part_stride;
    ScatterLevels sl;                                               // per-level partial-table counts
    // lazy EMA (large tables; `lazy` set by the launcher's caller): per 8-parameter chunk, the optimizer step its EMA is current for -- in word 28 (the pad) of the
    // chunk's record when `rec` is set (ema_step is nullptr then: nothing may index it), in this array otherwise.  set_params / upload_master leave the steps
    // alone: new weights enter the average from the next step on, chunks that sat steps out catch up with the weights they find (as the arrays always did)
    uint32_t* ema_step; int lazy;
    // lazy EMA + fused backend: one byte per 8-parameter chunk, set by whoever adds into ggrid (k_fused_train's atomics, k_big_accum),
    uint8_t* touched; uint32_t first_flag_chunk;
                                                                    // read and cleared by k_optimizer instead of scanning ggrid; chunks below first_flag_chunk
                                                                    // (MLP, LDS-scattered levels) are always visited
    int all_levels_dense;                                           // every level is LDS-scattered (no global-atomic table in use)
    // level-tile encode: the fp16 grid a second time, in LDS-tile order (tile_slot below), kept current by k_optimizer; nullptr = unused
    uint16_t* half_tiles;
};

// Tile image of the level-tile encode (kernels_encode.hip): a level that fits the CU's LDS whole keeps its entry order; a larger one (two parity tiles)
// stores its even entries first, then its odd ones, so that either tile is one contiguous copy.  Entry e of a level (offset off, size entries) sits at:
constexpr uint32_t kEncWholeMax = 163840u / 4u;                     // entries (half2) of a level that fits in one 160 KB tile
__host__ __device__ inline uint32_t tile_slot(uint32_t off, uint32_t size, uint32_t e_rel) {
    return size <= kEncWholeMax ? off + e_rel : off + (e_rel & 1u) * (size >> 1) + (e_rel >> 1); }

struct OptimConst {
    float beta1, beta2, epsilon, l2_reg, ema_decay, loss_scale, decay_base, log2_beta1, log2_beta2, log2_decay;
    int decay_start, decay_interval;
    uint32_t n_mlp, n_params;
};

// Occupancy-grid skipping on the level-tile chain (north_star N1; cfg.occupancy_skip): the position pass looks every sample's cell up in the bit grid, leaves
// the ray's 32 live bits in word 11 of its record (k_fused_train<PRE, OCC> takes them from there: both kernels see the grid as the position pass saw it) and
// compacts the LIVE samples of each of k_encode_tiles' sample partitions into an index list, so that the encode walks the live samples only.
//   idx [B]: partition w's list at [w * spw, w * spw + count_w), in arrival order (irrelevant: a sample's features go to its own slot).  WHICH list a sample
//     is on is free for the same reason: position block b (256 consecutive samples) appends to list b mod n_parts, so the lists come out equally long
//     whatever part of the batch is live (by sample range the longest list of the bench scene's late batches had 2355 entries against a mean of 2010: three
//     rounds of the encode's 1024 threads instead of two)
//   cnt [2][kLiveMaxParts][kLiveCntStride]: count_w of the iteration with parity p at [p][w][0] -- a 64-byte line each (one returning atomic per position
//     block lands there); the position pass of iteration j counts in set j & 1, k_encode_tiles(j) reads it and clears the other one for iteration j + 1
constexpr uint32_t kLiveMaxParts = 64, kLiveCntStride = 16;
struct LiveArgs { const uint32_t* occ_bits; uint32_t* idx; uint32_t* cnt; uint32_t spw, n_parts; };      // occ_bits == nullptr: every sample is evaluated, no lists

// What k_optimizer prepares for the next iteration of the fused backend (all zero = nothing): candidate rays and the A-fragment image.
struct OptimNext { uint32_t cand_blocks; uint16_t* frag_image; FragDims fd; BatchPtrs b; DatasetPtrs ds; ObjectConst oc;
                   // pos_blocks > 0 (level-tile encode): `b` holds the NEXT iteration's candidates already (k_encode_tiles generated them), these blocks sample
                   // its positions
                   uint32_t pos_blocks; float* x_all; LiveArgs live; };

// debug buffer ids for mon_object_debug_read (stable numbering, see binding.py BUF)
enum {
    MON_BUF_MASTER = 0, MON_BUF_HALF = 1, MON_BUF_EMA = 2, MON_BUF_M1 = 3, MON_BUF_M2 = 4, MON_BUF_STEPS = 5,
    MON_BUF_GMLP = 6, MON_BUF_GGRID_H = 9, MON_BUF_PTS = 10, MON_BUF_TDIST = 11, MON_BUF_E = 12, MON_BUF_HID = 13,
    MON_BUF_O = 14, MON_BUF_DO = 15, MON_BUF_DHID = 16, MON_BUF_DE = 17, MON_BUF_RGB_RAY = 18, MON_BUF_DEPTH_RAY = 19,
    MON_BUF_MASK_RAY = 20, MON_BUF_LOSS_RAY = 21, MON_BUF_RAY_O = 22, MON_BUF_RAY_D = 23, MON_BUF_RAY_T0 = 24,
    MON_BUF_RAY_T1 = 25, MON_BUF_TARGET = 26, MON_BUF_TARGET_DEPTH = 27, MON_BUF_BGCOL = 28, MON_BUF_RAY_FLAG = 29,
    MON_BUF_RAY_DN = 31, MON_BUF_MASK = 32, MON_BUF_STATE = 33,
    MON_BUF_FRAG_TRAIN = 34,    // the A-fragment image the next fused iteration will use (64 x 512 halves)
    MON_BUF_FRAG_REF = 35,      // the same image rebuilt from the current fp16 weights by k_build_frag_image (layout test)
    MON_BUF_X_ALL = 36,         // level-tile encode: positions float4 [B] of the batch the next / last iteration uses
    MON_BUF_E_SOA = 37,         // level-tile encode: encoded features half2 [L][B] of the last iteration
    MON_BUF_HALF_TILES = 38,    // level-tile encode: the fp16 grid in tile order (ParamPtrs::half_tiles)
    MON_BUF_LIVE_CNT = 40,      // occupancy-grid skipping on the level tiles: the live-sample counters [2][kLiveMaxParts][kLiveCntStride] (LiveArgs)
    MON_BUF_GGRID_F32 = 39      // the grid gradient the next k_optimizer will form, fp32: gradient table + the partial tables summed in the kernel's order
};

// Process-wide test and tuning switches (mon_set_option, include/mon_core.h); defaults are the product behaviour.
struct Options {      // (atomics: tests and tools flip options while object threads read them)
    std::atomic<long> backend{ -1 }, use_graph{ 0 }, big_switch{ 16384 }, lds_encode{ 1 }, train_lanes{ 2 }, roctx{ 0 }, step_variant{ 0 },
         keep_zero_samples{ 0 },   // 1: k_fused_train hands zero-gradient samples to the scatter too (the exactness test's A/B; same parameters, slower)
         tile_render{ 1 };      // inference on feature-planar level tiles: 0 never (gathers), 1 crops of 4096 rays and more + point queries, 2 always
    // NerfManagerOffline's 10 x 500 iterations (nerf_manager.cu:89): mon_offline_set_schedule, read by mon_offline_init
    std::atomic<long> offline_outer{ 10 }, offline_inner{ 500 };
#ifdef MON_OVERLAP_PROBE        // variant build only (tools/variant_build.sh ovl -DMON_OVERLAP_PROBE; HISTORY 7.9): k_optimizer(i) next to a throw-away k_encode_tiles
    std::atomic<long> overlap{ 0 }, enc_lds_kb{ 0 };
#endif
};
// Round 6: the A/B switches whose losing setting only a measurement ever wanted are VARIANT BUILDS now (tools/variant_build.sh <tag> -DMON_VARIANT_...; the
// oracle tests of the large-table optimizer run against each, tools/gpu_variants_large.sh), not runtime options of the shipping library:
//   MON_VARIANT_STEPS32   per-parameter step counters always 32 bits (shipping: saturating 16-bit ones where that is exact)
//   MON_VARIANT_ARRAYS    tables above 8 M parameters keep master / m1 / m2 / steps as four arrays (shipping: 128-byte chunk records)
//   MON_VARIANT_NO_FLAGS  the lazy optimizer finds touched chunks by scanning the gradient table (shipping: byte flags next to it)
#ifdef MON_VARIANT_STEPS32
constexpr bool kSteps16 = false;
#else
constexpr bool kSteps16 = true;
#endif
#ifdef MON_VARIANT_ARRAYS
constexpr bool kStateRecords = false;
#else
constexpr bool kStateRecords = true;
#endif
#ifdef MON_VARIANT_NO_FLAGS
constexpr bool kTouchedFlags = false;
#else
constexpr bool kTouchedFlags = true;
#endif
constexpr int kLaneChunk = 16;          // iterations an object enqueues per turn on a training lane (measured best, HISTORY 7.2)
constexpr long kOnlineSliceMin = 2;     // shortest training slice of the online manager (iterations)
Options& options();
int option_set(const char* name, long value);
int option_get(const char* name, long* value);

hipError_t use_device(int logical_device);      // hipSetDevice through the logical-device map (model.cpp)
int set_logical_devices(int n);

// ---- kernel launchers (kernels_*.hip)
void launch_gen_candidates(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st);
void launch_build_rays(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st);
void launch_gen_samples(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, const DevState* st, uint32_t S, uint32_t n_samples, uint32_t stream_id,
        uint32_t idx_base, int render, float* x4_or_null = nullptr);
void launch_render_rays(hipStream_t s, const BatchPtrs& b, const Intrinsics& K, const ObjectConst& oc, mon_frame_bbox box, const Mat4& pose, int pose_is_Toc,
        uint32_t pix0, uint32_t n);
void launch_grid_points(hipStream_t s, float* pts, int rx, int ry, int rz, uint32_t p0, uint32_t n);

// unfused network path (kernels_net.hip)
void launch_encode(hipStream_t s, const LevelTable& lt, const NetDims& nd, const uint16_t* params, const float* pts, uint16_t* E, uint32_t n,
        const DevState* st_or_null);
void launch_mlp_forward(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid_or_null, uint16_t* O, uint32_t n,
        const DevState* st_or_null);
void launch_mlp_backward(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st);
void launch_weight_grads(hipStream_t s, const NetDims& nd, const uint16_t* E, const uint16_t* Hid, const uint16_t* dHid, const uint16_t* dO, float* gmlp,
        uint32_t n, const DevState* st);
void launch_grid_backward(hipStream_t s, const LevelTable& lt, const NetDims& nd, const float* pts, const uint16_t* dE, uint16_t* ggrid, uint32_t n,
        const DevState* st);

// MFMA layer-at-a-time kernels (kernels_layers.hip): the shapes the fused kernels do not take.  ws_T: the T-layout copies the weight gradients read
// (layers_workspace_halves(nd, n) halves; nullptr = forward only, nothing is kept).  false: shape / batch not covered, nothing was launched
size_t layers_workspace_halves(const NetDims& nd, uint32_t n);
bool launch_mlp_forward_layers(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* E, uint16_t* Hid, uint16_t* O, uint32_t n,
        const DevState* st_or_null, uint16_t* ws_T_or_null,
        // the features as k_encode_tiles wrote them ([L][n] half2) instead of row-major E; E_out: where the row-major copy goes
        const uint16_t* e_soa_or_null = nullptr, uint16_t* E_out = nullptr,
        // false (whole steps): the activations go out in T layout + one ReLU mask bit each; row-major Hid is not written and launch_mlp_backward_layers must be
        // called with keep_rowmajor = false too
        bool keep_rowmajor = true);
bool launch_mlp_backward_layers(hipStream_t s, const NetDims& nd, const uint16_t* params, const uint16_t* Hid, const uint16_t* dO, uint16_t* dHid, uint16_t* dE,
        uint32_t n, const DevState* st, uint16_t* ws_T,
        bool keep_rowmajor = true /* false: dHid is written in T layout only -- the weight gradients' copy; the debug read-back then sees stale rows */,
        // != nullptr (whole steps of the hybrid scatter): dL/dE and the positions go straight into k_grid_scatter's hand-over layout (what k_rows_to_bins writes:
        // every sample at its natural slot of its ray's bin, every bin counter = its capacity) instead of row-major dE
        const struct BinsOut* bins = nullptr);
struct BinsOut { uint16_t* de_soa; float* x_soa; const float* pts; uint32_t n_bins; float clampv; DevState* st; };
void launch_weight_grads_layers(hipStream_t s, const NetDims& nd, float* gmlp, uint32_t n, const DevState* st, uint16_t* ws_T);
struct Model;
// inference forward of n samples E -> O with the layer kernels where the object has them (Hid of the training batch as scratch, piece by piece), else k_mlp_forward
void mlp_forward_inference(Model& m, hipStream_t s, const uint16_t* params, const uint16_t* E, uint16_t* O, uint32_t n);

// composite / loss gradient (kernels_composite.hip)
void launch_composite_grad(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st);
void launch_composite_render(hipStream_t s, const BatchPtrs& b, uint32_t S, uint32_t n_rays, float* rgb, float* depth, float* mask);
void launch_extract_density(hipStream_t s, const uint16_t* O, float* out, uint32_t n);
void launch_master_to_half(hipStream_t s, const float* master, uint16_t* half, uint32_t n);
// ParamPtrs::rec -> a flat array (0 master, 1 m1, 2 m2, 3 step counters as uint32)
void launch_state_unpack(hipStream_t s, const float* rec, int which, void* dst, uint32_t n);
void launch_state_pack_master(hipStream_t s, const float* master, float* rec, uint32_t n);
void launch_copy_params(hipStream_t s, const uint16_t* src, uint16_t* dst, uint32_t n);
void model_leave_lane(struct Model& m);      // non-training work goes to the object's own stream (model.cpp, training lanes)
// host (pinned) images -> packed RGBA8 | instance << 24
void launch_pack_frame(hipStream_t s, const uint8_t* rgb, int ch, int ri, int bi, const uint8_t* inst, uint32_t* dst, uint32_t px);
// source: pinned host memory the host rewrites (system-scope loads)
void launch_copy_from_host(hipStream_t s, const void* src, void* dst, uint32_t n_words);

// NeRF_Model::Step's schedule (kernels_step.hip; option step_variant): sample compaction + rollover of the batch between the two network passes
void launch_step_compaction(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st, uint32_t* steps, float* pts_compacted);

// optimizer (kernels_optim.hip)
// lazy_below: gradient-carrying samples at or below which the dense-table optimizer requests Adam state per touched chunk only
void launch_optimizer(hipStream_t s, const ParamPtrs& p, const OptimConst& oc, const DevState* st, DevState* st_next, const OptimNext& nx, uint32_t lazy_below);
void launch_ema_finalize(hipStream_t s, const ParamPtrs& p, const OptimConst& oc, const DevState* st);
void launch_reduce_partials(hipStream_t s, const float* partials, uint32_t n_partials, const NetDims& nd, float* gmlp, DevState* st);
uint32_t fused_partial_cols(const NetDims& nd);      // columns of a k_fused_train dW partial row (accumulator layout), the loss partial follows

// fused MFMA path (kernels_fused.hip)
constexpr uint32_t kMaxFusedGrid = 512;       // workgroups of k_fused_train (= dW partial rows per step): two per CU
bool fused_supported(const NetDims& nd, uint32_t S, uint32_t R);
uint32_t fused_train_grid(const NetDims& nd, uint32_t R);
void launch_fused_train(hipStream_t s, const LevelFast& lt, const NetDims& nd, const ParamPtrs& p, const BatchPtrs& b, const ObjectConst& oc, DevState* st,
        float* dw_partials, int debug_dump,
                        uint16_t* de_soa, float* x_soa, uint32_t lds_level_mask, uint16_t* frag_image, uint32_t big_switch, uint8_t* touched,
                                const uint32_t* occ_bits, uint32_t n_bins, const uint16_t* e_soa_or_null);
// level-tile encode (kernels_encode.hip): the forward gathers as LDS reads of a level tile, one workgroup per (level, sample partition)
bool encode_tiles_supported(const LevelTable& lt, const NetDims& nd);
void encode_tiles_setup_device();
void launch_sample_points(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st, float* x_all, const LiveArgs& live = LiveArgs{});
void launch_encode_tiles(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* half_tiles, const float* x_all, uint16_t* e_soa, uint32_t B,
        const DevState* st,
                         // b_next: the candidate set GenerateRays of the next iteration goes to
                         const BatchPtrs* b_next_or_null, const DatasetPtrs& ds, const ObjectConst& oc, uint32_t lds_bytes = 0, const LiveArgs& live = LiveArgs{});
uint32_t encode_tiles_spw(uint32_t B);          // samples per sample partition of k_encode_tiles (a partition's live list starts at w * spw)
// XORWOW sample stream (kernels_encode.hip k_xorwow_fill): one thread per lane, the generate calls of one iteration / one Render in the reference's order
void launch_xorwow_fill(hipStream_t s, void* lane_states, uint32_t lanes, int flavour, uint32_t start_lane, float* out0, uint32_t n0, float* out1, uint32_t n1,
        float* out2, uint32_t n2);
void launch_build_tiles_image(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* params, uint16_t* half_tiles);
void launch_occupancy_update(hipStream_t s, const LevelFast& lt, const NetDims& nd, const uint16_t* params, const ObjectConst& oc, uint16_t* frag_image,
        float raw_threshold, uint32_t* tmp, uint32_t* bits);
uint32_t scatter_plan(const LevelTable& lt, const NetDims& nd, ScatterLevels& sl);
uint32_t scatter_level_mask(const LevelTable& lt, const NetDims& nd);
bool grid_scatter_sums_partials(const LevelTable& lt, const NetDims& nd);
void launch_grid_scatter(hipStream_t s, const LevelTable& lt, const LevelFast& lf, const NetDims& nd, const uint16_t* de_soa, const float* x_soa, uint32_t B,
        uint32_t n_bins, uint16_t* gpart, uint32_t part_stride_entries, DevState* st,
                         // partials != null: also sums the dW partial rows (k_reduce_partials folded in)
                         const float* partials_or_null, uint32_t n_partials, float* gmlp, DevState* st_next);
// layer-at-a-time backend: dL/dE rows + positions of every sample into k_grid_scatter's hand-over layout (kernels_scatter.hip)
void launch_rows_to_bins(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* dE, const float* pts, uint32_t R, uint32_t S, uint32_t n_bins,
        uint16_t* de_soa, float* x_soa, DevState* st);
size_t big_scatter_workspace_bytes(const LevelTable& lt, const NetDims& nd, uint32_t lds_mask, uint32_t B);
void launch_big_scatter(hipStream_t s, const LevelTable& lt, const LevelFast& lf, const NetDims& nd, uint32_t lds_mask, const uint16_t* de_soa,
        const float* x_soa, uint32_t B,
                        uint32_t n_bins, const DevState* st, uint32_t big_switch, void* workspace, uint16_t* ggrid, uint8_t* touched_grid);
void launch_fused_render(hipStream_t s, const LevelFast& lt, const NetDims& nd, const uint16_t* params, const BatchPtrs& b, const ObjectConst& oc,
        uint32_t n_rays, uint32_t idx_base, float* rgb, float* depth, float* mask, uint16_t* frag_image, int build_image);
void launch_build_frag_image(hipStream_t s, const uint16_t* params, const NetDims& nd, uint16_t* image);
// inference on feature-planar level tiles (kernels_tilerender.hip): Render / RenderVideo, GetDensityOnGrid, mesh vertex colours
constexpr uint32_t kTileChunkJobs = 32768;          // rays (jobs of 2S = 64 samples) per chunk of the tile render
bool tile_render_supported(const LevelTable& lt, const NetDims& nd);
void launch_build_feat_image(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* params, uint16_t* image, uint32_t* zero_counter);
void launch_forward_frag_image(hipStream_t s, const NetDims& nd, const uint16_t* params, uint16_t* image);
void launch_render_rays_jobs(hipStream_t s, const Intrinsics& K, const ObjectConst& oc, mon_frame_bbox box, const Mat4& pose, int pose_is_Toc, uint32_t n_pix,
                             float* rec, uint32_t* count, uint32_t* next_count, float* rgb, float* depth, float* mask);
void launch_render_points(hipStream_t s, const ObjectConst& oc, const float* rec, const uint32_t* count, uint32_t job_base, uint32_t jobs_cap, float* x);
void launch_grid_points4(hipStream_t s, float* x, int rx, int ry, int rz, uint32_t p0, uint32_t n);
void launch_mesh_warp4(hipStream_t s, const float* verts, float* x, uint32_t v0, uint32_t n, const Aabb& box);
void launch_encode_feat(hipStream_t s, const LevelFast& lf, const NetDims& nd, const uint16_t* image, const float* x, uint16_t* e, uint32_t cap,
                        uint32_t n_host, const uint32_t* count, uint32_t job_base, uint32_t jobs_cap, uint32_t spj);
void launch_tile_render(hipStream_t s, const NetDims& nd, const ObjectConst& oc, const uint16_t* frag_image, const float* rec, const uint32_t* count,
                        uint32_t job_base, uint32_t jobs_cap, const float* x, const uint16_t* e, uint32_t cap, uint32_t n_pix, float* rgb, float* depth,
                                float* mask);
void launch_tile_points_mlp(hipStream_t s, const NetDims& nd, const uint16_t* frag_image, const uint16_t* e, uint32_t cap, uint32_t n_points, uint16_t* O);
void launch_candidates_and_frags(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st, const uint16_t* params,
        const NetDims& nd, uint16_t* frag_image);
int selftest_mfma(int device, const uint16_t* A, const uint16_t* B, float* D);

// ---- host classes
struct Dataset {
    int device = 0; Intrinsics K{}; uint32_t max_frames = 0, n_frames = 0; bool use_depth = false;
    uint32_t* d_rgba = nullptr; float* d_depth = nullptr; float* d_poses = nullptr;
    std::vector<uint8_t> present;      // present[id]: frame id has been uploaded (a new id lands in memory no kernel reads yet)
    // pinned staging of one incoming frame; the device's high-priority stream (model.cpp InferShared)
    uint8_t* h_stage = nullptr; size_t stage_bytes = 0; void* upload = nullptr;
    DatasetPtrs ptrs() const { return DatasetPtrs{ d_rgba, d_depth, d_poses, K }; }
};

struct MeshState;   // mesh.cpp
struct InferState;  // model.cpp: inference stream, published weight snapshots and the render workspace of their own

struct Model {
    MeshState* mesh = nullptr;
    InferState* infer = nullptr;                // mpInferenceStream (nerf_model.cu:1269): renders for viewers that never queue behind training
    Dataset* ds = nullptr; mon_config cfg{}; int device = 0;
    LevelTable lt{}; LevelFast lf{}; NetDims nd{}; ObjectConst oc{}; OptimConst opt{};
    uint32_t n_grid = 0, n_params = 0;
    hipStream_t train_stream = nullptr;      // mpTrainStream :1268; inference (render, mesh) runs on the same stream: the reference's mpInferenceStream is only
                                             // ever used from the object's own thread between training calls
    // level-tile encode: the second candidate set (cand_*, mask differ from B; everything else is shared); B and B_alt swap with the DevStates
    BatchPtrs B_alt{};
    // iteration i runs on one DevState and prepares the other for i + 1 (k_optimizer); swapped when an optimizer step is enqueued
    ParamPtrs P{}; BatchPtrs B{}; DevState* d_state = nullptr; DevState* d_state_next = nullptr;
    mon_frame_bbox* d_boxes = nullptr;
    uint32_t boxes_cap = 0, n_boxes = 0; uint32_t ws_samples = 0, ws_rays = 0;
    // fused backend
    // [512][fused_partial_cols + 64] fp32 dW partial rows of k_fused_train, accumulator layout (frag_layout.h acc_param)
    float* d_dw_partials = nullptr;
    uint16_t* d_de_soa = nullptr; float* d_x_soa = nullptr;       // compacted dL/dE rows [L][B] and positions [B] float4 for the scatter kernels
    // level-tile encode: positions [B] float4 of every sample, encoded features [L][B] half2 (nullptr: the fused kernel gathers)
    float* d_x_all = nullptr; uint16_t* d_e_soa = nullptr; uint16_t* d_half_tiles = nullptr;
    // network shapes outside the fused kernels (backend 0 by necessity) whose levels all fit the LDS scatter plan: whole training steps scatter through k_grid_scatter
    bool hybrid_scatter = false;
    bool b0_tiles_current = false;            // layer-kernel shapes on the level-tile encode: the tile image matches the fp16 weights (k_optimizer keeps it so in whole steps)
    uint16_t* d_layers_T = nullptr;          // T-layout workspace of the MFMA layer kernels (shapes outside the fused kernels)
    uint16_t* d_gpart = nullptr; ScatterLevels scatter{}; uint32_t lds_mask = 0;   // k_grid_scatter: partial tables, plan, levels it covers
    // halves of ONE partial table: the grid parameters of the LDS-scattered levels (a prefix of the levels), not of the whole table
    uint32_t part_halves = 0;
    uint16_t* d_frag_train = nullptr; uint16_t* d_frag_render = nullptr;           // MFMA A-fragment images (training weights / inference weights)
    uint32_t n_bins = 16;                                         // ray bins of the compacted gradient rows (scatter_bins(R) unless the option caps it)
    uint32_t* d_ema_step = nullptr; uint8_t* d_touched = nullptr;                  // lazy EMA bookkeeping, chunk flags (ParamPtrs)
    // kernels_bigscatter.hip: workspace, switch point, launched in this train call
    uint8_t* d_big_ws = nullptr; uint32_t big_switch = 0; bool big_active = false;
    // occupancy grid (cfg.occupancy_skip)
    uint32_t *d_occ = nullptr, *d_occ_tmp = nullptr; uint16_t* d_frag_occ = nullptr; float occ_raw_threshold = 0.f;
    uint32_t occ_refreshed_iter = 0, occ_next_refresh = 0;
    uint32_t *d_live_idx = nullptr, *d_live_cnt = nullptr;      // live-sample lists of the level-tile chain (LiveArgs)
    // whole-crop render outputs: ONE grow-only buffer, rgb | depth | mask of the current crop back to back
    float *d_out_all = nullptr, *d_out_rgb = nullptr, *d_out_depth = nullptr, *d_out_mask = nullptr; size_t out_cap = 0;
    // pinned staging of a crop on its way to the caller's (pageable) buffers
    float* h_out = nullptr; size_t h_out_cap = 0;
    std::vector<void*> allocs;
    DevState h_state{}; DevState* h_state_pinned = nullptr; int backend = 0; bool profiling = false; int fused_dump = 0;
    bool lazy_ema = false, ema_pending = false;   // large tables: EMA of untouched chunks is brought up to date on demand (k_ema_finalize)
    bool scatter_pending = false;   // a fused forward/backward was enqueued whose slot counter has not been reset by an optimizer step yet
    // XORWOW sample stream: lane states of the training generator (device), the two per-parity array sets, the iteration the fills have reached, the per-Render
    // generator xw_offset: values the training generator has produced
    void* d_xw_states = nullptr; float* d_xw = nullptr; uint32_t xw_filled = 0, xw_lanes = 0; int xw_flavour = 0; uint32_t enq_iter = 0; uint64_t xw_offset = 0;
    void* d_xw_render_states = nullptr; void* d_xw_render_init = nullptr; float* d_xw_render = nullptr; size_t xw_render_cap = 0;
    // NeRF_Model::Step schedule (option step_variant): per-ray sample counts / slots, the compacted positions
    uint32_t* d_step_counts = nullptr; float* d_step_pts = nullptr;
    // level-tile encode: used by the iteration being enqueued / the next batch's positions were written by the last k_optimizer / this train call runs the
    // gather chain (occupancy grid + few live samples)
    bool pre_active = false, points_ready = false, gathers_preferred = false;
    bool tile_counted = false;   // this object is counted in its device's tile workspace (freed with the device's last such object)
    bool tile_ok = false;        // the inference side may run on feature-planar level tiles (tile_render_supported)
    uint64_t weights_epoch = 0;  // process-wide unique stamp of the weights' current content (a new one after every train call / set_params / EMA catch-up):
                                 // the tile render's per-device workspace keeps its tile image while the stamp it was built for is current
    bool next_ready = false;     // fused backend: candidates + fragment image of the coming iteration were already produced by the last k_optimizer
    mon_profile prof{}; std::vector<hipEvent_t> ev_pool; std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_pending;
    // per-device training lanes (model.cpp): the lane and completion event of this object's last chunk
    struct TrainLanes* lanes = nullptr; int lane = -1; hipEvent_t lane_event = nullptr, switch_event = nullptr, sync_event = nullptr;
    // the object's private stream; train_stream is the one its work currently goes to (this one or a lane's)
    bool tail_marked = false; hipStream_t own_stream = nullptr;
#ifdef MON_OVERLAP_PROBE
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
#endif
    // (the state the captured pair of iterations starts on)
    hipGraphExec_t graph_exec = nullptr; int graph_backend = -1; const DevState* graph_state = nullptr; const void* graph_mask = nullptr;
};

int ensure_ema_current(Model& m);
uint64_t next_weights_epoch();
// Per-device workspace of the tile render, shared by the objects on the device; `side` 0: train-stream users (model_render, density lattice, mesh), 1: the
// inference stream.  A user holds `mu` from its first launch until its stream is synchronised.
struct TileWs {
    std::mutex mu;
    float* rec = nullptr; size_t rec_cap = 0;                   // job records of a crop: 12 floats per pixel
    uint32_t* counters = nullptr; uint32_t flip = 0;            // two job counters, used by alternate render calls (a call's ray kernel clears the next call's)
    float* x = nullptr; uint16_t* e = nullptr; uint16_t* O = nullptr; uint32_t cap = 0; int L_cap = 0;      // chunk buffers: positions, features, raw outputs
    uint16_t* image = nullptr; size_t image_cap = 0; uint16_t* frag = nullptr;      // feature-planar tile image + forward A fragments of the weights in use
    const void* key_params = nullptr; uint64_t key_epoch = ~0ull;
};
int tile_ws_get(Model& m, int side, size_t n_pix, TileWs** out);
// the weights `prm` (stamp `epoch`) as tile image + fragments in `ws` (rebuilt only when the stamp changed); caller holds ws.mu
void tile_ws_weights(Model& m, TileWs& ws, hipStream_t s, const uint16_t* prm, uint64_t epoch);
// ws.x holds n points -> ws.O (raw fp16 outputs [n][4]); caller holds ws.mu and has called tile_ws_weights
void tile_points_forward(Model& m, TileWs& ws, hipStream_t s, uint32_t n);
int model_publish_snapshot(Model& m);
int model_render_snapshot(Model& m, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, uint32_t* snapshot_step);
int level_table_build(const mon_config& c, LevelTable& lt, NetDims& nd, uint32_t& n_grid);
void level_fast_build(const LevelTable& lt, const NetDims& nd, LevelFast& lf);
void init_params_host(const mon_config& c, const NetDims& nd, uint32_t n_params, std::vector<float>& master);

}  // namespace mon

// opaque C handles of include/mon_core.h
struct mon_dataset { mon::Dataset* d; };
struct mon_object { mon::Model* m; };
