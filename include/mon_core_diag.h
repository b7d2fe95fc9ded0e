/* mon_core_diag.h -- C ABI of libmon_core_diag.so: diagnostics and test scaffolding of the MI355X Multi-Object-NeRF core.
 *
 * NOT part of the drop-in boundary (include/mon_core.h, libmon_core.so): nothing here is needed to run RO-MAP.  The library links against
 * libmon_core.so and looks into its objects (ro-map_amd/csrc/model.h) for the parity tests, the layout self-tests and the micro-benchmarks
 * that drove the design (profiles/r01_microbench.md). */
#ifndef MON_CORE_DIAG_H
#define MON_CORE_DIAG_H
#include "mon_core.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Copy an internal device buffer to the host; ids in ro-map_amd/csrc/model.h (MON_BUF_*). */
int mon_object_debug_read(mon_object* obj, int which, void* dst, size_t bytes);
/* Copy one uploaded frame of a dataset back to the host: rgba[H*W] packed r | g << 8 | b << 16 | instance << 24, depth[H*W] (NULL or a dataset without depth:
 * skipped), pose[16] (Twc, column-major).  The upload test compares what arrived with what was sent, frame by frame. */
int mon_dataset_debug_read(mon_dataset* ds, uint32_t frame, uint32_t* rgba, float* depth, float* pose16);
/* Diagnostic micro-benchmarks of scatter strategies (ro-map_amd/csrc/microbench.hip); *ms = best of 3 runs. */
int mon_microbench(int device, int mode, int pattern, uint32_t n_entries, uint32_t n_ops, float* ms);
/* Host-side check hook: corner index of the fused kernels' closed form (device_common.h:fast_grid_index) for level `level`
 * of configuration cfg; *size = entries of that level.  No device needed. */
int mon_debug_fast_index(const mon_config* cfg, int level, uint32_t x, uint32_t y, uint32_t z, uint32_t* index, uint32_t* size);
/* Layout of the MFMA A-fragment image of the fused kernels (ro-map_amd/csrc/frag_layout.h), both directions, for the layout test:
 * source[n_image] = MLP parameter index held by each image element (-1 = structural zero); slots[2 * n_mlp] = the (<= 2) image elements
 * each parameter feeds (-1 = none).  Either pointer may be NULL. */
int mon_debug_frag_layout(int encoded_width_padded, int n_neurons, int n_hidden_layers, int n_levels, int* source, int* slots, int* n_image, int* n_mlp);
/* Accumulator layout of k_fused_train's dW partial rows (frag_layout.h acc_param): param[n_cols] = MLP parameter index each column sums into
 * (-1 = pad column of a narrow encoding); the loss partial follows at column n_cols.  param may be NULL. */
int mon_debug_acc_layout(int encoded_width_padded, int n_neurons, int n_hidden_layers, int n_levels, int* param, int* n_cols);
/* MFMA fragment-layout self-test: D[32x32] = A[32x16] * B[16x32], fp16 in / fp32 out, through the lane mapping the fused kernels rely on. */
int mon_selftest_mfma(int device, const uint16_t* A, const uint16_t* B, float* D);
/* config.yaml key look-up of the sequence reader (cv::FileStorage semantics: exact key at line start); returns MON_ERR_IO when absent. */
int mon_debug_yaml_number(const char* text, const char* key, double* value);
/* Occupancy-grid bookkeeping of an object created with occupancy_skip (model.cpp maybe_refresh_occupancy): out[0] = iteration of the last refresh
 * (0 = none yet), out[1] = first iteration at or after which the next one is due. */
int mon_debug_occupancy_state(mon_object* obj, uint32_t out[2]);

/* Tile render bookkeeping (ro-map_amd/csrc/kernels_tilerender.hip): jobs (rays that hit the object's box, 2S samples each) of the LAST crop rendered on the
 * object's device through the per-device workspace of `side` (0: train-stream renders, 1: the inference stream); what bench.py's evaluated-sample count is. */
int mon_debug_render_jobs(mon_object* obj, int side, uint32_t* jobs);

#ifdef __cplusplus
}
#endif
#endif
