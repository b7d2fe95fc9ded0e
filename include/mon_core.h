/*
 * mon_core.h -- C ABI of libmon_core.so, the MI355X-native Multi-Object-NeRF core.
 *
 * This is the drop-in boundary for the one hot path of XiaoHan-Git/RO-MAP: the per-object NeRF
 * inside dependencies/Multi-Object-NeRF/Core (libMON.so).  The reference exposes a C++ ABI
 * (namespace nerf, Eigen / cv::Mat types); every entry point below cites the reference member it
 * replaces.  CORE = /root/reference/dependencies/Multi-Object-NeRF/Core.  The C++ shim that
 * reproduces the reference's class names on top of this ABI is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns an int status
 * (MON_OK == 0); matrices are 4x4 float, column-major (Eigen::Matrix4f memory order); images are
 * row-major HxW.  A handle may be used from one thread at a time; different handles are
 * independent (the reference runs one std::thread per object, CORE/src/nerf_manager.cu:89,259).
 * There is NO CPU fallback: every compute entry point fails with MON_ERR_NO_DEVICE when no gfx950
 * device is visible.
 */
#ifndef MON_CORE_H
#define MON_CORE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MON_OK              0
#define MON_ERR_ARG         1   /* bad argument                                        */
#define MON_ERR_NO_DEVICE   2   /* no HIP device (reference: cerr + exit(0), nerf_manager.cu:21-25) */
#define MON_ERR_HIP         3   /* HIP runtime error (reference: CUDA_CHECK_THROW)     */
#define MON_ERR_IO          4   /* file / parse error                                  */
#define MON_ERR_STATE       5   /* call not valid in the current state                 */
#define MON_ERR_NO_RAYS     6   /* a training step found zero rays inside the 3-D box (reference: i % 0 UB, nerf_model.cu:287) */

/* Network / training configuration.  Fields follow CORE/configs/base.json (tcnn JSON schema) and
 * the compile-time constants of CORE/include/nerf_model.h:145,166,172-175 promoted to runtime. */
typedef struct mon_config {
    int32_t  n_levels;             /* encoding.n_levels (16)                                 */
    int32_t  n_features;           /* encoding.n_features_per_level; must be 2               */
    int32_t  log2_hashmap_size;    /* encoding.log2_hashmap_size (16)                        */
    int32_t  base_resolution;      /* encoding.base_resolution (16)                          */
    float    per_level_scale;      /* encoding.per_level_scale; tcnn default 2.0             */
    int32_t  n_neurons;            /* network.n_neurons: 16, 32, 64 or 128 (tcnn FullyFusedMLP) */
    int32_t  n_hidden_layers;      /* network.n_hidden_layers: 1 or 2; 3 or 4 up to 64 neurons */
    int32_t  rays_per_batch;       /* mnRaysPerBatch (4096); multiple of 64                  */
    int32_t  n_samples;            /* mnSampleNum (32); render uses 2x (mnRenderSampleNum)   */
    float    loss_scale;           /* mLoss_Scale (128)                                      */
    float    learning_rate;        /* optimizer...nested.nested.learning_rate (1e-2)         */
    float    beta1, beta2, epsilon, l2_reg;
    float    ema_decay;            /* optimizer.decay (0.95)                                 */
    int32_t  decay_start, decay_interval;
    float    decay_base;
    uint32_t param_seed;           /* m_seed (1337)                                          */
    /* rng_flags: "same inputs" mode for a comparison with the CUDA build; 0 (default) = this repo's own streams:
     *   bits 0-1   sample stream: 0 counter RNG keyed by sample_seed | 1 XORWOW in the reference's order of draws, cuRAND flavour
     *              (nerf_model.cu:1432,1434,1468 and :1781; default seed) | 2 the same with rocRAND's seeding / float map
     *   bit  4     parameter init in tcnn's generate_random_uniform element order (pcg32 draws interleaved per thread)
     *   bits 16-31 XORWOW lanes in units of 1024 (0 = 4: cuRAND's 4096 subsequences)   -- ro-map_amd/csrc/xorwow.h */
    uint32_t rng_flags;
    uint64_t sample_seed;          /* key of the counter RNG that replaces the cuRAND XORWOW stream by default */
    int32_t  use_depth;            /* NeRF_Model::mbUseDepth                                 */
    /* occupancy_skip: 0 (default, the reference's behaviour: every one of the 32 samples of a ray is evaluated) | 1: occupancy-grid skipping -- a 64^3 bit
     * grid over the object's box, refreshed from the training weights (every 32 iterations at first, every 512 later) after 256 warm-up iterations and
     * dilated by one cell; samples in empty cells are not evaluated (no table reads, no contribution, no gradient).  An approximation the reference does
     * not have: parity runs leave it off. */
    int32_t  occupancy_skip;
} mon_config;

/* CORE/include/common.h:18-23 (note: h before w). */
typedef struct mon_frame_bbox { uint32_t FrameId, x, y, h, w; } mon_frame_bbox;

typedef struct mon_dataset mon_dataset;   /* nerf::NeRF_Dataset, one per device (nerf_data.h:19-71)   */
typedef struct mon_object  mon_object;    /* nerf::NeRF + nerf::NeRF_Model (nerf.h:19-89, nerf_model.h:92-184) */

typedef struct mon_object_info {
    uint32_t n_params, n_mlp_params, n_grid_params, encoded_width;
    uint32_t train_step;           /* mnTrainingStep                                        */
    uint32_t n_boxes;              /* mnBbox                                                */
    uint32_t last_n_valid;         /* rays inside the 3-D box in the last batch             */
    int32_t  device;               /* mGPUid                                                */
    float    last_loss;            /* mfPerTrainLoss                                        */
    float    learning_rate;        /* after ExponentialDecay                                */
    int32_t  backend;              /* 0 unfused kernels, 1 fused MFMA kernel                */
    uint32_t skipped_batches;      /* iterations skipped because no ray hit the 3-D box     */
} mon_object_info;

/* Kernel classes timed with HIP events on the object's train stream when profiling is on. */
enum { MON_K_BATCH = 0, MON_K_FWDBWD = 1, MON_K_OPTIM = 2, MON_K_RENDER = 3, MON_K_SCATTER = 4, MON_K_REDUCE = 5, MON_K_ENCODE = 6 /* k_encode_tiles */,
       MON_K_POINTS = 7 /* k_sample_points */, MON_K_COUNT = 8 };
/* FWDBWD = k_fused_train alone (fused backend) or the unfused forward/backward kernel group; SCATTER = k_grid_scatter; REDUCE = k_reduce_partials. */
typedef struct mon_profile { double ms[MON_K_COUNT]; uint64_t launches[MON_K_COUNT]; } mon_profile;

const char* mon_last_error(void);                 /* thread-local message for the last non-OK status */
int mon_version(void);

/* NerfManager{Offline,Online}::Init -- device discovery (nerf_manager.cu:16-38, :136-158). */
int mon_device_count(int* n_devices);
/* Device numbering of everything below is LOGICAL.  Default: the physical HIP devices.  n > 0: n logical devices mapped round-robin onto the
 * physical ones (logical d -> physical d mod physical count), so the reference's object k -> device k mod nGPU placement with one dataset replica
 * per device (nerf.cu:27-33, nerf_manager.cu:44-55) can be driven -- oversubscribed -- on a box with fewer GPUs; 0 restores the default.  Call before
 * creating datasets / managers. */
int mon_set_logical_devices(int n);
/* the HIP device a logical device runs on (what a multi-device consumer groups objects by) */
int mon_physical_device(int logical_device, int* physical_device);

/* NeRF_Model::ReadNetworkConfig (nerf_model.cu:1272-1284): tcnn JSON with comments. */
int mon_config_default(mon_config* cfg);                       /* CORE/configs/base.json values */
int mon_config_from_json(const char* path, mon_config* cfg);

/* NerfManagerOnline::DatasetInit / NeRF_Dataset::InitDataToGPU (nerf_manager.cu:160-187, nerf_data.cu:237-271):
 * intrinsics + capacity for max_frames frames resident in HBM on `device`. */
int mon_dataset_create(int device, int H, int W, float fx, float fy, float cx, float cy,
                       uint32_t max_frames, int use_depth, mon_dataset** out);
/* NerfManagerOnline::NewFrameToDataset / NeRF_Dataset::FrameDataToGPU (nerf_manager.cu:189-218,
 * nerf_data.cu:273-339) and the per-image body of DataToGPU (nerf_data.cu:151-221).
 * rgb: HxWxchannels 8-bit (channels 3 or 4), is_bgr as delivered by cv::imread / the SLAM frontend;
 * instance: HxW 8-bit instance ids (0 = background); depth: HxW float metres (z-depth, 0 = none) or NULL;
 * Twc16: camera-to-world pose. */
int mon_dataset_add_frame(mon_dataset* ds, uint32_t frame_id, const uint8_t* rgb, int channels, int is_bgr,
                          const uint8_t* instance, const float* depth, const float* Twc16);
int mon_dataset_n_frames(const mon_dataset* ds, uint32_t* n);
int mon_dataset_destroy(mon_dataset* ds);

/* NerfManagerOnline::CreateNeRF / NeRF::SetAttributes + CreateModelOnline + ResetNetwork
 * (nerf_manager.cu:237-261, nerf.cu:155-185, nerf_model.cu:1259-1342) and NerfManagerOffline::CreateNeRF
 * (nerf_manager.cu:64-92).  class_id doubles as the instance id (nerf.cu:75,158).  The 1.1x/1.2x box
 * inflation of SetAttributes is the caller's business (the shim applies it); aabb is used as given. */
int mon_object_create(mon_dataset* ds, const mon_config* cfg, int class_id, const float* Tow16,
                      const float* aabb_min3, const float* aabb_max3, mon_object** out);
/* NeRF_Model::UpdateFrameIdAndBbox / UpdateFrameIdAndBboxOnline (nerf_model.cu:1609-1628): append. */
int mon_object_add_boxes(mon_object* obj, const mon_frame_bbox* boxes, size_t n);
/* NeRF_Model::Train_Step / Train_Step_Online (nerf_model.cu:1630-1699): `iters` iterations of
 * GenerateBatch -> Step_No_Compacted -> optimizer_step; returns after the train stream drained.
 * *loss (may be NULL) = mean per-ray loss of the last iteration (mfPerTrainLoss). */
int mon_object_train(mon_object* obj, int iters, float* loss);
/* NeRF_Model::Render (nerf_model.cu:1702-1830; pose_is_Toc=0, pose=Twc) and one pose of RenderVideo
 * (:1916-1976; pose_is_Toc=1).  Outputs box.h x box.w: rgb[3*h*w] float RGB, depth, mask; host
 * pointers unless dst_on_device != 0 (then device pointers on the object's device). */
int mon_object_render(mon_object* obj, mon_frame_bbox box, const float* pose16, int pose_is_Toc,
                      float* rgb, float* depth, float* mask, int dst_on_device);
/* The same render on the object's INFERENCE stream (mpInferenceStream, nerf_model.cu:1269) from the inference weights the training side
 * published last (at the end of a mon_object_train call / online training slice -- of every call of 64 or more iterations, otherwise when a
 * viewer has asked since the last publication or 10 ms have passed): callable from any thread WHILE another thread trains
 * the object -- no model lock, nothing queued behind training.  *snapshot_step (may be NULL) = optimizer steps the weights had.  Host
 * outputs.  MON_ERR_STATE before the first publication and for objects without an inference side (tables above 8 M parameters, unfused
 * backend): use mon_object_render under the caller's own serialisation there. */
int mon_object_render_snapshot(mon_object* obj, mon_frame_bbox box, const float* pose16, int pose_is_Toc,
                               float* rgb, float* depth, float* mask, uint32_t* snapshot_step);
/* NeRF_Model::GetDensityOnGrid (nerf_model.cu:2007-2048): raw density channel on an rx*ry*rz lattice. */
int mon_object_density_grid(mon_object* obj, int rx, int ry, int rz, float* out_host);

/* ---- mesh extraction: NeRF_Model::GenerateMesh + TransCPUMesh (CORE/src/nerf_model.cu:1993-2095), MarchingCubes and
 * compute_mesh_1ring (CORE/src/marching_cubes.cu:478-509, 655-665), SaveMesh (nerf_model.cu:2181-2184 -> save_mesh :511-653).
 * res <= 0 selects the reference's 64 (marching_cubes.h:30); the reference's threshold is 2.0 on the PRE-activation density
 * (marching_cubes.h:31).  n_verts is rounded up to a multiple of 128 with all-zero padding vertices (marching_cubes.cu:496);
 * vertex / face numbering is deterministic here (lattice order), the reference's is atomicAdd order.
 * The result is kept with the object as CPUMeshData (CORE/include/common.h:32-41): verts / normals float[3n], colors u8[3n],
 * indices u32; get_mesh with try_lock_only != 0 behaves like DrawCPUMesh's try_lock (nerf.cu:486-490). */
int mon_object_generate_mesh(mon_object* obj, int res, float thresh, uint32_t* n_verts, uint32_t* n_indices);
int mon_object_mesh_counts(mon_object* obj, uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices);
int mon_object_get_mesh(mon_object* obj, float* verts, float* normals, uint8_t* colors, uint32_t* indices, int try_lock_only);
/* Counts + data under one hold of the mesh mutex, bounded by the caller's capacities (in vertices / indices): the form a viewer thread uses
 * while the object's thread trains and republishes the mesh (counts-then-get is only safe once the threads have ended).  MON_ERR_ARG with
 * the needed counts in n_* when a buffer is too small; MON_ERR_STATE when try_lock_only and the trainer holds the mesh, or no mesh yet. */
int mon_object_copy_mesh(mon_object* obj, uint32_t cap_verts, uint32_t cap_indices, float* verts, float* normals, uint8_t* colors, uint32_t* indices,
                         uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices, int try_lock_only);
/* lock-free: number of meshes published so far (0 = none); a viewer copies again only when it changed */
int mon_object_mesh_generation(mon_object* obj, uint64_t* generation);
int mon_object_get_mesh_raw(mon_object* obj, float* normals_raw, float* colors_f32);      /* un-normalised normals, float colours (parity tests) */
int mon_object_save_mesh(mon_object* obj, const char* path);                              /* ".ply" -> ASCII ply, anything else -> obj */
/* Marching cubes + normals on a caller-supplied lattice (x fastest); buffers may be NULL to query the counts. */
int mon_marching_cubes(int device, const float* density, int rx, int ry, int rz, float thresh, const float* aabb_min3, const float* aabb_max3,
                       float* verts, float* normals_raw, uint32_t* indices, uint32_t cap_verts, uint32_t cap_indices,
                       uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices);

/* the configuration the object was created with (borrowed objects of the managers: base.json as read) */
int mon_object_get_config(mon_object* obj, mon_config* cfg);
int mon_object_info_get(mon_object* obj, mon_object_info* info);
/* Parameter I/O (the reference has none; needed for fixtures/checkpoints).
 * which: 0 fp32 master, 1 fp16 working copy, 2 fp16 EMA (inference) copy. */
int mon_object_get_params(mon_object* obj, int which, void* dst, size_t bytes);
int mon_object_set_params(mon_object* obj, const float* master, size_t n);
/* Test hooks: split one iteration so intermediate buffers can be compared with the oracle.
 * stage bits: 1 GenerateBatch, 2 forward+backward, 4 optimizer_step(+step counter). */
int mon_object_train_stages(mon_object* obj, int stage_bits);
/* Backend selector for forward/backward: 0 = unfused reference kernels, 1 = fused MFMA kernel. */
int mon_object_set_backend(mon_object* obj, int backend);
int mon_object_set_profiling(mon_object* obj, int enable);
int mon_object_get_profile(mon_object* obj, mon_profile* out, int reset);
int mon_object_destroy(mon_object* obj);

/* ---- nerf::NerfManagerOffline (CORE/include/nerf_manager.h:21-50, CORE/src/nerf_manager.cu:9-131) without OpenCV/Eigen:
 * reads the reference's on-disk sequence layout (nerf_data.cu:27-121: config.yaml, img.txt, groundtruth.txt, rgb|depth|instance
 * PNGs) and object files (nerf.cu:58-118), one dataset replica per device, one thread per object, object k on device k mod nGPU,
 * 10 x 500 training iterations per object (nerf_manager.cu:89, nerf_model.cu:1635; MON_OFFLINE_OUTER / MON_OFFLINE_INNER override). */
typedef struct mon_offline mon_offline;
int mon_offline_create(const char* dataset_path, const char* network_config_file, int use_dense_depth, mon_offline** out);
int mon_offline_init(mon_offline* mgr);                                   /* Init()            */
/* The offline training schedule, process-wide, read by mon_offline_init: `outer` Train_Step calls of `inner` iterations per object (the reference hard-codes
 * 10 x 500: nerf_manager.cu:89, nerf_model.cu:1635; a mesh every 2nd outer step).  Both >= 1. */
int mon_offline_set_schedule(int outer, int inner);
int mon_offline_read_dataset(mon_offline* mgr);                           /* ReadDataset()     */
int mon_offline_create_nerf(mon_offline* mgr, const char* object_file);   /* CreateNeRF(file): starts the object's training thread */
int mon_offline_wait_threads_end(mon_offline* mgr);                       /* WaitThreadsEnd()  */
int mon_offline_n_objects(mon_offline* mgr, int* n);
int mon_offline_object_loss(mon_offline* mgr, int idx, float* loss, int* device);
/* test images for the first max_views (0 = all) training boxes of object idx: <out_dir>/<id>/test_{img,depth,mask}/<stamp>.png (nerf.cu:335-349) */
int mon_offline_render_test(mon_offline* mgr, int idx, const char* out_dir, int max_views);
/* the "Save Object Mesh" step of RenderTestImg alone (nerf.cu:397-403): <out_dir>/<id>/obj.ply if the object has a mesh; mon_offline_render_test includes it */
int mon_offline_save_mesh(mon_offline* mgr, int idx, const char* out_dir);
/* GetIntrinsics(), GetAllTwc() and, per object, GetObjTow() / GetBoundingBox() / GetFrameIdAndBBox() (MON/main.cpp:55,149-151,334-336: the viewer's inputs).
 * Buffers may be NULL to query the counts. */
int mon_offline_get_intrinsics(mon_offline* mgr, float* fx, float* fy, float* cx, float* cy, int* H, int* W);
int mon_offline_get_poses(mon_offline* mgr, float* Twc16s, size_t capacity_frames, size_t* n_frames);
int mon_offline_object_meta(mon_offline* mgr, int idx, int* class_id, float* Tow16, float* aabb_min3, float* aabb_max3, mon_frame_bbox* boxes,
        size_t capacity_boxes, size_t* n_boxes);
/* the timestamp string of the object's box_index-th observation (the test images' file names) */
int mon_offline_object_stamp(mon_offline* mgr, int idx, size_t box_index, char* buf, size_t capacity);
/* where the training thread saves <id>.ply (default "./output", nerf.cu:148; "" = do not save) */
int mon_offline_set_output_dir(mon_offline* mgr, const char* dir);
/* GetAllNeRF()[idx]: owned by the manager, do not destroy; only mon_object_copy_mesh(try_lock) is safe while its thread trains */
int mon_offline_object(mon_offline* mgr, int idx, mon_object** borrowed);
int mon_offline_destroy(mon_offline* mgr);
/* ---- nerf::NerfManagerOnline (CORE/include/nerf_manager.h:54-90, CORE/src/nerf_manager.cu:133-312) + the online half of nerf::NeRF
 * (nerf.cu:155-253, 406-448): per-object training thread sleeping on a condition variable, training gated on > 10 boxes,
 * per-object dataset mutex, finish protocol.  cv::Mat arguments become raw pointers (8-bit BGR(A), 8-bit instance, float depth). */
typedef struct mon_online mon_online;
int mon_online_create(const char* network_config_file, int use_sparse_depth, int train_step_iterations, mon_online** out);
int mon_online_init(mon_online* mgr);
int mon_online_dataset_init(mon_online* mgr, float fx, float fy, float cx, float cy, int H, int W, size_t imgs);
int mon_online_new_frame(mon_online* mgr, uint32_t img_id, const char* timestamp, const uint8_t* bgr, int channels, const uint8_t* instance,
                         const float* depth, const float* Twc16);                                   /* NewFrameToDataset */
/* CreateNeRF: 1.1x / 1.2x box inflation applied */
int mon_online_create_nerf(mon_online* mgr, int cls, const float* Tow16, const float* aabb_min3, const float* aabb_max3, size_t* idx_out);
int mon_online_update_nerf_bbox(mon_online* mgr, size_t idx, const mon_frame_bbox* boxes, size_t n, int train_step);                        /* UpdateNeRFBbox */
int mon_online_get_frame_idx(mon_online* mgr, const char* timestamp, int* idx);                     /* GetFrameIdx (-1 if unknown) */
/* NerfManagerOnline::UpdateDataset -> NeRF_Dataset::UpdateDataGPU (nerf_manager.cu:220-235, nerf_data.cu:341-353): the poses of frames
 * [cur_id - frame_num, cur_id) are replaced on every device (bundle adjustment moved them) while every object's training is excluded.
 * Twc16s: frame_num column-major 4x4.  The reference only calls it from commented-out code (LocalMapping.cc:1128-1150). */
int mon_online_update_dataset(mon_online* mgr, uint32_t cur_id, uint32_t frame_num, const float* Twc16s);
/* pose (Twc, column-major) the dataset holds for frame `frame_id` (NeRF::GetTwc reads these, nerf.cu:450-462) */
int mon_online_get_pose(mon_online* mgr, uint32_t frame_id, float* Twc16);
int mon_online_wait_threads_end(mon_online* mgr);                                                  /* WaitThreadsEnd: request finish + join */
int mon_online_object_info(mon_online* mgr, size_t idx, float* loss, int* train_calls, int* device, uint32_t* n_boxes);
/* one view of RenderNeRFsTest */
int mon_online_render(mon_online* mgr, size_t idx, mon_frame_bbox box, const float* Twc16, float* rgb, float* depth, float* mask);
/* RenderNeRFsTest(out_path, idx, stamps, boxes, Twcs, radius) -> NeRF::RenderTestImg (nerf.cu:255-404): test images + test.txt +
 * train.txt + the 60-view 360-degree video (RenderVideo, nerf_model.cu:1832-1990) + obj.ply under <out_path>/<id>/ */
int mon_online_render_nerfs_test(mon_online* mgr, const char* out_path, size_t idx, const char* const* timestamps, const mon_frame_bbox* boxes,
                                 const float* Twcs16, size_t n, float radius);
int mon_generate_toc(float theta_deg, float phi_deg, float radius, float* Toc16);   /* NeRF_Model::GenerateToc, nerf_model.cu:2186-2205 */
/* DrawMesh(idx) reads this object's CPUMeshData through mon_object_copy_mesh(try_lock) */
int mon_online_object(mon_online* mgr, size_t idx, mon_object** borrowed);
int mon_online_destroy(mon_online* mgr);

/* PNG codec used for the sequence layout (8/16-bit, gray/RGB/RGBA in; gray/RGB out; 16-bit samples big-endian as in the file).
 * pixels may be NULL to query the header only. */
int mon_png_read(const char* path, int* width, int* height, int* channels, int* bit_depth, uint8_t* pixels, size_t capacity);
int mon_png_write(const char* path, int width, int height, int channels, int bit_depth, const uint8_t* pixels_big_endian);
/* A rendered crop as the reference stores it (nerf.cu:335-349: img.convertTo(CV_8UC3, 255), depth.convertTo(CV_16UC1, 20000), mask.convertTo(CV_8UC1, 255);
 * saturating, round half to even); mask / mask_path may be NULL (RenderVideo writes none). */
int mon_write_render_pngs(const char* img_path, const char* depth_path, const char* mask_path, uint32_t w, uint32_t h, const float* rgb, const float* depth,
        const float* mask);

/* Process-wide test and tuning switches (none is needed for normal operation; defaults are the product behaviour).  Read when an object is
 * created or a training call is enqueued -- set them before.  Nine names (round 6; the A/B switches whose losing setting only a measurement wanted --
 * record / array optimizer state, 16- / 32-bit step counters, chunk flags, lane chunk, slice length -- are variant builds now, model.h):
 *   "backend"            -1 auto, 0 layer-at-a-time kernels, 1 fused
 *   "use_graph"          1: replay an iteration pair as a hipGraph
 *   "big_switch"         gradient-carrying samples below which the large-table levels scatter with global atomics (0 = always atomics, 1 = always binned)
 *   "lds_encode"         forward hash-grid encode from LDS-resident level tiles (kernels_encode.hip): 1 = from 3072 rays per batch (default), 2 = always,
 *                        0 = never (gathers inside k_fused_train) -- bit-identical parameters either way
 *   "tile_render"        inference on feature-planar level tiles: 0 never, 1 crops of 4096 rays and more + point queries, 2 always -- bit-identical images
 *   "step_variant"       1: NeRF_Model::Step's sample-compaction schedule (nerf_model.cu:1504-1550) on the layer-at-a-time kernels -- the reference's own
 *                        "unavailable, for reference only" path, kept checkable; 0 = Step_No_Compacted, what both drivers train with
 *   "keep_zero_samples"  1: zero-gradient samples are scattered too -- the exactness test's A/B
 *   "train_lanes"        per-device training lanes (0 = every object on its own stream)
 *   "roctx"              1: roctx ranges per phase
 * Unknown names return MON_ERR_ARG. */
int mon_set_option(const char* name, long value);
int mon_get_option(const char* name, long* value);

/* Whole-device helpers used by bench.py. */
int mon_device_synchronize(int device);
/* hipMemGetInfo: sizing how many object NeRFs a device takes (measured: base.json 198 MB each incl. workspaces, T = 2^22 2.5 GB; + ~190 MB once per device
 * for the render workspace) */
int mon_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* Staged-execution companion (tests): the fused backend also writes its intermediate activations into the debug buffers (slower);
 * they are read back through libmon_core_diag.so (include/mon_core_diag.h).  enable = 1: on the gather chain (every level's grid gradient through
 * global atomics, readable as one table); 2: on the chain the object would run anyway (level tiles -> k_fused_train<PRE> -> k_grid_scatter). */
int mon_object_set_debug_dump(mon_object* obj, int enable);

#ifdef __cplusplus
}
#endif
#endif /* MON_CORE_H */
