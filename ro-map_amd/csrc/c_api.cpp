// c_api.cpp -- extern "C" surface declared in include/mon_core.h.
#include <cstring>
#include "model.h"
#include "frag_layout.h"

namespace mon {
void set_error(const char* fmt, ...);
const char* last_error();
int device_count(int* n);
int physical_device(int logical, int* phys_out);
void config_default(mon_config& c);
int config_from_json(const char* path, mon_config& c);
int dataset_create(int device, int H, int W, float fx, float fy, float cx, float cy, uint32_t max_frames, int use_depth, Dataset** out);
int dataset_add_frame(Dataset* d, uint32_t id, const uint8_t* rgb, int ch, int is_bgr, const uint8_t* inst, const float* depth, const float* Twc);
int dataset_destroy(Dataset* d);
int model_create(Dataset* ds, const mon_config& cfg, int class_id, const float* Tow, const float* amin, const float* amax, Model** out);
int model_destroy(Model* m);
int model_add_boxes(Model& m, const mon_frame_bbox* boxes, size_t n);
int model_train(Model& m, int iters, float* loss, int stages);
int model_render(Model& m, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, int dst_on_device);
int model_density_grid(Model& m, int rx, int ry, int rz, float* out_host);
int model_get_params(Model& m, int which, void* dst, size_t bytes);
int model_set_params(Model& m, const float* master, size_t n);
int model_debug_read(Model& m, int which, void* dst, size_t bytes);
int model_generate_mesh(Model& m, int res, float thresh, uint32_t* n_verts, uint32_t* n_indices);
int model_mesh_counts(Model& m, uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices);
int model_get_mesh(Model& m, float* verts, float* normals, uint8_t* colors, uint32_t* indices, float* normals_raw, float* colors_f32, int try_only);
int model_save_mesh(Model& m, const char* path);
int model_mesh_generation(Model& m, uint64_t* gen);
int model_copy_mesh(Model& m, uint32_t cap_verts, uint32_t cap_indices, float* verts, float* normals, uint8_t* colors, uint32_t* indices,
                    uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices, int try_only);
int marching_cubes_host(int device, const float* density, int rx, int ry, int rz, float thresh, const float* amin, const float* amax,
                        float* verts, float* normals_raw, uint32_t* indices, uint32_t cap_verts, uint32_t cap_indices, uint32_t* n_verts,
                                uint32_t* n_verts_real, uint32_t* n_indices);
int microbench(int device, int mode, int pattern, uint32_t n_entries, uint32_t n_ops, float* ms_out);
}  // namespace mon

using namespace mon;


#define REQUIRE(p, what) do { if (!(p)) { set_error("%s: null %s", __func__, what); return MON_ERR_ARG; } } while (0)

extern "C" {

const char* mon_last_error(void) { return last_error(); }
int mon_version(void) { return 100; }
int mon_device_count(int* n) { REQUIRE(n, "n_devices"); return device_count(n); }
int mon_set_logical_devices(int n) { return set_logical_devices(n); }
int mon_offline_set_schedule(int outer, int inner) {
    if (outer < 1 || inner < 1) { set_error("offline_set_schedule: %d x %d", outer, inner); return MON_ERR_ARG; }
    options().offline_outer = outer; options().offline_inner = inner; return MON_OK;
}
int mon_set_option(const char* name, long value) { return option_set(name, value); }
int mon_get_option(const char* name, long* value) { return option_get(name, value); }
int mon_config_default(mon_config* cfg) { REQUIRE(cfg, "cfg"); config_default(*cfg); return MON_OK; }
int mon_config_from_json(const char* path, mon_config* cfg) { REQUIRE(path, "path"); REQUIRE(cfg, "cfg"); return config_from_json(path, *cfg); }

int mon_dataset_create(int device, int H, int W, float fx, float fy, float cx, float cy, uint32_t max_frames, int use_depth, mon_dataset** out) {
    REQUIRE(out, "out"); Dataset* d = nullptr;
    int rc = dataset_create(device, H, W, fx, fy, cx, cy, max_frames, use_depth, &d); if (rc) return rc;
    *out = new mon_dataset{ d }; return MON_OK;
}
int mon_dataset_add_frame(mon_dataset* ds, uint32_t frame_id, const uint8_t* rgb, int channels, int is_bgr, const uint8_t* instance, const float* depth,
        const float* Twc16) {
    REQUIRE(ds, "dataset"); return dataset_add_frame(ds->d, frame_id, rgb, channels, is_bgr, instance, depth, Twc16);
}
int mon_dataset_n_frames(const mon_dataset* ds, uint32_t* n) { REQUIRE(ds, "dataset"); REQUIRE(n, "n"); *n = ds->d->n_frames; return MON_OK; }
int mon_dataset_destroy(mon_dataset* ds) { if (!ds) return MON_OK; dataset_destroy(ds->d); delete ds; return MON_OK; }

int mon_object_create(mon_dataset* ds, const mon_config* cfg, int class_id, const float* Tow16, const float* aabb_min3, const float* aabb_max3,
        mon_object** out) {
    REQUIRE(ds, "dataset"); REQUIRE(cfg, "cfg"); REQUIRE(out, "out"); Model* m = nullptr;
    int rc = model_create(ds->d, *cfg, class_id, Tow16, aabb_min3, aabb_max3, &m); if (rc) return rc;
    *out = new mon_object{ m }; return MON_OK;
}
int mon_object_add_boxes(mon_object* o, const mon_frame_bbox* boxes, size_t n) { REQUIRE(o, "object"); return model_add_boxes(*o->m, boxes, n); }
int mon_object_train(mon_object* o, int iters, float* loss) { REQUIRE(o, "object"); return model_train(*o->m, iters, loss, 7); }
int mon_object_train_stages(mon_object* o, int stage_bits) { REQUIRE(o, "object"); return model_train(*o->m, 1, nullptr, stage_bits & 7); }
int mon_object_render(mon_object* o, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, int dst_on_device) {
    REQUIRE(o, "object"); return model_render(*o->m, box, pose16, pose_is_Toc, rgb, depth, mask, dst_on_device);
}
int mon_object_render_snapshot(mon_object* o, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask,
        uint32_t* snapshot_step) {
    REQUIRE(o, "object"); return model_render_snapshot(*o->m, box, pose16, pose_is_Toc, rgb, depth, mask, snapshot_step);
}
int mon_object_generate_mesh(mon_object* o, int res, float thresh, uint32_t* n_verts, uint32_t* n_indices) { REQUIRE(o, "object");
    return model_generate_mesh(*o->m, res, thresh, n_verts, n_indices); }
int mon_object_mesh_counts(mon_object* o, uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices) { REQUIRE(o, "object");
    return model_mesh_counts(*o->m, n_verts, n_verts_real, n_indices); }
int mon_object_get_mesh(mon_object* o, float* verts, float* normals, uint8_t* colors, uint32_t* indices, int try_lock_only) { REQUIRE(o, "object");
    return model_get_mesh(*o->m, verts, normals, colors, indices, nullptr, nullptr, try_lock_only); }
int mon_object_get_mesh_raw(mon_object* o, float* normals_raw, float* colors_f32) { REQUIRE(o, "object");
    return model_get_mesh(*o->m, nullptr, nullptr, nullptr, nullptr, normals_raw, colors_f32, 0); }
int mon_object_copy_mesh(mon_object* o, uint32_t cap_verts, uint32_t cap_indices, float* verts, float* normals, uint8_t* colors, uint32_t* indices,
                         uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices, int try_lock_only) {
    REQUIRE(o, "object");
    return model_copy_mesh(*o->m, cap_verts, cap_indices, verts, normals, colors, indices, n_verts, n_verts_real, n_indices, try_lock_only);
}
int mon_object_mesh_generation(mon_object* o, uint64_t* generation) { REQUIRE(o, "object"); REQUIRE(generation, "generation");
    return model_mesh_generation(*o->m, generation); }
int mon_object_save_mesh(mon_object* o, const char* path) { REQUIRE(o, "object"); REQUIRE(path, "path"); return model_save_mesh(*o->m, path); }
int mon_marching_cubes(int device, const float* density, int rx, int ry, int rz, float thresh, const float* aabb_min3, const float* aabb_max3,
                       float* verts, float* normals_raw, uint32_t* indices, uint32_t cap_verts, uint32_t cap_indices, uint32_t* n_verts,
                               uint32_t* n_verts_real, uint32_t* n_indices) {
    return marching_cubes_host(device, density, rx, ry, rz, thresh, aabb_min3, aabb_max3, verts, normals_raw, indices, cap_verts, cap_indices, n_verts,
            n_verts_real, n_indices);
}
int mon_object_density_grid(mon_object* o, int rx, int ry, int rz, float* out_host) { REQUIRE(o, "object");
    return model_density_grid(*o->m, rx, ry, rz, out_host); }
int mon_object_get_config(mon_object* o, mon_config* cfg) { REQUIRE(o, "object"); REQUIRE(cfg, "cfg"); *cfg = o->m->cfg; return MON_OK; }
int mon_object_info_get(mon_object* o, mon_object_info* info) {
    REQUIRE(o, "object"); REQUIRE(info, "info"); Model& m = *o->m;
    info->n_params = m.n_params; info->n_mlp_params = m.nd.n_mlp; info->n_grid_params = m.n_grid; info->encoded_width = (uint32_t)m.nd.Epad;
    info->train_step = m.h_state.step; info->n_boxes = m.n_boxes; info->last_n_valid = m.h_state.n_valid; info->device = m.device;
    info->last_loss = m.h_state.loss_sum / (float)m.oc.R; info->learning_rate = m.h_state.lr; info->backend = m.backend;
    info->skipped_batches = m.h_state.skipped; return MON_OK;
}
int mon_object_get_params(mon_object* o, int which, void* dst, size_t bytes) { REQUIRE(o, "object"); return model_get_params(*o->m, which, dst, bytes); }
int mon_object_set_params(mon_object* o, const float* master, size_t n) { REQUIRE(o, "object"); return model_set_params(*o->m, master, n); }
int mon_object_set_backend(mon_object* o, int backend) {
    REQUIRE(o, "object");
    if (backend == 1 && !fused_supported(o->m->nd, o->m->oc.S, o->m->oc.R)) { set_error("fused backend does not support this network shape");
        return MON_ERR_ARG; }
    if (backend != 0 && backend != 1) { set_error("backend must be 0 or 1"); return MON_ERR_ARG; }
    o->m->backend = backend; o->m->next_ready = false; o->m->b0_tiles_current = false; return MON_OK;
}
int mon_object_set_debug_dump(mon_object* o, int enable) { REQUIRE(o, "object"); o->m->fused_dump = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    o->m->graph_backend = -1; return MON_OK; }
int mon_object_set_profiling(mon_object* o, int enable) { REQUIRE(o, "object"); o->m->profiling = enable != 0; return MON_OK; }
int mon_object_get_profile(mon_object* o, mon_profile* out, int reset) {
    REQUIRE(o, "object"); REQUIRE(out, "out"); *out = o->m->prof; if (reset) std::memset(&o->m->prof, 0, sizeof(mon_profile)); return MON_OK;
}
int mon_object_destroy(mon_object* o) { if (!o) return MON_OK; model_destroy(o->m); delete o; return MON_OK; }

int mon_physical_device(int logical_device, int* physical_device) { return mon::physical_device(logical_device, physical_device); }
int mon_device_synchronize(int device) {
    if (use_device(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { set_error("device synchronize failed"); return MON_ERR_HIP; }
    return MON_OK;
}
int mon_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
    REQUIRE(free_bytes, "free_bytes"); REQUIRE(total_bytes, "total_bytes");
    if (use_device(device) != hipSuccess || hipMemGetInfo(free_bytes, total_bytes) != hipSuccess) { set_error("hipMemGetInfo failed on device %d", device);
        return MON_ERR_HIP; }
    return MON_OK;
}

}  // extern "C"
