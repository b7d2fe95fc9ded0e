// diag.cpp -- libmon_core_diag.so: diagnostics and test scaffolding (include/mon_core_diag.h).  Links against libmon_core.so and reads its objects
// through the internal headers; nothing here is on the product path.
#include <cstring>
#include <string>
#include <vector>
#include "model.h"
#include "frag_layout.h"
#include "../../include/mon_core_diag.h"

namespace mon {
void set_error(const char* fmt, ...);
int ensure_ema_current(Model& m);
int microbench(int device, int mode, int pattern, uint32_t n_entries, uint32_t n_ops, float* ms);
int selftest_mfma(int device, const uint16_t* A, const uint16_t* B, float* D);
bool read_yaml_number(const std::string& text, const char* key, double& v);

#define HIPCHECK(expr)                                                                                         \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                                       \
        set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return MON_ERR_HIP; } } while (0)

int model_debug_read(Model& m, int which, void* dst, size_t bytes) {
    model_leave_lane(m);
    if (which == MON_BUF_EMA) { int rc = ensure_ema_current(m); if (rc) return rc; }
    const size_t R = m.oc.R, B = R * m.oc.S, n = m.n_params; const void* src = nullptr; size_t sz = 0;
    // optimizer state kept as chunk records
    if (m.P.rec && (which == MON_BUF_MASTER || which == MON_BUF_M1 || which == MON_BUF_M2 || which == MON_BUF_STEPS)) {
        if (!dst || bytes < n * 4) { set_error("debug_read: buffer too small"); return MON_ERR_ARG; }
        HIPCHECK(use_device(m.device)); void* tmp = nullptr; HIPCHECK(hipMalloc(&tmp, n * 4));
        launch_state_unpack(m.train_stream, m.P.rec, which == MON_BUF_MASTER ? 0 : which == MON_BUF_M1 ? 1 : which == MON_BUF_M2 ? 2 : 3, tmp, (uint32_t)n);
        hipError_t e = hipStreamSynchronize(m.train_stream); if (e == hipSuccess) e = hipMemcpy(dst, tmp, n * 4, hipMemcpyDeviceToHost);
        (void)hipFree(tmp); HIPCHECK(e); return MON_OK;
    }
    switch (which) {
        case MON_BUF_MASTER: src = m.P.master; sz = n * 4; break;       case MON_BUF_HALF: src = m.P.half; sz = n * 2; break;
        case MON_BUF_EMA: src = m.P.ema; sz = n * 2; break;             case MON_BUF_M1: src = m.P.m1; sz = n * 4; break;
        // (16-bit counters are widened below)
        case MON_BUF_M2: src = m.P.m2; sz = n * 4; break;
        case MON_BUF_STEPS: src = m.P.steps ? (const void*)m.P.steps : (const void*)m.P.steps16; sz = n * 4; break;
        case MON_BUF_GMLP: src = m.P.gmlp; sz = (size_t)m.nd.n_mlp * 4; break;
        case MON_BUF_GGRID_H: src = m.P.ggrid; sz = (size_t)m.n_grid * 2; break;
        case MON_BUF_GGRID_F32: src = m.P.ggrid; sz = (size_t)m.n_grid * 4; break;
        case MON_BUF_PTS: src = m.B.pts; sz = B * 12; break;            case MON_BUF_TDIST: src = m.B.tdist; sz = B * 4; break;
        case MON_BUF_E: src = m.B.E; sz = B * m.nd.Epad * 2; break;     case MON_BUF_HID: src = m.B.Hid; sz = B * m.nd.W * m.nd.NH * 2; break;
        case MON_BUF_O: src = m.B.O; sz = B * 8; break;                 case MON_BUF_DO: src = m.B.dO; sz = B * 8; break;
        case MON_BUF_DHID: src = m.B.dHid; sz = B * m.nd.W * m.nd.NH * 2; break;
        case MON_BUF_DE: src = m.B.dE; sz = B * m.nd.Epad * 2; break;
        case MON_BUF_RGB_RAY: src = m.B.rgb_ray; sz = R * 12; break;    case MON_BUF_DEPTH_RAY: src = m.B.depth_ray; sz = R * 4; break;
        case MON_BUF_MASK_RAY: src = m.B.mask_ray; sz = R * 4; break;   case MON_BUF_LOSS_RAY: src = m.B.loss_ray; sz = R * 4; break;
        case MON_BUF_RAY_O: src = m.B.ray_o; sz = R * 12; break;        case MON_BUF_RAY_D: src = m.B.ray_d; sz = R * 12; break;
        case MON_BUF_RAY_T0: src = m.B.ray_t0; sz = R * 4; break;       case MON_BUF_RAY_T1: src = m.B.ray_t1; sz = R * 4; break;
        case MON_BUF_TARGET: src = m.B.target; sz = R * 12; break;      case MON_BUF_TARGET_DEPTH: src = m.B.target_depth; sz = R * 4; break;
        case MON_BUF_BGCOL: src = m.B.bgcol; sz = R * 12; break;        case MON_BUF_RAY_FLAG: src = m.B.ray_flag; sz = R; break;
        case MON_BUF_RAY_DN: src = m.B.ray_dn; sz = R * 4; break;       case MON_BUF_MASK: src = m.B.mask; sz = (R / 64) * 8; break;
        case MON_BUF_STATE: src = m.d_state; sz = sizeof(DevState); break;
        case MON_BUF_FRAG_TRAIN: src = m.d_frag_train; sz = 64 * 512 * 2; break;
        case MON_BUF_X_ALL: if (!m.d_x_all) { set_error("debug_read: level-tile encode not in use"); return MON_ERR_STATE; } src = m.d_x_all; sz = B * 16;
        break;
        case MON_BUF_E_SOA: if (!m.d_e_soa) { set_error("debug_read: level-tile encode not in use"); return MON_ERR_STATE; } src = m.d_e_soa;
        sz = B * (size_t)m.nd.L * 4; break;
        case MON_BUF_HALF_TILES: if (!m.d_half_tiles) { set_error("debug_read: level-tile encode not in use"); return MON_ERR_STATE; } src = m.d_half_tiles;
        sz = (size_t)m.n_grid * 2; break;
        case MON_BUF_LIVE_CNT: if (!m.d_live_cnt) { set_error("debug_read: no live-sample lists (occupancy_skip off, or not on the level tiles)"); return MON_ERR_STATE; }
            src = m.d_live_cnt; sz = 2u * kLiveMaxParts * kLiveCntStride * 4u; break;
        case MON_BUF_FRAG_REF:
            if (!m.d_frag_render) { set_error("debug_read: fused backend not available"); return MON_ERR_STATE; }
            HIPCHECK(use_device(m.device)); HIPCHECK(hipMemsetAsync(m.d_frag_render, 0, 64 * 512 * 2, m.train_stream));
            launch_build_frag_image(m.train_stream, m.P.half, m.nd, m.d_frag_render); src = m.d_frag_render; sz = 64 * 512 * 2; break;
        default: set_error("debug_read: unknown buffer id %d", which); return MON_ERR_ARG;
    }
    if (!dst || bytes < sz) { set_error("debug_read: buffer too small (%zu < %zu)", bytes, sz); return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device)); HIPCHECK(hipStreamSynchronize(m.train_stream));
    // saturating 16-bit counters on the device (ParamPtrs::steps16): hand out uint32 like before
    if (which == MON_BUF_STEPS && !m.P.steps) {
        std::vector<uint16_t> h16(n); HIPCHECK(hipMemcpy(h16.data(), src, n * 2, hipMemcpyDeviceToHost));
        uint32_t* out = reinterpret_cast<uint32_t*>(dst); for (size_t i = 0; i < n; ++i) out[i] = h16[i];
        return MON_OK;
    }
    // the grid gradient as k_optimizer forms it: the fp16 gradient table (large levels: binned sums or tcnn's atomics) plus the fp16 partial tables of the
    // LDS-scattered levels, summed in fp32 in the kernel's order -- pairs of partitions, (a + b) added to the running sum (update_chunk, kernels_optim.hip)
    if (which == MON_BUF_GGRID_H || which == MON_BUF_GGRID_F32) {
        std::vector<uint16_t> tab(m.n_grid); HIPCHECK(hipMemcpy(tab.data(), m.P.ggrid, (size_t)m.n_grid * 2, hipMemcpyDeviceToHost));
        std::vector<float> acc(m.n_grid);
        for (uint32_t i = 0; i < m.n_grid; ++i) { _Float16 h; std::memcpy(&h, &tab[i], 2); acc[i] = (float)h; }
        // whole steps of the shapes outside the fused kernels keep their partial tables too (hybrid_scatter); the dense optimizer then never reads the table
        const bool parts = (m.backend == 1 && m.lds_mask) || (m.backend == 0 && m.hybrid_scatter && which == MON_BUF_GGRID_F32);
        if (parts) {
            // partial tables are planar: [partition][feature][parity][entry / 2], over the LDS-scattered levels' entries
            const uint32_t n_ent = m.part_halves / 2, n_half = n_ent / 2;
            std::vector<uint16_t> pa(m.part_halves), pb(m.part_halves);
            auto val = [&](const std::vector<uint16_t>& part, uint32_t e, uint32_t f) { _Float16 h;
                std::memcpy(&h, &part[((size_t)f * 2 + (e & 1u)) * n_half + (e >> 1)], 2); return (float)h; };
            for (uint32_t q = 0; q < m.scatter.max_P; q += 2) {
                HIPCHECK(hipMemcpy(pa.data(), m.d_gpart + (size_t)q * m.part_halves, (size_t)m.part_halves * 2, hipMemcpyDeviceToHost));
                if (q + 1 < m.scatter.max_P) HIPCHECK(hipMemcpy(pb.data(), m.d_gpart + (size_t)(q + 1) * m.part_halves, (size_t)m.part_halves * 2,
                        hipMemcpyDeviceToHost));
                for (int l = 0; l < m.nd.L; ++l) {
                    // (a level with fewer partial tables: the rest of the buffer is not its data)
                    const bool lds_level = m.backend == 1 ? ((m.lds_mask >> l) & 1u) != 0u : true;
                    if (q >= m.scatter.P[l] || !lds_level) continue;
                    const bool two = q + 1 < m.scatter.P[l];
                    for (uint32_t e = m.lt.offset[l]; e < m.lt.offset[l + 1]; ++e) for (uint32_t f = 0; f < 2; ++f)
                        acc[2 * e + f] += two ? val(pa, e, f) + val(pb, e, f) : val(pa, e, f);
                }
            }
        }
        if (which == MON_BUF_GGRID_F32) {
            if (!dst || bytes < (size_t)m.n_grid * 4) { set_error("debug_read: buffer too small"); return MON_ERR_ARG; }
            std::memcpy(dst, acc.data(), (size_t)m.n_grid * 4); return MON_OK;
        }
        uint16_t* out = reinterpret_cast<uint16_t*>(dst);
        for (uint32_t i = 0; i < m.n_grid; ++i) { const _Float16 h = (_Float16)acc[i]; std::memcpy(&out[i], &h, 2); }
        return MON_OK;
    }
    HIPCHECK(hipMemcpy(dst, src, sz, hipMemcpyDeviceToHost));
    return MON_OK;
}


}  // namespace mon

using namespace mon;
#define REQUIRE(p, what) do { if (!(p)) { set_error("%s: null %s", __func__, what); return MON_ERR_ARG; } } while (0)

extern "C" {
int mon_object_debug_read(mon_object* o, int which, void* dst, size_t bytes) { REQUIRE(o, "object"); return model_debug_read(*o->m, which, dst, bytes); }
int mon_dataset_debug_read(mon_dataset* ds, uint32_t frame, uint32_t* rgba, float* depth, float* pose16) {
    REQUIRE(ds, "dataset"); Dataset& d = *ds->d;
    if (frame >= d.max_frames) { set_error("dataset_debug_read: frame %u >= capacity %u", frame, d.max_frames); return MON_ERR_ARG; }
    const size_t px = (size_t)d.K.H * d.K.W;
    HIPCHECK(use_device(d.device)); HIPCHECK(hipDeviceSynchronize());
    if (rgba) HIPCHECK(hipMemcpy(rgba, d.d_rgba + px * frame, px * 4, hipMemcpyDeviceToHost));
    if (depth && d.d_depth) HIPCHECK(hipMemcpy(depth, d.d_depth + px * frame, px * 4, hipMemcpyDeviceToHost));
    if (pose16) HIPCHECK(hipMemcpy(pose16, d.d_poses + 16 * (size_t)frame, 64, hipMemcpyDeviceToHost));
    return MON_OK;
}
int mon_microbench(int device, int mode, int pattern, uint32_t n_entries, uint32_t n_ops, float* ms) { REQUIRE(ms, "ms");
    return microbench(device, mode, pattern, n_entries, n_ops, ms); }
int mon_debug_fast_index(const mon_config* cfg, int level, uint32_t x, uint32_t y, uint32_t z, uint32_t* index, uint32_t* size) {
    REQUIRE(cfg, "cfg"); REQUIRE(index, "index"); REQUIRE(size, "size");
    LevelTable lt{}; NetDims nd{}; uint32_t n_grid = 0; int rc = level_table_build(*cfg, lt, nd, n_grid); if (rc) return rc;
    if (level < 0 || level >= nd.L) { set_error("level out of range"); return MON_ERR_ARG; }
    LevelFast lf{}; level_fast_build(lt, nd, lf);
    *index = fast_grid_index(lf, level, x, y, z); *size = lf.size[level]; return MON_OK;
}
int mon_debug_frag_layout(int epad, int W, int NH, int L, int* source, int* slots, int* n_image, int* n_mlp) {
    if (!(epad == 16 || epad == 32) || !((((W == 32 || W == 64) && (NH == 1 || NH == 2)) || (W == 128 && NH == 1))) || L < 1 || 2 * L > epad) { set_error("frag_layout: unsupported shape");
        return MON_ERR_ARG; }
    const FragDims d{ epad, W, NH, L };
    if (n_image) *n_image = d.N_FRAGS() * 512;
    if (n_mlp) *n_mlp = d.N_MLP();
    if (source) for (int i = 0; i < d.N_FRAGS() * 512; ++i) source[i] = frag_source(d, i);
    if (slots) for (int p = 0; p < d.N_MLP(); ++p) { int o[2] = { -1, -1 }; const int n = frag_slots(d, p, o); slots[2 * p] = n > 0 ? o[0] : -1;
        slots[2 * p + 1] = n > 1 ? o[1] : -1; }
    return MON_OK;
}
int mon_debug_acc_layout(int epad, int W, int NH, int L, int* param, int* n_cols) {
    if (!(epad == 16 || epad == 32) || !((((W == 32 || W == 64) && (NH == 1 || NH == 2)) || (W == 128 && NH == 1))) || L < 1 || 2 * L > epad) { set_error("acc_layout: unsupported shape");
        return MON_ERR_ARG; }
    const FragDims d{ epad, W, NH, L };
    if (n_cols) *n_cols = acc_cols(d);
    if (param) for (int i = 0; i < acc_cols(d); ++i) param[i] = acc_param(d, i);
    return MON_OK;
}
int mon_selftest_mfma(int device, const uint16_t* A, const uint16_t* B, float* D) { REQUIRE(A, "A"); REQUIRE(B, "B"); REQUIRE(D, "D");
    return selftest_mfma(device, A, B, D); }
int mon_debug_occupancy_state(mon_object* o, uint32_t out[2]) {
    if (!o || !o->m || !out) { mon::set_error("debug_occupancy_state: null argument"); return MON_ERR_ARG; }
    out[0] = o->m->occ_refreshed_iter; out[1] = o->m->occ_next_refresh; return MON_OK;
}
int mon_debug_render_jobs(mon_object* o, int side, uint32_t* jobs) {
    if (!o || !o->m || !jobs) { mon::set_error("debug_render_jobs: null argument"); return MON_ERR_ARG; }
    if (!o->m->tile_ok) { mon::set_error("debug_render_jobs: this object does not render on level tiles"); return MON_ERR_STATE; }
    mon::TileWs* ws = nullptr; const int rc = mon::tile_ws_get(*o->m, side, 0, &ws); if (rc) return rc;
    std::lock_guard<std::mutex> l(ws->mu);
    if (!ws->counters) { mon::set_error("debug_render_jobs: no tile workspace on the object's device"); return MON_ERR_STATE; }
    if (mon::use_device(o->m->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { mon::set_error("debug_render_jobs: device error");
        return MON_ERR_HIP; }
    if (hipMemcpy(jobs, ws->counters + 16u * ((ws->flip + 1u) & 1u), 4, hipMemcpyDeviceToHost) != hipSuccess) {
        mon::set_error("debug_render_jobs: copy failed"); return MON_ERR_HIP; }
    return MON_OK;
}
int mon_debug_yaml_number(const char* text, const char* key, double* value) {
    REQUIRE(text, "text"); REQUIRE(key, "key"); REQUIRE(value, "value");
    if (!read_yaml_number(text, key, *value)) { set_error("config.yaml: %s missing or not a number", key); return MON_ERR_IO; }
    return MON_OK;
}
}
