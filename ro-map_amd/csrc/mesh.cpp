// mesh.cpp -- host side of mesh extraction: NeRF_Model::GenerateMesh / TransCPUMesh / SaveMesh
// (CORE/src/nerf_model.cu:1993-2095, 2181-2184), MarchingCubes' count -> allocate -> emit flow (CORE/src/marching_cubes.cu:478-509)
// and the ASCII ply / obj writer (marching_cubes.cu:511-653, the non-unwrapped branch).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "model.h"

namespace mon {
void set_error(const char* fmt, ...);

#define HIPCHECK(expr)                                                                                         \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                                       \
        set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return MON_ERR_HIP; } } while (0)

void launch_mc_count(hipStream_t s, const float* density, int rx, int ry, int rz, float thresh, uint64_t* block_sums);
void launch_mc_emit(hipStream_t s, const float* density, int rx, int ry, int rz, float thresh, const float* amin, const float* amax, const uint64_t* block_offs,
                    int32_t* vertidx, float* verts, uint32_t* indices, float* normals_raw, float* normals, uint32_t n_verts_real, uint32_t n_indices);
void launch_mesh_warp(hipStream_t s, const float* verts, float* pts, uint32_t v0, uint32_t n, const Aabb& box);
void launch_mesh_colors(hipStream_t s, const uint16_t* O, float* colf, uint8_t* col8, uint32_t v0, uint32_t n);

// Device buffers (grow-only) + the CPU copy the viewers read (CPUMeshData, CORE/include/common.h:32-41).
struct MeshState {
    int device = 0;
    float* d_density = nullptr; size_t cap_lattice = 0; int32_t* d_vertidx = nullptr; uint64_t* d_blocks = nullptr;
    float *d_verts = nullptr, *d_nraw = nullptr, *d_normals = nullptr, *d_colf = nullptr; uint8_t* d_col8 = nullptr; size_t cap_verts = 0;
    uint32_t* d_indices = nullptr; size_t cap_indices = 0;
    uint32_t n_verts = 0, n_verts_real = 0, n_indices = 0;      // device-side result (n_verts is rounded up to 128, :496)
    std::mutex mu; bool have_result = false;                    // CPUMeshData::mesh_mutex / have_reslult
    std::atomic<uint64_t> generation{ 0 };                      // bumped whenever a new mesh is published to the host copy (viewers skip unchanged meshes)
    std::vector<float> verts, normals, normals_raw, colors_f32; std::vector<uint8_t> colors; std::vector<uint32_t> indices; uint32_t cpu_n_real = 0;
};

template <class T> static int grow(T*& p, size_t& cap, size_t need, size_t elems_per_unit = 1) {
    if (need <= cap && p) return MON_OK;
    if (p) hipFree(p);
    p = nullptr; const size_t n = need + need / 4 + 128;
    if (hipMalloc((void**)&p, n * elems_per_unit * sizeof(T)) != hipSuccess) { cap = 0;
        set_error("mesh: hipMalloc of %zu bytes failed", n * elems_per_unit * sizeof(T)); return MON_ERR_HIP; }
    cap = n; return MON_OK;
}

MeshState* mesh_state_create(int device) { MeshState* ms = new MeshState(); ms->device = device; return ms; }
void mesh_state_destroy(MeshState* ms) {
    if (!ms) return;
    use_device(ms->device);
    for (void* p : { (void*)ms->d_density, (void*)ms->d_vertidx, (void*)ms->d_blocks, (void*)ms->d_verts, (void*)ms->d_nraw, (void*)ms->d_normals,
            (void*)ms->d_colf, (void*)ms->d_col8, (void*)ms->d_indices })
        if (p) hipFree(p);
    delete ms;
}

int mesh_reserve_lattice(MeshState& ms, size_t res3) {
    if (res3 <= ms.cap_lattice && ms.d_density) return MON_OK;
    for (void* p : { (void*)ms.d_density, (void*)ms.d_vertidx, (void*)ms.d_blocks }) if (p) hipFree(p);
    ms.d_density = nullptr; ms.d_vertidx = nullptr; ms.d_blocks = nullptr; ms.cap_lattice = 0;
    HIPCHECK(hipMalloc((void**)&ms.d_density, res3 * 4)); HIPCHECK(hipMalloc((void**)&ms.d_vertidx, res3 * 12));
    HIPCHECK(hipMalloc((void**)&ms.d_blocks, ((res3 + 255) / 256 + 1) * 8));
    ms.cap_lattice = res3; return MON_OK;
}

// MarchingCubes + compute_mesh_1ring on a density lattice already resident in ms.d_density.
int mesh_extract(MeshState& ms, hipStream_t s, int rx, int ry, int rz, float thresh, const float* amin, const float* amax) {
    const size_t res3 = (size_t)rx * ry * rz; const uint32_t nb = (uint32_t)((res3 + 255) / 256);
    launch_mc_count(s, ms.d_density, rx, ry, rz, thresh, ms.d_blocks);
    uint64_t totals = 0;
    // the reference's count-pass read-back :492-494
    HIPCHECK(hipMemcpyAsync(&totals, ms.d_blocks + nb, 8, hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s));
    ms.n_verts_real = (uint32_t)(totals & 0xffffffffu); ms.n_indices = (uint32_t)(totals >> 32);
    ms.n_verts = (ms.n_verts_real + 127u) & ~127u;                                                                         // "round for later nn stuff" :496
    if (ms.n_verts > ms.cap_verts || !ms.d_verts) {
        size_t c0 = ms.cap_verts, c1 = ms.cap_verts, c2 = ms.cap_verts, c3 = ms.cap_verts, c4 = ms.cap_verts; int rc;
        if ((rc = grow(ms.d_verts, c0, ms.n_verts, 3)) || (rc = grow(ms.d_nraw, c1, ms.n_verts, 3)) || (rc = grow(ms.d_normals, c2, ms.n_verts, 3)) ||
            (rc = grow(ms.d_colf, c3, ms.n_verts, 3)) || (rc = grow(ms.d_col8, c4, ms.n_verts, 3))) { ms.cap_verts = 0; return rc; }
        ms.cap_verts = c0;
    }
    { int rc = grow(ms.d_indices, ms.cap_indices, ms.n_indices); if (rc) return rc; }
    if (ms.n_verts) {
        HIPCHECK(hipMemsetAsync(ms.d_verts, 0, (size_t)ms.n_verts * 12, s)); HIPCHECK(hipMemsetAsync(ms.d_nraw, 0, (size_t)ms.n_verts * 12, s));
        HIPCHECK(hipMemsetAsync(ms.d_normals, 0, (size_t)ms.n_verts * 12, s));
    }
    launch_mc_emit(s, ms.d_density, rx, ry, rz, thresh, amin, amax, ms.d_blocks, ms.d_vertidx, ms.d_verts, ms.d_indices, ms.d_nraw, ms.d_normals,
            ms.n_verts_real, ms.n_indices);
    HIPCHECK(hipGetLastError());
    return MON_OK;
}

// TransCPUMesh nerf_model.cu:2071-2095
int mesh_to_cpu(MeshState& ms, hipStream_t s, bool with_colors) {
    std::unique_lock<std::mutex> lock(ms.mu);
    const size_t n = ms.n_verts;
    ms.verts.resize(3 * n); ms.normals.resize(3 * n); ms.normals_raw.resize(3 * n); ms.colors.assign(3 * n, 0); ms.colors_f32.assign(3 * n, 0.f);
    ms.indices.resize(ms.n_indices);
    if (n) {
        HIPCHECK(hipMemcpyAsync(ms.verts.data(), ms.d_verts, n * 12, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(ms.normals.data(), ms.d_normals, n * 12, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(ms.normals_raw.data(), ms.d_nraw, n * 12, hipMemcpyDeviceToHost, s));
        if (with_colors) { HIPCHECK(hipMemcpyAsync(ms.colors.data(), ms.d_col8, n * 3, hipMemcpyDeviceToHost, s));
            HIPCHECK(hipMemcpyAsync(ms.colors_f32.data(), ms.d_colf, n * 12, hipMemcpyDeviceToHost, s)); }
    }
    if (ms.n_indices) HIPCHECK(hipMemcpyAsync(ms.indices.data(), ms.d_indices, (size_t)ms.n_indices * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    ms.cpu_n_real = ms.n_verts_real; ms.have_result = true; ms.generation.fetch_add(1);
    return MON_OK;
}

// save_mesh marching_cubes.cu:511-653, unwrap_it = false, nerf_scale = 1, nerf_offset = 0 (nerf_model.h:136-137)
int mesh_save(MeshState& ms, const char* path) {
    std::unique_lock<std::mutex> lock(ms.mu);
    if (!ms.have_result) { set_error("SaveMesh: no mesh has been generated"); return MON_ERR_STATE; }
    const std::string name(path);
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_error("Failed to open %s for writing.", path); return MON_ERR_IO; }
    const size_t nv = ms.verts.size() / 3, nf = ms.indices.size() / 3;
    auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
    if (name.size() >= 3 && name.substr(name.size() - 3) == "ply") {
        std::fprintf(f, "ply\nformat ascii 1.0\ncomment Multi-Object-NeRF object mesh (gfx950 core)\nelement vertex %u\n"
                        "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
                        "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face %u\nproperty list uchar int vertex_index\nend_header\n",
                     (unsigned)nv, (unsigned)nf);
        for (size_t i = 0; i < nv; ++i) {
            const float* p = &ms.verts[3 * i]; const float* n = &ms.normals[3 * i]; const float* c = &ms.colors_f32[3 * i];
            std::fprintf(f, "%0.5f %0.5f %0.5f %0.3f %0.3f %0.3f %d %d %d\n", p[0], p[1], p[2], n[0], n[1], n[2],
                         (int)(unsigned char)clampf(c[0] * 255.f, 0.f, 255.f), (int)(unsigned char)clampf(c[1] * 255.f, 0.f, 255.f),
                                 (int)(unsigned char)clampf(c[2] * 255.f, 0.f, 255.f));
        }
        // reversed winding :609
        for (size_t i = 0; i < 3 * nf; i += 3) std::fprintf(f, "3 %d %d %d\n", (int)ms.indices[i + 2], (int)ms.indices[i + 1], (int)ms.indices[i]);
    } else {                                                                                                                                          // obj
        for (size_t i = 0; i < nv; ++i) {
            const float* p = &ms.verts[3 * i]; const float* c = &ms.colors_f32[3 * i];
            std::fprintf(f, "v %0.5f %0.5f %0.5f %0.3f %0.3f %0.3f\n", p[0], p[1], p[2], clampf(c[0], 0.f, 1.f), clampf(c[1], 0.f, 1.f),
                    clampf(c[2], 0.f, 1.f));
        }
        for (size_t i = 0; i < nv; ++i) { const float* n = &ms.normals[3 * i]; std::fprintf(f, "vn %0.5f %0.5f %0.5f\n", n[0], n[1], n[2]); }
        for (size_t i = 0; i < 3 * nf; i += 3)
            std::fprintf(f, "f %u//%u %u//%u %u//%u\n", ms.indices[i + 2] + 1, ms.indices[i + 2] + 1, ms.indices[i + 1] + 1, ms.indices[i + 1] + 1,
                    ms.indices[i] + 1, ms.indices[i] + 1);
    }
    std::fclose(f);
    return MON_OK;
}

// GenerateMesh nerf_model.cu:1993-2004 + TransCPUMesh: density lattice (inference weights, raw channel 3) -> marching cubes ->
// normals -> vertex colours -> CPU copy.  Runs on the object's stream; the batch workspace doubles as the inference scratch.
int model_generate_mesh(Model& m, int res, float thresh, uint32_t* n_verts, uint32_t* n_indices) {
    if (res <= 0) res = 64;                                                   // marching_cubes.h:30
    if (res < 2 || res > 512) { set_error("generate_mesh: res must be in [2, 512]"); return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device));
    // created with the object (no lazily published pointer for readers to race on)
    if (!m.mesh) { set_error("mesh: object has no mesh state"); return MON_ERR_STATE; }
    model_leave_lane(m);
    MeshState& ms = *m.mesh; hipStream_t s = m.train_stream;
    const size_t res3 = (size_t)res * res * res;
    int rc = mesh_reserve_lattice(ms, res3); if (rc) return rc;
    rc = ensure_ema_current(m); if (rc) return rc;
    HIPCHECK(hipStreamSynchronize(s));
    HIPCHECK(hipMemcpy(&m.h_state, m.d_state, offsetof(DevState, n_scatter), hipMemcpyDeviceToHost));
    const uint16_t* prm = (m.h_state.step > 0) ? m.P.ema : m.P.half;
    // the two network passes: on feature-planar level tiles in LDS (kernels_tilerender.hip, the device's train-side workspace) or, for tables beyond the
    // tiles, layer by layer.  The workspace is shared by the device's objects: it is held for each PASS only -- the reference meshes objects concurrently, one
    // thread each (nerf.cu:138-145), and marching cubes with its host round trips needs none of it -- and never released while kernels that read it may still
    // be queued (TileWs::mu: "a user holds mu until its stream is synchronised"), on the error paths either.
    struct TileHold {
        TileWs* ws; hipStream_t s; std::unique_lock<std::mutex> lock;
        TileHold(TileWs* w, hipStream_t st) : ws(w), s(st) { if (ws) lock = std::unique_lock<std::mutex>(ws->mu); }
        ~TileHold() { if (lock.owns_lock()) { (void)hipStreamSynchronize(s); lock.unlock(); } }
    };
    TileWs* ws = nullptr;
    if (m.backend == 1 && m.tile_ok && options().tile_render != 0) { rc = tile_ws_get(m, 0, 0, &ws); if (rc) return rc; }
    const uint32_t chunk = ws ? ws->cap : m.ws_samples;
    {   TileHold hold(ws, s);
        if (ws) tile_ws_weights(m, *ws, s, prm, m.weights_epoch);
        for (size_t p0 = 0; p0 < res3; p0 += chunk) {                         // GetDensityOnGrid :2007-2048
            const uint32_t n = (uint32_t)((res3 - p0) < chunk ? (res3 - p0) : chunk);
            if (ws) {
                launch_grid_points4(s, ws->x, res, res, res, (uint32_t)p0, n);
                tile_points_forward(m, *ws, s, n);
                launch_extract_density(s, ws->O, ms.d_density + p0, n);
                continue;
            }
            launch_grid_points(s, m.B.pts, res, res, res, (uint32_t)p0, n);
            launch_encode(s, m.lt, m.nd, prm, m.B.pts, m.B.E, n, nullptr);
            mlp_forward_inference(m, s, prm, m.B.E, m.B.O, n);
            launch_extract_density(s, m.B.O, ms.d_density + p0, n);
        }
    }
    rc = mesh_extract(ms, s, res, res, res, thresh, m.oc.aabb.mn, m.oc.aabb.mx); if (rc) return rc;
    {   TileHold hold(ws, s);
        if (ws) tile_ws_weights(m, *ws, s, prm, m.weights_epoch);             // (another object may have used the workspace meanwhile: rebuilt only then)
        for (uint32_t v0 = 0; v0 < ms.n_verts; v0 += chunk) {                 // compute_mesh_vertex_colors :2050-2069 (padding vertices included)
            const uint32_t n = (ms.n_verts - v0) < chunk ? (ms.n_verts - v0) : chunk;
            if (ws) {
                launch_mesh_warp4(s, ms.d_verts, ws->x, v0, n, m.oc.aabb);
                tile_points_forward(m, *ws, s, n);
                launch_mesh_colors(s, ws->O, ms.d_colf, ms.d_col8, v0, n);
                continue;
            }
            launch_mesh_warp(s, ms.d_verts, m.B.pts, v0, n, m.oc.aabb);
            launch_encode(s, m.lt, m.nd, prm, m.B.pts, m.B.E, n, nullptr);
            mlp_forward_inference(m, s, prm, m.B.E, m.B.O, n);
            launch_mesh_colors(s, m.B.O, ms.d_colf, ms.d_col8, v0, n);
        }
    }
    HIPCHECK(hipGetLastError());
    rc = mesh_to_cpu(ms, s, true); if (rc) return rc;
    if (n_verts) *n_verts = ms.n_verts;
    if (n_indices) *n_indices = ms.n_indices;
    return MON_OK;
}

int model_mesh_counts(Model& m, uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices) {
    if (!m.mesh) { if (n_verts) *n_verts = 0; if (n_verts_real) *n_verts_real = 0; if (n_indices) *n_indices = 0; return MON_OK; }
    std::unique_lock<std::mutex> lock(m.mesh->mu);
    if (n_verts) *n_verts = (uint32_t)(m.mesh->verts.size() / 3);
    if (n_verts_real) *n_verts_real = m.mesh->cpu_n_real;
    if (n_indices) *n_indices = (uint32_t)m.mesh->indices.size();
    return MON_OK;
}

// Copy-out of CPUMeshData; try_only mirrors DrawCPUMesh's try_lock (nerf.cu:486-490): returns MON_ERR_STATE when the trainer holds the mesh.
int model_get_mesh(Model& m, float* verts, float* normals, uint8_t* colors, uint32_t* indices, float* normals_raw, float* colors_f32, int try_only) {
    if (!m.mesh) { set_error("get_mesh: no mesh has been generated"); return MON_ERR_STATE; }
    std::unique_lock<std::mutex> lock(m.mesh->mu, std::defer_lock);
    if (try_only) { if (!lock.try_lock()) { set_error("get_mesh: mesh is being updated"); return MON_ERR_STATE; } } else lock.lock();
    MeshState& ms = *m.mesh;
    if (!ms.have_result) { set_error("get_mesh: no mesh has been generated"); return MON_ERR_STATE; }
    if (verts) std::memcpy(verts, ms.verts.data(), ms.verts.size() * 4);
    if (normals) std::memcpy(normals, ms.normals.data(), ms.normals.size() * 4);
    if (colors) std::memcpy(colors, ms.colors.data(), ms.colors.size());
    if (indices) std::memcpy(indices, ms.indices.data(), ms.indices.size() * 4);
    if (normals_raw) std::memcpy(normals_raw, ms.normals_raw.data(), ms.normals_raw.size() * 4);
    if (colors_f32) std::memcpy(colors_f32, ms.colors_f32.data(), ms.colors_f32.size() * 4);
    return MON_OK;
}
// DrawCPUMesh-safe copy-out: counts and data under ONE hold of the mesh mutex, bounded by the caller's capacities (the training thread may
// publish a larger mesh between a mon_object_mesh_counts call and the copy).  MON_ERR_ARG + the needed counts when a buffer is too small.
int model_copy_mesh(Model& m, uint32_t cap_verts, uint32_t cap_indices, float* verts, float* normals, uint8_t* colors, uint32_t* indices,
                    uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices, int try_only) {
    if (!m.mesh) { set_error("copy_mesh: no mesh has been generated"); return MON_ERR_STATE; }
    std::unique_lock<std::mutex> lock(m.mesh->mu, std::defer_lock);
    if (try_only) { if (!lock.try_lock()) { set_error("copy_mesh: mesh is being updated"); return MON_ERR_STATE; } } else lock.lock();
    MeshState& ms = *m.mesh;
    if (!ms.have_result) { set_error("copy_mesh: no mesh has been generated"); return MON_ERR_STATE; }
    const uint32_t nv = (uint32_t)(ms.verts.size() / 3), ni = (uint32_t)ms.indices.size();
    if (n_verts) *n_verts = nv; if (n_verts_real) *n_verts_real = ms.cpu_n_real; if (n_indices) *n_indices = ni;
    if (nv > cap_verts || ni > cap_indices) {
        set_error("copy_mesh: buffers hold %u vertices / %u indices, the mesh has %u / %u", cap_verts, cap_indices, nv, ni); return MON_ERR_ARG; }
    if (verts) std::memcpy(verts, ms.verts.data(), ms.verts.size() * 4);
    if (normals) std::memcpy(normals, ms.normals.data(), ms.normals.size() * 4);
    if (colors) std::memcpy(colors, ms.colors.data(), ms.colors.size());
    if (indices) std::memcpy(indices, ms.indices.data(), ms.indices.size() * 4);
    return MON_OK;
}
int model_mesh_generation(Model& m, uint64_t* gen) { *gen = m.mesh ? m.mesh->generation.load() : 0; return MON_OK; }
int model_save_mesh(Model& m, const char* path) {
    if (!m.mesh) { set_error("SaveMesh: no mesh has been generated"); return MON_ERR_STATE; }
    return mesh_save(*m.mesh, path);
}
void model_mesh_free(Model& m) { mesh_state_destroy(m.mesh); m.mesh = nullptr; }

// Marching cubes on a caller-supplied lattice (test / tooling entry: analytic fields, non-cubic lattices).
int marching_cubes_host(int device, const float* density, int rx, int ry, int rz, float thresh, const float* amin, const float* amax,
                        float* verts, float* normals_raw, uint32_t* indices, uint32_t cap_verts, uint32_t cap_indices, uint32_t* n_verts,
                                uint32_t* n_verts_real, uint32_t* n_indices) {
    if (!density || !amin || !amax || rx < 2 || ry < 2 || rz < 2) { set_error("marching_cubes: bad argument"); return MON_ERR_ARG; }
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available"); return MON_ERR_NO_DEVICE; }
    HIPCHECK(use_device(device));
    MeshState* ms = mesh_state_create(device); const size_t res3 = (size_t)rx * ry * rz;
    int rc = mesh_reserve_lattice(*ms, res3);
    if (!rc) { if (hipMemcpy(ms->d_density, density, res3 * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("marching_cubes: upload failed");
            rc = MON_ERR_HIP; } }
    if (!rc) rc = mesh_extract(*ms, nullptr, rx, ry, rz, thresh, amin, amax);
    if (!rc) rc = mesh_to_cpu(*ms, nullptr, false);
    if (!rc) {
        if (n_verts) *n_verts = ms->n_verts; if (n_verts_real) *n_verts_real = ms->n_verts_real; if (n_indices) *n_indices = ms->n_indices;
        if (verts && cap_verts >= ms->n_verts) std::memcpy(verts, ms->verts.data(), ms->verts.size() * 4);
        if (normals_raw && cap_verts >= ms->n_verts) std::memcpy(normals_raw, ms->normals_raw.data(), ms->normals_raw.size() * 4);
        if (indices && cap_indices >= ms->n_indices) std::memcpy(indices, ms->indices.data(), ms->indices.size() * 4);
        if ((verts || normals_raw) && cap_verts < ms->n_verts) { set_error("marching_cubes: vertex buffer too small (%u < %u)", cap_verts, ms->n_verts);
            rc = MON_ERR_ARG; }
        if (indices && cap_indices < ms->n_indices) { set_error("marching_cubes: index buffer too small (%u < %u)", cap_indices, ms->n_indices);
            rc = MON_ERR_ARG; }
    }
    mesh_state_destroy(ms);
    return rc;
}

}  // namespace mon
