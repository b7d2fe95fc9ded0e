// rccl_gather.cpp -- libmon_core_rccl.so (include/mon_core_rccl.h): gather-to-root of the final renders over RCCL for one process whose objects sit on
// several devices.  Written against the public boundary (include/mon_core.h) only; the core library does not depend on RCCL.
//
// training has no collective; the final render is gathered from the owner GPUs to the GPU that composites / writes the images.  On the
// 8-GPU xGMI mesh every peer has a direct link to the root, so the peers' messages travel side by side (no ring): a single-process communicator
// (ncclCommInitAll over the visible devices), one grouped batch of ncclSend / ncclRecv of the TRUE message sizes, one message per device per call
// (all of a device's crops packed: at these sizes the transfers are latency-bound).  The reference has no equivalent: each object's thread writes its
// own PNGs from host copies (CORE/src/nerf.cu:255-404).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mon_core_rccl.h"

namespace {

thread_local std::string g_rccl_err;
int fail(int code, const char* fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_rccl_err = buf; std::fprintf(stderr, "libmon_core_rccl: %s\n", buf); return code;
}
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(MON_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define NCCL_OK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(MON_ERR_HIP, "%s: %s", #expr, ncclGetErrorString(r_)); } while (0)

struct DeviceSide { hipStream_t stream = nullptr; float* msg = nullptr; size_t cap = 0; };      // a device's outgoing message (its objects' crops, packed)

}  // namespace

struct mon_gather {
    int n_dev = 0, root = 0;                      // physical devices in the communicator, the root's physical id
    std::vector<ncclComm_t> comms; std::vector<DeviceSide> dev;
    float* recv = nullptr; size_t recv_cap = 0;   // on the root: the peers' messages back to back
    float* h_stage = nullptr; size_t h_cap = 0;   // pinned: everything the root hands to the host
    uint64_t bytes_links = 0, bytes_root = 0; int senders = 0; double transfer_ms = 0.0;
};

extern "C" {

int mon_gather_plan(const int* object_device, const uint32_t* n_pix, int n, int n_devices, uint64_t* floats_per_device, uint64_t* offset_of_object) {
    if (n < 0 || n_devices < 1 || (n && (!object_device || !n_pix)) || !floats_per_device) return fail(MON_ERR_ARG, "gather_plan: bad argument");
    for (int d = 0; d < n_devices; ++d) floats_per_device[d] = 0;
    for (int i = 0; i < n; ++i) {
        const int d = object_device[i];
        if (d < 0 || d >= n_devices) return fail(MON_ERR_ARG, "gather_plan: object %d on device %d of %d", i, d, n_devices);
        if (offset_of_object) offset_of_object[i] = floats_per_device[d];
        floats_per_device[d] += 5ull * n_pix[i];                      // rgb (3) | depth | mask
    }
    return MON_OK;
}

int mon_gather_create(int root_device, mon_gather** out) {
    if (!out) return fail(MON_ERR_ARG, "gather_create: null argument");
    int n_phys = 0; HIP_OK(hipGetDeviceCount(&n_phys));
    if (n_phys < 1) return fail(MON_ERR_NO_DEVICE, "gather_create: no HIP device");
    int root_phys = 0; if (mon_physical_device(root_device, &root_phys) != MON_OK) return fail(MON_ERR_ARG, "gather_create: no logical device %d", root_device);
    mon_gather* g = new mon_gather(); g->n_dev = n_phys; g->root = root_phys; g->comms.resize(n_phys); g->dev.resize(n_phys);
    std::vector<int> ids(n_phys); for (int d = 0; d < n_phys; ++d) ids[d] = d;
    ncclResult_t r = ncclCommInitAll(g->comms.data(), n_phys, ids.data());
    if (r != ncclSuccess) { delete g; return fail(MON_ERR_HIP, "ncclCommInitAll over %d devices: %s", n_phys, ncclGetErrorString(r)); }
    for (int d = 0; d < n_phys; ++d) { HIP_OK(hipSetDevice(d)); HIP_OK(hipStreamCreateWithFlags(&g->dev[d].stream, hipStreamNonBlocking)); }
    *out = g; return MON_OK;
}

int mon_gather_destroy(mon_gather* g) {
    if (!g) return MON_OK;
    for (int d = 0; d < g->n_dev; ++d) {
        (void)hipSetDevice(d);
        if (g->dev[d].stream) { (void)hipStreamSynchronize(g->dev[d].stream); (void)hipStreamDestroy(g->dev[d].stream); }
        if (g->dev[d].msg) (void)hipFree(g->dev[d].msg);
        if (g->comms[d]) ncclCommDestroy(g->comms[d]);
    }
    (void)hipSetDevice(g->root); if (g->recv) (void)hipFree(g->recv); if (g->h_stage) (void)hipHostFree(g->h_stage);
    delete g; return MON_OK;
}

int mon_gather_stats(mon_gather* g, uint64_t* bytes_over_links, uint64_t* bytes_on_root, int* sending_devices, double* transfer_ms) {
    if (!g) return fail(MON_ERR_ARG, "gather_stats: null argument");
    if (bytes_over_links) *bytes_over_links = g->bytes_links; if (bytes_on_root) *bytes_on_root = g->bytes_root;
    if (sending_devices) *sending_devices = g->senders; if (transfer_ms) *transfer_ms = g->transfer_ms;
    return MON_OK;
}

int mon_gather_renders(mon_gather* g, mon_object* const* objects, const mon_frame_bbox* boxes, const float* poses16, int pose_is_Toc, int n,
                       float* const* rgb, float* const* depth, float* const* mask) {
    if (!g || n < 0 || (n && (!objects || !boxes || !poses16 || !rgb || !depth || !mask))) return fail(MON_ERR_ARG, "gather_renders: bad argument");
    // ---- where every object lives, and the messages that follow from it
    std::vector<int> dev_of(n); std::vector<uint32_t> npix(n);
    for (int i = 0; i < n; ++i) {
        mon_object_info info;
        if (mon_object_info_get(objects[i], &info) != MON_OK) return fail(MON_ERR_ARG, "gather_renders: object %d: %s", i, mon_last_error());
        if (mon_physical_device(info.device, &dev_of[i]) != MON_OK) return fail(MON_ERR_ARG, "gather_renders: object %d: %s", i, mon_last_error());
        npix[i] = boxes[i].w * boxes[i].h;
        if (!npix[i] || !rgb[i] || !depth[i] || !mask[i]) return fail(MON_ERR_ARG, "gather_renders: object %d: empty box or null output", i);
    }
    std::vector<uint64_t> len(g->n_dev), off(n);
    { const int rc = mon_gather_plan(dev_of.data(), npix.data(), n, g->n_dev, len.data(), off.data()); if (rc) return rc; }
    for (int d = 0; d < g->n_dev; ++d) if (len[d] > g->dev[d].cap) {
        HIP_OK(hipSetDevice(d)); if (g->dev[d].msg) HIP_OK(hipFree(g->dev[d].msg));
        g->dev[d].msg = nullptr; g->dev[d].cap = 0; HIP_OK(hipMalloc((void**)&g->dev[d].msg, len[d] * 4)); g->dev[d].cap = len[d];
    }
    std::vector<uint64_t> recv_off(g->n_dev, 0); uint64_t recv_len = 0; g->senders = 0;
    for (int d = 0; d < g->n_dev; ++d) if (d != g->root && len[d]) { recv_off[d] = recv_len; recv_len += len[d]; ++g->senders; }
    if (recv_len > g->recv_cap) { HIP_OK(hipSetDevice(g->root)); if (g->recv) HIP_OK(hipFree(g->recv)); g->recv = nullptr; g->recv_cap = 0;
        HIP_OK(hipMalloc((void**)&g->recv, recv_len * 4)); g->recv_cap = recv_len; }
    uint64_t total = 0; for (int d = 0; d < g->n_dev; ++d) total += len[d];
    if (total > g->h_cap) { HIP_OK(hipSetDevice(g->root)); if (g->h_stage) HIP_OK(hipHostFree(g->h_stage)); g->h_stage = nullptr; g->h_cap = 0;
        HIP_OK(hipHostMalloc((void**)&g->h_stage, total * 4, hipHostMallocDefault)); g->h_cap = total; }
    // ---- render: every device's objects one after the other into the device's message (device-resident: dst_on_device = 1), the devices side by side
    std::vector<int> rcs(g->n_dev, MON_OK); std::vector<std::string> errs(g->n_dev); std::vector<std::thread> th;
    for (int d = 0; d < g->n_dev; ++d) {
        if (!len[d]) continue;
        th.emplace_back([&, d] {
            for (int i = 0; i < n; ++i) {
                if (dev_of[i] != d) continue;
                float* base = g->dev[d].msg + off[i];
                const int rc = mon_object_render(objects[i], boxes[i], poses16 + 16 * (size_t)i, pose_is_Toc, base, base + 3 * (size_t)npix[i],
                        base + 4 * (size_t)npix[i], 1);
                if (rc != MON_OK) { rcs[d] = rc; errs[d] = mon_last_error(); return; }
            }
        });
    }
    for (auto& t : th) t.join();
    for (int d = 0; d < g->n_dev; ++d) if (rcs[d] != MON_OK) return fail(rcs[d], "gather_renders: render on device %d: %s", d, errs[d].c_str());
    // ---- the peers' messages to the root: one grouped batch, every transfer on its own direct link
    const auto t0 = std::chrono::steady_clock::now();
    g->bytes_links = 0; g->bytes_root = len[g->root] * 4;
    if (g->senders) {
        NCCL_OK(ncclGroupStart());
        for (int d = 0; d < g->n_dev; ++d) {
            if (d == g->root || !len[d]) continue;
            NCCL_OK(ncclSend(g->dev[d].msg, len[d], ncclFloat, g->root, g->comms[d], g->dev[d].stream));
            NCCL_OK(ncclRecv(g->recv + recv_off[d], len[d], ncclFloat, d, g->comms[g->root], g->dev[g->root].stream));
            g->bytes_links += len[d] * 4;
        }
        NCCL_OK(ncclGroupEnd());
        for (int d = 0; d < g->n_dev; ++d) if (d != g->root && len[d]) { HIP_OK(hipSetDevice(d)); HIP_OK(hipStreamSynchronize(g->dev[d].stream)); }
    }
    // ---- root -> host, one pass: its own message, then the received ones
    HIP_OK(hipSetDevice(g->root));
    std::vector<uint64_t> host_off(g->n_dev, 0); uint64_t h = 0;
    for (int d = 0; d < g->n_dev; ++d) {
        if (!len[d]) continue;
        host_off[d] = h;
        HIP_OK(hipMemcpyAsync(g->h_stage + h, d == g->root ? g->dev[d].msg : g->recv + recv_off[d], len[d] * 4, hipMemcpyDeviceToHost, g->dev[g->root].stream));
        h += len[d];
    }
    HIP_OK(hipStreamSynchronize(g->dev[g->root].stream));
    g->transfer_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n; ++i) {
        const float* src = g->h_stage + host_off[dev_of[i]] + off[i]; const size_t p = npix[i];
        std::memcpy(rgb[i], src, 12 * p); std::memcpy(depth[i], src + 3 * p, 4 * p); std::memcpy(mask[i], src + 4 * p, 4 * p);
    }
    return MON_OK;
}

int mon_offline_render_test_gathered(mon_gather* g, mon_offline* mgr, const char* out_dir, int max_views) {
    if (!g || !mgr || !out_dir) return fail(MON_ERR_ARG, "render_test_gathered: null argument");
    int n_obj = 0; if (mon_offline_n_objects(mgr, &n_obj) != MON_OK) return fail(MON_ERR_ARG, "%s", mon_last_error());
    size_t n_frames = 0; mon_offline_get_poses(mgr, nullptr, 0, &n_frames);
    std::vector<float> poses(16 * n_frames);
    if (n_frames && mon_offline_get_poses(mgr, poses.data(), n_frames, &n_frames) != MON_OK) return fail(MON_ERR_STATE, "%s", mon_last_error());
    std::vector<mon_object*> objs(n_obj); std::vector<std::vector<mon_frame_bbox>> boxes(n_obj); std::vector<int> ids(n_obj); size_t views = 0;
    ::mkdir(out_dir, 0755);
    for (int k = 0; k < n_obj; ++k) {
        if (mon_offline_object(mgr, k, &objs[k]) != MON_OK) return fail(MON_ERR_ARG, "%s", mon_last_error());
        size_t nb = 0; mon_offline_object_meta(mgr, k, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &nb);
        boxes[k].resize(nb);
        if (nb && mon_offline_object_meta(mgr, k, nullptr, nullptr, nullptr, nullptr, boxes[k].data(), nb, &nb) != MON_OK) return fail(MON_ERR_STATE, "%s",
                mon_last_error());
        if (max_views > 0 && boxes[k].size() > (size_t)max_views) boxes[k].resize((size_t)max_views);
        views = std::max(views, boxes[k].size());
        ids[k] = k;                                                           // (object ids of OfflineNeRF are the creation order, nerf.cu:22-25)
        const std::string root = std::string(out_dir) + "/" + std::to_string(ids[k]);
        for (const char* sub : { "", "/test_img", "/test_depth", "/test_mask" }) ::mkdir((root + sub).c_str(), 0755);
    }
    for (size_t v = 0; v < views; ++v) {                                      // view v of every object that has one: one gather
        std::vector<mon_object*> o; std::vector<mon_frame_bbox> b; std::vector<float> p; std::vector<int> who;
        for (int k = 0; k < n_obj; ++k) if (v < boxes[k].size()) {
            const mon_frame_bbox bb = boxes[k][v];
            if (bb.FrameId >= n_frames) return fail(MON_ERR_STATE, "render_test_gathered: frame %u of %zu", bb.FrameId, n_frames);
            o.push_back(objs[k]); b.push_back(bb); who.push_back(k);
            p.insert(p.end(), poses.begin() + 16 * (size_t)bb.FrameId, poses.begin() + 16 * (size_t)bb.FrameId + 16);
        }
        std::vector<std::vector<float>> rgb(o.size()), depth(o.size()), mask(o.size()); std::vector<float*> pr, pd, pm;
        for (size_t i = 0; i < o.size(); ++i) { const size_t px = (size_t)b[i].w * b[i].h; rgb[i].resize(3 * px); depth[i].resize(px); mask[i].resize(px);
            pr.push_back(rgb[i].data()); pd.push_back(depth[i].data()); pm.push_back(mask[i].data()); }
        const int rc = mon_gather_renders(g, o.data(), b.data(), p.data(), 0, (int)o.size(), pr.data(), pd.data(), pm.data()); if (rc) return rc;
        for (size_t i = 0; i < o.size(); ++i) {
            char stamp[128]; if (mon_offline_object_stamp(mgr, who[i], v, stamp, sizeof stamp) != MON_OK) return fail(MON_ERR_STATE, "%s", mon_last_error());
            const std::string root = std::string(out_dir) + "/" + std::to_string(ids[who[i]]);
            const int wrc = mon_write_render_pngs((root + "/test_img/" + stamp + ".png").c_str(), (root + "/test_depth/" + stamp + ".png").c_str(),
                    (root + "/test_mask/" + stamp + ".png").c_str(),
                                                  b[i].w, b[i].h, rgb[i].data(), depth[i].data(), mask[i].data());
            if (wrc != MON_OK) return fail(wrc, "%s", mon_last_error());
        }
    }
    return MON_OK;
}

}  // extern "C"
