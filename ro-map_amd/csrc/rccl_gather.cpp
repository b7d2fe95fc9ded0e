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

// A RANK of the gather = one LOGICAL device of the core library (mon_device_count; mon_set_logical_devices maps several onto one GPU): its physical GPU, a stream
// there and its outgoing message (its objects' crops, packed).  The communicator itself spans the PHYSICAL devices (RCCL refuses two ranks per device).
struct Rank { int phys = 0; hipStream_t stream = nullptr; float* msg = nullptr; size_t cap = 0; };

// an open ncclGroupStart is always closed, also on the error paths (a thread left in group mode queues every later RCCL call of that thread)
struct NcclGroup {
    bool open = false;
    ncclResult_t start() { const ncclResult_t r = ncclGroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { open = false; return ncclGroupEnd(); }
    ~NcclGroup() { if (open) (void)ncclGroupEnd(); }
};

// grow-only device / pinned buffers: the pointer is cleared BEFORE the old block is freed, so a failing free cannot leave it dangling for a later destroy
int grow_device(float*& p, size_t& cap, size_t floats, int phys) {
    if (floats <= cap) return MON_OK;
    HIP_OK(hipSetDevice(phys)); float* old = p; p = nullptr; cap = 0; if (old) HIP_OK(hipFree(old));
    HIP_OK(hipMalloc((void**)&p, floats * 4)); cap = floats; return MON_OK;
}
int grow_pinned(float*& p, size_t& cap, size_t floats, int phys) {
    if (floats <= cap) return MON_OK;
    HIP_OK(hipSetDevice(phys)); float* old = p; p = nullptr; cap = 0; if (old) HIP_OK(hipHostFree(old));
    HIP_OK(hipHostMalloc((void**)&p, floats * 4, hipHostMallocDefault)); cap = floats; return MON_OK;
}

}  // namespace

struct mon_gather {
    int n_ranks = 0, root = 0, n_phys = 0, root_phys = 0;      // ranks = logical devices; the root rank and its GPU
    int transport = MON_GATHER_AUTO;
    std::vector<ncclComm_t> comms;                // one per PHYSICAL device (ncclCommInitAll)
    std::vector<Rank> rk;
    float* recv = nullptr; size_t recv_cap = 0;   // on the root's GPU: the other ranks' messages back to back
    float* h_stage = nullptr; size_t h_cap = 0;   // pinned: everything the root hands to the host
    uint64_t bytes_links = 0, bytes_root = 0, bytes_rccl = 0, bytes_copy = 0; int senders = 0, msgs_rccl = 0, msgs_copy = 0; double transfer_ms = 0.0;
};

extern "C" {

const char* mon_gather_last_error(void) { return g_rccl_err.c_str(); }

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
    int n_ranks = 0; if (mon_device_count(&n_ranks) != MON_OK || n_ranks < 1) return fail(MON_ERR_NO_DEVICE, "gather_create: %s", mon_last_error());
    if (root_device < 0 || root_device >= n_ranks) return fail(MON_ERR_ARG, "gather_create: no logical device %d (of %d)", root_device, n_ranks);
    mon_gather* g = new mon_gather(); g->n_ranks = n_ranks; g->root = root_device; g->n_phys = n_phys; g->comms.assign(n_phys, nullptr); g->rk.resize(n_ranks);
    // (from here on every failure goes through mon_gather_destroy: nothing created so far leaks)
    for (int d = 0; d < n_ranks; ++d) if (mon_physical_device(d, &g->rk[d].phys) != MON_OK) { mon_gather_destroy(g);
        return fail(MON_ERR_ARG, "gather_create: %s", mon_last_error()); }
    g->root_phys = g->rk[root_device].phys;
    std::vector<int> ids(n_phys); for (int d = 0; d < n_phys; ++d) ids[d] = d;
    ncclResult_t r = ncclCommInitAll(g->comms.data(), n_phys, ids.data());
    if (r != ncclSuccess) { for (auto& c : g->comms) c = nullptr; mon_gather_destroy(g);
        return fail(MON_ERR_HIP, "ncclCommInitAll over %d devices: %s", n_phys, ncclGetErrorString(r)); }
    // once per communicator set: what RCCL itself says it built (rank count, device of every communicator) -- the line an 8-GPU run's log should carry
    {   std::string desc; int cnt = 0;
        for (int d = 0; d < n_phys; ++d) { int c = 0, dev = -1, rk = -1; (void)ncclCommCount(g->comms[d], &c); (void)ncclCommCuDevice(g->comms[d], &dev);
            (void)ncclCommUserRank(g->comms[d], &rk); cnt = c; char b[48]; snprintf(b, sizeof b, "%s%d@gpu%d", d ? " " : "", rk, dev); desc += b; }
        fprintf(stderr, "[mon_gather] RCCL communicators over %d GPU(s): ncclCommCount %d, rank@device %s; %d logical rank(s), root rank %d on GPU %d\n", n_phys, cnt,
                desc.c_str(), n_ranks, root_device, g->root_phys);
    }
    for (int d = 0; d < n_ranks; ++d) {
        hipError_t e = hipSetDevice(g->rk[d].phys); if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->rk[d].stream, hipStreamNonBlocking);
        if (e != hipSuccess) { mon_gather_destroy(g); return fail(MON_ERR_HIP, "gather_create: stream of rank %d: %s", d, hipGetErrorString(e)); }
    }
    *out = g; return MON_OK;
}

int mon_gather_destroy(mon_gather* g) {
    if (!g) return MON_OK;
    for (auto& k : g->rk) {
        (void)hipSetDevice(k.phys);
        if (k.stream) { (void)hipStreamSynchronize(k.stream); (void)hipStreamDestroy(k.stream); k.stream = nullptr; }
        if (k.msg) { float* m = k.msg; k.msg = nullptr; (void)hipFree(m); }
    }
    for (auto& c : g->comms) if (c) { ncclCommDestroy(c); c = nullptr; }
    (void)hipSetDevice(g->root_phys);
    if (g->recv) { float* m = g->recv; g->recv = nullptr; (void)hipFree(m); }
    if (g->h_stage) { float* m = g->h_stage; g->h_stage = nullptr; (void)hipHostFree(m); }
    delete g; return MON_OK;
}

int mon_gather_set_transport(mon_gather* g, int transport) {
    if (!g || transport < MON_GATHER_AUTO || transport > MON_GATHER_PEER_COPY) return fail(MON_ERR_ARG, "gather_set_transport: bad argument");
    g->transport = transport; return MON_OK;
}

int mon_gather_stats(mon_gather* g, uint64_t* bytes_over_links, uint64_t* bytes_on_root, int* sending_devices, double* transfer_ms) {
    if (!g) return fail(MON_ERR_ARG, "gather_stats: null argument");
    if (bytes_over_links) *bytes_over_links = g->bytes_links; if (bytes_on_root) *bytes_on_root = g->bytes_root;
    if (sending_devices) *sending_devices = g->senders; if (transfer_ms) *transfer_ms = g->transfer_ms;
    return MON_OK;
}

int mon_gather_transport_stats(mon_gather* g, int* n_ranks, uint64_t* bytes_rccl, int* messages_rccl, uint64_t* bytes_peer_copy, int* messages_peer_copy) {
    if (!g) return fail(MON_ERR_ARG, "gather_transport_stats: null argument");
    if (n_ranks) *n_ranks = g->n_ranks; if (bytes_rccl) *bytes_rccl = g->bytes_rccl; if (messages_rccl) *messages_rccl = g->msgs_rccl;
    if (bytes_peer_copy) *bytes_peer_copy = g->bytes_copy; if (messages_peer_copy) *messages_peer_copy = g->msgs_copy;
    return MON_OK;
}

int mon_gather_renders(mon_gather* g, mon_object* const* objects, const mon_frame_bbox* boxes, const float* poses16, int pose_is_Toc, int n,
                       float* const* rgb, float* const* depth, float* const* mask) {
    if (!g || n < 0 || (n && (!objects || !boxes || !poses16 || !rgb || !depth || !mask))) return fail(MON_ERR_ARG, "gather_renders: bad argument");
    // ---- where every object lives (its LOGICAL device = its rank), and the messages that follow from it
    std::vector<int> rank_of(n); std::vector<uint32_t> npix(n);
    for (int i = 0; i < n; ++i) {
        mon_object_info info;
        if (mon_object_info_get(objects[i], &info) != MON_OK) return fail(MON_ERR_ARG, "gather_renders: object %d: %s", i, mon_last_error());
        rank_of[i] = info.device;
        if (rank_of[i] < 0 || rank_of[i] >= g->n_ranks) return fail(MON_ERR_STATE, "gather_renders: object %d on logical device %d, the gather was created "
                "over %d (mon_set_logical_devices changed since?)", i, rank_of[i], g->n_ranks);
        npix[i] = boxes[i].w * boxes[i].h;
        if (!npix[i] || !rgb[i] || !depth[i] || !mask[i]) return fail(MON_ERR_ARG, "gather_renders: object %d: empty box or null output", i);
    }
    std::vector<uint64_t> len(g->n_ranks), off(n);
    { const int rc = mon_gather_plan(rank_of.data(), npix.data(), n, g->n_ranks, len.data(), off.data()); if (rc) return rc; }
    for (int d = 0; d < g->n_ranks; ++d) { const int rc = grow_device(g->rk[d].msg, g->rk[d].cap, len[d], g->rk[d].phys); if (rc) return rc; }
    std::vector<uint64_t> recv_off(g->n_ranks, 0); uint64_t recv_len = 0; g->senders = 0;
    for (int d = 0; d < g->n_ranks; ++d) if (d != g->root && len[d]) { recv_off[d] = recv_len; recv_len += len[d]; ++g->senders; }
    { const int rc = grow_device(g->recv, g->recv_cap, recv_len, g->root_phys); if (rc) return rc; }
    uint64_t total = 0; for (int d = 0; d < g->n_ranks; ++d) total += len[d];
    { const int rc = grow_pinned(g->h_stage, g->h_cap, total, g->root_phys); if (rc) return rc; }
    // ---- render: every rank's objects one after the other into the rank's message (device-resident: dst_on_device = 1), the ranks side by side
    std::vector<int> rcs(g->n_ranks, MON_OK); std::vector<std::string> errs(g->n_ranks); std::vector<std::thread> th;
    for (int d = 0; d < g->n_ranks; ++d) {
        if (!len[d]) continue;
        th.emplace_back([&, d] {
            for (int i = 0; i < n; ++i) {
                if (rank_of[i] != d) continue;
                float* base = g->rk[d].msg + off[i];
                const int rc = mon_object_render(objects[i], boxes[i], poses16 + 16 * (size_t)i, pose_is_Toc, base, base + 3 * (size_t)npix[i],
                        base + 4 * (size_t)npix[i], 1);
                if (rc != MON_OK) { rcs[d] = rc; errs[d] = mon_last_error(); return; }
            }
        });
    }
    for (auto& t : th) t.join();
    for (int d = 0; d < g->n_ranks; ++d) if (rcs[d] != MON_OK) return fail(rcs[d], "gather_renders: render on logical device %d: %s", d, errs[d].c_str());
    // ---- the other ranks' messages to the root.  Two transports behind one bookkeeping (recv_off, the unpack below):
    //   RCCL       ranks on another GPU than the root's: one grouped batch of ncclSend / ncclRecv of the true sizes, every transfer on its own direct xGMI link
    //   peer copy  ranks that share the root's GPU (logical devices; RCCL has one rank per GPU) -- or every rank with MON_GATHER_PEER_COPY: hipMemcpyPeerAsync
    //              on the sender's stream into the same slot of the root's receive buffer
    const auto t0 = std::chrono::steady_clock::now();
    g->bytes_links = 0; g->bytes_root = len[g->root] * 4; g->bytes_rccl = g->bytes_copy = 0; g->msgs_rccl = g->msgs_copy = 0;
    if (g->senders) {
        std::vector<char> via_rccl(g->n_ranks, 0);
        for (int d = 0; d < g->n_ranks; ++d) if (d != g->root && len[d])
            via_rccl[d] = g->transport != MON_GATHER_PEER_COPY && g->rk[d].phys != g->root_phys && g->comms[g->rk[d].phys] && g->comms[g->root_phys];
        // (one stream per communicator inside a group: the ranks of one GPU send on the stream of that GPU's first rank; their messages were rendered
        // synchronously, nothing is pending on the others)
        std::vector<int> first_rank_of(g->n_phys, -1);
        for (int d = g->n_ranks - 1; d >= 0; --d) first_rank_of[g->rk[d].phys] = d;
        {   NcclGroup grp; ncclResult_t r = ncclSuccess; bool any = false;
            for (int d = 0; d < g->n_ranks && r == ncclSuccess; ++d) {
                if (!via_rccl[d]) continue;
                if (!any) { r = grp.start(); any = true; if (r != ncclSuccess) break; }
                r = ncclSend(g->rk[d].msg, len[d], ncclFloat, g->root_phys, g->comms[g->rk[d].phys], g->rk[first_rank_of[g->rk[d].phys]].stream);
                if (r == ncclSuccess) r = ncclRecv(g->recv + recv_off[d], len[d], ncclFloat, g->rk[d].phys, g->comms[g->root_phys], g->rk[g->root].stream);
                if (r == ncclSuccess) { g->bytes_rccl += len[d] * 4; ++g->msgs_rccl; }      // (counted once both calls were accepted: ADVICE r05)
            }
            if (r == ncclSuccess && any) r = grp.end();
            if (r != ncclSuccess) return fail(MON_ERR_HIP, "gather_renders: RCCL send / receive: %s", ncclGetErrorString(r));
        }
        for (int d = 0; d < g->n_ranks; ++d) {
            if (d == g->root || !len[d] || via_rccl[d]) continue;
            HIP_OK(hipSetDevice(g->rk[d].phys));
            HIP_OK(hipMemcpyPeerAsync(g->recv + recv_off[d], g->root_phys, g->rk[d].msg, g->rk[d].phys, len[d] * 4, g->rk[d].stream));
            g->bytes_copy += len[d] * 4; ++g->msgs_copy;
        }
        for (int d = 0; d < g->n_ranks; ++d) if (d != g->root && len[d]) { HIP_OK(hipSetDevice(g->rk[d].phys));
            HIP_OK(hipStreamSynchronize(via_rccl[d] ? g->rk[first_rank_of[g->rk[d].phys]].stream : g->rk[d].stream)); }
        HIP_OK(hipSetDevice(g->root_phys)); HIP_OK(hipStreamSynchronize(g->rk[g->root].stream));      // (the receives)
        g->bytes_links = g->bytes_rccl + g->bytes_copy;
    }
    g->transfer_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();      // send / receive + synchronisation only
    // ---- root -> host, one pass: its own message, then the received ones
    HIP_OK(hipSetDevice(g->root_phys));
    std::vector<uint64_t> host_off(g->n_ranks, 0); uint64_t h = 0;
    for (int d = 0; d < g->n_ranks; ++d) {
        if (!len[d]) continue;
        host_off[d] = h;
        HIP_OK(hipMemcpyAsync(g->h_stage + h, d == g->root ? g->rk[d].msg : g->recv + recv_off[d], len[d] * 4, hipMemcpyDeviceToHost, g->rk[g->root].stream));
        h += len[d];
    }
    HIP_OK(hipStreamSynchronize(g->rk[g->root].stream));
    for (int i = 0; i < n; ++i) {
        const float* src = g->h_stage + host_off[rank_of[i]] + off[i]; const size_t p = npix[i];
        std::memcpy(rgb[i], src, 12 * p); std::memcpy(depth[i], src + 3 * p, 4 * p); std::memcpy(mask[i], src + 4 * p, 4 * p);
    }
    return MON_OK;
}

int mon_offline_render_test_gathered(mon_gather* g, mon_offline* mgr, const char* out_dir, int max_views) {
    if (!g || !mgr || !out_dir) return fail(MON_ERR_ARG, "render_test_gathered: null argument");
    // The FINAL render: the objects' training threads are joined first (mon_object_render is the caller-serialised entry point; mon_offline_render_test takes the
    // object's model mutex instead because it may run beside training).  Idempotent when the caller has already waited.
    if (mon_offline_wait_threads_end(mgr) != MON_OK) return fail(MON_ERR_STATE, "render_test_gathered: %s", mon_last_error());
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
    // "Save Object Mesh" (nerf.cu:397-403): <out>/<id>/obj.ply, like mon_offline_render_test; the mesh lives on the object's device and goes to disk from there
    for (int k = 0; k < n_obj; ++k) if (mon_offline_save_mesh(mgr, k, out_dir) != MON_OK) return fail(MON_ERR_STATE, "render_test_gathered: mesh of object %d: %s",
            k, mon_last_error());
    return MON_OK;
}

}  // extern "C"
