// model.cpp -- host side of one object NeRF on one gfx950 device.
// Mirrors nerf::NeRF_Model (CORE/src/nerf_model.cu:1259-1830) and nerf::NeRF_Dataset
// (CORE/src/nerf_data.cu:123-339) without Eigen/OpenCV/tcnn types.  Differences by design:
//   * the whole iteration is enqueued on one HIP stream with no host synchronisation (the
//     reference syncs 3x per iteration: nerf_model.cu:1459,1469,1645); counters live in DevState;
//   * an iteration can be replayed as a hipGraph;
//   * the dataset is one packed RGBA8+instance slab per device instead of per-frame float buffers.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <dlfcn.h>
#include "model.h"
#include "xorwow.h"

namespace mon {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); g_err = buf;
}
const char* last_error() { return g_err.c_str(); }

#define HIPCHECK(expr)                                                                                         \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                                       \
        set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return MON_ERR_HIP; } } while (0)

static Options g_options;
Options& options() { return g_options; }
static std::atomic<long>* option_slot(const char* name) {
    static const struct { const char* n; std::atomic<long> Options::*f; } tab[] = {
        { "backend", &Options::backend }, { "use_graph", &Options::use_graph }, { "big_switch", &Options::big_switch }, { "lds_encode", &Options::lds_encode },
        { "roctx", &Options::roctx }, { "step_variant", &Options::step_variant }, { "keep_zero_samples", &Options::keep_zero_samples },
        { "train_lanes", &Options::train_lanes }, { "tile_render", &Options::tile_render },
#ifdef MON_OVERLAP_PROBE
        { "overlap", &Options::overlap }, { "enc_lds_kb", &Options::enc_lds_kb },
#endif
    };
    for (const auto& e : tab) if (name && std::strcmp(name, e.n) == 0) return &(g_options.*(e.f));
    return nullptr;
}
int option_set(const char* name, long value) { std::atomic<long>* p = option_slot(name); if (!p) {
        set_error("set_option: unknown option '%s'", name ? name : "(null)"); return MON_ERR_ARG; } *p = value; return MON_OK; }
int option_get(const char* name, long* value) { std::atomic<long>* p = option_slot(name); if (!p || !value) {
        set_error("get_option: unknown option '%s'", name ? name : "(null)"); return MON_ERR_ARG; } *value = *p; return MON_OK; }

void config_default(mon_config& c);
int config_from_json(const char* path, mon_config& c);

// Logical devices: what the managers and the C ABI number 0 .. n-1.  By default they are the physical HIP devices; mon_set_logical_devices(n)
// maps n logical devices round-robin onto the physical ones, so the multi-device code paths (object k -> device k mod nGPU, one dataset replica and
// one stream pool per device, CORE/src/nerf.cu:27-33, nerf_manager.cu:44-55) also run -- oversubscribed -- on a box with fewer GPUs.
static std::atomic<int> g_logical_devices{ 0 };
static int physical_count() {          // (asked once: use_device sits on every entry point, and the runtime call takes a process-wide lock)
    static const int n = [] { int c = 0; return (hipGetDeviceCount(&c) == hipSuccess) ? c : 0; }();
    return n;
}
hipError_t use_device(int logical) {
    const int phys = physical_count(); if (phys < 1) return hipErrorNoDevice;
    const int n = g_logical_devices.load(); if (logical < 0 || logical >= (n > 0 ? n : phys)) return hipErrorInvalidDevice;
    return hipSetDevice(logical % phys);
}
int physical_device(int logical, int* phys_out) {
    const int phys = physical_count(); const int n = g_logical_devices.load();
    if (phys < 1 || logical < 0 || logical >= (n > 0 ? n : phys) || !phys_out) { set_error("physical_device: no such logical device %d", logical);
        return MON_ERR_ARG; }
    *phys_out = logical % phys; return MON_OK;
}
int set_logical_devices(int n) {
    if (n < 0 || n > 64) { set_error("set_logical_devices: 0 (= the physical devices) .. 64"); return MON_ERR_ARG; }
    g_logical_devices.store(n); return MON_OK;
}
int device_count(int* n) {
    int c = 0; hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c < 1) { *n = 0; set_error("Can not Detect GPU: %s", hipGetErrorString(e)); return MON_ERR_NO_DEVICE; }
    const int l = g_logical_devices.load(); *n = l > 0 ? l : c; return MON_OK;
}

// ------------------------------------------------------------------ dataset
int dataset_destroy(Dataset* d);
// One high-priority stream and one pinned result buffer per DEVICE, shared by the objects on it (viewer renders) and by the device's dataset (frame uploads):
// created with the device's first object (CreateNeRF is a
// milliseconds call anyway; created by the first render it was a 10 ms spike in front of the viewer), and only one more hardware-queue client however many
// objects train (a high-priority queue per object measurably slowed sliced training).  Renders of one device take turns on it.
// h_cap only grows; h_out / growth belong to mu
struct InferShared { std::mutex mu; hipStream_t stream = nullptr; float* h_out = nullptr; std::atomic<size_t> h_cap{ 0 }; };
static std::mutex g_infer_mu; static std::map<int, InferShared*> g_infer_shared;
static int infer_shared_get(int device, size_t pixels_hint, InferShared** out) {
    InferShared* sh = nullptr;
    {   std::lock_guard<std::mutex> l(g_infer_mu);       // the map lookup only: a viewer render holds sh->mu across stream syncs, and object creation
        InferShared*& slot = g_infer_shared[device];     // (SLAM thread, any device) must not queue behind it with the global lock held
        if (!slot) {
            slot = new InferShared();
            int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&slot->stream, hipStreamNonBlocking, hi) != hipSuccess) {
                delete slot; slot = nullptr; set_error("inference stream creation failed on device %d", device); return MON_ERR_HIP;
            }
        }
        sh = slot;
    }
    if (5 * pixels_hint > sh->h_cap.load(std::memory_order_acquire)) {      // (grow-only: the common case takes no lock at all)
        std::lock_guard<std::mutex> l2(sh->mu);          // h_out / growth belong to sh->mu: model_render_snapshot resizes them under it
        if (5 * pixels_hint <= sh->h_cap.load(std::memory_order_relaxed)) { *out = sh; return MON_OK; }
        float* q = nullptr; if (hipHostMalloc((void**)&q, 5 * pixels_hint * sizeof(float), hipHostMallocDefault) != hipSuccess) {
            set_error("pinned render buffer allocation failed"); return MON_ERR_HIP; }
        if (sh->h_out) hipHostFree(sh->h_out);
        sh->h_out = q; sh->h_cap = 5 * pixels_hint;
    }
    *out = sh; return MON_OK;
}

// ---- tile render workspace (kernels_tilerender.hip): per device and side, grow-only, never freed (like the inference stream)
static std::atomic<uint64_t> g_weights_epoch{ 1 };
uint64_t next_weights_epoch() { return g_weights_epoch.fetch_add(1); }
// objects: tile-capable objects alive on the device (under g_tile_mu); the buffers go with the last one
struct TileWsPair { TileWs side[2]; int objects = 0; };
static std::mutex g_tile_mu; static std::map<int, TileWsPair*> g_tile_ws;
template <class T> static int ws_grow(T*& p, size_t n_elems) {      // (contents are scratch: nothing to carry over)
    void* q = nullptr;
    if (hipMalloc(&q, n_elems * sizeof(T)) != hipSuccess) { set_error("tile render workspace: allocation of %zu bytes failed", n_elems * sizeof(T));
        return MON_ERR_HIP; }
    if (p) hipFree(p);
    p = (T*)q; return MON_OK;
}
// caller must hold ws->mu before it touches the buffers; the capacity checks below run under it too (two objects of one device may ask at once)
int tile_ws_get(Model& m, int side, size_t n_pix, TileWs** out) {
    // (the workspace is filed under m.device: allocate it there whatever device the calling thread happens to have current)
    HIPCHECK(use_device(m.device));
    TileWsPair* pr = nullptr;
    {   std::lock_guard<std::mutex> l(g_tile_mu);
        TileWsPair*& slot = g_tile_ws[m.device]; if (!slot) slot = new TileWsPair(); pr = slot; }
    TileWs& ws = pr->side[side & 1];
    std::lock_guard<std::mutex> l(ws.mu);
    int rc;
    const uint32_t cap = kTileChunkJobs * 64u;
    if (!ws.counters) {
        if ((rc = ws_grow(ws.counters, 64))) return rc;
        // (hipMemset of device memory may return before the fill has run, and the renders' streams are non-blocking: without the synchronisation the fill can
        // land in the middle of the first render's ray kernel -- the job count restarts, k_tile_render reads job records nobody wrote and writes to the pixel
        // index it finds there: the memory fault of the object-churn test, seen whenever a device's last object had returned the workspace)
        HIPCHECK(hipMemset(ws.counters, 0, 256)); HIPCHECK(hipDeviceSynchronize());
    }
    if (ws.cap < cap || ws.L_cap < m.nd.L) {
        const int L = std::max(ws.L_cap, m.nd.L);
        if ((rc = ws_grow(ws.x, 4 * (size_t)cap)) || (rc = ws_grow(ws.e, (size_t)L * 2 * cap)) || (rc = ws_grow(ws.O, 4 * (size_t)cap))) return rc;
        ws.cap = cap; ws.L_cap = L;
    }
    if (!ws.frag && (rc = ws_grow(ws.frag, 64 * 512))) return rc;
    if (ws.image_cap < 2 * (size_t)m.n_grid) { if ((rc = ws_grow(ws.image, 2 * (size_t)m.n_grid + 64))) return rc; ws.image_cap = 2 * (size_t)m.n_grid;
        ws.key_epoch = ~0ull; }
    if (ws.rec_cap < n_pix) { const size_t c = std::max<size_t>(n_pix, 2 * ws.rec_cap); if ((rc = ws_grow(ws.rec, 12 * c))) return rc; ws.rec_cap = c; }
    *out = &ws; return MON_OK;
}
static void tile_ws_object_born(int device) { std::lock_guard<std::mutex> l(g_tile_mu); TileWsPair*& slot = g_tile_ws[device];
    if (!slot) slot = new TileWsPair(); ++slot->objects; }
static void tile_ws_object_gone(int device) {
    std::lock_guard<std::mutex> l(g_tile_mu);
    auto it = g_tile_ws.find(device); if (it == g_tile_ws.end() || --it->second->objects > 0) return;
    // the device's last object: a few hundred MB of scratch are returned (nobody can hold ws.mu: users are objects)
    for (TileWs& ws : it->second->side) {
        std::lock_guard<std::mutex> wl(ws.mu);
        for (void* q : { (void*)ws.rec, (void*)ws.counters, (void*)ws.x, (void*)ws.e, (void*)ws.O, (void*)ws.image, (void*)ws.frag }) if (q) hipFree(q);
        ws.rec = nullptr; ws.counters = nullptr; ws.x = nullptr; ws.e = nullptr; ws.O = nullptr; ws.image = nullptr; ws.frag = nullptr;
        ws.rec_cap = 0; ws.cap = 0; ws.L_cap = 0; ws.image_cap = 0; ws.flip = 0; ws.key_params = nullptr; ws.key_epoch = ~0ull;
    }
}
static int tile_ws_objects(int device) { std::lock_guard<std::mutex> l(g_tile_mu); auto it = g_tile_ws.find(device);
    return it == g_tile_ws.end() ? 0 : it->second->objects; }
void tile_ws_weights(Model& m, TileWs& ws, hipStream_t s, const uint16_t* prm, uint64_t epoch) {
    if (ws.key_params == prm && ws.key_epoch == epoch) return;
    launch_build_feat_image(s, m.lf, m.nd, prm, ws.image, nullptr);
    launch_forward_frag_image(s, m.nd, prm, ws.frag);
    ws.key_params = prm; ws.key_epoch = epoch;
}
void tile_points_forward(Model& m, TileWs& ws, hipStream_t s, uint32_t n) {
    launch_encode_feat(s, m.lf, m.nd, ws.image, ws.x, ws.e, ws.cap, n, nullptr, 0u, 0u, 1u);
    launch_tile_points_mlp(s, m.nd, ws.frag, ws.e, ws.cap, n, ws.O);
}
// NeRF_Model::Render's body (:1768-1828) for a whole crop on the tile path: rays + hit compaction, then per chunk of jobs points -> encode -> MLP + composite
// into the device buffers rgb / depth / mask (pixel order); caller holds ws.mu and has called tile_ws_weights
static void tile_render_crop(Model& m, TileWs& ws, hipStream_t s, const ObjectConst& oc, mon_frame_bbox box, const Mat4& pose, int pose_is_Toc,
                             float* rgb, float* depth, float* mask) {
    const uint32_t n_pix = box.w * box.h;
    uint32_t* cnt = ws.counters + 16u * (ws.flip & 1u); uint32_t* next = ws.counters + 16u * ((ws.flip + 1u) & 1u); ++ws.flip;
    launch_render_rays_jobs(s, m.ds->K, oc, box, pose, pose_is_Toc, n_pix, ws.rec, cnt, next, rgb, depth, mask);
    for (uint32_t j0 = 0; j0 < n_pix; j0 += kTileChunkJobs) {          // (the job count lives on the device: chunks past it return at once)
        const uint32_t jc = std::min(kTileChunkJobs, n_pix - j0);
        launch_render_points(s, oc, ws.rec, cnt, j0, jc, ws.x);
        launch_encode_feat(s, m.lf, m.nd, ws.image, ws.x, ws.e, ws.cap, 0u, cnt, j0, jc, 2u * oc.S);
        launch_tile_render(s, m.nd, oc, ws.frag, ws.rec, cnt, j0, jc, ws.x, ws.e, ws.cap, n_pix, rgb, depth, mask);
    }
}
// whether a crop of n_pix rays goes to the tile path (option tile_render: 0 never, 1 from 4096 rays up -- below that the tile copies cost what the gathers cost
// --, 2 always)
static bool tile_render_wanted(const Model& m, size_t n_pix) {
    const long o = options().tile_render;
    return m.backend == 1 && m.tile_ok && m.oc.S == 32u && o != 0 && (o >= 2 || n_pix >= 4096);
}

// ---- training lanes: the per-device scheduler behind "one host thread per object" (nerf_manager.cu:89,256-259).
// Measured (tools/multi_object.py, base.json objects): two objects training concurrently fall into anti-phase on their own -- one gathers (k_fused_train,
// bound by the L2 request path) while the other scatters and updates (LDS atomics, HBM) -- 1.75 G ray-samples/s against 1.38 G for one; with three or more streams in flight
// the dispatcher mixes workgroups of kernels that exclude each other on a CU (k_grid_scatter takes a CU's whole LDS; streams beyond the hardware queues share
// one and block each other) and the aggregate drops to 1.5 G.  So the training work of ALL objects of a device goes through `train_lanes` (2) shared streams: a
// chunk of an object's iterations is enqueued on the lane with the least work in flight (the lane of the object's previous chunk while that is still running:
// stream order then keeps the object's iterations in sequence; a change of lane is ordered by an event).  The device sees two streams of whole training steps
// whatever the number of objects.
// Lanes order work for speed only: no result depends on them.
constexpr int kMaxLanes = 4; constexpr uint32_t kLaneRing = 256;
// a lane's chunk events are only QUERIED (how much work is in flight): without the system-scope fence of a default event (an L2 write-back per chunk)
constexpr unsigned kLaneEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
struct TrainLanes {
    std::mutex mu; std::atomic<int> objects{ 0 };      // live objects of the device
    // per lane: `mu` orders the enqueueing of whole chunks; ev[tail .. head) = completion events of the chunks in flight (head: written by the enqueuer, tail:
    // by whoever picks a lane, under TrainLanes::mu); pending = chunks that have picked the lane and not finished enqueueing
    struct Lane { std::mutex mu; hipStream_t stream = nullptr; hipEvent_t ev[kLaneRing] = {}; std::atomic<uint32_t> head{ 0 }, pending{ 0 },
            tail{ 0 }; } lane[kMaxLanes];
};
static std::mutex g_lanes_mu; static std::map<int, TrainLanes*> g_lanes;
// (the device is current.)  The lane streams are created TOGETHER, with the device's dataset and before any object's own stream: the runtime places a new
// stream on the hardware queue with the fewest users, so two streams created back to back get different queues -- created lazily, with object streams in
// between, both lanes could land on one queue and run strictly one after the other.
static TrainLanes* lanes_get(int device) {
    std::lock_guard<std::mutex> l(g_lanes_mu); TrainLanes*& t = g_lanes[device];
    if (!t) {
        t = new TrainLanes();
        const long want_lanes = options().train_lanes; const int n = want_lanes < kMaxLanes ? (int)want_lanes : kMaxLanes;
        for (int i = 0; i < n; ++i) if (hipStreamCreateWithFlags(&t->lane[i].stream, hipStreamNonBlocking) != hipSuccess) { t->lane[i].stream = nullptr;
            (void)hipGetLastError(); }
    }
    return t;
}
#ifdef MON_VARIANT_STALE_MARK          // (variant build for the regression test: the bug it guards against)
#define MON_INVALIDATE_MARK(m) ((void)0)
#else
#define MON_INVALIDATE_MARK(m) ((m).tail_marked = false)
#endif
// marks the end of what the object has enqueued so far on its current stream (called where an entry point returns with work still in flight: the end of a train
// call)
static void mark_tail(Model& m) {
    // (a DEFAULT event, with its release fence: this one orders the object's kernels across two hardware queues)
    if (!m.switch_event && hipEventCreateWithFlags(&m.switch_event, hipEventDisableTiming) != hipSuccess) { m.switch_event = nullptr; m.tail_marked = false;
        return; }
    m.tail_marked = hipEventRecord(m.switch_event, m.train_stream) == hipSuccess;
}
// moves the object's work to stream `to`: everything it has enqueued so far is ordered before whatever follows on the new stream.  The wait is for the object's
// OWN last work (mark_tail) -- an event recorded now would also stand behind every chunk other objects have queued on the old lane since, and tie the two lanes
// together.
static void switch_stream(Model& m, hipStream_t to);
// Everything that is not a training chunk (renders on the train stream, density grids, meshes, parameter access, box uploads) runs on the object's OWN stream:
// on a lane it would queue -- and its synchronisation would wait -- behind every chunk other objects have enqueued there.
// (work follows: an earlier mark no longer stands for the object's last work)
void model_leave_lane(Model& m) { switch_stream(m, m.own_stream); MON_INVALIDATE_MARK(m); }
static void switch_stream(Model& m, hipStream_t to) {
    if (m.train_stream == to) return;
    if (!m.tail_marked) mark_tail(m);
    if (m.tail_marked) (void)hipStreamWaitEvent(to, m.switch_event, 0);
    else (void)hipStreamSynchronize(m.train_stream);
    m.train_stream = to; m.tail_marked = false;
}
// One chunk of an object's iterations on a lane.  The lane is picked under the device-wide lock (short); the chunk is then ENQUEUED under the lane's own lock
// (a few microseconds per launch), so the chunks of different objects do not interleave within a lane while the host threads of different lanes enqueue side by
// side.
struct LaneChunk {
    Model& m; TrainLanes* tl = nullptr; std::unique_lock<std::mutex> lock; int l = -1;
    explicit LaneChunk(Model& mm, bool enabled) : m(mm) {
        const long want_lanes = options().train_lanes; const int n = want_lanes < kMaxLanes ? (int)want_lanes : kMaxLanes;
        // (up to `n` objects: their own streams ARE the lanes)
        if (!enabled || n <= 0 || !m.lanes || m.lanes->objects.load() <= n) { switch_stream(m, m.own_stream); MON_INVALIDATE_MARK(m); return; }
        tl = m.lanes;
        {   std::lock_guard<std::mutex> pick(tl->mu);
            // previous chunk still in flight: same lane
            if (m.lane >= 0 && m.lane < n && m.lane_event && m.train_stream == tl->lane[m.lane].stream
                    && hipEventQuery(m.lane_event) == hipErrorNotReady) l = m.lane;
            else {
                uint32_t best = ~0u, mine = ~0u;
                for (int i = 0; i < n; ++i) {
                    TrainLanes::Lane& L = tl->lane[i];
                    const uint32_t head = L.head.load(std::memory_order_acquire);
                    uint32_t tail = L.tail.load(std::memory_order_relaxed);
                    while (tail != head && hipEventQuery(L.ev[tail % kLaneRing]) == hipSuccess) ++tail;          // retire finished chunks
                    L.tail.store(tail, std::memory_order_relaxed);
                    const uint32_t load = head - tail + L.pending.load();
                    if (i == m.lane && m.train_stream == L.stream) mine = load;
                    if (load < best) { best = load; l = i; }
                }
                if (mine != ~0u && mine < best + 2u) l = m.lane;                                               // stay unless the other lane is clearly shorter
            }
            // (a hipErrorNotReady would otherwise be reported by the next hipGetLastError)
            (void)hipGetLastError();
            tl->lane[l].pending.fetch_add(1);
        }
        TrainLanes::Lane& L = tl->lane[l];
        lock = std::unique_lock<std::mutex>(L.mu);
        if (!L.stream && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) { L.stream = nullptr; L.pending.fetch_sub(1); tl = nullptr;
            lock.unlock(); switch_stream(m, m.own_stream); MON_INVALIDATE_MARK(m); return; }
        switch_stream(m, L.stream);
        // work follows on this stream: the mark of an earlier call no longer stands for the object's last work
        MON_INVALIDATE_MARK(m);
    }
    ~LaneChunk() {
        if (!tl) return;
        TrainLanes::Lane& L = tl->lane[l];
        const uint32_t head = L.head.load(std::memory_order_relaxed);
        // (a full ring -- 256 chunks in flight on one lane -- goes uncounted)
        if (head - L.tail.load(std::memory_order_relaxed) < kLaneRing) {
            hipEvent_t& e = L.ev[head % kLaneRing];
            if ((e || hipEventCreateWithFlags(&e, kLaneEventFlags) == hipSuccess) && hipEventRecord(e, m.train_stream) == hipSuccess) {
                L.head.store(head + 1u, std::memory_order_release); m.lane_event = e; }
        }
        m.lane = l; L.pending.fetch_sub(1);
    }
};

int dataset_create(int device, int H, int W, float fx, float fy, float cx, float cy, uint32_t max_frames, int use_depth, Dataset** out) {
    int n = 0; int rc = device_count(&n); if (rc) return rc;
    if (device < 0 || device >= n || H <= 0 || W <= 0 || max_frames == 0) { set_error("dataset_create: bad argument"); return MON_ERR_ARG; }
    HIPCHECK(use_device(device));
    Dataset* d = new Dataset();
    d->device = device; d->K = Intrinsics{ fx, fy, cx, cy, H, W }; d->max_frames = max_frames; d->use_depth = use_depth != 0;
    const size_t px = (size_t)H * W;
    const auto alloc = [&]() -> int {
        HIPCHECK(hipMalloc((void**)&d->d_rgba, px * 4 * max_frames));
        // a frame id never uploaded reads as black / instance 0, not as whatever the allocation held
        HIPCHECK(hipMemset(d->d_rgba, 0, px * 4 * max_frames));
        if (d->use_depth) { HIPCHECK(hipMalloc((void**)&d->d_depth, px * 4 * max_frames)); HIPCHECK(hipMemset(d->d_depth, 0, px * 4 * max_frames)); }
        HIPCHECK(hipMalloc((void**)&d->d_poses, 64 * (size_t)max_frames));
        HIPCHECK(hipMemset(d->d_poses, 0, 64 * (size_t)max_frames));
        // hipMemset returns before the fill has run (null stream), and the upload stream is non-blocking: without this wait the fill can land AFTER the first
        // frames' packing kernels and wipe them (seen as frames that read as black / instance 0 now and then -- more valid candidate rays than the oracle has)
        HIPCHECK(hipStreamSynchronize(nullptr));
        return MON_OK;
    };
    if ((rc = alloc())) { dataset_destroy(d); return rc; }          // e.g. out of memory for max_frames images: free what was taken
    d->present.assign(max_frames, 0);
    // frames arrive through a pinned staging buffer and are packed by a kernel on the device's high-priority stream (see dataset_add_frame)
    { InferShared* sh = nullptr; if ((rc = infer_shared_get(device, px, &sh))) { dataset_destroy(d); return rc; } d->upload = sh; }
    (void)lanes_get(device);                                            // the device's training lanes exist before its first object
    // (coherent pinned memory, and the packing kernels read it with system-scope loads: the same host addresses are rewritten for every frame)
    d->stage_bytes = px * 9 + 128;                                      // raw colour (<= 4 B/pixel), instance (1 B), depth (4 B), pose
    if (hipHostMalloc((void**)&d->h_stage, d->stage_bytes, hipHostMallocCoherent) != hipSuccess) {
        set_error("dataset_create: pinned staging allocation failed"); dataset_destroy(d); return MON_ERR_HIP; }
    *out = d; return MON_OK;
}
int dataset_add_frame(Dataset* d, uint32_t id, const uint8_t* rgb, int ch, int is_bgr, const uint8_t* inst, const float* depth, const float* Twc) {
    if (!d || !rgb || !inst || !Twc || (ch != 3 && ch != 4)) { set_error("dataset_add_frame: bad argument"); return MON_ERR_ARG; }
    if (id >= d->max_frames) { set_error("dataset_add_frame: frame id %u >= capacity %u", id, d->max_frames); return MON_ERR_ARG; }
    if (d->use_depth && !depth) { set_error("depth img error: dataset was created with use_depth"); return MON_ERR_ARG; }      // nerf_data.cu:296-300
    HIPCHECK(use_device(d->device));
    const size_t px = (size_t)d->K.H * d->K.W;
    const int ri = is_bgr ? 2 : 0, bi = is_bgr ? 0 : 2;             // cv::COLOR_BGR2RGB, nerf_data.cu:167,286
    // The caller's (pageable) images are copied into pinned memory and packed to 4 B/pixel by a kernel that reads them across PCIe, on the device's
    // high-priority stream: a synchronous hipMemcpy from pageable memory goes through the runtime's blit path at normal priority and, with a dozen objects
    // training, kept the SLAM thread 4 ms per frame (8 ms worst case); the host-side pack loop alone cost 0.4 ms.
    InferShared* sh = static_cast<InferShared*>(d->upload); std::lock_guard<std::mutex> one(sh->mu);
    uint8_t* st_rgb = d->h_stage, *st_inst = st_rgb + px * 4; float* st_depth = reinterpret_cast<float*>(st_inst + ((px + 15) & ~(size_t)15));
    float* st_pose = st_depth + (d->use_depth ? px : 0);
    std::memcpy(st_rgb, rgb, px * (size_t)ch); std::memcpy(st_inst, inst, px); std::memcpy(st_pose, Twc, 64);
    launch_pack_frame(sh->stream, st_rgb, ch, ri, bi, st_inst, d->d_rgba + px * id, (uint32_t)px);
    if (d->use_depth) { std::memcpy(st_depth, depth, px * 4); launch_copy_from_host(sh->stream, st_depth, d->d_depth + px * id, (uint32_t)px); }
    launch_copy_from_host(sh->stream, st_pose, d->d_poses + 16 * (size_t)id, 16u);
    HIPCHECK(hipStreamSynchronize(sh->stream)); HIPCHECK(hipGetLastError());
    if (id + 1 > d->n_frames) d->n_frames = id + 1;                 // mFrameDataNum, nerf_data.cu:338
    d->present[id] = 1;
    return MON_OK;
}
// UpdateDataGPU (nerf_data.cu:341-353): overwrite the poses of n consecutive frames
int dataset_update_poses(Dataset* d, uint32_t first, uint32_t n, const float* Twc16s) {
    if (!d || !Twc16s || first + n > d->max_frames) {
        set_error("update_poses: frames %u..%u outside the dataset (%u frames)", first, first + n, d ? d->max_frames : 0u); return MON_ERR_ARG; }
    HIPCHECK(use_device(d->device));
    HIPCHECK(hipMemcpy(d->d_poses + 16 * (size_t)first, Twc16s, 64 * (size_t)n, hipMemcpyHostToDevice));
    return MON_OK;
}
int dataset_destroy(Dataset* d) {
    if (!d) return MON_OK;
    use_device(d->device);
    if (d->d_rgba) hipFree(d->d_rgba);
    if (d->d_depth) hipFree(d->d_depth);
    if (d->d_poses) hipFree(d->d_poses);
    if (d->h_stage) hipHostFree(d->h_stage);
    delete d; return MON_OK;
}

// ------------------------------------------------------------------ model
// hipStreamCreate costs ~8 ms (a hardware queue is set up), and CreateNeRF runs on the SLAM thread: streams of destroyed objects are
// kept for the next object of that device, and a manager can reserve some ahead of time (mon_online_dataset_init).
static std::mutex g_stream_mu; static std::map<int, std::vector<hipStream_t>> g_stream_pool;
static int stream_acquire(int device, hipStream_t* out) {
    { std::lock_guard<std::mutex> l(g_stream_mu); auto& v = g_stream_pool[device]; if (!v.empty()) { *out = v.back(); v.pop_back(); return MON_OK; } }
    HIPCHECK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));       // mpTrainStream, nerf_model.cu:1268
    return MON_OK;
}
static void stream_release(int device, hipStream_t s) { std::lock_guard<std::mutex> l(g_stream_mu); g_stream_pool[device].push_back(s); }
int stream_pool_reserve(int device, int n) {
    HIPCHECK(use_device(device));
    std::vector<hipStream_t> fresh;
    { std::lock_guard<std::mutex> l(g_stream_mu); n -= (int)g_stream_pool[device].size(); }
    int rc = MON_OK;
    for (int i = 0; i < n; ++i) {
        hipStream_t s; const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e)); rc = MON_ERR_HIP; break; }
        fresh.push_back(s);
    }
    std::lock_guard<std::mutex> l(g_stream_mu); for (hipStream_t s : fresh) g_stream_pool[device].push_back(s);      // what was created is kept either way
    return rc;
}

// ---- inference side of a model (the reference's second stream, nerf_model.cu:1268-1269).  The training thread PUBLISHES the inference weights
// at the end of every train call / online slice: a device-to-device copy into one of two snapshot buffers, ordered on the train stream, with an
// event.  A viewer thread renders from the latest published snapshot on the inference stream (created with the highest priority) and in a
// workspace of its own: it takes no model mutex, never touches the train stream, and its kernels do not queue behind training slices.
struct InferState {
    InferShared* shared = nullptr;                                      // the device's inference stream (highest priority) and pinned result buffer
    uint16_t* snap[2] = { nullptr, nullptr }; hipEvent_t ready[2] = { nullptr, nullptr }; uint32_t step_of[2] = { 0, 0 }; bool written[2] = { false, false };
    int latest = -1, readers[2] = { 0, 0 }; std::mutex mu;              // which snapshot is current, who is reading which
    // weights stamp of each snapshot (next_weights_epoch at publication: the tile render's image key)
    uint64_t epoch_of[2] = { 0, 0 };
    std::atomic<bool> wanted{ false }; std::chrono::steady_clock::time_point last_pub{};      // a viewer asked since the last publication; when that was
    BatchPtrs rb{}; float *out_all = nullptr, *out_rgb = nullptr, *out_depth = nullptr, *out_mask = nullptr; size_t out_cap = 0; uint16_t* frag = nullptr;
    std::vector<void*> grown;                                           // superseded output buffers, freed with the object
};

template <class T> static int dev_alloc(Model& m, T*& p, size_t n, bool zero = true) {
    void* q = nullptr; const size_t bytes = (n ? n : 1) * sizeof(T);
    HIPCHECK(hipMalloc(&q, bytes));
    if (zero) HIPCHECK(hipMemset(q, 0, bytes));
    m.allocs.push_back(q); p = (T*)q; return MON_OK;
}

static constexpr uint32_t kRenderChunkRays = 16384;   // rays per render pass (x 2S samples)

// fp32 master weights from the host into the model (the arrays, or the chunk records through a staging buffer) + the fp16 working copy h(master)
static int upload_master(Model& m, const float* master) {
    const size_t n = m.n_params;
    if (!m.P.rec) {
        HIPCHECK(hipMemcpy(m.P.master, master, n * 4, hipMemcpyHostToDevice));
        launch_master_to_half(m.train_stream, m.P.master, m.P.half, (uint32_t)n);        // h(master), same rounding as every later update
        return MON_OK;
    }
    float* tmp = nullptr; HIPCHECK(hipMalloc((void**)&tmp, n * 4));
    hipError_t e = hipMemcpy(tmp, master, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_state_pack_master(m.train_stream, tmp, m.P.rec, (uint32_t)n);
        launch_master_to_half(m.train_stream, tmp, m.P.half, (uint32_t)n); e = hipStreamSynchronize(m.train_stream); }
    (void)hipFree(tmp); HIPCHECK(e); return MON_OK;
}
MeshState* mesh_state_create(int device);
static int model_init(Model& m, Dataset* ds, const mon_config& cfg, int class_id, const float* Tow, const float* amin, const float* amax) {
    m.ds = ds; m.cfg = cfg; m.device = ds->device;
    int rc = level_table_build(cfg, m.lt, m.nd, m.n_grid);
    if (rc) return rc;
    m.n_params = m.nd.n_mlp + m.n_grid;
    level_fast_build(m.lt, m.nd, m.lf);
    {   // fixed-point unit of the exact LDS gradient accumulation: 2^-24 (every fp16 value is a multiple of it) up to the reference's loss scale of
        // 128, coarser by the next power of two of loss_scale / 128 beyond it, so that the int32 range always spans un-scaled gradient sums below 1.0
        int shift = 0; while (shift < 23 && 128.0f * (float)(1u << shift) < cfg.loss_scale) ++shift;
        m.lf.fix_scale = 16777216.0f / (float)(1u << shift); m.lf.fix_clamp = 100.0f * (float)(1u << shift);
    }
    HIPCHECK(use_device(m.device));
    std::memcpy(m.oc.Tow.m, Tow, 64);
    for (int a = 0; a < 3; ++a) { m.oc.aabb.mn[a] = amin[a]; m.oc.aabb.mx[a] = amax[a]; }
    m.oc.instance_id = (uint32_t)(uint8_t)class_id;                 // nerf.cu:75,158
    m.oc.R = (uint32_t)cfg.rays_per_batch; m.oc.S = (uint32_t)cfg.n_samples; m.oc.use_depth = cfg.use_depth && ds->use_depth;
    m.oc.sample_seed = cfg.sample_seed; m.oc.loss_scale = cfg.loss_scale;
    m.n_bins = kDefaultScatterBins;
    m.opt = OptimConst{ cfg.beta1, cfg.beta2, cfg.epsilon, cfg.l2_reg, cfg.ema_decay, cfg.loss_scale, cfg.decay_base, std::log2(cfg.beta1),
            std::log2(cfg.beta2), std::log2(cfg.ema_decay), cfg.decay_start, cfg.decay_interval, m.nd.n_mlp, m.n_params };
    { const int rcs = stream_acquire(m.device, &m.own_stream); if (rcs) return rcs; }
    m.train_stream = m.own_stream; m.lanes = lanes_get(m.device); m.lanes->objects.fetch_add(1);
    // ---- parameters (ResetNetwork :1286-1342; Trainer init)
    const size_t n = m.n_params;
    // per-parameter step counters in 16 bits, saturating, where that is EXACT: 1 - beta^t == 1.0f (beta^t < 2^-25) for every t >= 65535 and both betas
    // (variant build MON_VARIANT_STEPS32: always 32 bits).  The device evaluates 1 - exp2f(t * log2(beta)) in fp32; the host test runs in double, so it keeps a
    // margin of four binades (beta^65535 < 2^-29, i.e. beta <= 0.99969) instead of sitting on the rounding boundary.
    const double kSteps16Bound = std::ldexp(1.0, -29);
    const bool steps16 = kSteps16 && std::pow((double)cfg.beta1, 65535.0) < kSteps16Bound && std::pow((double)cfg.beta2, 65535.0) < kSteps16Bound;
    // lazy EMA: only where the optimizer is not the dense variant anyway and the table is large (> 8 M parameters)
    m.lazy_ema = m.n_grid > (8u << 20);
    // large tables (lazy EMA, 16-bit step counters): the optimizer state as one 128-byte record per chunk (ParamPtrs::rec) instead of four arrays
    const bool records = m.lazy_ema && steps16 && kStateRecords && (n & 7u) == 0u;
    if (records) { if ((rc = dev_alloc(m, m.P.rec, 4 * n))) return rc; }
    else if ((rc = dev_alloc(m, m.P.master, n, false)) || (rc = dev_alloc(m, m.P.m1, n)) || (rc = dev_alloc(m, m.P.m2, n)) ||
             (rc = steps16 ? dev_alloc(m, m.P.steps16, n + 8) : dev_alloc(m, m.P.steps, n))) return rc;
    if ((rc = dev_alloc(m, m.P.half, n, false)) || (rc = dev_alloc(m, m.P.ema, n)) || (rc = records ? MON_OK : dev_alloc(m, m.d_ema_step, n / 8 + 1)) ||      // (chunk records keep the EMA step in their pad word: no array, ParamPtrs::lazy says "lazy")
        
        (rc = dev_alloc(m, m.P.gmlp, m.nd.n_mlp)) || (rc = dev_alloc(m, m.P.ggrid, m.n_grid))) return rc;
    {
        std::vector<float> master; init_params_host(cfg, m.nd, m.n_params, master);
        const int urc = upload_master(m, master.data()); if (urc) return urc;
    }
    // ---- workspace (AllocateBatchWorkspace :1344-1427), sized for max(train batch, render chunk)
    const uint32_t R = m.oc.R, S = m.oc.S;
    m.ws_rays = R > kRenderChunkRays ? R : kRenderChunkRays;
    // the layer-at-a-time buffers (pts, tdist, E, O) serve a render pass of the unfused backend; an object whose inference runs on the fused kernels / level
    // tiles only needs them at the training batch's size (64 + 12 + 8 + 4 MB less per base.json object: a render of the unfused backend then takes passes of
    // ws_samples / 2S rays)
    const bool fused_inference = fused_supported(m.nd, S, R) && options().backend != 0;
    const uint32_t Btrain = R * S, Brender = fused_inference ? Btrain : kRenderChunkRays * 2 * S;
    m.ws_samples = Btrain > Brender ? Btrain : Brender;
    BatchPtrs& B = m.B;
    if ((rc = dev_alloc(m, B.cand_o, 3 * (size_t)R)) || (rc = dev_alloc(m, B.cand_d, 3 * (size_t)R)) || (rc = dev_alloc(m, B.cand_dn, R)) ||
        (rc = dev_alloc(m, B.cand_t0, R)) || (rc = dev_alloc(m, B.cand_t1, R)) || (rc = dev_alloc(m, B.cand_depth, R)) ||
        (rc = dev_alloc(m, B.cand_rgba, R)) || (rc = dev_alloc(m, B.mask, (R + 63) / 64 + 64)) ||
        (rc = dev_alloc(m, B.ray_o, 3 * (size_t)m.ws_rays)) || (rc = dev_alloc(m, B.ray_d, 3 * (size_t)m.ws_rays))
                || (rc = dev_alloc(m, B.ray_dn, m.ws_rays)) ||
        (rc = dev_alloc(m, B.ray_t0, m.ws_rays)) || (rc = dev_alloc(m, B.ray_t1, m.ws_rays)) || (rc = dev_alloc(m, B.target, 3 * (size_t)R)) ||
        (rc = dev_alloc(m, B.target_depth, R)) || (rc = dev_alloc(m, B.bgcol, 3 * (size_t)R)) || (rc = dev_alloc(m, B.ray_flag, m.ws_rays)) ||
        (rc = dev_alloc(m, B.pts, 3 * (size_t)m.ws_samples)) || (rc = dev_alloc(m, B.tdist, m.ws_samples)) ||
        (rc = dev_alloc(m, B.E, (size_t)m.ws_samples * m.nd.Epad)) || (rc = dev_alloc(m, B.O, (size_t)m.ws_samples * kOut)) ||
        (rc = dev_alloc(m, B.Hid, (size_t)Btrain * m.nd.W * m.nd.NH)) || (rc = dev_alloc(m, B.dO, (size_t)Btrain * kOut)) ||
        (rc = dev_alloc(m, B.dHid, (size_t)Btrain * m.nd.W * m.nd.NH)) || (rc = dev_alloc(m, B.dE, (size_t)Btrain * m.nd.Epad)) ||
        (rc = dev_alloc(m, B.rgb_ray, 3 * (size_t)R)) || (rc = dev_alloc(m, B.depth_ray, R)) || (rc = dev_alloc(m, B.mask_ray, R))
                || (rc = dev_alloc(m, B.loss_ray, R)) ||
        (rc = dev_alloc(m, m.d_state, 2)) || (rc = dev_alloc(m, m.d_dw_partials, (size_t)(fused_partial_cols(m.nd) + 64) * kMaxFusedGrid)) ||
        (rc = dev_alloc(m, m.d_out_all, 5 * (size_t)kRenderChunkRays))) return rc;
    m.out_cap = kRenderChunkRays;
    if (rng_stream_mode(cfg.rng_flags)) {          // "same inputs" mode: the reference's XORWOW stream (xorwow.h) instead of the counter RNG
        m.xw_lanes = rng_xorwow_lanes(cfg.rng_flags); m.xw_flavour = rng_stream_mode(cfg.rng_flags) == 2 ? kXorwowRocrand : kXorwowCurand;
        std::vector<XorwowState> st; xorwow_lane_states(0ull /* the generator's default seed: nerf_model.cu never sets one */, m.xw_flavour, m.xw_lanes, st);
        XorwowState *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;
        if ((rc = dev_alloc(m, d_a, m.xw_lanes, false)) || (rc = dev_alloc(m, d_b, m.xw_lanes, false)) || (rc = dev_alloc(m, d_c, m.xw_lanes, false)) ||
            (rc = dev_alloc(m, m.d_xw, 2 * (size_t)(5 + S) * R))) return rc;
        HIPCHECK(hipMemcpy(d_a, st.data(), sizeof(XorwowState) * m.xw_lanes, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(d_c, st.data(), sizeof(XorwowState) * m.xw_lanes, hipMemcpyHostToDevice));
        m.d_xw_states = d_a; m.d_xw_render_states = d_b; m.d_xw_render_init = d_c;
        m.oc.xw[0] = m.d_xw; m.oc.xw[1] = m.d_xw + (size_t)(5 + S) * R;
    }
    if (fused_supported(m.nd, S, m.oc.R)) {
        if ((rc = dev_alloc(m, m.d_frag_train, 64 * 512)) || (rc = dev_alloc(m, m.d_frag_render, 64 * 512))) return rc;       // <= 30 fragments of 512 halves
        m.lds_mask = scatter_plan(m.lt, m.nd, m.scatter);
        // a partial table spans the entries up to the end of the LAST LDS-scattered level (the plan covers a prefix of the levels: sizes grow with the level);
        // sized by the whole table it was 16 x 211 MB = 3.4 GB of a T = 2^22 object for the 37 k entries of its two small levels
        {
            int last = -1; for (int l = 0; l < m.nd.L; ++l) if ((m.lds_mask >> l) & 1u) last = l;
            m.part_halves = last < 0 ? 0u : ((2u * m.lt.offset[last + 1] + 15u) & ~15u);
        }
        if (m.lds_mask && ((rc = dev_alloc(m, m.d_de_soa, (size_t)m.nd.L * Btrain * 2)) || (rc = dev_alloc(m, m.d_x_soa, 4 * (size_t)Btrain)) ||
                           (rc = dev_alloc(m, m.d_gpart, (size_t)m.scatter.max_P * m.part_halves)))) return rc;
        // Levels beyond the LDS plan (more than 2^18 entries): binned exact scatter while many samples carry a gradient (kernels_bigscatter.hip).
        // MON_BIG_SWITCH = gradient-carrying samples below which the global-atomic path takes over (0: atomics always).
        const size_t big_bytes = m.lds_mask ? big_scatter_workspace_bytes(m.lt, m.nd, m.lds_mask, Btrain) : 0;
        const uint32_t big_switch = (uint32_t)options().big_switch;
        if (big_bytes && big_switch) { if ((rc = dev_alloc(m, m.d_big_ws, big_bytes))) return rc; m.big_switch = big_switch; }
        // level-tile encode (kernels_encode.hip): every level must fit two LDS tiles and go through the LDS scatter (option lds_encode = 0: gathers inside
        // k_fused_train)
        // Batch size: a workgroup's two tile copies, four barriers and the launch cost the same whatever it walks -- measured (tools/kernel_times.py, both
        // chains, same box):
        // R = 1024 (C1) 48.9 vs 41.3 us per step for the gather chain, R = 2048 75.1 vs 75.4, R = 4096 99 vs 107, R = 8192 166 vs 172.  Option lds_encode = 1
        // (default) takes the tile chain from 3072 rays (98 304 samples) up, 2 always (tests), 0 never.
        const bool tiles_pay = options().lds_encode >= 2 || Btrain >= 98304u;
        if (options().lds_encode && tiles_pay && m.lds_mask == ((1u << m.nd.L) - 1u) && encode_tiles_supported(m.lt, m.nd)) {
            if ((rc = dev_alloc(m, B.ray_rec, 12 * (size_t)R))) return rc;
            m.B_alt = B;                             // (cand_* / mask replaced below, after the workspace pointers are final)
            if ((rc = dev_alloc(m, m.B_alt.cand_o, 3 * (size_t)R)) || (rc = dev_alloc(m, m.B_alt.cand_d, 3 * (size_t)R))
                    || (rc = dev_alloc(m, m.B_alt.cand_dn, R)) ||
                (rc = dev_alloc(m, m.B_alt.cand_t0, R)) || (rc = dev_alloc(m, m.B_alt.cand_t1, R)) || (rc = dev_alloc(m, m.B_alt.cand_depth, R)) ||
                (rc = dev_alloc(m, m.B_alt.cand_rgba, R)) || (rc = dev_alloc(m, m.B_alt.mask, (R + 63) / 64 + 64))) return rc;
            if ((rc = dev_alloc(m, m.d_x_all, 4 * (size_t)Btrain)) || (rc = dev_alloc(m, m.d_e_soa, (size_t)m.nd.L * Btrain * 2))
                    || (rc = dev_alloc(m, m.d_half_tiles, (size_t)m.n_grid + 64))) return rc;
            encode_tiles_setup_device();
        }
        // chunk flags for the lazy optimizer (tables above 8 M parameters with levels outside the LDS plan); MON_TOUCHED_FLAGS=0: scan the gradient table
        const bool flags_on = kTouchedFlags;
        if (flags_on && m.lazy_ema && m.lds_mask && m.lds_mask != ((m.nd.L >= 32) ? 0xffffffffu : ((1u << m.nd.L) - 1u))
                && (rc = dev_alloc(m, m.d_touched, (m.n_params >> 3) + 16))) return rc;
    }
    if (!fused_supported(m.nd, S, m.oc.R) && (Btrain & 31u) == 0u) {
        // shapes outside the fused kernels: the T-layout workspace of the MFMA layer kernels (kernels_layers.hip); without it kernels_net.hip's one-sample-per-
        // thread kernels run
        if ((rc = dev_alloc(m, m.d_layers_T, layers_workspace_halves(m.nd, Btrain), false))) return rc;
    }
    if (fused_supported(m.nd, S, m.oc.R)) { }
    else if (S == 32u && !m.lazy_ema) {
        // Shapes the fused kernels do not take (16 neurons, 2 x 128, three / four hidden layers): the layer-at-a-time kernels, but their grid backward through
        // k_grid_scatter when the plan covers every level (tables up to 2^18 entries per level) -- see k_rows_to_bins
        ScatterLevels plan{}; const uint32_t mask = scatter_plan(m.lt, m.nd, plan);
        if (mask == ((1u << m.nd.L) - 1u) && (R % kDefaultScatterBins) == 0u) {
            m.scatter = plan; m.part_halves = (2u * m.lt.offset[m.nd.L] + 15u) & ~15u;
            if ((rc = dev_alloc(m, m.d_de_soa, (size_t)m.nd.L * Btrain * 2)) || (rc = dev_alloc(m, m.d_x_soa, 4 * (size_t)Btrain)) ||
                (rc = dev_alloc(m, m.d_gpart, (size_t)m.scatter.max_P * m.part_halves))) return rc;
            m.hybrid_scatter = true;
            // ... and their forward encode from LDS level tiles like the fused chain's (k_encode_tiles; same batch-size rule, option lds_encode)
            const bool tiles_pay_b0 = options().lds_encode >= 2 || Btrain >= 98304u;
            if (m.d_layers_T && options().lds_encode && tiles_pay_b0 && encode_tiles_supported(m.lt, m.nd)) {
                if ((rc = dev_alloc(m, m.d_x_all, 4 * (size_t)Btrain)) || (rc = dev_alloc(m, m.d_e_soa, (size_t)m.nd.L * Btrain * 2))
                        || (rc = dev_alloc(m, m.d_half_tiles, (size_t)m.n_grid + 64))) return rc;
                encode_tiles_setup_device();
            }
        }
    }
    if (cfg.occupancy_skip && fused_supported(m.nd, S, m.oc.R)) {
        constexpr size_t words = (size_t)kOccRes * kOccRes * kOccRes / 32;
        if ((rc = dev_alloc(m, m.d_occ, words, false)) || (rc = dev_alloc(m, m.d_occ_tmp, words, false))
                || (rc = dev_alloc(m, m.d_frag_occ, 64 * 512))) return rc;
        HIPCHECK(hipMemset(m.d_occ, 0xff, words * 4));                       // warm-up: every cell counts as occupied
        // a cell is empty when one sample interval through it would be transparent: alpha = 1 - exp(-sigma * dt) < 1e-3 with dt = box diagonal / samples
        float diag2 = 0.f; for (int a = 0; a < 3; ++a) diag2 += (amax[a] - amin[a]) * (amax[a] - amin[a]);
        const float dt = std::sqrt(diag2) / (float)S;
        m.occ_raw_threshold = std::log(1e-3f / std::max(dt, 1e-6f));
        // the level-tile chain with the grid in use: live-sample lists for k_encode_tiles (LiveArgs, model.h) -- a position block's 256 samples must lie in
        // one of the encode's sample partitions
        // (a list holds at most ceil(blocks / n_parts) * 256 entries: it must fit the partition's spw slots)
        const uint32_t spw_l = encode_tiles_spw(Btrain), parts_l = (Btrain + spw_l - 1u) / spw_l, blocks_l = (Btrain + 255u) / 256u;
        if (m.d_e_soa && Btrain % 256u == 0u && parts_l <= kLiveMaxParts && ((blocks_l + parts_l - 1u) / parts_l) * 256u <= spw_l) {
            if ((rc = dev_alloc(m, m.d_live_idx, Btrain)) || (rc = dev_alloc(m, m.d_live_cnt, 2u * kLiveMaxParts * kLiveCntStride))) return rc;
        }
    }
    m.boxes_cap = 1024;
    if ((rc = dev_alloc(m, m.d_boxes, m.boxes_cap))) return rc;
    B.boxes = m.d_boxes; m.B_alt.boxes = m.d_boxes;
    m.h_state = DevState{}; m.h_state.lr = cfg.learning_rate;
    m.h_state.ema_deb_old = 0.0f; m.h_state.ema_deb_new = 1.0f / (1.0f - (float)std::pow((double)cfg.ema_decay, 1.0));   // step 1
    m.d_state_next = m.d_state + 1;                          // two states: iteration i runs on one, k_optimizer(i) writes the other for iteration i + 1
    HIPCHECK(hipMemcpy(m.d_state, &m.h_state, sizeof(DevState), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(m.d_state_next, &m.h_state, sizeof(DevState), hipMemcpyHostToDevice));
    HIPCHECK(hipHostMalloc((void**)&m.h_state_pinned, sizeof(DevState), hipHostMallocDefault));
    m.backend = fused_supported(m.nd, S, m.oc.R) ? 1 : 0;
    if (options().backend >= 0) m.backend = options().backend ? (fused_supported(m.nd, S, m.oc.R) ? 1 : 0) : 0;
    m.mesh = mesh_state_create(m.device);
    m.tile_ok = fused_supported(m.nd, S, m.oc.R) && !m.lazy_ema && tile_render_supported(m.lt, m.nd);
    m.weights_epoch = next_weights_epoch();
    if (m.tile_ok) { tile_ws_object_born(m.device); m.tile_counted = true; }
    // (the XORWOW mode renders on the train stream: one generator per Render, like the reference; tables above 8 M parameters keep their EMA lazily and would
    // cost 2 x 200 MB of snapshots: they render on the train stream)
    if (m.backend == 1 && !m.lazy_ema && !m.d_xw) {
        InferState* is = new InferState(); m.infer = is;
        // (a whole frame fits: no growth in front of a viewer)
        if ((rc = infer_shared_get(m.device, (size_t)ds->K.W * (size_t)ds->K.H, &is->shared))) return rc;
        for (int k = 0; k < 2; ++k) { if ((rc = dev_alloc(m, is->snap[k], n, false))) return rc;
            HIPCHECK(hipEventCreateWithFlags(&is->ready[k], hipEventDisableTiming)); }
        is->rb = m.B;
        if ((rc = dev_alloc(m, is->rb.ray_o, 3 * (size_t)kRenderChunkRays)) || (rc = dev_alloc(m, is->rb.ray_d, 3 * (size_t)kRenderChunkRays))
                || (rc = dev_alloc(m, is->rb.ray_dn, kRenderChunkRays)) ||
            (rc = dev_alloc(m, is->rb.ray_t0, kRenderChunkRays)) || (rc = dev_alloc(m, is->rb.ray_t1, kRenderChunkRays))
                    || (rc = dev_alloc(m, is->rb.ray_flag, kRenderChunkRays)) ||
            (rc = dev_alloc(m, is->out_all, 5 * (size_t)kRenderChunkRays)) || (rc = dev_alloc(m, is->frag, 64 * 512))) return rc;
        is->out_cap = kRenderChunkRays;
    }
    // what this object's creation enqueued: the fills of its allocations (null stream) and its own stream's kernels.  NOT the device: with other objects'
    // training threads running, a device-wide wait stands behind everything they have queued -- a whole Train_Step of 500 iterations each in the offline
    // manager (CreateNeRF of the 8th object of a job took 125-140 ms, 9-23 ms with an idle device; eight objects 0.71-0.80 s -> 0.28-0.32 s of the caller's
    // time, same PSNRs, same job wall: tools/offline_job.py, A/B/A/B on one box)
    HIPCHECK(hipStreamSynchronize(nullptr)); HIPCHECK(hipStreamSynchronize(m.train_stream));
    return MON_OK;
}

// Owner thread, after a train call's state read-back: copy the inference weights (EMA once a step has been taken) into the snapshot buffer no reader
// holds and make it the current one.  ~4 us of device-to-device copy at base.json size, ordered on the train stream.
// A publication costs the owner thread ~15 us of host time (copy + event), which matters when the online manager trains in slices of a few
// iterations: unless forced (long train calls, set_params), it happens when a viewer has asked since the last one or 10 ms have passed.
static int publish_snapshot(Model& m, bool force = true) {
    InferState* is = m.infer; if (!is) return MON_OK;
    const auto now = std::chrono::steady_clock::now();
    if (!force && is->latest >= 0 && !is->wanted.load() && now - is->last_pub < std::chrono::milliseconds(10)) return MON_OK;
    int w;
    {   std::lock_guard<std::mutex> l(is->mu); w = is->latest == 0 ? 1 : 0;
        // a render still reads the older buffer: keep the current snapshot this round (the viewer's request stays standing)
        if (is->readers[w] > 0) return MON_OK;
        // from here until the new copy's event is recorded the buffer is not a valid fall-back for a render: its event still shows the PREVIOUS copy as
        // complete
        // (two publications back to back, the first copy still queued behind other objects' chunks: a render fell back to this buffer while it was rewritten)
        is->written[w] = false; }
    const uint16_t* src = (m.h_state.step > 0) ? m.P.ema : m.P.half;
    launch_copy_params(m.train_stream, src, is->snap[w], m.n_params);
    HIPCHECK(hipEventRecord(is->ready[w], m.train_stream));
    { std::lock_guard<std::mutex> l(is->mu); is->step_of[w] = m.h_state.step; is->epoch_of[w] = next_weights_epoch(); is->written[w] = true; is->latest = w; }
    is->wanted.store(false); is->last_pub = now;          // (only now: a publication skipped above must not discard the viewer's request)
    return MON_OK;
}

int model_destroy(Model* mp);
int model_create(Dataset* ds, const mon_config& cfg, int class_id, const float* Tow, const float* amin, const float* amax, Model** out) {
    if (!ds || !Tow || !amin || !amax) { set_error("object_create: bad argument"); return MON_ERR_ARG; }
    if (cfg.rays_per_batch < 64 || (cfg.rays_per_batch % 64) != 0 || cfg.n_samples < 1 || cfg.n_samples > 64) {
        set_error("rays_per_batch must be a multiple of 64, n_samples 1..64"); return MON_ERR_ARG; }
    if ((cfg.rng_flags & 3u) == 3u || (cfg.rng_flags & ~0xffff0013u) != 0u || (cfg.rng_flags >> 16) > 1024u) {
        set_error("rng_flags: bits 0-1 = 0 (counter RNG) | 1 (XORWOW, cuRAND flavour) | 2 (XORWOW, rocRAND flavour), bit 4 = tcnn init order, "
                  "bits 16-31 = XORWOW lanes / 1024 (at most 1024)");
        return MON_ERR_ARG;
    }
    if (!(cfg.loss_scale > 0.f) || !(cfg.loss_scale <= 65536.f)) { set_error("loss_scale must be in (0, 65536] (fp16 gradients; the reference uses 128)");
        return MON_ERR_ARG; }
    Model* mp = new Model();
    const int rc = model_init(*mp, ds, cfg, class_id, Tow, amin, amax);
    if (rc) { model_destroy(mp); return rc; }          // a failed allocation half-way must not leak what came before it
    *out = mp; return MON_OK;
}

static void drop_graph(Model& m) { if (m.graph_exec) { hipGraphExecDestroy(m.graph_exec); m.graph_exec = nullptr; m.graph_backend = -1; } }

void model_mesh_free(Model& m);
int model_destroy(Model* mp) {
    if (!mp) return MON_OK;
    Model& m = *mp; use_device(m.device);
    model_mesh_free(m);
    if (m.own_stream) model_leave_lane(m);
    if (m.train_stream) hipStreamSynchronize(m.train_stream);
    if (m.infer) {
        // (the stream and the pinned buffer stay with the device)
        InferState* is = m.infer; if (is->shared) { std::lock_guard<std::mutex> l(is->shared->mu); hipStreamSynchronize(is->shared->stream); }
        for (int k = 0; k < 2; ++k) if (is->ready[k]) hipEventDestroy(is->ready[k]);
        for (void* p : is->grown) hipFree(p);
        delete is; m.infer = nullptr;
    }
    drop_graph(m);
    if (m.tile_counted) { tile_ws_object_gone(m.device); m.tile_counted = false; }
    for (auto& e : m.ev_pool) hipEventDestroy(e);
    for (void* p : m.allocs) hipFree(p);
    if (m.h_state_pinned) hipHostFree(m.h_state_pinned);
    if (m.h_out) (void)hipHostFree(m.h_out);
    if (m.lanes) m.lanes->objects.fetch_sub(1);
    if (m.switch_event) hipEventDestroy(m.switch_event);
    if (m.sync_event) hipEventDestroy(m.sync_event);
    if (m.own_stream) { hipStreamSynchronize(m.own_stream); stream_release(m.device, m.own_stream); }        // idle: the next object of this device takes it
    delete mp; return MON_OK;
}

int model_add_boxes(Model& m, const mon_frame_bbox* boxes, size_t n) {
    if (!boxes || n == 0) { set_error("add_boxes: empty"); return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device));
    for (size_t i = 0; i < n; ++i) {
        const mon_frame_bbox& b = boxes[i];
        if (b.FrameId >= m.ds->max_frames || b.w == 0 || b.h == 0 || b.x + b.w > (uint32_t)m.ds->K.W || b.y + b.h > (uint32_t)m.ds->K.H) {
            set_error("add_boxes: box %zu (frame %u, x %u y %u h %u w %u) outside the %dx%d image / dataset capacity", i, b.FrameId, b.x, b.y, b.h, b.w,
                    m.ds->K.W, m.ds->K.H);
            return MON_ERR_ARG;
        }
        // the reference's callers always hand the frame over first (LocalMapping.cc:1175 before :1242); rays of an absent frame would train on nothing
        if (!m.ds->present[b.FrameId]) {
            set_error("add_boxes: box %zu names frame %u, which has not been added to the dataset", i, b.FrameId);
            return MON_ERR_STATE;
        }
    }
    model_leave_lane(m); HIPCHECK(hipStreamSynchronize(m.train_stream));
    if (m.n_boxes + n > m.boxes_cap) {
        uint32_t cap = m.boxes_cap; while (cap < m.n_boxes + n) cap *= 2;
        mon_frame_bbox* nb = nullptr; int rc = dev_alloc(m, nb, cap); if (rc) return rc;
        HIPCHECK(hipMemcpy(nb, m.d_boxes, sizeof(mon_frame_bbox) * m.n_boxes, hipMemcpyDeviceToDevice));
        m.d_boxes = nb; m.B.boxes = nb; m.B_alt.boxes = nb; m.boxes_cap = cap; drop_graph(m);       // old buffer stays in allocs until destroy
    }
    HIPCHECK(hipMemcpy(m.d_boxes + m.n_boxes, boxes, sizeof(mon_frame_bbox) * n, hipMemcpyHostToDevice));   // nerf_model.cu:1625
    m.n_boxes += (uint32_t)n;
    HIPCHECK(hipMemcpy(&m.d_state->n_boxes, &m.n_boxes, 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(&m.d_state_next->n_boxes, &m.n_boxes, 4, hipMemcpyHostToDevice));
    // (the copies above ran on the null stream, which the object's non-blocking streams do not wait for: a grown box list's zero-fill and device-to-device copy
    // are done before the next batch reads it)
    HIPCHECK(hipStreamSynchronize(nullptr));
    m.next_ready = false;                                   // candidates pre-generated for the next iteration used the old box list
    return MON_OK;
}

// ---- profiling helpers: HIP events on the train stream around each kernel class
static hipEvent_t get_event(Model& m) {
    if (!m.ev_pool.empty()) { hipEvent_t e = m.ev_pool.back(); m.ev_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
// roctx ranges per phase (option "roctx" = 1): the phases of an iteration show up by name in a rocprofv3 --marker-trace of the host side.  The
// library is looked up at run time (librocprofiler-sdk-roctx.so, then libroctx64.so) -- nothing links against it, and without the option nothing is loaded.
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        for (const char* lib : { "librocprofiler-sdk-roctx.so", "libroctx64.so" }) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL); if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA")); pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) break;
            push = nullptr; pop = nullptr;
        }
    }
};
static Roctx* roctx() { if (!options().roctx) return nullptr; static Roctx r; return r.push ? &r : nullptr; }
static const char* const kPhaseName[MON_K_COUNT] = { "mon.batch (GenerateBatch)", "mon.fwd_bwd (k_fused_train)", "mon.optimizer (k_optimizer)", "mon.render",
        "mon.scatter (k_grid_scatter)", "mon.reduce_partials", "mon.encode (k_encode_tiles)", "mon.points (k_sample_points)" };
struct ProfScope {
    Model& m; int cls; hipEvent_t a = nullptr, b = nullptr; Roctx* rx;
    ProfScope(Model& mm, int c) : m(mm), cls(c), rx(roctx()) { if (rx) rx->push(kPhaseName[c]); if (m.profiling) { a = get_event(m); b = get_event(m);
            hipEventRecord(a, m.train_stream); } }
    ~ProfScope() { if (m.profiling) { hipEventRecord(b, m.train_stream); m.ev_pending.push_back({ cls, { a, b } }); } if (rx) rx->pop(); }
};
static void collect_profile(Model& m) {
    for (auto& p : m.ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) { m.prof.ms[p.first] += ms; m.prof.launches[p.first] += 1; }
        m.ev_pool.push_back(p.second.first); m.ev_pool.push_back(p.second.second);
    }
    m.ev_pending.clear();
}

void mlp_forward_inference(Model& m, hipStream_t s, const uint16_t* params, const uint16_t* E, uint16_t* O, uint32_t n) {
    // (shapes outside the fused kernels: the whole-network MFMA forward, nothing but O written; a tail of fewer than 32 samples on the per-sample kernel)
    const uint32_t body = m.d_layers_T ? (n & ~31u) : 0u; uint32_t done = 0;
    if (body && launch_mlp_forward_layers(s, m.nd, params, E, nullptr, O, body, nullptr, nullptr)) done = body;
    if (done < n) launch_mlp_forward(s, m.nd, params, E + (size_t)done * m.nd.Epad, nullptr, O + (size_t)done * kOut, n - done, nullptr);
}

// occupancy-grid skipping on the level-tile chain: what the position pass, k_encode_tiles and k_fused_train<PRE, OCC> share once the grid is in use
static LiveArgs live_args(const Model& m) {
    if (!(m.d_occ && m.occ_refreshed_iter && m.d_live_idx)) return LiveArgs{};
    const uint32_t B = m.oc.R * m.oc.S, spw = encode_tiles_spw(B);
    return LiveArgs{ m.d_occ, m.d_live_idx, m.d_live_cnt, spw, (B + spw - 1u) / spw };
}

// One iteration of Train_Step's loop body (nerf_model.cu:1637-1646), enqueued without host syncs.
static void enqueue_iteration(Model& m, int stages) {
    hipStream_t s = m.train_stream; const uint32_t B = m.oc.R * m.oc.S;
    // XORWOW mode: the generate calls of this iteration and of the next one (whose candidates and positions are prepared during this one), in order, once each
    if (m.d_xw) {
        const uint32_t R = m.oc.R, n_it = (5u + m.oc.S) * R;
        while (m.xw_filled <= m.enq_iter + 1u) {
            float* set = m.d_xw + (size_t)(m.xw_filled & 1u) * n_it;
            // :1432, :1434, :1468
            launch_xorwow_fill(s, m.d_xw_states, m.xw_lanes, m.xw_flavour, (uint32_t)(m.xw_offset % m.xw_lanes), set, 2u * R, set + 2u * R, 3u * R,
                    set + 5u * R, m.oc.S * R);
            ++m.xw_filled; m.xw_offset += n_it;
        }
    }
    if (stages & 1) {      // GenerateBatch :1429-1502
        ProfScope ps(m, MON_K_BATCH);
        // otherwise the last k_optimizer already did all of it
        if (m.backend == 1) { if (!m.next_ready) { launch_candidates_and_frags(s, m.B, m.ds->ptrs(), m.oc, m.d_state, m.P.half, m.nd, m.d_frag_train);
                if (m.d_half_tiles) launch_build_tiles_image(s, m.lf, m.nd, m.P.half, m.d_half_tiles); } }
        else launch_gen_candidates(s, m.B, m.ds->ptrs(), m.oc, m.d_state);
        if (m.backend == 0) {                       // the fused kernel compacts the rays itself
            launch_build_rays(s, m.B, m.oc, m.d_state);
            // (the layer-kernel shapes on the level-tile encode: the positions also as k_encode_tiles' float4)
            launch_gen_samples(s, m.B, m.oc, m.d_state, m.oc.S, B, kStreamDt, 0u, 0, (m.d_layers_T && m.d_e_soa && m.d_half_tiles) ? m.d_x_all : nullptr);
        }
    }
    // whole steps of a shape outside the fused kernels scatter through k_rows_to_bins -> k_grid_scatter into partial tables that the DENSE optimizer sums; the
    // Step() schedule and the stage-wise debugging entry keep tcnn's global atomics into ggrid, which only the non-dense optimizer reads and clears.  ONE flag for
    // both sites (the scatter site and the optimizer site disagreed under step_variant, and the grid stopped training)
    const bool hybrid = m.backend == 0 && m.hybrid_scatter && stages == 7 && !options().step_variant;
    if (stages & 2) {      // Step_No_Compacted :1552-1607
        if (m.backend == 0 && options().step_variant) {
            // NeRF_Model::Step (nerf_model.cu:1504-1550): inference of every sample, per-ray sample compaction + rollover (kernels_step.hip), then
            // forward + backward of the compacted batch.  B.pts / B.dO hold the compacted batch afterwards.
            ProfScope ps(m, MON_K_FWDBWD);
            launch_encode(s, m.lt, m.nd, m.P.half, m.B.pts, m.B.E, B, m.d_state);
            // :1509 inference_mixed_precision_impl, training weights
            launch_mlp_forward(s, m.nd, m.P.half, m.B.E, nullptr, m.B.O, B, m.d_state);
            launch_step_compaction(s, m.B, m.oc, m.d_state, m.d_step_counts, m.d_step_pts);
            hipMemcpyAsync(m.B.pts, m.d_step_pts, 12 * (size_t)B, hipMemcpyDeviceToDevice, s);
            launch_encode(s, m.lt, m.nd, m.P.half, m.B.pts, m.B.E, B, m.d_state);                             // :1545 forward of the compacted batch
            // (this schedule's gradient scatter reads dE / dO row-major through tcnn-style atomics and is a checkable curiosity, not a fast path: row-major throughout)
            if (!(m.d_layers_T && launch_mlp_forward_layers(s, m.nd, m.P.half, m.B.E, m.B.Hid, m.B.O, B, m.d_state, m.d_layers_T)))
                launch_mlp_forward(s, m.nd, m.P.half, m.B.E, m.B.Hid, m.B.O, B, m.d_state);
            if (m.d_layers_T && launch_mlp_backward_layers(s, m.nd, m.P.half, m.B.Hid, m.B.dO, m.B.dHid, m.B.dE, B, m.d_state, m.d_layers_T, true))
                launch_weight_grads_layers(s, m.nd, m.P.gmlp, B, m.d_state, m.d_layers_T);
            else {
                launch_mlp_backward(s, m.nd, m.P.half, m.B.Hid, m.B.dO, m.B.dHid, m.B.dE, B, m.d_state);          // :1547
                launch_weight_grads(s, m.nd, m.B.E, m.B.Hid, m.B.dHid, m.B.dO, m.P.gmlp, B, m.d_state);
            }
            if (m.hybrid_scatter) hipMemsetAsync(m.P.ggrid, 0, (size_t)m.n_grid * 2, s);           // (a whole step of the other schedule may have left partial sums)
            launch_grid_backward(s, m.lt, m.nd, m.B.pts, m.B.dE, m.P.ggrid, B, m.d_state);
        } else if (m.backend == 0) {
            ProfScope ps(m, MON_K_FWDBWD);
            // the layer-kernel shapes at base.json-sized batches: the encode from LDS level tiles (k_encode_tiles, bit-identical to the gathers) -- the tile image
            // is kept current by k_optimizer in whole steps and rebuilt here after anything else touched the weights
            const bool tiles_b0 = m.d_layers_T && m.d_e_soa && m.d_half_tiles && options().lds_encode != 0;
            if (tiles_b0) {
                if (!m.b0_tiles_current) { launch_build_tiles_image(s, m.lf, m.nd, m.P.half, m.d_half_tiles); m.b0_tiles_current = true; }
                launch_encode_tiles(s, m.lf, m.nd, m.d_half_tiles, m.d_x_all, m.d_e_soa, B, m.d_state, nullptr, m.ds->ptrs(), m.oc);
            } else
            launch_encode(s, m.lt, m.nd, m.P.half, m.B.pts, m.B.E, B, m.d_state);
            // (shapes outside the fused kernels: whole-network MFMA kernels, kernels_layers.hip)
            if (tiles_b0) {
                if (!launch_mlp_forward_layers(s, m.nd, m.P.half, m.B.E, m.B.Hid, m.B.O, B, m.d_state, m.d_layers_T, m.d_e_soa, m.B.E, stages != 7))
                    launch_mlp_forward(s, m.nd, m.P.half, m.B.E, m.B.Hid, m.B.O, B, m.d_state);         // (not reached: the shapes with a T workspace are the kernels' shapes)
            }
            else if (!(m.d_layers_T && launch_mlp_forward_layers(s, m.nd, m.P.half, m.B.E, m.B.Hid, m.B.O, B, m.d_state, m.d_layers_T, nullptr, nullptr, stages != 7)))
                launch_mlp_forward(s, m.nd, m.P.half, m.B.E, m.B.Hid, m.B.O, B, m.d_state);
            launch_composite_grad(s, m.B, m.oc, m.d_state);
            // (whole steps of the hybrid scatter with S = 32: the backward kernel writes k_grid_scatter's hand-over itself)
            const bool fold_bins = hybrid && m.d_layers_T && m.oc.S == 32u;
            const BinsOut bins{ m.d_de_soa, m.d_x_soa, m.B.pts, m.n_bins, m.lf.fix_clamp, m.d_state };
            bool bins_done = false;
            if (m.d_layers_T && launch_mlp_backward_layers(s, m.nd, m.P.half, m.B.Hid, m.B.dO, m.B.dHid, m.B.dE, B, m.d_state, m.d_layers_T, stages != 7,
                    fold_bins ? &bins : nullptr)) {
                launch_weight_grads_layers(s, m.nd, m.P.gmlp, B, m.d_state, m.d_layers_T); bins_done = fold_bins; }
            else {
                launch_mlp_backward(s, m.nd, m.P.half, m.B.Hid, m.B.dO, m.B.dHid, m.B.dE, B, m.d_state);
                launch_weight_grads(s, m.nd, m.B.E, m.B.Hid, m.B.dHid, m.B.dO, m.P.gmlp, B, m.d_state);
            }
            // whole steps of a shape outside the fused kernels: the exact LDS scatter (partial tables, summed by the optimizer).  Stage-wise calls (the debugging
            // entry that stops before the optimizer) keep tcnn's global atomics into ggrid, which only the non-dense optimizer clears: start from zeros there
            if (hybrid) {
                if (!bins_done) launch_rows_to_bins(s, m.lf, m.nd, m.B.dE, m.B.pts, m.oc.R, m.oc.S, m.n_bins, m.d_de_soa, m.d_x_soa, m.d_state);
                launch_grid_scatter(s, m.lt, m.lf, m.nd, m.d_de_soa, m.d_x_soa, B, m.n_bins, m.d_gpart, m.part_halves / 2, m.d_state, nullptr, 0u, m.P.gmlp,
                        m.d_state_next);
            } else {
                if (m.hybrid_scatter) hipMemsetAsync(m.P.ggrid, 0, (size_t)m.n_grid * 2, s);
                launch_grid_backward(s, m.lt, m.nd, m.B.pts, m.B.dE, m.P.ggrid, B, m.d_state);
            }
        } else {
            // stage-wise debugging: a forward/backward without an optimizer step after it
            if (m.scatter_pending) hipMemsetAsync(m.d_state->n_scatter, 0, sizeof(m.d_state->n_scatter), s);
            // the encode as LDS reads of level tiles; the fused kernel then loads the features
            // (the grid in use without live-sample lists -- a batch size whose partitions do not hold whole position blocks --: the gather chain masks its loads)
            const bool pre = m.d_e_soa && m.fused_dump != 1 && options().lds_encode && !m.gathers_preferred && !(m.d_occ && m.occ_refreshed_iter && !m.d_live_idx);
            m.pre_active = pre;
            const LiveArgs live = live_args(m);
            if (pre) {
                // positions of this batch: normally the last k_optimizer's position blocks already wrote them (and k_encode_tiles of the last iteration the
                // candidates)
                if (!(m.next_ready && m.points_ready)) { ProfScope pp(m, MON_K_POINTS); launch_sample_points(s, m.B, m.oc, m.d_state, m.d_x_all, live); }
                // (the next iteration is always prepared ahead: the stand-alone kernels run after an invalidation only)
                const bool gen_next = true;
                { ProfScope pe(m, MON_K_ENCODE); launch_encode_tiles(s, m.lf, m.nd, m.d_half_tiles, m.d_x_all, m.d_e_soa, B, m.d_state, gen_next ? &m.B_alt
                        : nullptr, m.ds->ptrs(), m.oc,
#ifdef MON_OVERLAP_PROBE
                        (uint32_t)options().enc_lds_kb * 1024u,
#else
                        0u,
#endif
                        live); }
            }
            ProfScope ps(m, MON_K_FWDBWD);
            // (no grid look-ups before the first refresh: every cell is live during the warm-up)
            launch_fused_train(s, m.lf, m.nd, m.P, m.B, m.oc, m.d_state, m.d_dw_partials, m.fused_dump, m.d_de_soa, m.d_x_soa, m.lds_mask, m.d_frag_train,
                    m.big_active ? m.big_switch : 0u, m.d_touched, m.occ_refreshed_iter ? m.d_occ : nullptr, m.n_bins, pre ? m.d_e_soa : nullptr);
            m.scatter_pending = true;
        }
    }
    if ((stages & 2) && m.backend == 1) {
        const bool folded = m.lds_mask && grid_scatter_sums_partials(m.lt, m.nd);   // the scatter workgroups also sum the dW partial rows
        if (m.lds_mask) { ProfScope ps(m, MON_K_SCATTER);
            launch_grid_scatter(s, m.lt, m.lf, m.nd, m.d_de_soa, m.d_x_soa, B, m.n_bins, m.d_gpart, m.part_halves / 2, m.d_state,
                                                                                folded ? m.d_dw_partials : nullptr, fused_train_grid(m.nd, m.oc.R), m.P.gmlp,
                                                                                        m.d_state_next); }
        if (m.big_active) { ProfScope ps(m, MON_K_SCATTER);
            launch_big_scatter(s, m.lt, m.lf, m.nd, m.lds_mask, m.d_de_soa, m.d_x_soa, B, m.n_bins, m.d_state, m.big_switch, m.d_big_ws, m.P.ggrid, m.d_touched
                ? m.d_touched + (m.nd.n_mlp >> 3) : nullptr); }
        if (!folded) { ProfScope ps(m, MON_K_REDUCE); launch_reduce_partials(s, m.d_dw_partials, fused_train_grid(m.nd, m.oc.R), m.nd, m.P.gmlp, m.d_state); }
    }
    if (stages & 4) {      // Trainer::optimizer_step :1644
        ProfScope ps(m, MON_K_OPTIM);
        ParamPtrs P = m.P;
        if (m.backend == 1 && m.lds_mask) { P.gpart = m.d_gpart; P.part_stride = m.part_halves; P.sl = m.scatter;
            P.all_levels_dense = (m.lds_mask == ((1u << m.nd.L) - 1u)) ? 1 : 0; }
        if (hybrid) { P.gpart = m.d_gpart; P.part_stride = m.part_halves; P.sl = m.scatter; P.all_levels_dense = 1; }
        P.half_tiles = ((m.backend == 1 || hybrid) && P.gpart && P.all_levels_dense) ? m.d_half_tiles : nullptr;
        if (m.backend == 0) m.b0_tiles_current = hybrid && P.half_tiles != nullptr;        // (any other optimizer leaves the tile image behind the weights)
        const bool lazy = m.lazy_ema && !(P.gpart && P.all_levels_dense);
        P.ema_step = lazy ? m.d_ema_step : nullptr; P.lazy = lazy ? 1 : 0; if (lazy) m.ema_pending = true;
        // (the LDS-scattered levels are a prefix: sizes grow with the level)              // every writer of ggrid on the fused path sets the chunk flags; the
        // unfused grid backward does not
        if (lazy && m.backend == 1 && m.d_touched && (m.lds_mask & (m.lds_mask + 1u)) == 0u) {
            P.touched = m.d_touched; uint32_t first_big = 0; while (first_big < (uint32_t)m.nd.L && ((m.lds_mask >> first_big) & 1u)) ++first_big;
            P.first_flag_chunk = (m.nd.n_mlp + 2u * m.lt.offset[first_big]) >> 3;
        }
        OptimNext nx{};
        // (see gen_next above: options fold_next / fold_reduce / lds_scatter were measurement switches of rounds 1-2 and are gone)
        const bool fold = true;
        // k_encode_tiles generated the next candidates into B_alt; sample their positions here
        const bool pos_mode = m.backend == 1 && fold && m.pre_active && m.d_e_soa;
        if (m.backend == 1 && fold) {
            nx.cand_blocks = pos_mode ? 0u : (m.oc.R + 255) / 256; nx.frag_image = m.d_frag_train; nx.fd = FragDims{ m.nd.Epad, m.nd.W, m.nd.NH, m.nd.L };
            nx.b = pos_mode ? m.B_alt : m.B; nx.ds = m.ds->ptrs(); nx.oc = m.oc;
            // one sample per thread up to 131 072 samples (two beyond: as many position blocks as optimizer blocks made the kernel 10 us longer at R = 8192): a
            // thread's chain is select -> candidate loads -> store, ~5 us of latency that several samples per thread put in series (64 blocks of 8 samples per
            // thread made these blocks the kernel's tail)
            if (pos_mode) { nx.pos_blocks = std::min((B + 255u) / 256u, 512u); nx.x_all = m.d_x_all; nx.live = live_args(m); }
        }
#ifdef MON_OVERLAP_PROBE
        // PROBE (option overlap): a second, throw-away k_encode_tiles of the CURRENT batch next to k_optimizer -- what would the pair cost side by side?
        //   1 behind the optimizer on the same stream; 2 / 3 on a side stream, enqueued before / after the optimizer; 4 / 5 the same with a high-priority side
        //   stream (its own hardware queue); 6 behind the optimizer on the same stream WITHOUT the barrier bit (hipExtAnyOrderLaunch)
        const long ovl = pos_mode ? (long)options().overlap : 0; const uint32_t enc_lds = (uint32_t)options().enc_lds_kb * 1024u;
        auto dummy_encode = [&](hipStream_t q, uint32_t any) { launch_encode_tiles(q, m.lf, m.nd, m.d_half_tiles, m.d_x_all, m.d_e_soa, B, m.d_state, nullptr,
                m.ds->ptrs(), m.oc, enc_lds | any); };
        const bool side = ovl >= 2 && ovl <= 5, enc_first = ovl == 2 || ovl == 4;
        if (side && !m.side_stream) {
            if (ovl >= 4) hipStreamCreateWithPriority(&m.side_stream, hipStreamNonBlocking, -1); else hipStreamCreateWithFlags(&m.side_stream, hipStreamNonBlocking);
            hipEventCreateWithFlags(&m.ev_fork, hipEventDisableTiming); hipEventCreateWithFlags(&m.ev_join, hipEventDisableTiming); }
        if (side) { hipEventRecord(m.ev_fork, s); hipStreamWaitEvent(m.side_stream, m.ev_fork, 0); }
        if (side && enc_first) { dummy_encode(m.side_stream, 0u); hipEventRecord(m.ev_join, m.side_stream); }
#endif
        launch_optimizer(s, P, m.opt, m.d_state, m.d_state_next, nx, (m.oc.R * m.oc.S) / 8u); m.scatter_pending = false;
#ifdef MON_OVERLAP_PROBE
        if (side && !enc_first) { dummy_encode(m.side_stream, 0u); hipEventRecord(m.ev_join, m.side_stream); }
        if (side) hipStreamWaitEvent(s, m.ev_join, 0);
        if (ovl == 1) dummy_encode(s, 0u);
        if (ovl == 6) dummy_encode(s, 1u);
#endif
        std::swap(m.d_state, m.d_state_next);                // the next iteration (and the host's read-back) uses the state this launch prepares
        if (pos_mode) std::swap(m.B, m.B_alt);               // ... and the candidate set k_encode_tiles filled for it
        m.next_ready = (m.backend == 1 && fold); m.points_ready = pos_mode; ++m.enq_iter;
    }
}

// Occupancy grid refresh (cfg.occupancy_skip): before iteration `iter` when it is due.  Stream-ordered between two iterations, from the training weights.
static void maybe_refresh_occupancy(Model& m, uint32_t iter) {
    // due at the first iteration it is asked for at or after the next multiple of kOccInterval (the hipGraph path only asks at the start of a captured PAIR:
    // after an odd number of iterations an exact "iter % interval == 0" test was never true again and the grid was never refreshed)
    if (!m.d_occ || m.backend != 1 || iter < (uint32_t)kOccWarmup || iter < m.occ_next_refresh) return;
    launch_occupancy_update(m.train_stream, m.lf, m.nd, m.P.half, m.oc, m.d_frag_occ, m.occ_raw_threshold, m.d_occ_tmp, m.d_occ);
    // the density field settles: every kOccInterval iterations at first, every 4th / 16th of that rate later (a refresh costs ~60 us, 1.9 us per step at the
    // early rate -- more than the skipping saves once the level-tile chain has taken the gathers out of the forward pass; tools/occ_timing.py)
    const uint32_t every = (uint32_t)kOccInterval * (iter < 512u ? 1u : iter < 2048u ? 4u : 16u);
    m.occ_refreshed_iter = iter; m.occ_next_refresh = (iter / every + 1u) * every;
    // the positions already sampled for this iteration carry the OLD grid's live bits and lists (or none: the first refresh): sample them again
    if (m.d_live_idx) m.points_ready = false;
}

static int sync_state(Model& m) {
    // :1645 (once per call instead of once per iteration); the state rides the same sync in a pinned buffer -- the online manager trains
    // in slices of a few iterations, where a second blocking copy would be a visible share of the slice
    // (only the head: the slot counters behind it are 16 KB the host never reads; written by a one-block kernel rather than hipMemcpyAsync, whose small-copy
    // path costs the slicing online thread ~10 us per call)
    launch_copy_params(m.train_stream, reinterpret_cast<const uint16_t*>(m.d_state), reinterpret_cast<uint16_t*>(m.h_state_pinned),
            (uint32_t)(offsetof(DevState, n_scatter) / 2));
    // (an event, not hipStreamSynchronize: the stream may be a lane other objects keep feeding)
    if (!m.sync_event) HIPCHECK(hipEventCreateWithFlags(&m.sync_event, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(m.sync_event, m.train_stream)); HIPCHECK(hipEventSynchronize(m.sync_event));
    std::memcpy(&m.h_state, m.h_state_pinned, offsetof(DevState, n_scatter));
    collect_profile(m);
    return MON_OK;
}

int model_train(Model& m, int iters, float* loss, int stages) {
    if (iters < 0) { set_error("train: negative iteration count"); return MON_ERR_ARG; }
    if (m.n_boxes == 0) { set_error("train: no 2-D boxes (UpdateFrameIdAndBbox was never called)"); return MON_ERR_STATE; }
    HIPCHECK(use_device(m.device));
    // workspace of the Step() schedule (enqueue_iteration cannot report a failed allocation)
    if (m.backend == 0 && options().step_variant && !m.d_step_counts) {
        int rc;
        if ((rc = dev_alloc(m, m.d_step_counts, (size_t)m.oc.R + 1)) || (rc = dev_alloc(m, m.d_step_pts, 3 * (size_t)m.oc.R * m.oc.S))) return rc;
    }
    // Large-table scatter: the device picks binned / atomic per iteration from the previous iteration's gradient-carrying sample count;
    // once the host has seen that count well below the switch point it stops launching the (then empty) binning kernels at all.
    m.big_active = m.big_switch && (m.h_state.n_scatter_last == 0u || m.h_state.n_scatter_last > m.big_switch / 2u);
    // Occupancy-grid skipping (opt-in): once the grid is in use and few samples are left, the gather chain wins -- k_fused_train skips the gathers of the
    // samples in empty cells, k_encode_tiles encodes every sample (kernel_times, late window: 63.3 against 65.4 us per step; early, 99.6 against 92.1).  Both
    // chains leave bit-identical parameters, so the choice is free per call; the host knows the regime from the last call's read-back.
    m.gathers_preferred = m.d_occ && m.occ_refreshed_iter && m.h_state.n_scatter_last != 0u && 8u * m.h_state.n_scatter_last < m.oc.R * m.oc.S;
#ifndef MON_OCC_PREFER_GATHERS      // (variant build for the A/B)
    // round 6: with the live-sample lists k_encode_tiles walks the live samples only and the level tiles win in every regime
    if (m.d_live_idx) m.gathers_preferred = false;
#endif
    const bool use_graph_env = options().use_graph != 0;
    if (iters > 0) m.weights_epoch = next_weights_epoch();
    // (nothing of this object is in flight between calls: the read-back at the end of the last one is current)
    m.enq_iter = m.h_state.iter;
    // (the first occupancy refresh changes a kernel argument)
    const bool use_graph = use_graph_env && !m.profiling && stages == 7 && iters >= 2 && !(m.d_occ && !m.occ_refreshed_iter) && !m.d_xw;
    // chunks of iterations go through the device's training lanes (whole steps only; big-table objects are HBM-bound in their optimizer and gain from more
    // overlap, not less)
    const bool lanes_on = stages == 7 && m.big_switch == 0u; const int chunk = kLaneChunk;
    if (use_graph) {
        // (the last bit: which forward chain the captured pair runs)
        const int graph_key = m.backend | (m.big_active ? 256 : 0) | (m.occ_refreshed_iter ? 512 : 0)
                | ((m.d_e_soa && options().lds_encode && !m.gathers_preferred) ? 1024 : 0);
        // (the captured pair starts on this DevState and this candidate set)
        if (!m.graph_exec || m.graph_backend != graph_key || m.graph_state != m.d_state || m.graph_mask != m.B.mask) {
            drop_graph(m);
            hipGraph_t g = nullptr;
            // captured on the object's own stream (a lane is shared with other host threads), replayed on the current one
            const hipStream_t cur = m.train_stream; m.train_stream = m.own_stream;
            HIPCHECK(hipStreamBeginCapture(m.train_stream, hipStreamCaptureModeThreadLocal));
            m.next_ready = false;                           // the captured iterations are self-contained
            m.graph_state = m.d_state; m.graph_mask = m.B.mask;
            // a PAIR: the two DevStates swap roles every iteration, after two the captured pointers are current again
            enqueue_iteration(m, 7); enqueue_iteration(m, 7);
            const hipError_t ce = hipStreamEndCapture(m.train_stream, &g); m.train_stream = cur; HIPCHECK(ce);
            HIPCHECK(hipGraphInstantiate(&m.graph_exec, g, nullptr, nullptr, 0));
            hipGraphDestroy(g); m.graph_backend = graph_key;
        }
        int i = 0;
        while (i + 2 <= iters) {
            LaneChunk lc(m, lanes_on);
            for (int k = 0; k < chunk && i + 2 <= iters; k += 2, i += 2) { maybe_refresh_occupancy(m, m.h_state.iter + (uint32_t)i);
                HIPCHECK(hipGraphLaunch(m.graph_exec, m.train_stream)); }
        }
        for (; i < iters; ++i) { LaneChunk lc(m, lanes_on); maybe_refresh_occupancy(m, m.h_state.iter + (uint32_t)i); m.next_ready = false;
            enqueue_iteration(m, 7); }
    } else {
        for (int i = 0; i < iters; ) {
            LaneChunk lc(m, lanes_on);
            for (int k = 0; k < chunk && i < iters; ++k, ++i) { if (stages == 7) maybe_refresh_occupancy(m, m.h_state.iter + (uint32_t)i);
                enqueue_iteration(m, stages); }
        }
    }
    HIPCHECK(hipGetLastError());
    int rc = sync_state(m); if (rc) return rc;
    if (loss) *loss = m.h_state.loss_sum / (float)m.oc.R;       // :1650-1658
    if (stages == 7 && iters > 0) rc = publish_snapshot(m, iters >= 64);
    mark_tail(m);
    return rc;
}

// owner thread: the end of a whole Train_Step_Online
int model_publish_snapshot(Model& m) { HIPCHECK(use_device(m.device)); const int rc = publish_snapshot(m, true); mark_tail(m); return rc; }

// Render of the latest PUBLISHED inference weights on the inference stream: callable from any thread while the owner trains (no model mutex,
// no train-stream work).  MON_ERR_STATE when nothing has been published yet (or the model has no inference side): the caller falls back to
// model_render under the model mutex.
int model_render_snapshot(Model& m, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, uint32_t* snapshot_step) {
    if (!pose16 || !rgb || !depth || !mask || box.w == 0 || box.h == 0) { set_error("render: bad argument"); return MON_ERR_ARG; }
    InferState* is = m.infer; if (!is) { set_error("render_snapshot: this object renders on its train stream"); return MON_ERR_STATE; }
    InferShared* sh = is->shared; std::lock_guard<std::mutex> one(sh->mu);      // one snapshot render per device at a time
    is->wanted.store(true);                                             // the training side refreshes the snapshot at the end of its current slice
    int r;
    {   std::lock_guard<std::mutex> l(is->mu); r = is->latest; if (r < 0) { set_error("render_snapshot: no weights published yet"); return MON_ERR_STATE; }
        // the newest snapshot's copy may still be queued behind other objects' training kernels (it runs on the train stream, at normal priority: 1-2 ms on a
        // busy device); the one before it is complete, and nobody writes it before the train stream has been synchronised again -- by which time the newest is
        // complete and chosen here.  A viewer prefers a finished snapshot one slice older to waiting.
        if (is->written[1 - r] && hipEventQuery(is->ready[r]) != hipSuccess && hipEventQuery(is->ready[1 - r]) == hipSuccess) r = 1 - r;
        ++is->readers[r]; if (snapshot_step) *snapshot_step = is->step_of[r]; }
    struct Release { InferState* is; int r; ~Release() { std::lock_guard<std::mutex> l(is->mu); --is->readers[r]; } } release{ is, r };
    HIPCHECK(use_device(m.device));
    hipStream_t s = sh->stream;
    HIPCHECK(hipStreamWaitEvent(s, is->ready[r], 0));
    Mat4 pose; std::memcpy(pose.m, pose16, 64);
    const uint32_t n_pix = box.w * box.h, S2 = 2 * m.oc.S;
    // one buffer of 5 floats per pixel: rgb | depth | mask laid out back to back for THIS crop, so one copy brings them home
    if (n_pix > is->out_cap) {
        const size_t cap = std::max<size_t>(n_pix, 2 * is->out_cap); void* q = nullptr;
        HIPCHECK(hipMalloc(&q, 20 * cap));
        is->grown.push_back(q);                                         // (freed with the object; the superseded ones are smaller than the live one)
        is->out_all = (float*)q; is->out_cap = cap;
    }
    is->out_rgb = is->out_all; is->out_depth = is->out_all + 3 * (size_t)n_pix; is->out_mask = is->out_all + 4 * (size_t)n_pix;
    // A viewer's render competes with the training kernels of every object on the device.  k_encode_feat's workgroups need a whole CU each (160 KB of LDS) and
    // wait until training workgroups have drained from one; the gather render's small workgroups slip in anywhere.  Measured (tools/online_replay.py,
    // 60 keyframes every 50 ms, mean / p99 of the viewer's crop): 1 object 0.85 / 1.13 ms on tiles against 1.20 / 1.93 ms through the gathers, 4 objects 0.51 / 2.0 against 0.68 / 1.7,
    // 12 objects 0.58 / 3.1 against 0.67 / 1.7 -- so the tiles serve the viewer while few objects live on the device, the gathers once many do
    // (tile_ws_objects counts the tile-capable objects ALIVE on the device: in the online manager every live object has a training thread).
    TileWs* tws = nullptr; std::unique_lock<std::mutex> tile_lock;
    // level tiles in LDS (kernels_tilerender.hip); the inference side's own workspace
    if (tile_render_wanted(m, n_pix) && (options().tile_render >= 2 || tile_ws_objects(m.device) <= 4)) {
        { const int rc = tile_ws_get(m, 1, n_pix, &tws); if (rc) return rc; }
        tile_lock = std::unique_lock<std::mutex>(tws->mu);                // (held until the stream is synchronised below)
        uint64_t ep; { std::lock_guard<std::mutex> l(is->mu); ep = is->epoch_of[r]; }
        tile_ws_weights(m, *tws, s, is->snap[r], ep);
        tile_render_crop(m, *tws, s, m.oc, box, pose, pose_is_Toc, is->out_rgb, is->out_depth, is->out_mask);
    } else {
        for (uint32_t p0 = 0; p0 < n_pix; p0 += kRenderChunkRays) {
            const uint32_t n = (n_pix - p0) < kRenderChunkRays ? (n_pix - p0) : kRenderChunkRays;
            launch_render_rays(s, is->rb, m.ds->K, m.oc, box, pose, pose_is_Toc, p0, n);
            launch_fused_render(s, m.lf, m.nd, is->snap[r], is->rb, m.oc, n, p0 * S2, is->out_rgb + 3 * (size_t)p0, is->out_depth + p0, is->out_mask + p0,
                    is->frag, p0 == 0u);
        }
    }
    // results: through a PINNED staging buffer of the inference side.  A device-to-host copy into the caller's pageable memory is done by the runtime's own
    // blit path, which queues at normal priority behind every training kernel on the device (12 objects training: 5-6 ms for 1 MB, one render in five); into
    // pinned memory it is ordered on this high-priority stream.
    if (5 * (size_t)n_pix > sh->h_cap) {                                  // (a crop larger than a frame: not produced by the managers)
        float* q = nullptr; HIPCHECK(hipHostMalloc((void**)&q, 5 * (size_t)n_pix * sizeof(float), hipHostMallocDefault));
        if (sh->h_out) hipHostFree(sh->h_out);
        sh->h_out = q; sh->h_cap = 5 * (size_t)n_pix;
    }
    // (a copy KERNEL on this stream writing the pinned buffer over PCIe, not hipMemcpyAsync: the runtime's copy path has its own queueing, 1-2 ms on a device
    //  busy with a dozen training objects, while a kernel inherits the stream's priority like the render kernels before it)
    launch_copy_params(s, reinterpret_cast<const uint16_t*>(is->out_all), reinterpret_cast<uint16_t*>(sh->h_out), (uint32_t)(10 * (size_t)n_pix));
    HIPCHECK(hipStreamSynchronize(s));
    HIPCHECK(hipGetLastError());
    std::memcpy(rgb, sh->h_out, 12 * (size_t)n_pix); std::memcpy(depth, sh->h_out + 3 * (size_t)n_pix, 4 * (size_t)n_pix);
    std::memcpy(mask, sh->h_out + 4 * (size_t)n_pix, 4 * (size_t)n_pix);
    return MON_OK;
}

// Render / RenderVideo body :1768-1828, chunked; inference (EMA) weights once training has run.
// Lazy EMA: apply the steps untouched chunks sat out before the inference weights are read.
int ensure_ema_current(Model& m) {
    if (!m.ema_pending) return MON_OK;
    HIPCHECK(use_device(m.device)); model_leave_lane(m);
    ParamPtrs P = m.P; P.ema_step = m.d_ema_step; P.lazy = 1;
    launch_ema_finalize(m.train_stream, P, m.opt, m.d_state);
    HIPCHECK(hipGetLastError()); m.ema_pending = false; m.weights_epoch = next_weights_epoch(); return MON_OK;
}

int model_render(Model& m, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, int dst_on_device) {
    if (!pose16 || !rgb || !depth || !mask || box.w == 0 || box.h == 0) { set_error("render: bad argument"); return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device));
    model_leave_lane(m);
    { int rc = ensure_ema_current(m); if (rc) return rc; }
    hipStream_t s = m.train_stream;
    // (no state read-back: every call that advances the optimizer ends with sync_state, so the host's copy of the step counter is current whenever this thread
    // gets here -- the read-back and the synchronisation in front of it were two host round trips per render)
    const uint16_t* prm = (m.h_state.step > 0) ? m.P.ema : m.P.half;
    Mat4 pose; std::memcpy(pose.m, pose16, 64);
    const uint32_t n_pix = box.w * box.h, S2 = 2 * m.oc.S;
    // whole-crop output buffer (grow-only): all chunks are enqueued back to back, one copy-out and one sync per call
    if (n_pix > m.out_cap) {
        // doubling: the superseded buffers (freed with the object) add up to less than the live one
        const size_t cap = std::max<size_t>(n_pix, 2 * m.out_cap);
        int rc; if ((rc = dev_alloc(m, m.d_out_all, 5 * cap, false))) return rc;
        m.out_cap = cap;
    }
    // rgb | depth | mask of THIS crop, back to back
    m.d_out_rgb = m.d_out_all; m.d_out_depth = m.d_out_all + 3 * (size_t)n_pix; m.d_out_mask = m.d_out_all + 4 * (size_t)n_pix;
    if (!dst_on_device && 5 * (size_t)n_pix > m.h_out_cap) {          // pinned staging: one device-to-host copy instead of three into pageable memory
        const size_t cap = std::max<size_t>(5 * (size_t)n_pix, 2 * m.h_out_cap); float* q = nullptr;
        HIPCHECK(hipHostMalloc((void**)&q, cap * sizeof(float), hipHostMallocDefault));
        if (m.h_out) (void)hipHostFree(m.h_out);
        m.h_out = q; m.h_out_cap = cap;
    }
    if (m.d_xw) {          // XORWOW mode: a NEW generator per Render (default seed) draws the whole crop's RandDt in one call (nerf_model.cu:1725-1728, :1781)
        const size_t need = (size_t)n_pix * S2;
        if (need > m.xw_render_cap) { int rc = dev_alloc(m, m.d_xw_render, need, false); if (rc) return rc; m.xw_render_cap = need; }
        HIPCHECK(hipMemcpyAsync(m.d_xw_render_states, m.d_xw_render_init, sizeof(XorwowState) * m.xw_lanes, hipMemcpyDeviceToDevice, s));
        launch_xorwow_fill(s, m.d_xw_render_states, m.xw_lanes, m.xw_flavour, 0u, m.d_xw_render, (uint32_t)need, nullptr, 0u, nullptr, 0u);
        m.oc.xw_render = m.d_xw_render;
    }
    TileWs* tws = nullptr; std::unique_lock<std::mutex> tile_lock;
    // level tiles in LDS (kernels_tilerender.hip), the device's train-side workspace: held until the stream is synchronised below
    if (tile_render_wanted(m, n_pix)) {
        { const int rc = tile_ws_get(m, 0, n_pix, &tws); if (rc) return rc; }
        tile_lock = std::unique_lock<std::mutex>(tws->mu);
        ProfScope ps(m, MON_K_RENDER);
        tile_ws_weights(m, *tws, s, prm, m.weights_epoch);
        tile_render_crop(m, *tws, s, m.oc, box, pose, pose_is_Toc, m.d_out_rgb, m.d_out_depth, m.d_out_mask);
    } else
    {
      // rays per pass: the fused kernel takes kRenderChunkRays; the layer-at-a-time kernels what their sample buffers hold
      const uint32_t pass = m.backend == 0 ? std::max(1u, std::min(kRenderChunkRays, m.ws_samples / S2)) : kRenderChunkRays;
      for (uint32_t p0 = 0; p0 < n_pix; p0 += pass) {
        const uint32_t n = (n_pix - p0) < pass ? (n_pix - p0) : pass;
        ProfScope ps(m, MON_K_RENDER);
        launch_render_rays(s, m.B, m.ds->K, m.oc, box, pose, pose_is_Toc, p0, n);
        if (m.backend == 0) {
            launch_gen_samples(s, m.B, m.oc, m.d_state, S2, n * S2, kStreamRender, p0 * S2, 1);
            launch_encode(s, m.lt, m.nd, prm, m.B.pts, m.B.E, n * S2, nullptr);
            mlp_forward_inference(m, s, prm, m.B.E, m.B.O, n * S2);
            launch_composite_render(s, m.B, S2, n, m.d_out_rgb + 3 * (size_t)p0, m.d_out_depth + p0, m.d_out_mask + p0);
        } else {
            launch_fused_render(s, m.lf, m.nd, prm, m.B, m.oc, n, p0 * S2, m.d_out_rgb + 3 * (size_t)p0, m.d_out_depth + p0, m.d_out_mask + p0,
                    m.d_frag_render, p0 == 0u);
        }
      }
    }
    if (dst_on_device) {
        HIPCHECK(hipMemcpyAsync(rgb, m.d_out_rgb, 12 * (size_t)n_pix, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(depth, m.d_out_depth, 4 * (size_t)n_pix, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(mask, m.d_out_mask, 4 * (size_t)n_pix, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipStreamSynchronize(s));
    } else {
        HIPCHECK(hipMemcpyAsync(m.h_out, m.d_out_all, 20 * (size_t)n_pix, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        std::memcpy(rgb, m.h_out, 12 * (size_t)n_pix); std::memcpy(depth, m.h_out + 3 * (size_t)n_pix, 4 * (size_t)n_pix);
        std::memcpy(mask, m.h_out + 4 * (size_t)n_pix, 4 * (size_t)n_pix);
    }
    HIPCHECK(hipGetLastError());
    collect_profile(m);
    return MON_OK;
}

// GetDensityOnGrid :2007-2048
int model_density_grid(Model& m, int rx, int ry, int rz, float* out_host) {
    if (rx < 2 || ry < 2 || rz < 2 || !out_host || (uint64_t)rx * (uint64_t)ry * (uint64_t)rz > (1ull << 31)) { set_error("density_grid: bad argument");
        return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device));
    model_leave_lane(m);
    { int rc = ensure_ema_current(m); if (rc) return rc; }
    hipStream_t s = m.train_stream;
    HIPCHECK(hipStreamSynchronize(s));
    // (the head: the slot counters behind it are 16 KB the host never reads)
    HIPCHECK(hipMemcpy(&m.h_state, m.d_state, offsetof(DevState, n_scatter), hipMemcpyDeviceToHost));
    const uint16_t* prm = (m.h_state.step > 0) ? m.P.ema : m.P.half;
    const uint32_t total = (uint32_t)rx * ry * rz, chunk = m.ws_samples;
    if (m.backend == 1 && m.tile_ok && options().tile_render != 0) {      // level tiles in LDS: the device's train-side workspace
        TileWs* ws = nullptr; { const int rc = tile_ws_get(m, 0, 0, &ws); if (rc) return rc; }
        std::lock_guard<std::mutex> wl(ws->mu);
        tile_ws_weights(m, *ws, s, prm, m.weights_epoch);
        const uint32_t tchunk = std::min(ws->cap, m.ws_samples);
        for (uint32_t p0 = 0; p0 < total; p0 += tchunk) {
            const uint32_t n = std::min(total - p0, tchunk);
            launch_grid_points4(s, ws->x, rx, ry, rz, p0, n);
            tile_points_forward(m, *ws, s, n);
            launch_extract_density(s, ws->O, m.B.tdist, n);
            HIPCHECK(hipMemcpyAsync(out_host + p0, m.B.tdist, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
            HIPCHECK(hipStreamSynchronize(s));
        }
        return MON_OK;
    }
    for (uint32_t p0 = 0; p0 < total; p0 += chunk) {
        const uint32_t n = (total - p0) < chunk ? (total - p0) : chunk;
        launch_grid_points(s, m.B.pts, rx, ry, rz, p0, n);
        launch_encode(s, m.lt, m.nd, prm, m.B.pts, m.B.E, n, nullptr);
        mlp_forward_inference(m, s, prm, m.B.E, m.B.O, n);
        launch_extract_density(s, m.B.O, m.B.tdist, n);
        HIPCHECK(hipMemcpyAsync(out_host + p0, m.B.tdist, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
    }
    return MON_OK;
}

int model_get_params(Model& m, int which, void* dst, size_t bytes) {
    const void* src = nullptr; size_t need = 0;
    if (which == 0 && m.P.rec) {                             // chunk records: the master weights through a staging buffer
        if (!dst || bytes < (size_t)m.n_params * 4) { set_error("get_params: buffer too small"); return MON_ERR_ARG; }
        HIPCHECK(use_device(m.device)); model_leave_lane(m);
        float* tmp = nullptr; HIPCHECK(hipMalloc((void**)&tmp, (size_t)m.n_params * 4));
        launch_state_unpack(m.train_stream, m.P.rec, 0, tmp, m.n_params);
        hipError_t e = hipStreamSynchronize(m.train_stream); if (e == hipSuccess) e = hipMemcpy(dst, tmp, (size_t)m.n_params * 4, hipMemcpyDeviceToHost);
        (void)hipFree(tmp); HIPCHECK(e); return MON_OK;
    }
    switch (which) { case 0: src = m.P.master; need = (size_t)m.n_params * 4; break; case 1: src = m.P.half; need = (size_t)m.n_params * 2; break;
                     case 2: src = m.P.ema; need = (size_t)m.n_params * 2; break; default: set_error("get_params: which must be 0..2"); return MON_ERR_ARG; }
    if (!dst || bytes < need) { set_error("get_params: buffer too small (%zu < %zu)", bytes, need); return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device)); model_leave_lane(m);
    if (which == 2) { int rc = ensure_ema_current(m); if (rc) return rc; }
    HIPCHECK(hipStreamSynchronize(m.train_stream));
    HIPCHECK(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost)); return MON_OK;
}
int model_set_params(Model& m, const float* master, size_t n) {
    if (!master || n != m.n_params) { set_error("set_params: expected %u values", m.n_params); return MON_ERR_ARG; }
    HIPCHECK(use_device(m.device)); model_leave_lane(m); HIPCHECK(hipStreamSynchronize(m.train_stream));
    { const int urc = upload_master(m, master); if (urc) return urc; }
    HIPCHECK(hipStreamSynchronize(m.train_stream));
    m.next_ready = false;                                   // the fragment image no longer matches the weights
    m.b0_tiles_current = false;
    m.weights_epoch = next_weights_epoch();
    { const int rc = publish_snapshot(m); if (rc) return rc; }   // (viewers of an untrained object see the weights just set:
    // the snapshot is complete before the call returns, so no render prefers the one before it)
    HIPCHECK(hipStreamSynchronize(m.train_stream)); return MON_OK;
}

}  // namespace mon
