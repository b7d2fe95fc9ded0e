// manager.cpp -- host mirror of nerf::NerfManagerOffline / nerf::NeRF (offline part) without Eigen / OpenCV:
// CORE/src/nerf_manager.cu:9-131 (Init, ReadDataset, CreateNeRF, WaitThreadsEnd), CORE/src/nerf_data.cu:27-235
// (sequence layout: config.yaml, img.txt, groundtruth.txt, rgb|depth|instance PNGs), CORE/src/nerf.cu:58-152
// (object file, thread body: 10 x 500 iterations) and the test-image writer of nerf.cu:255-349.
// One std::thread per object, object k on device k mod nGPU (nerf.cu:27-33), one dataset replica per device.
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include "model.h"
#include "png_io.h"

namespace mon {

void set_error(const char* fmt, ...);
const char* last_error();
int device_count(int* n);
int config_from_json(const char* path, mon_config& c);
int dataset_create(int device, int H, int W, float fx, float fy, float cx, float cy, uint32_t max_frames, int use_depth, Dataset** out);
int dataset_add_frame(Dataset* d, uint32_t id, const uint8_t* rgb, int ch, int is_bgr, const uint8_t* inst, const float* depth, const float* Twc);
int dataset_destroy(Dataset* d);
int model_create(Dataset* ds, const mon_config& cfg, int class_id, const float* Tow, const float* amin, const float* amax, Model** out);
int model_destroy(Model* m);
int model_add_boxes(Model& m, const mon_frame_bbox* boxes, size_t n);
int model_generate_mesh(Model& m, int res, float thresh, uint32_t* n_verts, uint32_t* n_indices);
int model_save_mesh(Model& m, const char* path);
int model_mesh_counts(Model& m, uint32_t* n_verts, uint32_t* n_verts_real, uint32_t* n_indices);
int model_train(Model& m, int iters, float* loss, int stages);
int stream_pool_reserve(int device, int n);
int dataset_update_poses(Dataset* d, uint32_t first, uint32_t n, const float* Twc16s);
int model_render(Model& m, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, int dst_on_device);
int model_render_snapshot(Model& m, mon_frame_bbox box, const float* pose16, int pose_is_Toc, float* rgb, float* depth, float* mask, uint32_t* snapshot_step);

// Eigen::Quaternionf(w,x,y,z).toRotationMatrix() + translation -> column-major 4x4 (nerf_data.cu:100-106)
static void pose_from_tq(const float* t, float qx, float qy, float qz, float qw, float* M) {
    const float n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw); qx /= n; qy /= n; qz /= n; qw /= n;
    const float R[9] = { 1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                         2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                         2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy) };      // row-major
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[c * 4 + r] = R[r * 3 + c];
    M[3] = M[7] = M[11] = 0.f; M[12] = t[0]; M[13] = t[1]; M[14] = t[2]; M[15] = 1.f;
}
// inverse of a rigid transform (the reference calls Matrix4f::inverse() on Two, nerf.cu:89)
static void rigid_inverse(const float* M, float* Inv) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Inv[c * 4 + r] = M[r * 4 + c];
    for (int r = 0; r < 3; ++r) Inv[12 + r] = -(Inv[r] * M[12] + Inv[4 + r] * M[13] + Inv[8 + r] * M[14]);
    Inv[3] = Inv[7] = Inv[11] = 0.f; Inv[15] = 1.f;
}

struct OfflineObject {
    int id = 0, device = 0, cls = 0; float Tow[16]; float amin[3], amax[3];
    std::vector<mon_frame_bbox> boxes; std::vector<std::string> stamps;
    Model* model = nullptr; mon_object handle{ nullptr }; float last_loss = 0.f; int rc = 0; std::string err;
    std::mutex mu_model;            // a Model is single-threaded: held by the training thread per train / mesh call and by every caller-side use of the model
};

struct OfflineManager {
    std::string dataset, cfg_path; bool use_depth = false; int n_dev = 0; mon_config cfg{};
    float fx = 0, fy = 0, cx = 0, cy = 0, depth_scale = 1.f; int H = 0, W = 0;
    std::vector<std::string> names, stamps; std::map<std::string, uint32_t> stamp_to_idx; std::vector<float> poses;   // [n][16]
    std::vector<Dataset*> ds; std::vector<OfflineObject*> objs; std::vector<std::thread> threads; bool joined = false;
    int outer_iters = 10, inner_iters = 500;       // nerf_manager.cu:89, nerf_model.cu:1635
    std::string mesh_dir = "./output";              // nerf.cu:148
    int mesh_res = 64; float mesh_thresh = 2.0f;   // marching_cubes.h:30-31
};

// cv::FileStorage looks keys up exactly (nerf_data.cu:39-46): the key must start a line (after blanks), be followed by blanks and
// ':', and carry a number; comment lines and longer keys with the same prefix (Camera.Height vs Camera.H) do not match.
bool read_yaml_number(const std::string& text, const char* key, double& v) {
    const size_t klen = std::strlen(key); size_t ls = 0;
    while (ls < text.size()) {
        size_t le = text.find('\n', ls); if (le == std::string::npos) le = text.size();
        size_t p = ls; while (p < le && (text[p] == ' ' || text[p] == '\t')) ++p;
        if (p + klen <= le && text.compare(p, klen, key) == 0) {
            size_t q = p + klen; while (q < le && (text[q] == ' ' || text[q] == '\t')) ++q;
            if (q < le && text[q] == ':') {
                const char* b = text.c_str() + q + 1; char* e = nullptr; const double x = std::strtod(b, &e);
                if (e != b && e <= text.c_str() + le) { v = x; return true; }
                return false;                                   // the key is there but carries no number
            }
        }
        ls = le + 1;
    }
    return false;
}

int offline_init(OfflineManager& m) {                                   // nerf_manager.cu:16-38
    int rc = device_count(&m.n_dev); if (rc) return rc;
    rc = config_from_json(m.cfg_path.c_str(), m.cfg); if (rc) return rc;
    m.cfg.use_depth = m.use_depth ? 1 : 0;
    // (10 x 500 unless a test shortened the job: mon_set_option)
    m.outer_iters = (int)options().offline_outer; m.inner_iters = (int)options().offline_inner;
    return MON_OK;
}

int offline_read_dataset(OfflineManager& m) {                           // nerf_manager.cu:40-62, nerf_data.cu:27-235
    std::ifstream fc(m.dataset + "/config.yaml");
    if (!fc) { set_error("Failed to open settings file at: %s/config.yaml", m.dataset.c_str()); return MON_ERR_IO; }
    std::stringstream ss; ss << fc.rdbuf(); const std::string y = ss.str(); double v;
    const struct { const char* key; float* f; int* i; } fields[] = { { "Camera.fx", &m.fx, nullptr }, { "Camera.fy", &m.fy, nullptr }, { "Camera.cx", &m.cx,
            nullptr }, { "Camera.cy", &m.cy, nullptr },
                                                                     { "Camera.H", nullptr, &m.H }, { "Camera.W", nullptr, &m.W } };
    for (const auto& fd : fields) {
        if (!read_yaml_number(y, fd.key, v)) { set_error("config.yaml: %s missing or not a number", fd.key); return MON_ERR_IO; }
        if (fd.f) *fd.f = (float)v; else *fd.i = (int)v;
    }
    if (m.H <= 0 || m.W <= 0 || !(m.fx > 0.f) || !(m.fy > 0.f)) {
        set_error("config.yaml: bad intrinsics (fx %g fy %g H %d W %d)", (double)m.fx, (double)m.fy, m.H, m.W); return MON_ERR_IO; }
    if (m.use_depth) {
        if (!read_yaml_number(y, "DepthMapFactor", v)) { set_error("config.yaml: DepthMapFactor missing or not a number"); return MON_ERR_IO; }
        m.depth_scale = (float)v;
    }
    std::ifstream fi(m.dataset + "/img.txt"), fg(m.dataset + "/groundtruth.txt"); std::string line;
    if (!fi || !fg) { set_error("Load dataset error: img.txt / groundtruth.txt missing in %s", m.dataset.c_str()); return MON_ERR_IO; }
    std::getline(fi, line);                                             // skip comments
    while (std::getline(fi, line)) { if (line.empty()) continue; std::stringstream s2(line); std::string st, nm; s2 >> st >> nm;
        m.stamp_to_idx[st] = (uint32_t)m.names.size(); m.names.push_back(nm); m.stamps.push_back(st); }
    std::getline(fg, line);
    while (std::getline(fg, line)) {
        if (line.empty()) continue; std::stringstream s2(line); std::string st; float t[3], qx, qy, qz, qw;
        s2 >> st >> t[0] >> t[1] >> t[2] >> qx >> qy >> qz >> qw;
        float M[16]; pose_from_tq(t, qx, qy, qz, qw, M); m.poses.insert(m.poses.end(), M, M + 16);
    }
    const size_t n = m.poses.size() / 16;
    if (n == 0 || n != m.names.size()) { set_error("Load dataset error...No images (%zu poses, %zu image names)", n, m.names.size()); return MON_ERR_IO; }
    const size_t px = (size_t)m.H * m.W;
    for (int g = 0; g < m.n_dev; ++g) { Dataset* d = nullptr; int rc = dataset_create(g, m.H, m.W, m.fx, m.fy, m.cx, m.cy, (uint32_t)n, m.use_depth, &d);
        if (rc) return rc; m.ds.push_back(d); }
    // The PNGs of a batch of frames are decoded by a few host threads at once (inflate + unfilter: ~10 ms per 640x480 frame, the whole read of a sequence
    // otherwise), then handed to the device(s) in frame order.  The reference reads them one by one with cv::imread (nerf_data.cu:151-221).
    struct Decoded { std::vector<uint8_t> rgb, inst; std::vector<float> depth; std::string err; };
    const auto decode = [&](size_t i, Decoded& o) {
        PngImage c, s, z; o.err.clear(); o.rgb.resize(px * 3); o.inst.resize(px); if (m.use_depth) o.depth.resize(px);
        if (!png_read(m.dataset + "/rgb/" + m.names[i], c, o.err) || !png_read(m.dataset + "/instance/" + m.names[i], s, o.err)) return;
        if (c.width != m.W || c.height != m.H || s.width != m.W || s.height != m.H || s.bit_depth != 8) {
            o.err = "image " + m.names[i] + " does not match config.yaml"; return; }
        {   // cv::imread(IMREAD_COLOR), nerf_data.cu:158: gray is replicated to three channels, alpha dropped, 16-bit samples reduced to their high byte
            const size_t sb = c.bit_depth == 16 ? 2 : 1, pb = (size_t)c.channels * sb; const bool gray = c.channels < 3;
            for (size_t p = 0; p < px; ++p) {
                const uint8_t* q = &c.data[p * pb];
                o.rgb[3 * p] = q[0]; o.rgb[3 * p + 1] = gray ? q[0] : q[sb]; o.rgb[3 * p + 2] = gray ? q[0] : q[2 * sb];
            }
            // instance: IMREAD_UNCHANGED, first byte of every pixel (nerf_data.cu:196-207 reads the buffer as one u8 per pixel); OpenCV orders colour
            // pixels B,G,R, so for a colour / palette mask that byte is the blue sample
            const int ic = s.channels >= 3 ? 2 : 0;
            for (size_t p = 0; p < px; ++p) o.inst[p] = s.data[p * s.channels + ic];
        }
        if (m.use_depth) {
            if (!png_read(m.dataset + "/depth/" + m.names[i], z, o.err)) return;
            if (z.width != m.W || z.height != m.H || z.bit_depth != 16) {
                o.err = "depth image " + m.names[i] + " must be 16-bit " + std::to_string(m.W) + "x" + std::to_string(m.H); return; }
            // convertTo(CV_32FC1, mfDepthScale), nerf_data.cu:187
            for (size_t p = 0; p < px; ++p) o.depth[p] = (float)((z.data[2 * p * z.channels] << 8) | z.data[2 * p * z.channels + 1]) * m.depth_scale;
        }
    };
    unsigned nt = std::thread::hardware_concurrency(); nt = nt < 1u ? 1u : (nt > 8u ? 8u : nt);
    std::vector<Decoded> batch(nt);
    for (size_t base = 0; base < n; base += nt) {
        const size_t cnt = std::min<size_t>(nt, n - base);
        std::vector<std::thread> th;
        for (size_t t = 1; t < cnt; ++t) th.emplace_back([&, t] { decode(base + t, batch[t]); });
        decode(base, batch[0]);
        for (std::thread& t : th) t.join();
        for (size_t t = 0; t < cnt; ++t) {
            if (!batch[t].err.empty()) { set_error("%s", batch[t].err.c_str()); return MON_ERR_IO; }
            for (int g = 0; g < m.n_dev; ++g) {                         // PNG stores RGB; is_bgr = 0 (cv::imread would hand BGR)
                int rc = dataset_add_frame(m.ds[g], (uint32_t)(base + t), batch[t].rgb.data(), 3, 0, batch[t].inst.data(), m.use_depth ? batch[t].depth.data()
                        : nullptr, &m.poses[16 * (base + t)]); if (rc) return rc;
            }
        }
    }
    return MON_OK;
}

static void train_offline_thread(OfflineManager* m, OfflineObject* o) {  // NeRF::TrainOffline, nerf.cu:120-152
    { std::lock_guard<std::mutex> l(o->mu_model); o->rc = model_add_boxes(*o->model, o->boxes.data(), o->boxes.size()); }
    for (int i = 1; i <= m->outer_iters && o->rc == MON_OK; ++i) {
        std::lock_guard<std::mutex> l(o->mu_model);
        o->rc = model_train(*o->model, m->inner_iters, &o->last_loss, 7);
        if (o->rc == MON_OK) std::printf("Id: %d Step: %d loss: %f\n", o->id, i * m->inner_iters, o->last_loss);
        // GenerateMesh + TransCPUMesh, nerf.cu:138-145
        if (o->rc == MON_OK && i % 2 == 0) o->rc = model_generate_mesh(*o->model, m->mesh_res, m->mesh_thresh, nullptr, nullptr);
    }
    std::lock_guard<std::mutex> l(o->mu_model);
    if (o->rc == MON_OK && !m->mesh_dir.empty()) {                        // SaveMesh("./output/<id>.ply"), nerf.cu:148-149
        uint32_t nv = 0; model_mesh_counts(*o->model, &nv, nullptr, nullptr);
        ::mkdir(m->mesh_dir.c_str(), 0755);
        if (nv && model_save_mesh(*o->model, (m->mesh_dir + "/" + std::to_string(o->id) + ".ply").c_str()) != MON_OK) std::fprintf(stderr,
                "Id: %d mesh not saved: %s\n", o->id, last_error());
    }
    if (o->rc != MON_OK) o->err = last_error();            // the message is thread-local: hand it to whoever joins this thread
}

int offline_create_nerf(OfflineManager& m, const char* object_file) {    // nerf_manager.cu:64-92, nerf.cu:58-118
    std::ifstream f(object_file);
    if (!f) { set_error("object file error... %s", object_file); return MON_ERR_IO; }
    if (m.ds.empty()) { set_error("CreateNeRF before ReadDataset"); return MON_ERR_STATE; }
    OfflineObject* o = new OfflineObject(); o->id = (int)m.objs.size(); o->device = o->id % m.n_dev;
    std::string line; std::getline(f, line); std::getline(f, line); std::stringstream ss(line); float v[10] = {}; ss >> o->cls; for (float& x : v) ss >> x;
    if (ss.fail()) { delete o; set_error("object file error... %s: expected 'class tx ty tz qx qy qz qw ex ey ez' on line 2", object_file); return MON_ERR_IO; }
    float Two[16]; pose_from_tq(v, v[3], v[4], v[5], v[6], Two); rigid_inverse(Two, o->Tow);
    for (int a = 0; a < 3; ++a) { o->amin[a] = -v[7 + a]; o->amax[a] = v[7 + a]; }
    while (std::getline(f, line)) {
        if (line.empty()) continue; std::stringstream s2(line); std::string st; mon_frame_bbox b{}; s2 >> st >> b.x >> b.y >> b.h >> b.w;
        auto it = m.stamp_to_idx.find(st);
        if (it == m.stamp_to_idx.end()) { delete o; set_error("object file %s references unknown stamp %s", object_file, st.c_str()); return MON_ERR_IO; }
        b.FrameId = it->second; o->boxes.push_back(b); o->stamps.push_back(st);
    }
    int rc = model_create(m.ds[o->device], m.cfg, o->cls, o->Tow, o->amin, o->amax, &o->model);
    if (rc) { delete o; return rc; }
    o->handle.m = o->model;
    m.objs.push_back(o);
    m.threads.emplace_back(train_offline_thread, &m, o);                // one thread per model, nerf_manager.cu:89
    return MON_OK;
}

int offline_wait(OfflineManager& m) {                                    // nerf_manager.cu:94-102
    // (a second call after the threads were joined reports the objects' results again: the gathered test images wait on their own, whether or not the caller did)
    if (m.threads.empty() && !m.joined) { set_error("WaitThreadsEnd: no threads"); return MON_ERR_STATE; }
    for (auto& t : m.threads) if (t.joinable()) t.join();
    m.threads.clear(); m.joined = true;
    for (auto* o : m.objs) if (o->rc != MON_OK) { set_error("object %d: %s", o->id, o->err.c_str()); return o->rc; }
    return MON_OK;
}

// img.convertTo(CV_8UC3, 255) / depth.convertTo(CV_16UC1, 20000) / mask.convertTo(CV_8UC1, 255) + imwrite, nerf.cu:335-349 (saturating, round half to even)
static int write_render_pngs(const std::string& img_path, const std::string& depth_path, const std::string& mask_path, uint32_t w, uint32_t h,
                             const float* rgb, const float* depth, const float* mask) {
    const size_t n = (size_t)w * h; std::string err; std::vector<uint8_t> c8(3 * n), m8(n), d16(2 * n);
    for (size_t p = 0; p < 3 * n; ++p) { const float q = rgb[p] * 255.f; c8[p] = (uint8_t)(q < 0.f ? 0.f : (q > 255.f ? 255.f : std::nearbyint(q))); }
    for (size_t p = 0; p < n; ++p) {
        if (mask) { const float q = mask[p] * 255.f; m8[p] = (uint8_t)(q > 0.f ? (q > 255.f ? 255.f : std::nearbyint(q)) : 0.f); }
        const float q = depth[p] * 20000.f; const uint32_t u = (uint32_t)(q < 0.f ? 0.f : (q > 65535.f ? 65535.f : std::nearbyint(q)));
        d16[2 * p] = (uint8_t)(u >> 8); d16[2 * p + 1] = (uint8_t)u;
    }
    if (!png_write(img_path, (int)w, (int)h, 3, 8, c8.data(), err) || !png_write(depth_path, (int)w, (int)h, 1, 16, d16.data(), err) ||
        (mask && !png_write(mask_path, (int)w, (int)h, 1, 8, m8.data(), err))) { set_error("%s", err.c_str()); return MON_ERR_IO; }
    return MON_OK;
}

// NeRF_Model::GenerateToc nerf_model.cu:2186-2205: camera on a sphere of radius r around the object, looking at its centre
void generate_toc(float theta, float phi, float r, float* Toc16) {
    const double d2r = M_PI / 180.0;
    const float t[3] = { (float)(r * std::cos(phi * d2r) * std::cos(theta * d2r)), (float)(r * std::cos(phi * d2r) * std::sin(theta * d2r)),
            (float)(r * std::sin(phi * d2r)) };
    float z[3] = { -t[0], -t[1], -t[2] }; const float zn = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]); if (zn > 0.f) for (float& v : z) v /= zn;
    const float rv = (float)((theta + 90.0f) * d2r); float x[3] = { std::cos(rv), std::sin(rv), 0.f };
    const float xn = std::sqrt(x[0] * x[0] + x[1] * x[1]); if (xn > 0.f) for (float& v : x) v /= xn;
    float y[3] = { z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0] };
    const float yn = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]); if (yn > 0.f) for (float& v : y) v /= yn;
    const float M[16] = { x[0], x[1], x[2], 0.f, y[0], y[1], y[2], 0.f, z[0], z[1], z[2], 0.f, t[0], t[1], t[2], 1.f };
    std::memcpy(Toc16, M, 64);
}

// NeRF_Model::RenderVideo nerf_model.cu:1832-1990: 60 views (6 degree steps, 30 degree elevation), central half of the image, object-frame poses
static int render_video(Model& m, int H, int W, float radius, const std::string& img_dir, const std::string& depth_dir) {
    mon_frame_bbox box{ 0u, (uint32_t)(W / 4), (uint32_t)(H / 4), (uint32_t)(H / 2), (uint32_t)(W / 2) };
    const size_t n = (size_t)box.w * box.h; std::vector<float> rgb(3 * n), depth(n), mask(n);
    const int theta_num = 60; const float theta = 360 / (float)theta_num; float cur = 0.f;
    for (int i = 0; i < theta_num; ++i) {
        cur += theta; float Toc[16]; generate_toc(cur, 30.f, radius, Toc);
        int rc = model_render(m, box, Toc, 1, rgb.data(), depth.data(), mask.data(), 0); if (rc) return rc;
        rc = write_render_pngs(img_dir + "/" + std::to_string(i) + ".png", depth_dir + "/" + std::to_string(i) + ".png", "", box.w, box.h, rgb.data(),
                depth.data(), nullptr); if (rc) return rc;
    }
    return MON_OK;
}

// Eigen::Quaternionf(Matrix3f) for a column-major 4x4 (x y z w); trace / largest-diagonal branches
static void quat_from_pose(const float* M, float* q) {
    const float m00 = M[0], m11 = M[5], m22 = M[10], tr = m00 + m11 + m22;
    if (tr > 0.f) { float t = std::sqrt(tr + 1.f); q[3] = 0.5f * t; t = 0.5f / t; q[0] = (M[6] - M[9]) * t; q[1] = (M[8] - M[2]) * t;
        q[2] = (M[1] - M[4]) * t; }
    else {
        int i = 0; if (m11 > m00) i = 1; if (m22 > (i ? m11 : m00)) i = 2; const int j = (i + 1) % 3, k = (j + 1) % 3;
        auto R = [&](int r, int c) { return M[c * 4 + r]; };
        float t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.f); q[i] = 0.5f * t; t = 0.5f / t;
        q[3] = (R(k, j) - R(j, k)) * t; q[j] = (R(j, i) + R(i, j)) * t; q[k] = (R(k, i) + R(i, k)) * t;
    }
}
static void mat4_mul(const float* A, const float* B, float* C) {       // column-major C = A * B
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) { float v = 0.f; for (int k = 0; k < 4; ++k) v += A[k * 4 + r] * B[c * 4 + k]; C[c * 4 + r] = v; }
}
static void write_pose_line(std::ofstream& f, const std::string& stamp, const mon_frame_bbox& b, const float* Tow, const float* Twc) {
    float Toc[16], q[4]; mat4_mul(Tow, Twc, Toc); quat_from_pose(Toc, q);
    f << stamp << " " << b.x << " " << b.y << " " << b.h << " " << b.w << " " << Toc[12] << " " << Toc[13] << " " << Toc[14] << " " << q[0] << " " << q[1]
            << " " << q[2] << " " << q[3] << std::endl;
}

// (caller holds o.mu_model)
static int offline_save_mesh_locked(OfflineManager& m, OfflineObject& o, const std::string& root) {
    uint32_t n_mesh = 0; model_mesh_counts(*o.model, &n_mesh, nullptr, nullptr);
    if (n_mesh) {                                                        // "Save Object Mesh", nerf.cu:397-403
        int rc = model_generate_mesh(*o.model, m.mesh_res, m.mesh_thresh, nullptr, nullptr); if (rc) return rc;
        rc = model_save_mesh(*o.model, (root + "/obj.ply").c_str()); if (rc) return rc;
    }
    return MON_OK;
}

// Test images of one object for each of its training boxes: <out>/<id>/test_img|test_depth|test_mask/<stamp>.png,
// 8-bit colour, 16-bit depth x 20000, 8-bit mask (nerf.cu:335-349).
int offline_render_test(OfflineManager& m, int idx, const char* out_dir, int max_views) {
    if (idx < 0 || idx >= (int)m.objs.size()) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    OfflineObject* o = m.objs[idx]; const std::string root = std::string(out_dir) + "/" + std::to_string(o->id);
    ::mkdir(out_dir, 0755);                                              // mkdir -p of nerf.cu:258-283, one level at a time
    for (const char* sub : { "", "/test_img", "/test_depth", "/test_mask" }) ::mkdir((root + sub).c_str(), 0755);
    std::lock_guard<std::mutex> lm(o->mu_model);                         // may be called while the object is still training
    std::string err; const size_t nv = max_views > 0 && (size_t)max_views < o->boxes.size() ? (size_t)max_views : o->boxes.size();
    for (size_t i = 0; i < nv; ++i) {
        const mon_frame_bbox b = o->boxes[i]; const size_t n = (size_t)b.w * b.h;
        std::vector<float> rgb(3 * n), depth(n), mask(n);
        int rc = model_render(*o->model, b, &m.poses[16 * (size_t)b.FrameId], 0, rgb.data(), depth.data(), mask.data(), 0); if (rc) return rc;
        rc = write_render_pngs(root + "/test_img/" + o->stamps[i] + ".png", root + "/test_depth/" + o->stamps[i] + ".png",
                root + "/test_mask/" + o->stamps[i] + ".png", b.w, b.h, rgb.data(), depth.data(), mask.data());
        if (rc) return rc;
    }
    return offline_save_mesh_locked(m, *o, root);
}

// "Save Object Mesh" of RenderTestImg (nerf.cu:397-403) on its own: <out>/<id>/obj.ply when the object has a mesh (the gathered test images write the PNGs from
// one place and still need every object's mesh from its own device)
int offline_save_mesh(OfflineManager& m, int idx, const char* out_dir) {
    if (idx < 0 || idx >= (int)m.objs.size()) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    OfflineObject* o = m.objs[idx]; const std::string root = std::string(out_dir) + "/" + std::to_string(o->id);
    ::mkdir(out_dir, 0755); ::mkdir(root.c_str(), 0755);
    std::lock_guard<std::mutex> lm(o->mu_model);
    return offline_save_mesh_locked(m, *o, root);
}

int offline_destroy(OfflineManager* m) {
    if (!m) return MON_OK;
    for (auto& t : m->threads) if (t.joinable()) t.join();
    for (auto* o : m->objs) { if (o->model) model_destroy(o->model); delete o; }
    for (auto* d : m->ds) dataset_destroy(d);
    delete m; return MON_OK;
}

// ------------------------------------------------------------------ online manager
// nerf::NerfManagerOnline + the online half of nerf::NeRF (CORE/src/nerf_manager.cu:133-312, CORE/src/nerf.cu:155-253,
// 406-448): frames arrive one at a time from the SLAM frontend, every object has its own training thread that sleeps on a
// condition variable until new 2-D boxes arrive, trains `train_step` x TrainStepIterations iterations once it has more than
// 10 boxes, and does one last round when asked to finish.
struct OnlineObject {
    int id = 0, device = 0, cls = 0; float Tow[16]; float amin[3], amax[3];
    std::vector<mon_frame_bbox> boxes; size_t n_boxes = 0, n_uploaded = 0; int pending_train_step = 0, iterations = 500;
    std::mutex mu_boxes, mu_finish; std::condition_variable cond; bool finish = false;
    std::mutex* dataset_mutex = nullptr;
    Model* model = nullptr; mon_object handle{ nullptr }; float last_loss = 0.f; int train_calls = 0; int rc = 0; std::string err;
    int mesh_res = 64; float mesh_thresh = 2.0f;
    std::mutex mu_model;            // a Model is single-threaded: the training thread holds it per train slice / box upload / mesh call, the SLAM-side
                                    // calls (render, object_info, RenderNeRFsTest) while they use the model or read what the thread writes
    const std::atomic<int>* device_objects = nullptr;      // objects training on the same device (slice length, see train_sliced)
    std::atomic<int> waiters{ 0 };  // callers blocked on mu_model / the dataset mutex: the training thread lets them in between two slices
                                    // (std::mutex is not fair -- a thread that unlocks and relocks in a loop would starve them)
};

// Caller-side lock of one of an object's mutexes, announced to its training thread.
struct AnnouncedLock {
    std::unique_lock<std::mutex> lock;
    AnnouncedLock(OnlineObject* o, std::mutex& mu) : lock(mu, std::defer_lock) { o->waiters.fetch_add(1); lock.lock(); o->waiters.fetch_sub(1); }
};

struct OnlineManager {
    std::string cfg_path; bool use_depth = false; int iters = 500, n_dev = 0, next_dev = 0; mon_config cfg{};
    size_t n_images = 0; std::vector<Dataset*> ds; std::vector<std::vector<std::unique_ptr<std::mutex>>> ds_mutex;
    std::vector<std::unique_ptr<std::atomic<int>>> dev_objects;
    std::map<std::string, uint32_t> stamp_to_idx; std::vector<OnlineObject*> objs; std::vector<std::thread> threads;
    // objs grows on the SLAM thread (CreateNeRF) while the viewer looks objects up (DrawMesh, renders): look-ups copy the pointer under this lock
    std::mutex mu_objs;
    // host copy of the poses for train.txt (nerf.cu:369-373 reads them back from the device)
    int H = 0, W = 0; std::map<uint32_t, std::vector<float>> poses;
};

// The training thread tests `finish` and then waits on `cond` under mu_boxes; passing through mu_boxes between setting the flag and
// notifying closes the window in which the notification could fall between that test and the wait.
static void request_finish(OnlineObject* o) {
    { std::unique_lock<std::mutex> l(o->mu_finish); o->finish = true; }
    { std::unique_lock<std::mutex> l(o->mu_boxes); }
    o->cond.notify_all();
}
static bool online_check_finish(OnlineObject* o) { std::unique_lock<std::mutex> l(o->mu_finish); return o->finish; }

// Train_Step_Online takes the per-object dataset mutex around GenerateBatch of every iteration (nerf_model.cu:1675-1678), so the SLAM
// thread's NewFrameToDataset never waits longer than one batch generation.  Here an iteration is three stream-ordered launches without
// host involvement, so the mutexes are held for slices of up to kOnlineSlice iterations (~1.5 ms at base.json size, one host sync per
// slice) and anybody waiting for them is let in between two slices.  With several objects on a device the slices shrink (16 / n, at
// least option `online_slice_min` = 2): the other objects' queued work keeps the GPU busy across this thread's syncs (the device's training
// lanes, model.cpp, hold a chunk of every object), and a frame upload -- which takes every object's dataset mutex -- waits for short
// slices only.  Measured with 12 objects (tools/online_replay.py): a minimum of 4 / 8 / 16 iterations trains 5 % more in the same time but
// NewFrameToDataset's p99 goes 1.7 -> 2.1 / 5.4 / 10 ms (and the viewer's crop 0.9 -> 4.4 ms at 16).
static constexpr int kOnlineSlice = 16;
static int train_sliced(OnlineObject* o) {
    int rc = MON_OK;
    for (int done = 0; done < o->iterations && rc == MON_OK; ) {
        while (o->waiters.load() > 0) std::this_thread::yield();
        const int sharing = o->device_objects ? o->device_objects->load() : 1;
        int n = kOnlineSlice / (sharing > 0 ? sharing : 1); const long slice_min = kOnlineSliceMin;
        const int n_min = slice_min > 0 ? (int)slice_min : 2; if (n < n_min) n = n_min; if (n > o->iterations - done) n = o->iterations - done;
        std::unique_lock<std::mutex> dl(*o->dataset_mutex); std::lock_guard<std::mutex> lm(o->mu_model);
        rc = model_train(*o->model, n, &o->last_loss, 7); done += n;
        if (rc == MON_OK && done >= o->iterations) rc = model_publish_snapshot(*o->model);       // viewers see the end of every Train_Step_Online
    }
    return rc;
}

static OnlineObject* online_object(OnlineManager& m, size_t idx) { std::lock_guard<std::mutex> l(m.mu_objs);
    return idx < m.objs.size() ? m.objs[idx] : nullptr; }
static std::vector<OnlineObject*> online_objects(OnlineManager& m) { std::lock_guard<std::mutex> l(m.mu_objs); return m.objs; }

static void train_online_thread(OnlineObject* o) {                       // NeRF::TrainOnline, nerf.cu:187-253
    int train_step_count = 0;
    for (;;) {
        int train_step = 0;
        {
            std::unique_lock<std::mutex> lock(o->mu_boxes);
            if (o->n_boxes == o->n_uploaded && !online_check_finish(o)) o->cond.wait(lock);          // no update: wait (:209-212)
            if (o->n_boxes > o->n_uploaded) {
                std::lock_guard<std::mutex> lm(o->mu_model);
                o->rc = model_add_boxes(*o->model, o->boxes.data() + o->n_uploaded, o->n_boxes - o->n_uploaded);   // UpdateFrameIdAndBboxOnline
                o->n_uploaded = o->n_boxes; train_step = o->pending_train_step; o->pending_train_step = 0;
            }
        }
        if (o->rc == MON_OK && o->n_uploaded > 10) {                      // :223
            for (int i = 0; i < train_step && o->rc == MON_OK; ++i) {
                o->rc = train_sliced(o); ++train_step_count;
                std::lock_guard<std::mutex> lm(o->mu_model); ++o->train_calls;
                // :228-236
                if (o->rc == MON_OK && train_step_count % 2 == 0) o->rc = model_generate_mesh(*o->model, o->mesh_res, o->mesh_thresh, nullptr, nullptr);
            }
        }
        if (online_check_finish(o) || o->rc != MON_OK) break;
        ::usleep(3000);
    }
    if (o->rc == MON_OK && o->n_uploaded > 0) {                           // last time (:246)
        o->rc = train_sliced(o);
        std::lock_guard<std::mutex> lm(o->mu_model); ++o->train_calls;
        if (o->rc == MON_OK) o->rc = model_generate_mesh(*o->model, o->mesh_res, o->mesh_thresh, nullptr, nullptr);       // :247-249
    }
    if (o->rc != MON_OK) o->err = last_error();
    std::printf("Id: %d finished! \n", o->id);
}

int online_destroy(OnlineManager* m) {
    if (!m) return MON_OK;
    for (auto* o : m->objs) request_finish(o);
    for (auto& t : m->threads) if (t.joinable()) t.join();
    for (auto* o : m->objs) { if (o->model) model_destroy(o->model); delete o; }
    for (auto* d : m->ds) dataset_destroy(d);
    delete m; return MON_OK;
}

}  // namespace mon

using namespace mon;
struct mon_offline { OfflineManager* m; };
struct mon_online { OnlineManager* m; };
#define REQ(p) do { if (!(p)) { set_error("%s: null argument", __func__); return MON_ERR_ARG; } } while (0)

extern "C" {
int mon_offline_create(const char* dataset_path, const char* network_config_file, int use_dense_depth, mon_offline** out) {   // NerfManagerOffline ctor
    REQ(dataset_path); REQ(network_config_file); REQ(out);
    OfflineManager* m = new OfflineManager(); m->dataset = dataset_path; m->cfg_path = network_config_file; m->use_depth = use_dense_depth != 0;
    *out = new mon_offline{ m }; return MON_OK;
}
int mon_offline_init(mon_offline* h) { REQ(h); return offline_init(*h->m); }
int mon_offline_read_dataset(mon_offline* h) { REQ(h); return offline_read_dataset(*h->m); }
int mon_offline_create_nerf(mon_offline* h, const char* object_file) { REQ(h); REQ(object_file); return offline_create_nerf(*h->m, object_file); }
int mon_offline_wait_threads_end(mon_offline* h) { REQ(h); return offline_wait(*h->m); }
int mon_offline_n_objects(mon_offline* h, int* n) { REQ(h); REQ(n); *n = (int)h->m->objs.size(); return MON_OK; }
int mon_offline_object_loss(mon_offline* h, int idx, float* loss, int* device) { REQ(h); REQ(loss); if (idx < 0 || idx >= (int)h->m->objs.size()) {
        set_error("NeRF Idx error ..."); return MON_ERR_ARG; } *loss = h->m->objs[idx]->last_loss; if (device) *device = h->m->objs[idx]->device;
        return MON_OK; }
int mon_offline_render_test(mon_offline* h, int idx, const char* out_dir, int max_views) { REQ(h); REQ(out_dir);
    return offline_render_test(*h->m, idx, out_dir, max_views); }
int mon_offline_save_mesh(mon_offline* h, int idx, const char* out_dir) { REQ(h); REQ(out_dir); return offline_save_mesh(*h->m, idx, out_dir); }
// GetIntrinsics / GetAllTwc / NeRF::GetObjTow, GetBoundingBox, GetFrameIdAndBBox -- what MON/main.cpp:55,149-151,334-336 reads for its viewer
int mon_offline_get_intrinsics(mon_offline* h, float* fx, float* fy, float* cx, float* cy, int* H, int* W) {
    REQ(h); OfflineManager& m = *h->m; if (fx) *fx = m.fx; if (fy) *fy = m.fy; if (cx) *cx = m.cx; if (cy) *cy = m.cy; if (H) *H = m.H; if (W) *W = m.W;
    return MON_OK;
}
int mon_offline_get_poses(mon_offline* h, float* Twc16s, size_t capacity_frames, size_t* n_frames) {
    REQ(h); OfflineManager& m = *h->m; const size_t n = m.poses.size() / 16; if (n_frames) *n_frames = n;
    if (Twc16s) { if (capacity_frames < n) { set_error("get_poses: buffer holds %zu of %zu frames", capacity_frames, n); return MON_ERR_ARG;
            } std::memcpy(Twc16s, m.poses.data(), n * 64); }
    return MON_OK;
}
int mon_offline_object_meta(mon_offline* h, int idx, int* class_id, float* Tow16, float* aabb_min3, float* aabb_max3, mon_frame_bbox* boxes,
        size_t capacity_boxes, size_t* n_boxes) {
    REQ(h); if (idx < 0 || idx >= (int)h->m->objs.size()) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    OfflineObject* o = h->m->objs[idx];
    if (class_id) *class_id = o->cls; if (Tow16) std::memcpy(Tow16, o->Tow, 64); if (aabb_min3) std::memcpy(aabb_min3, o->amin, 12);
    if (aabb_max3) std::memcpy(aabb_max3, o->amax, 12);
    if (n_boxes) *n_boxes = o->boxes.size();
    if (boxes) { if (capacity_boxes < o->boxes.size()) { set_error("object_meta: buffer holds %zu of %zu boxes", capacity_boxes, o->boxes.size());
            return MON_ERR_ARG; } std::memcpy(boxes, o->boxes.data(), o->boxes.size() * sizeof(mon_frame_bbox)); }
    return MON_OK;
}
int mon_offline_object_stamp(mon_offline* h, int idx, size_t box_index, char* buf, size_t capacity) {
    REQ(h); REQ(buf); if (idx < 0 || idx >= (int)h->m->objs.size()) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    OfflineObject* o = h->m->objs[idx];
    if (box_index >= o->stamps.size() || capacity < o->stamps[box_index].size() + 1) { set_error("object_stamp: box index or buffer size");
        return MON_ERR_ARG; }
    std::memcpy(buf, o->stamps[box_index].c_str(), o->stamps[box_index].size() + 1); return MON_OK;
}
int mon_write_render_pngs(const char* img_path, const char* depth_path, const char* mask_path, uint32_t w, uint32_t h, const float* rgb, const float* depth,
        const float* mask) {
    if (!img_path || !depth_path || !rgb || !depth || (mask && !mask_path) || !w || !h) { set_error("write_render_pngs: bad argument"); return MON_ERR_ARG; }
    return write_render_pngs(img_path, depth_path, mask_path ? mask_path : "", w, h, rgb, depth, mask);
}
int mon_offline_set_output_dir(mon_offline* h, const char* dir) { REQ(h); h->m->mesh_dir = dir ? dir : ""; return MON_OK; }
int mon_offline_object(mon_offline* h, int idx, mon_object** borrowed) { REQ(h); REQ(borrowed); if (idx < 0 || idx >= (int)h->m->objs.size()) {
        set_error("NeRF Idx error ..."); return MON_ERR_ARG; } *borrowed = &h->m->objs[idx]->handle; return MON_OK; }
int mon_offline_destroy(mon_offline* h) { if (!h) return MON_OK; offline_destroy(h->m); delete h; return MON_OK; }

// ---- NerfManagerOnline
int mon_online_create(const char* network_config_file, int use_sparse_depth, int train_step_iterations, mon_online** out) {
    REQ(network_config_file); REQ(out);
    OnlineManager* m = new OnlineManager(); m->cfg_path = network_config_file; m->use_depth = use_sparse_depth != 0; m->iters = train_step_iterations;
    *out = new mon_online{ m }; return MON_OK;
}
int mon_online_init(mon_online* h) {                                      // nerf_manager.cu:136-158
    REQ(h); OnlineManager& m = *h->m;
    int rc = device_count(&m.n_dev); if (rc) return rc;
    rc = config_from_json(m.cfg_path.c_str(), m.cfg); if (rc) return rc;
    m.cfg.use_depth = m.use_depth ? 1 : 0; return MON_OK;
}
int mon_online_dataset_init(mon_online* h, float fx, float fy, float cx, float cy, int H, int W, size_t imgs) {   // :160-187
    REQ(h); OnlineManager& m = *h->m;
    if (m.n_dev == 0) { set_error("DatasetInit before Init"); return MON_ERR_STATE; }
    if (!m.ds.empty()) { set_error("DatasetInit called twice"); return MON_ERR_STATE; }
    m.n_images = imgs; m.ds_mutex.resize(m.n_dev); m.H = H; m.W = W;
    for (int g = 0; g < m.n_dev; ++g) m.dev_objects.emplace_back(new std::atomic<int>(0));
    // CreateNeRF runs on the SLAM thread later: take the ~8 ms per stream now
    for (int g = 0; g < m.n_dev; ++g) { const int rcs = stream_pool_reserve(g, 8); if (rcs) return rcs; }
    for (int g = 0; g < m.n_dev; ++g) { Dataset* d = nullptr; int rc = dataset_create(g, H, W, fx, fy, cx, cy, (uint32_t)imgs, m.use_depth, &d);
        if (rc) return rc; m.ds.push_back(d); }
    return MON_OK;
}
// :189-218
int mon_online_new_frame(mon_online* h, uint32_t img_id, const char* timestamp, const uint8_t* bgr, int channels, const uint8_t* instance, const float* depth,
        const float* Twc16) {
    REQ(h); REQ(timestamp); OnlineManager& m = *h->m;
    m.stamp_to_idx[timestamp] = img_id;                                   // nerf_data.cu:284
    if (Twc16) m.poses[img_id].assign(Twc16, Twc16 + 16);
    for (int g = 0; g < m.n_dev; ++g) {
        // The reference locks every object's dataset mutex on the device (its per-frame pointer table is rewritten, nerf_manager.cu:204-216).
        // Here a NEW frame id lands in a slab row that no kernel can be reading -- 2-D boxes naming it only arrive afterwards
        // (UpdateNeRFBbox) -- so training is only excluded when an id that is already in use is overwritten.
        const bool overwrite = img_id < m.ds[g]->max_frames && m.ds[g]->present[img_id];
        std::vector<std::unique_ptr<AnnouncedLock>> held;
        if (overwrite) for (auto* o : online_objects(m)) if (o->device == g) held.emplace_back(new AnnouncedLock(o, *o->dataset_mutex));
        const int rc = dataset_add_frame(m.ds[g], img_id, bgr, channels, 1, instance, m.use_depth ? depth : nullptr, Twc16);
        if (rc) return rc;
    }
    return MON_OK;
}
int mon_online_update_dataset(mon_online* h, uint32_t cur_id, uint32_t frame_num, const float* Twc16s) {   // :220-235
    REQ(h); REQ(Twc16s); OnlineManager& m = *h->m;
    if (frame_num == 0) return MON_OK;
    if (frame_num > cur_id) { set_error("UpdateDataset: %u frames before frame %u", frame_num, cur_id); return MON_ERR_ARG; }
    const uint32_t head = cur_id - frame_num;
    for (int g = 0; g < m.n_dev; ++g) {
        // every object's dataset mutex on the device (nerf_data.cu:345-347), announced so a training slice lets the update in; the candidate rays
        // an object prepared for its next iteration used the old poses: they are regenerated
        std::vector<std::unique_ptr<AnnouncedLock>> held, models;
        for (auto* o : online_objects(m)) if (o->device == g) held.emplace_back(new AnnouncedLock(o, *o->dataset_mutex));
        const int rc = dataset_update_poses(m.ds[g], head, frame_num, Twc16s); if (rc) return rc;
        for (auto* o : online_objects(m)) if (o->device == g && o->model) { AnnouncedLock lm(o, o->mu_model); o->model->next_ready = false; }
    }
    for (uint32_t i = 0; i < frame_num; ++i) m.poses[head + i].assign(Twc16s + 16 * (size_t)i, Twc16s + 16 * (size_t)i + 16);
    return MON_OK;
}
int mon_online_get_pose(mon_online* h, uint32_t frame_id, float* Twc16) {
    REQ(h); REQ(Twc16); auto it = h->m->poses.find(frame_id);
    if (it == h->m->poses.end()) { set_error("get_pose: frame %u has not been added", frame_id); return MON_ERR_ARG; }
    std::memcpy(Twc16, it->second.data(), 64); return MON_OK;
}
// :237-261 + SetAttributes nerf.cu:155-185
int mon_online_create_nerf(mon_online* h, int cls, const float* Tow16, const float* aabb_min3, const float* aabb_max3, size_t* idx_out) {
    REQ(h); REQ(Tow16); REQ(aabb_min3); REQ(aabb_max3); REQ(idx_out); OnlineManager& m = *h->m;
    if (m.ds.empty()) { set_error("CreateNeRF before DatasetInit"); return MON_ERR_STATE; }
    OnlineObject* o = new OnlineObject(); o->id = (int)m.objs.size(); o->device = m.next_dev; m.next_dev = (m.next_dev + 1) % m.n_dev; o->cls = cls;
    o->iterations = m.iters;
    std::memcpy(o->Tow, Tow16, 64);
    const float k = (cls == 41 || cls == 73) ? 1.2f : 1.1f;              // appropriately expand the 3-D box (nerf.cu:163-172)
    for (int a = 0; a < 3; ++a) { o->amin[a] = k * aabb_min3[a]; o->amax[a] = k * aabb_max3[a]; }
    o->boxes.resize(m.n_images);                                          // mFrameIdBbox.resize(maxnumBbox) :182
    m.ds_mutex[o->device].emplace_back(new std::mutex()); o->dataset_mutex = m.ds_mutex[o->device].back().get();
    int rc = model_create(m.ds[o->device], m.cfg, cls, o->Tow, o->amin, o->amax, &o->model);
    if (rc) { delete o; return rc; }
    o->handle.m = o->model; o->device_objects = m.dev_objects[o->device].get(); m.dev_objects[o->device]->fetch_add(1);
    { std::lock_guard<std::mutex> l(m.mu_objs); *idx_out = m.objs.size(); m.objs.push_back(o); }
    m.threads.emplace_back(train_online_thread, o);                      // thread per object, nerf_manager.cu:259
    return MON_OK;
}
// :298-303 + UpdateFrameBBox nerf.cu:406-421
int mon_online_update_nerf_bbox(mon_online* h, size_t idx, const mon_frame_bbox* boxes, size_t n, int train_step) {
    REQ(h);
    OnlineObject* o = online_object(*h->m, idx);
    if (!o) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    // the manager drops an EMPTY update before it reaches the NeRF (nerf_manager.cu:300-301): pending_train_step of an update
    // the object's thread has not consumed yet must not be overwritten by it
    if (n == 0) return MON_OK;
    REQ(boxes);
    std::unique_lock<std::mutex> lock(o->mu_boxes);
    if (o->n_boxes + n > o->boxes.size()) o->boxes.resize(o->n_boxes + n);
    for (size_t i = 0; i < n; ++i) o->boxes[o->n_boxes + i] = boxes[i];
    o->n_boxes += n; o->pending_train_step = train_step; o->cond.notify_all();
    return MON_OK;
}
int mon_online_get_frame_idx(mon_online* h, const char* timestamp, int* idx) {   // GetFrameIdx :288-296 (key = std::to_string(double) on the caller's side)
    REQ(h); REQ(timestamp); REQ(idx); auto it = h->m->stamp_to_idx.find(timestamp); *idx = it == h->m->stamp_to_idx.end() ? -1 : (int)it->second; return MON_OK;
}
int mon_online_wait_threads_end(mon_online* h) {                           // :263-278
    REQ(h); OnlineManager& m = *h->m;
    if (m.threads.empty()) { set_error("WaitThreadsEnd: no threads"); return MON_ERR_STATE; }
    for (auto* o : online_objects(m)) request_finish(o);                     // RequestFinish nerf.cu:443-448
    for (auto& t : m.threads) if (t.joinable()) t.join();
    m.threads.clear(); std::puts("All NeRF threads completed ...");
    for (auto* o : online_objects(m)) if (o->rc != MON_OK) { set_error("object %d: %s", o->id, o->err.c_str()); return o->rc; }
    return MON_OK;
}
int mon_online_object_info(mon_online* h, size_t idx, float* loss, int* train_calls, int* device, uint32_t* n_boxes) {
    REQ(h); OnlineObject* o = online_object(*h->m, idx); if (!o) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    AnnouncedLock lm(o, o->mu_model); if (loss) *loss = o->last_loss; if (train_calls) *train_calls = o->train_calls; if (device) *device = o->device;
    if (n_boxes) *n_boxes = (uint32_t)o->n_uploaded; return MON_OK;
}
// one view of RenderNeRFsTest :280-285
int mon_online_render(mon_online* h, size_t idx, mon_frame_bbox box, const float* Twc16, float* rgb, float* depth, float* mask) {
    REQ(h); OnlineObject* o = online_object(*h->m, idx); if (!o) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    // a viewer's render: the latest published inference weights on the object's inference stream -- no model mutex, nothing queued behind training
    if (model_render_snapshot(*o->model, box, Twc16, 0, rgb, depth, mask, nullptr) == MON_OK) return MON_OK;
    AnnouncedLock lm(o, o->mu_model);         // nothing published yet / no inference side: let in between two slices of a running training step
    return model_render(*o->model, box, Twc16, 0, rgb, depth, mask, 0);
}
// NerfManagerOnline::RenderNeRFsTest -> NeRF::RenderTestImg, nerf.cu:255-404: <out>/<id>/{test_img,test_depth,test_mask}/<stamp>.png,
// test.txt, train.txt (object-centric poses), 60-view video_img / video_depth, obj.ply
int mon_online_render_nerfs_test(mon_online* h, const char* out_path, size_t idx, const char* const* timestamps, const mon_frame_bbox* boxes,
        const float* Twcs16, size_t n, float radius) {
    REQ(h); REQ(out_path); OnlineManager& m = *h->m;
    if (online_objects(m).empty()) return MON_OK;                          // nerf_manager.cu:282
    if (!online_object(m, idx)) { set_error("NeRF Idx error ..."); return MON_ERR_ARG; }
    if (n && (!timestamps || !boxes || !Twcs16)) { set_error("RenderNeRFsTest: null argument"); return MON_ERR_ARG; }
    OnlineObject* o = online_object(m, idx); const std::string root = std::string(out_path) + "/" + std::to_string(o->id);
    std::vector<mon_frame_bbox> trained;                                  // snapshot of the uploaded boxes (lock order everywhere: mu_boxes, then mu_model)
    { std::lock_guard<std::mutex> lb(o->mu_boxes); trained.assign(o->boxes.begin(), o->boxes.begin() + (ptrdiff_t)o->n_uploaded); }
    AnnouncedLock lm(o, o->mu_model);
    ::mkdir(out_path, 0755);
    for (const char* sub : { "", "/test_img", "/test_depth", "/test_mask", "/video_img", "/video_depth" }) ::mkdir((root + sub).c_str(), 0755);
    std::ofstream f(root + "/test.txt");
    if (!f) { set_error("mkdir error: %s", root.c_str()); return MON_ERR_IO; }
    f << std::fixed << "#stamp  box.x  box.y  box.h  box.w  tx  ty  tz  qx  qy  qz  qw (object-centric)" << std::endl;
    std::printf("Render Object %d test imgs to %s/test_img ... please wait...\n", o->id, root.c_str());
    for (size_t i = 0; i < n; ++i) {
        const mon_frame_bbox b = boxes[i]; const float* Twc = Twcs16 + 16 * i; const std::string st = timestamps[i]; const size_t px = (size_t)b.w * b.h;
        write_pose_line(f, st, b, o->Tow, Twc);
        std::vector<float> rgb(3 * px), depth(px), mask(px);
        int rc = model_render(*o->model, b, Twc, 0, rgb.data(), depth.data(), mask.data(), 0); if (rc) return rc;
        rc = write_render_pngs(root + "/test_img/" + st + ".png", root + "/test_depth/" + st + ".png", root + "/test_mask/" + st + ".png", b.w, b.h,
                rgb.data(), depth.data(), mask.data()); if (rc) return rc;
    }
    f.close();
    f.open(root + "/train.txt");                                          // training data, nerf.cu:356-390
    f << std::fixed << "#class Bbox" << std::endl << o->cls << " " << o->amax[0] << " " << o->amax[1] << " " << o->amax[2] << " " << std::endl;
    f << "#stamp box.x box.y box.h box.w  tx  ty  tz  qx  qy  qz  qw (object-centric)" << std::endl;
    for (const mon_frame_bbox& b : trained) {
        std::string st;
        for (const auto& kv : m.stamp_to_idx) if (kv.second == b.FrameId) { st = kv.first; break; }
        auto it = m.poses.find(b.FrameId); if (it == m.poses.end()) continue;
        write_pose_line(f, st, b, o->Tow, it->second.data());
    }
    f.close();
    std::printf("Render Object %d 360 video imgs to %s/video_img ... please wait...\n", o->id, root.c_str());
    int rc = render_video(*o->model, m.H, m.W, radius, root + "/video_img", root + "/video_depth"); if (rc) return rc;
    uint32_t n_mesh = 0; model_mesh_counts(*o->model, &n_mesh, nullptr, nullptr);
    if (n_mesh) {                                                         // "Save Object Mesh", nerf.cu:397-403
        std::puts("Save Object Mesh ... please wait...");
        rc = model_generate_mesh(*o->model, o->mesh_res, o->mesh_thresh, nullptr, nullptr); if (rc) return rc;
        rc = model_save_mesh(*o->model, (root + "/obj.ply").c_str()); if (rc) return rc;
    }
    return MON_OK;
}
int mon_generate_toc(float theta_deg, float phi_deg, float radius, float* Toc16) { REQ(Toc16); generate_toc(theta_deg, phi_deg, radius, Toc16); return MON_OK; }
int mon_online_object(mon_online* h, size_t idx, mon_object** borrowed) { REQ(h); REQ(borrowed); OnlineObject* o = online_object(*h->m, idx); if (!o) {
        set_error("NeRF Idx error ..."); return MON_ERR_ARG; } *borrowed = &o->handle; return MON_OK; }
int mon_online_destroy(mon_online* h) { if (!h) return MON_OK; online_destroy(h->m); delete h; return MON_OK; }

int mon_png_read(const char* path, int* width, int* height, int* channels, int* bit_depth, uint8_t* pixels, size_t capacity) {
    REQ(path); REQ(width); REQ(height); REQ(channels); REQ(bit_depth);
    PngImage img; std::string err;
    if (!png_read(path, img, err)) { set_error("%s", err.c_str()); return MON_ERR_IO; }
    *width = img.width; *height = img.height; *channels = img.channels; *bit_depth = img.bit_depth;
    if (pixels) { if (capacity < img.data.size()) { set_error("png_read: buffer too small"); return MON_ERR_ARG;
            } std::memcpy(pixels, img.data.data(), img.data.size()); }
    return MON_OK;
}
int mon_png_write(const char* path, int width, int height, int channels, int bit_depth, const uint8_t* pixels_big_endian) {
    REQ(path); REQ(pixels_big_endian); std::string err;
    if (!png_write(path, width, height, channels, bit_depth, pixels_big_endian, err)) { set_error("%s", err.c_str()); return MON_ERR_IO; }
    return MON_OK;
}
}
