// compat/mon_compat.cpp -- the reference's manager / NeRF classes as a thin layer over the C ABI's manager entry points
// (include/mon_core.h: mon_offline_*, mon_online_*, mon_object_copy_mesh).  Compiled inside the RO-MAP tree (Eigen, OpenCV core
// for the cv::Mat signatures, GLEW for DrawCPUMesh); see INTEGRATION.md.  Threads, datasets, PNG I/O and training live in
// libmon_core.so; fatal errors keep the reference's convention: message on cerr, exit(0) (nerf_manager.cu:21-25).
// tests/test_compat_shim.py compiles this file against minimal stand-ins of the three third-party headers and runs the consumers'
// call sequences (MON/main.cpp:322-340, REF/src/System.cc:120-138,567,610, LocalMapping.cc:1172-1280) on the device.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <type_traits>
#include "nerf_manager.h"

namespace nerf {

// What the C ABI relies on when this layer hands the reference's types across it.  With the stand-in headers of tests/compat_stubs/ these hold by construction;
// tests/compat_real_deps.sh compiles this file against the machine's own Eigen3 / OpenCV / GLEW so that they are checked against the real types.
// FrameIdAndBbox (CORE/include/common.h:18-23) IS mon_frame_bbox: vectors of it are passed as mon_frame_bbox arrays (h BEFORE w)
static_assert(sizeof(FrameIdAndBbox) == 20 && sizeof(FrameIdAndBbox) == sizeof(mon_frame_bbox), "FrameIdAndBbox must stay layout-compatible with mon_frame_bbox");
static_assert(offsetof(FrameIdAndBbox, FrameId) == 0 && offsetof(FrameIdAndBbox, x) == 4 && offsetof(FrameIdAndBbox, y) == 8 && offsetof(FrameIdAndBbox, h) == 12
        && offsetof(FrameIdAndBbox, w) == 16, "FrameIdAndBbox: {FrameId, x, y, h, w}, four bytes each");
static_assert(offsetof(mon_frame_bbox, h) == 12 && offsetof(mon_frame_bbox, w) == 16, "mon_frame_bbox: h before w like the reference");
static_assert(std::is_standard_layout<FrameIdAndBbox>::value && std::is_trivially_copyable<FrameIdAndBbox>::value, "FrameIdAndBbox is a POD");
// BoundingBox (common.h:25-30): two Vector3f, read through .data() as 3 + 3 floats; EIGEN_MAKE_ALIGNED_OPERATOR_NEW adds operators, no members
static_assert(sizeof(Eigen::Vector3f) == 12, "Eigen::Vector3f = three packed floats");
static_assert(sizeof(BoundingBox) == 24, "BoundingBox = {Vector3f min, max} without padding");
// poses cross the boundary as 16 floats, COLUMN-major (Twc16 / Tow16 of include/mon_core.h): Eigen's default storage order, read through .data()
static_assert(sizeof(Eigen::Matrix4f) == 64 && !Eigen::Matrix4f::IsRowMajor, "Eigen::Matrix4f = 16 floats, column-major");

static void die(const char* what) { std::cerr << what << ": " << mon_last_error() << std::endl; exit(0); }

void NeRF::DrawCPUMesh() {                                                    // nerf.cu:484-507
    CPUMeshData& cm = mCPUMeshData;
    std::unique_lock<std::mutex> lock(cm.mesh_mutex, std::try_to_lock);
    if (!lock.owns_lock() || !mpObject) return;
    // refresh the host copy when the training thread has published a new mesh (TransCPUMesh fills CPUMeshData there, nerf.cu:138-145)
    uint64_t gen = 0; mon_object_mesh_generation(mpObject, &gen);
    if (gen != mMeshGeneration) {
        uint32_t nv = (uint32_t)(cm.verts.size() / 3), ni = (uint32_t)cm.indices.size(), need_v = 0, need_r = 0, need_i = 0;
        int rc = mon_object_copy_mesh(mpObject, nv, ni, cm.verts.data(), cm.normals.data(), cm.colors.data(), cm.indices.data(), &need_v, &need_r, &need_i, 1);
        if (rc == MON_ERR_ARG && (need_v > nv || need_i > ni)) {                   // the mesh grew: make room and try once more
            cm.verts.resize(3 * (size_t)need_v); cm.normals.resize(3 * (size_t)need_v); cm.colors.resize(3 * (size_t)need_v); cm.indices.resize(need_i);
            rc = mon_object_copy_mesh(mpObject, need_v, need_i, cm.verts.data(), cm.normals.data(), cm.colors.data(), cm.indices.data(), &need_v, &need_r,
                    &need_i, 1);
        }
        if (rc == MON_OK) {
            cm.verts.resize(3 * (size_t)need_v); cm.normals.resize(3 * (size_t)need_v); cm.colors.resize(3 * (size_t)need_v); cm.indices.resize(need_i);
            cm.have_reslult = true; mMeshGeneration = gen;
        }
    }
    if (!cm.have_reslult || cm.indices.empty()) return;
    glEnableClientState(GL_VERTEX_ARRAY); glEnableClientState(GL_NORMAL_ARRAY); glEnableClientState(GL_COLOR_ARRAY);
    glVertexPointer(3, GL_FLOAT, 0, cm.verts.data()); glColorPointer(3, GL_UNSIGNED_BYTE, 0, cm.colors.data());
    glNormalPointer(GL_FLOAT, 0, cm.normals.data());
    glDrawElements(GL_TRIANGLES, (GLsizei)cm.indices.size(), GL_UNSIGNED_INT, cm.indices.data());
    glDisableClientState(GL_VERTEX_ARRAY); glDisableClientState(GL_NORMAL_ARRAY); glDisableClientState(GL_COLOR_ARRAY);
}

vector<Eigen::Matrix4f> NeRF::GetTwc() {                                      // nerf.cu:450-462
    vector<Eigen::Matrix4f> out;
    if (mpOffline) {
        size_t n = 0; mon_offline_get_poses(mpOffline, nullptr, 0, &n); std::vector<float> flat(16 * n);
        if (n && mon_offline_get_poses(mpOffline, flat.data(), n, &n)) die("GetTwc");
        for (const FrameIdAndBbox& b : mFrameIdBbox) {
            Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
            if (b.FrameId < n) std::memcpy(T.data(), &flat[16 * (size_t)b.FrameId], 64);
            out.push_back(T);
        }
    } else if (mpOnline) {
        for (const FrameIdAndBbox& b : mFrameIdBbox) {
            Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
            if (mon_online_get_pose(mpOnline, b.FrameId, T.data())) die("GetTwc");
            out.push_back(T);
        }
    }
    return out;
}

// ------------------------------------------------------------------ offline manager (nerf_manager.cu:9-131)
NerfManagerOffline::NerfManagerOffline(const string datasetPath, const string cfg, bool useDenseDepth)
    : msNetworkConfigFile(cfg), msDatasetPath(datasetPath), mbUseDenseDepth(useDenseDepth) {
    if (mon_offline_create(datasetPath.c_str(), cfg.c_str(), useDenseDepth ? 1 : 0, &mpManager)) die("NerfManagerOffline");
}
NerfManagerOffline::~NerfManagerOffline() { mon_offline_destroy(mpManager); }
// "Can not Detect GPU" / config errors are fatal (:21-25)
bool NerfManagerOffline::Init() { if (mon_offline_init(mpManager)) die("Init"); return true; }
bool NerfManagerOffline::ReadDataset() {                                                                              // nerf_data.cu:27-235
    if (mon_offline_read_dataset(mpManager)) { std::cerr << "Load dataset error: " << mon_last_error() << std::endl; return false; }
    return true;
}
bool NerfManagerOffline::CreateNeRF(const string objectFile) {                                                        // nerf_manager.cu:64-92, nerf.cu:58-118
    if (mon_offline_create_nerf(mpManager, objectFile.c_str())) { std::cerr << "Create NeRF error: " << mon_last_error() << std::endl; return false; }
    auto n = std::make_shared<NeRF>(); const int idx = (int)mvpNeRFs.size(); n->mId = idx; n->mpOffline = mpManager;
    size_t nb = 0;
    if (mon_offline_object_meta(mpManager, idx, &n->mClass, n->mObjTow.data(), n->mBoundingBox.min.data(), n->mBoundingBox.max.data(), nullptr, 0,
            &nb)) die("CreateNeRF");
    n->mFrameIdBbox.resize(nb);
    if (mon_offline_object_meta(mpManager, idx, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<mon_frame_bbox*>(n->mFrameIdBbox.data()), nb, &nb) ||
        mon_offline_object(mpManager, idx, &n->mpObject)) die("CreateNeRF");
    n->mInstanceId = (uint8_t)n->mClass; n->mnBbox = nb;
    mvpNeRFs.push_back(n);
    return true;
}
bool NerfManagerOffline::WaitThreadsEnd() {                                                                           // nerf_manager.cu:94-102
    if (mvpNeRFs.empty()) return false;
    if (mon_offline_wait_threads_end(mpManager)) die("WaitThreadsEnd");
    return true;
}
vector<Eigen::Matrix4f> NerfManagerOffline::GetAllTwc() {
    size_t n = 0; mon_offline_get_poses(mpManager, nullptr, 0, &n);
    std::vector<float> flat(16 * n); vector<Eigen::Matrix4f> out(n);
    if (n && mon_offline_get_poses(mpManager, flat.data(), n, &n)) die("GetAllTwc");
    for (size_t i = 0; i < n; ++i) std::memcpy(out[i].data(), &flat[16 * i], 64);                                      // both sides column-major
    return out;
}
void NerfManagerOffline::GetIntrinsics(float& fx, float& fy, float& cx, float& cy) {
    mon_offline_get_intrinsics(mpManager, &fx, &fy, &cx, &cy, nullptr, nullptr); }

// ------------------------------------------------------------------ online manager (nerf_manager.cu:133-312)
NerfManagerOnline::NerfManagerOnline(const string cfg, bool UseSparseDepth, int iters)
    : mNetworkConfigFile(cfg), mbUseSparseDepth(UseSparseDepth), mnTrainStepIterations(iters) {
    if (mon_online_create(cfg.c_str(), UseSparseDepth ? 1 : 0, iters, &mpManager)) die("NerfManagerOnline");
}
NerfManagerOnline::~NerfManagerOnline() { mon_online_destroy(mpManager); }
bool NerfManagerOnline::Init() { if (mon_online_init(mpManager)) die("Init"); return true; }
void NerfManagerOnline::DatasetInit(float fx, float fy, float cx, float cy, int H, int W, size_t imgs) {
    if (mon_online_dataset_init(mpManager, fx, fy, cx, cy, H, W, imgs)) die("InitDataToGPU");
}
void NerfManagerOnline::NewFrameToDataset(unsigned int imgId, const string stamp, cv::Mat& img, cv::Mat& instance, const cv::Mat& depth,
        const Eigen::Matrix4f& pose) {
    // 8-bit BGR(A) colour, 8-bit instance ids, CV_32FC1 z-depth in metres with 0 = no sample (nerf_data.cu:279-339; callers pass clones,
    // LocalMapping.cc:1112-1113)
    const cv::Mat c = img.isContinuous() ? img : img.clone(), s = instance.isContinuous() ? instance : instance.clone();
    const cv::Mat z = (mbUseSparseDepth && !depth.isContinuous()) ? depth.clone() : depth;
    if (mon_online_new_frame(mpManager, imgId, stamp.c_str(), c.data, c.channels(), s.data, mbUseSparseDepth ? z.ptr<float>() : nullptr,
            pose.data())) die("FrameDataToGPU");
}
size_t NerfManagerOnline::CreateNeRF(const int Class, const Eigen::Matrix4f& ObjTow, const nerf::BoundingBox& box) {
    // the 1.1x / 1.2x inflation of SetAttributes (nerf.cu:163-172) is applied behind the C ABI
    size_t idx = 0;
    if (mon_online_create_nerf(mpManager, Class, ObjTow.data(), box.min.data(), box.max.data(), &idx)) die("Create NeRF error");
    auto n = std::make_shared<NeRF>(); n->mId = (int)idx; n->mClass = Class; n->mInstanceId = (uint8_t)Class; n->mObjTow = ObjTow; n->mpOnline = mpManager;
    const float k = (Class == 41 || Class == 73) ? 1.2f : 1.1f;
    for (int a = 0; a < 3; ++a) { n->mBoundingBox.min.data()[a] = k * box.min.data()[a]; n->mBoundingBox.max.data()[a] = k * box.max.data()[a]; }
    if (mon_online_object(mpManager, idx, &n->mpObject)) die("Create NeRF error");
    mvpNeRFs.push_back(n);
    return idx;
}
int NerfManagerOnline::GetFrameIdx(double t) {                                // nerf_manager.cu:288-296: the key is std::to_string(double) on both sides
    int idx = -1; mon_online_get_frame_idx(mpManager, std::to_string(t).c_str(), &idx); return idx;
}
void NerfManagerOnline::UpdateNeRFBbox(const size_t idx, const vector<nerf::FrameIdAndBbox>& v, const int train_step) {
    if (v.empty()) return;
    if (idx < mvpNeRFs.size()) { mvpNeRFs[idx]->mFrameIdBbox.insert(mvpNeRFs[idx]->mFrameIdBbox.end(), v.begin(), v.end());
        mvpNeRFs[idx]->mnBbox = mvpNeRFs[idx]->mFrameIdBbox.size(); }
    if (mon_online_update_nerf_bbox(mpManager, idx, reinterpret_cast<const mon_frame_bbox*>(v.data()), v.size(), train_step)) die("UpdateNeRFBbox");
}
void NerfManagerOnline::UpdateDataset(unsigned int CurId, unsigned int FrameNum, const vector<Eigen::Matrix4f>& Poses) {      // nerf_manager.cu:220-235
    if (FrameNum == 0 || Poses.size() < FrameNum) return;
    std::vector<float> flat(16 * (size_t)FrameNum);
    for (unsigned int i = 0; i < FrameNum; ++i) std::memcpy(&flat[16 * (size_t)i], Poses[i].data(), 64);
    if (mon_online_update_dataset(mpManager, CurId, FrameNum, flat.data())) die("UpdateDataGPU");
}
void NerfManagerOnline::DrawMesh(size_t idx) { if (idx < mvpNeRFs.size()) mvpNeRFs[idx]->DrawCPUMesh(); }
bool NerfManagerOnline::WaitThreadsEnd() {
    if (mvpNeRFs.empty()) return false;
    if (mon_online_wait_threads_end(mpManager)) die("WaitThreadsEnd");
    return true;
}
void NerfManagerOnline::RenderNeRFsTest(const string out_path, const size_t Idx, const vector<string>& ts, const vector<FrameIdAndBbox>& vb,
                                        const vector<Eigen::Matrix4f>& vT, const float radius) {
    if (mvpNeRFs.empty()) return;                                             // nerf_manager.cu:282
    std::vector<const char*> stamps(ts.size()); std::vector<float> poses(16 * vT.size());
    for (size_t i = 0; i < ts.size(); ++i) stamps[i] = ts[i].c_str();
    for (size_t i = 0; i < vT.size(); ++i) std::memcpy(&poses[16 * i], vT[i].data(), 64);
    if (mon_online_render_nerfs_test(mpManager, out_path.c_str(), Idx, stamps.data(), reinterpret_cast<const mon_frame_bbox*>(vb.data()), poses.data(),
            ts.size(), radius)) die("RenderNeRFsTest");
}

}  // namespace nerf
