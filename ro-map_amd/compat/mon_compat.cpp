// compat/mon_compat.cpp -- the reference's manager / NeRF classes implemented on the C ABI (include/mon_core.h).
// Compiled inside the RO-MAP tree (needs Eigen, OpenCV, GLEW); see INTEGRATION.md.  Behaviour follows
// CORE/src/nerf_manager.cu and CORE/src/nerf.cu: thread per object, round-robin device choice, fixed 10 x 500
// offline iterations, cond-var driven online training gated on more than 10 boxes, fatal errors = cerr + exit(0).
#include <unistd.h>
#include <fstream>
#include <iostream>
#include <sstream>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/imgproc.hpp>
#include "nerf_manager.h"

namespace nerf {

static void die(const char* what) { std::cerr << what << ": " << mon_last_error() << std::endl; exit(0); }   // nerf_manager.cu:21-25

NeRF::~NeRF() { if (mpObject) mon_object_destroy(mpObject); }
vector<FrameIdAndBbox> NeRF::GetFrameIdAndBBox() { return vector<FrameIdAndBbox>(mFrameIdBbox.begin(), mFrameIdBbox.begin() + mnBbox); }

void NeRF::UpdateCPUMesh() {                                                  // GenerateMesh + TransCPUMesh, nerf.cu:138-145 / 228-236
    uint32_t nv = 0, ni = 0;
    if (mon_object_generate_mesh(mpObject, 64, 2.0f, &nv, &ni)) die("GenerateMesh");      // marching_cubes.h:30-31
    std::unique_lock<std::mutex> lock(mCPUMeshData.mesh_mutex);
    mCPUMeshData.verts.resize(3 * nv); mCPUMeshData.normals.resize(3 * nv); mCPUMeshData.colors.resize(3 * nv); mCPUMeshData.indices.resize(ni);
    mon_object_get_mesh(mpObject, mCPUMeshData.verts.data(), mCPUMeshData.normals.data(), mCPUMeshData.colors.data(), mCPUMeshData.indices.data(), 0);
    mCPUMeshData.have_reslult = true;
}

void NeRF::DrawCPUMesh() {                                                    // nerf.cu:484-507
    std::unique_lock<std::mutex> lock(mCPUMeshData.mesh_mutex, std::try_to_lock);
    if (!lock.owns_lock() || !mCPUMeshData.have_reslult) return;
    glEnableClientState(GL_VERTEX_ARRAY); glEnableClientState(GL_NORMAL_ARRAY); glEnableClientState(GL_COLOR_ARRAY);
    glVertexPointer(3, GL_FLOAT, 0, mCPUMeshData.verts.data()); glColorPointer(3, GL_UNSIGNED_BYTE, 0, mCPUMeshData.colors.data());
    glNormalPointer(GL_FLOAT, 0, mCPUMeshData.normals.data());
    glDrawElements(GL_TRIANGLES, (GLsizei)mCPUMeshData.indices.size(), GL_UNSIGNED_INT, mCPUMeshData.indices.data());
    glDisableClientState(GL_VERTEX_ARRAY); glDisableClientState(GL_NORMAL_ARRAY); glDisableClientState(GL_COLOR_ARRAY);
}

void NeRF::TrainOffline(const int iterations) {                               // nerf.cu:120-152
    mon_object_add_boxes(mpObject, reinterpret_cast<const mon_frame_bbox*>(mFrameIdBbox.data()), mnBbox);
    for (int i = 1; i <= iterations; ++i) {
        float loss = 0.f;
        if (mon_object_train(mpObject, 500, &loss)) die("Train_Step");       // nerf_model.cu:1635
        std::cout << "Id: " << mId << " Step: " << i * 500 << " loss: " << loss << std::endl;
        if (i % 2 == 0) UpdateCPUMesh();
    }
    mon_object_save_mesh(mpObject, ("./output/" + std::to_string(mId) + ".ply").c_str());   // nerf.cu:148-149
}

void NeRF::UpdateFrameBBox(const vector<FrameIdAndBbox>& v, const int train_step) {   // nerf.cu:406-421
    std::unique_lock<std::mutex> lock(mUpdateBbox);
    for (size_t i = 0; i < v.size(); ++i) mFrameIdBbox[mnBbox + i] = v[i];
    mnBbox += v.size(); mnTrainStep = train_step; mCond.notify_all();
}
void NeRF::RequestFinish() {                                                 // nerf.cu:443-448; passing through mUpdateBbox keeps the notification from falling between TrainOnline's test and its wait
    { std::unique_lock<std::mutex> lock(mFinishMutex); mbFinishRequested = true; }
    { std::unique_lock<std::mutex> lock(mUpdateBbox); }
    mCond.notify_all();
}
bool NeRF::CheckFinish() { std::unique_lock<std::mutex> lock(mFinishMutex); return mbFinishRequested; }

void NeRF::TrainOnline() {                                                    // nerf.cu:187-253
    int train_step_count = 0;
    while (true) {
        int train_step = 0;
        {
            std::unique_lock<std::mutex> lock(mUpdateBbox);
            if (mnBbox == mnUploaded && !CheckFinish()) mCond.wait(lock);
            if (mnBbox > mnUploaded) {
                mon_object_add_boxes(mpObject, reinterpret_cast<const mon_frame_bbox*>(mFrameIdBbox.data() + mnUploaded), mnBbox - mnUploaded);
                mnUploaded = mnBbox; train_step = mnTrainStep; mnTrainStep = 0;
            }
        }
        if (mnUploaded > 10)
            for (int i = 0; i < train_step; ++i) {
                std::unique_lock<std::mutex> dl(*mpDatasetMutex);             // GenerateBatch under the dataset mutex (nerf_model.cu:1675-1678)
                float loss = 0.f; mon_object_train(mpObject, mnIteration, &loss); dl.unlock();
                if (++train_step_count % 2 == 0) UpdateCPUMesh();
            }
        if (CheckFinish()) break;
        usleep(3000);
    }
    float loss = 0.f; mon_object_train(mpObject, mnIteration, &loss); UpdateCPUMesh();
    std::cout << "Id: " << mId << " finished! " << std::endl;
}

void NeRF::RenderTestImg(const string out_path, const vector<string>& stamps, const vector<Eigen::Matrix4f>& Twcs, const vector<FrameIdAndBbox>& boxes, const float) {
    const string dir = out_path + "/" + std::to_string(mId);                  // nerf.cu:255-349 (test images + mesh; the 360 video is a "next" row)
    if (system(("mkdir -p " + dir + "/test_img " + dir + "/test_depth " + dir + "/test_mask").c_str()) != 0) throw std::runtime_error("mkdir error");
    for (size_t i = 0; i < stamps.size(); ++i) {
        const FrameIdAndBbox& b = boxes[i];
        cv::Mat img(b.h, b.w, CV_32FC3), depth(b.h, b.w, CV_32FC1), mask(b.h, b.w, CV_32FC1);
        mon_frame_bbox mb{ b.FrameId, b.x, b.y, b.h, b.w };
        if (mon_object_render(mpObject, mb, Twcs[i].data(), 0, img.ptr<float>(), depth.ptr<float>(), mask.ptr<float>(), 0)) die("Render");
        cv::cvtColor(img, img, cv::COLOR_RGB2BGR); img.convertTo(img, CV_8UC3, 255); cv::imwrite(dir + "/test_img/" + stamps[i] + ".png", img);
        depth.convertTo(depth, CV_16UC1, 20000); cv::imwrite(dir + "/test_depth/" + stamps[i] + ".png", depth);
        mask.convertTo(mask, CV_8UC1, 255); cv::imwrite(dir + "/test_mask/" + stamps[i] + ".png", mask);
    }
    if (mCPUMeshData.have_reslult) { UpdateCPUMesh(); mon_object_save_mesh(mpObject, (dir + "/obj.ply").c_str()); }   // nerf.cu:397-403
}

// ------------------------------------------------------------------ offline manager (nerf_manager.cu:9-131)
NerfManagerOffline::NerfManagerOffline(const string datasetPath, const string cfg, bool useDenseDepth)
    : msNetworkConfigFile(cfg), msDatasetPath(datasetPath), mbUseDenseDepth(useDenseDepth) {}
bool NerfManagerOffline::Init() {
    if (mon_device_count(&mNumGPU)) die("Can not Detect GPU");
    if (mon_config_from_json(msNetworkConfigFile.c_str(), &mConfig)) die("Read Network Config error");
    mConfig.use_depth = mbUseDenseDepth; return true;
}
bool NerfManagerOffline::ReadDataset() {                                      // nerf_data.cu:27-235
    cv::FileStorage fs(msDatasetPath + "/config.yaml", cv::FileStorage::READ);
    if (!fs.isOpened()) { std::cerr << "Failed to open settings file" << std::endl; exit(0); }
    mfx = fs["Camera.fx"]; mfy = fs["Camera.fy"]; mcx = fs["Camera.cx"]; mcy = fs["Camera.cy"]; mH = fs["Camera.H"]; mW = fs["Camera.W"];
    const float depthScale = mbUseDenseDepth ? (float)fs["DepthMapFactor"] : 1.f;
    std::ifstream fi(msDatasetPath + "/img.txt"), fg(msDatasetPath + "/groundtruth.txt"); string line; vector<string> names;
    std::getline(fi, line);
    while (std::getline(fi, line)) { if (line.empty()) continue; std::stringstream ss(line); string st, nm; ss >> st >> nm; mStampToIdx[st] = (uint32_t)names.size(); names.push_back(nm); }
    std::getline(fg, line);
    while (std::getline(fg, line)) {
        if (line.empty()) continue; std::stringstream ss(line); string st; float tx, ty, tz, qx, qy, qz, qw; ss >> st >> tx >> ty >> tz >> qx >> qy >> qz >> qw;
        Eigen::Matrix4f T = Eigen::Matrix4f::Identity(); T.topLeftCorner(3, 3) = Eigen::Quaternionf(qw, qx, qy, qz).toRotationMatrix(); T.col(3).head<3>() = Eigen::Vector3f(tx, ty, tz);
        mvTwc.push_back(T);
    }
    if (mvTwc.empty()) { std::cerr << "Load dataset error...No images..." << std::endl; return false; }
    for (int g = 0; g < mNumGPU; ++g) {                                       // one replica per device (nerf_manager.cu:44-55)
        mon_dataset* ds = nullptr;
        if (mon_dataset_create(g, mH, mW, mfx, mfy, mcx, mcy, (uint32_t)mvTwc.size(), mbUseDenseDepth, &ds)) die("DataToGPU");
        for (size_t i = 0; i < mvTwc.size(); ++i) {
            cv::Mat bgr = cv::imread(msDatasetPath + "/rgb/" + names[i], cv::IMREAD_COLOR), inst = cv::imread(msDatasetPath + "/instance/" + names[i], cv::IMREAD_UNCHANGED), depth;
            if (bgr.empty() || inst.empty()) { std::cerr << "Can not read image" << std::endl; exit(0); }
            if (mbUseDenseDepth) { cv::imread(msDatasetPath + "/depth/" + names[i], cv::IMREAD_UNCHANGED).convertTo(depth, CV_32FC1, depthScale); }
            if (mon_dataset_add_frame(ds, (uint32_t)i, bgr.data, 3, 1, inst.data, mbUseDenseDepth ? depth.ptr<float>() : nullptr, mvTwc[i].data())) die("DataToGPU");
        }
        mvpDataset.push_back(ds);
    }
    return true;
}
bool NerfManagerOffline::CreateNeRF(const string objectFile) {               // nerf_manager.cu:64-92, nerf.cu:58-118
    std::ifstream f(objectFile); if (!f) { std::cerr << "object file error..." << std::endl; return false; }
    auto n = std::make_shared<NeRF>(); n->mId = (int)mvpNeRFs.size(); n->mGPUid = n->mId % mNumGPU; mvpNeRFs.push_back(n);
    string line; std::getline(f, line); std::getline(f, line); std::stringstream ss(line); float v[10]; ss >> n->mClass; for (float& x : v) ss >> x;
    Eigen::Matrix4f Two = Eigen::Matrix4f::Identity(); Two.topLeftCorner(3, 3) = Eigen::Quaternionf(v[6], v[3], v[4], v[5]).toRotationMatrix(); Two.col(3).head<3>() = Eigen::Vector3f(v[0], v[1], v[2]);
    n->mObjTow = Two.inverse(); n->mBoundingBox.min = Eigen::Vector3f(-v[7], -v[8], -v[9]); n->mBoundingBox.max = Eigen::Vector3f(v[7], v[8], v[9]);
    while (std::getline(f, line)) { if (line.empty()) continue; std::stringstream s2(line); string st; FrameIdAndBbox b; s2 >> st >> b.x >> b.y >> b.h >> b.w; b.FrameId = mStampToIdx[st]; n->mFrameIdBbox.push_back(b); }
    n->mnBbox = n->mFrameIdBbox.size();
    if (mon_object_create(mvpDataset[n->mGPUid], &mConfig, n->mClass, n->mObjTow.data(), n->mBoundingBox.min.data(), n->mBoundingBox.max.data(), &n->mpObject)) die("Create NeRF error");
    mvThreads.emplace_back(&NeRF::TrainOffline, n, 10);
    return true;
}
bool NerfManagerOffline::WaitThreadsEnd() { if (mvThreads.empty()) return false; for (auto& t : mvThreads) t.join(); return true; }

// ------------------------------------------------------------------ online manager (nerf_manager.cu:133-312)
NerfManagerOnline::NerfManagerOnline(const string cfg, bool UseSparseDepth, int iters) : mNetworkConfigFile(cfg), mbUseSparseDepth(UseSparseDepth), mnTrainStepIterations(iters) {}
bool NerfManagerOnline::Init() {
    if (mon_device_count(&mNumGPU)) die("Can not Detect GPU");
    if (mon_config_from_json(mNetworkConfigFile.c_str(), &mConfig)) die("Read Network Config error");
    mConfig.use_depth = mbUseSparseDepth; return true;
}
void NerfManagerOnline::DatasetInit(float fx, float fy, float cx, float cy, int H, int W, size_t imgs) {
    mnImages = imgs; mvDatasetMutex.resize(mNumGPU);
    for (int g = 0; g < mNumGPU; ++g) { mon_dataset* ds = nullptr; if (mon_dataset_create(g, H, W, fx, fy, cx, cy, (uint32_t)imgs, mbUseSparseDepth, &ds)) die("InitDataToGPU"); mvpDataset.push_back(ds); }
}
void NerfManagerOnline::NewFrameToDataset(unsigned int imgId, const string stamp, cv::Mat& img, cv::Mat& instance, const cv::Mat& depth, const Eigen::Matrix4f& pose) {
    mStampToIdx[stamp] = imgId;                                               // nerf_data.cu:284
    for (int g = 0; g < mNumGPU; ++g) {
        for (auto& m : mvDatasetMutex[g]) m->lock();                          // writers exclude every object's GenerateBatch on that device
        const int rc = mon_dataset_add_frame(mvpDataset[g], imgId, img.data, img.channels(), 1, instance.data, mbUseSparseDepth ? depth.ptr<float>() : nullptr, pose.data());
        for (auto& m : mvDatasetMutex[g]) m->unlock();
        if (rc) die("FrameDataToGPU");
    }
}
size_t NerfManagerOnline::CreateNeRF(const int Class, const Eigen::Matrix4f& ObjTow, const nerf::BoundingBox& box) {
    auto n = std::make_shared<NeRF>(); const size_t idx = mvpNeRFs.size(); mvpNeRFs.push_back(n);
    n->mId = (int)idx; n->mGPUid = mNextGPU; mNextGPU = (mNextGPU + 1) % mNumGPU; n->mClass = Class; n->mObjTow = ObjTow; n->mnIteration = mnTrainStepIterations;
    const float k = (Class == 41 || Class == 73) ? 1.2f : 1.1f;               // SetAttributes, nerf.cu:163-172
    n->mBoundingBox.min = k * box.min; n->mBoundingBox.max = k * box.max; n->mFrameIdBbox.resize(mnImages);
    mvDatasetMutex[n->mGPUid].emplace_back(new std::mutex()); n->mpDatasetMutex = mvDatasetMutex[n->mGPUid].back().get();
    if (mon_object_create(mvpDataset[n->mGPUid], &mConfig, Class, n->mObjTow.data(), n->mBoundingBox.min.data(), n->mBoundingBox.max.data(), &n->mpObject)) die("Create NeRF error");
    mvThreads.emplace_back(&NeRF::TrainOnline, n);
    return idx;
}
int NerfManagerOnline::GetFrameIdx(double t) { auto it = mStampToIdx.find(std::to_string(t)); return it == mStampToIdx.end() ? -1 : (int)it->second; }   // nerf_manager.cu:288-296
void NerfManagerOnline::UpdateNeRFBbox(const size_t idx, const vector<nerf::FrameIdAndBbox>& v, const int train_step) { if (!v.empty()) mvpNeRFs[idx]->UpdateFrameBBox(v, train_step); }
void NerfManagerOnline::DrawMesh(size_t idx) { if (idx < mvpNeRFs.size()) mvpNeRFs[idx]->DrawCPUMesh(); }
bool NerfManagerOnline::WaitThreadsEnd() {
    if (mvThreads.empty()) return false;
    for (auto& n : mvpNeRFs) n->RequestFinish();
    for (auto& t : mvThreads) t.join();
    std::cout << "All NeRF threads completed ..." << std::endl; return true;
}
void NerfManagerOnline::RenderNeRFsTest(const string out_path, const size_t Idx, const vector<string>& ts, const vector<FrameIdAndBbox>& vb, const vector<Eigen::Matrix4f>& vT, const float radius) {
    if (!mvpNeRFs.empty()) mvpNeRFs[Idx]->RenderTestImg(out_path, ts, vT, vb, radius);
}

}  // namespace nerf
