// compat/nerf.h -- source-compatible nerf::NeRF (CORE/include/nerf.h:19-89) on top of the C ABI.
#pragma once
#include <condition_variable>
#include <thread>
#include "common.h"
#include "mon_core.h"

namespace nerf {

class NeRF {
public:
    NeRF() = default;
    ~NeRF();
    // callers: MON/main.cpp:55,149,151,217
    vector<FrameIdAndBbox> GetFrameIdAndBBox();
    Eigen::Matrix4f GetObjTow() { return mObjTow; }
    BoundingBox GetBoundingBox() { return mBoundingBox; }
    CPUMeshData& GetCPUMeshData() { return mCPUMeshData; }
    void DrawCPUMesh();
    void UpdateCPUMesh();
    // online protocol (nerf.cu:187-253, 406-448)
    void UpdateFrameBBox(const vector<FrameIdAndBbox>& vFrameBbox, const int train_step);
    void RequestFinish();
    bool CheckFinish();
    void TrainOffline(const int iterations);      // 10 x 500 iterations (nerf_manager.cu:89, nerf_model.cu:1635)
    void TrainOnline();
    void RenderTestImg(const string out_path, const vector<string>& timestamp, const vector<Eigen::Matrix4f>& testTwc,
                       const vector<FrameIdAndBbox>& testBbox, const float radius);

    int mId = -1, mGPUid = -1, mClass = 0, mnIteration = 500, mnTrainStep = 0;
    Eigen::Matrix4f mObjTow = Eigen::Matrix4f::Identity();
    BoundingBox mBoundingBox;
    std::vector<FrameIdAndBbox> mFrameIdBbox; size_t mnBbox = 0, mnUploaded = 0;
    std::mutex mUpdateBbox, mFinishMutex; std::condition_variable mCond; bool mbFinishRequested = false;
    CPUMeshData mCPUMeshData;
    mon_object* mpObject = nullptr;               // replaces shared_ptr<NeRF_Model>
    std::mutex* mpDatasetMutex = nullptr;         // per-object dataset mutex (nerf_manager.cu:245-247)
};

}  // namespace nerf
