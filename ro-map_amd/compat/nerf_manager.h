// compat/nerf_manager.h -- source-compatible nerf::NerfManagerOffline / NerfManagerOnline
// (CORE/include/nerf_manager.h:21-90).  Same names, signatures and the public member consumers touch
// (mbUseSparseDepth, REF/src/LocalMapping.cc:1172); everything below the signatures is libmon_core.so.
#pragma once
#include <map>
#include <opencv2/core.hpp>
#include "nerf.h"

namespace nerf {

class NerfManagerOffline {
public:
    NerfManagerOffline(const string datasetPath, const string networkConfigFile, bool useDenseDepth);
    bool Init();
    bool ReadDataset();
    bool CreateNeRF(const string objectFile);
    bool WaitThreadsEnd();
    std::shared_ptr<NeRF> GetNeRF(int idx) { return mvpNeRFs.at(idx); }
    vector<std::shared_ptr<NeRF>> GetAllNeRF() { return mvpNeRFs; }
    vector<Eigen::Matrix4f> GetAllTwc() { return mvTwc; }
    void GetIntrinsics(float& fx, float& fy, float& cx, float& cy) { fx = mfx; fy = mfy; cx = mcx; cy = mcy; }

    string msNetworkConfigFile, msDatasetPath; bool mbUseDenseDepth; int mNumGPU = 0;
    vector<std::shared_ptr<NeRF>> mvpNeRFs; vector<std::thread> mvThreads;
    vector<mon_dataset*> mvpDataset; mon_config mConfig; vector<Eigen::Matrix4f> mvTwc; std::map<string, uint32_t> mStampToIdx;
    float mfx = 0, mfy = 0, mcx = 0, mcy = 0; int mH = 0, mW = 0;
};

class NerfManagerOnline {
public:
    NerfManagerOnline(const string network_config_file, bool UseSparseDepth, int TrainStepIterations);
    bool Init();
    void DatasetInit(float fx, float fy, float cx, float cy, int H, int W, size_t imgs);
    void NewFrameToDataset(unsigned int imgId, const string timestamp, cv::Mat& img, cv::Mat& instance, const cv::Mat& depth_img, const Eigen::Matrix4f& pose);
    size_t CreateNeRF(const int Class, const Eigen::Matrix4f& ObjTow, const nerf::BoundingBox& BoundingBox);
    int GetFrameIdx(double timastamp);
    void UpdateNeRFBbox(const size_t idx, const vector<nerf::FrameIdAndBbox>& vFrameBbox, const int train_step);
    void DrawMesh(size_t idx);
    bool WaitThreadsEnd();
    void RenderNeRFsTest(const string out_path, const size_t Idx, const vector<string>& timestamp, const vector<FrameIdAndBbox>& vBbox,
                         const vector<Eigen::Matrix4f>& vTwc, const float radius);

    string mNetworkConfigFile; bool mbUseSparseDepth; int mnTrainStepIterations; int mNumGPU = 0;
    vector<std::shared_ptr<NeRF>> mvpNeRFs; vector<std::thread> mvThreads;
    vector<mon_dataset*> mvpDataset; vector<std::vector<std::unique_ptr<std::mutex>>> mvDatasetMutex; mon_config mConfig;
    std::map<string, uint32_t> mStampToIdx; size_t mnImages = 0; int mNextGPU = 0;
};

}  // namespace nerf
