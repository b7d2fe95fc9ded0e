// compat/nerf_manager.h -- source-compatible nerf::NerfManagerOffline / NerfManagerOnline
// (CORE/include/nerf_manager.h:21-90).  Same names, signatures and the public member consumers touch
// (mbUseSparseDepth, REF/src/LocalMapping.cc:1172); every method is one call into libmon_core.so's manager entry
// points (mon_offline_* / mon_online_*, include/mon_core.h), which own the datasets, objects and training threads.
#pragma once
#if defined(__has_include) && __has_include(<opencv/cv.hpp>)
#include <opencv/cv.hpp>                          // what CORE/include/nerf_manager.h:9 pulls in (OpenCV 3): consumers rely on its transitive includes
#else
#include <opencv2/core.hpp>
#endif
#include "nerf.h"

namespace nerf {

class NerfManagerOffline {
public:
    NerfManagerOffline(const string datasetPath, const string networkConfigFile, bool useDenseDepth);
    ~NerfManagerOffline();
    bool Init();
    bool ReadDataset();
    bool CreateNeRF(const string objectFile);
    bool WaitThreadsEnd();
    std::shared_ptr<NeRF> GetNeRF(int idx) { return mvpNeRFs.at(idx); }
    vector<std::shared_ptr<NeRF>> GetAllNeRF() { return mvpNeRFs; }
    vector<Eigen::Matrix4f> GetAllTwc();
    void GetIntrinsics(float& fx, float& fy, float& cx, float& cy);

    string msNetworkConfigFile, msDatasetPath; bool mbUseDenseDepth;
    vector<std::shared_ptr<NeRF>> mvpNeRFs;
    mon_offline* mpManager = nullptr;
};

class NerfManagerOnline {
public:
    NerfManagerOnline(const string network_config_file, bool UseSparseDepth, int TrainStepIterations);
    ~NerfManagerOnline();
    bool Init();
    void DatasetInit(float fx, float fy, float cx, float cy, int H, int W, size_t imgs);
    void NewFrameToDataset(unsigned int imgId, const string timestamp, cv::Mat& img, cv::Mat& instance, const cv::Mat& depth_img, const Eigen::Matrix4f& pose);
    size_t CreateNeRF(const int Class, const Eigen::Matrix4f& ObjTow, const nerf::BoundingBox& BoundingBox);
    int GetFrameIdx(double timastamp);
    void UpdateNeRFBbox(const size_t idx, const vector<nerf::FrameIdAndBbox>& vFrameBbox, const int train_step);
    void UpdateDataset(unsigned int CurId, unsigned int FrameNum, const vector<Eigen::Matrix4f>& Poses);     // nerf_manager.h:66
    void DrawMesh(size_t idx);
    bool WaitThreadsEnd();
    void RenderNeRFsTest(const string out_path, const size_t Idx, const vector<string>& timestamp, const vector<FrameIdAndBbox>& vBbox,
                         const vector<Eigen::Matrix4f>& vTwc, const float radius);

    string mNetworkConfigFile; bool mbUseSparseDepth; int mnTrainStepIterations;
    vector<std::shared_ptr<NeRF>> mvpNeRFs;
    mon_online* mpManager = nullptr;
};

}  // namespace nerf
