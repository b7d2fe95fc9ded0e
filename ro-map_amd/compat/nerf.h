// compat/nerf.h -- source-compatible nerf::NeRF (CORE/include/nerf.h:19-89): what the consumers of libMON touch on an object.
// The object itself (network, training thread, mesh) lives behind the C ABI (include/mon_core.h); this class is a view on it.
#pragma once
#include "common.h"
#include "mon_core.h"

namespace nerf {

class NeRF {
public:
    // callers: MON/main.cpp:55,149,151,217; REF/src/MapDrawer.cc:396 via NerfManagerOnline::DrawMesh
    vector<FrameIdAndBbox> GetFrameIdAndBBox() { return mFrameIdBbox; }
    Eigen::Matrix4f GetObjTow() { return mObjTow; }
    BoundingBox GetBoundingBox() { return mBoundingBox; }
    CPUMeshData& GetCPUMeshData() { return mCPUMeshData; }
    void DrawCPUMesh();                           // nerf.cu:484-507: try_lock, draw the last mesh the training thread published
    // nerf.cu:509-551 draws CUDA-GL interop buffers; here the same mesh from host arrays (no interop on this side)
    void DrawMesh() { DrawCPUMesh(); }
    vector<Eigen::Matrix4f> GetTwc();             // nerf.cu:450-462: the dataset's pose of every frame the object has a 2-D box in

    int mId = -1, mClass = 0;
    uint8_t mInstanceId = 0;                      // nerf.h:59 (= class id as stored in the instance images, nerf.cu:75,158)
    size_t mnBbox = 0;                            // nerf.h:64
    Eigen::Matrix4f mObjTow = Eigen::Matrix4f::Identity();
    BoundingBox mBoundingBox;
    std::vector<FrameIdAndBbox> mFrameIdBbox;
    CPUMeshData mCPUMeshData; uint64_t mMeshGeneration = 0;   // generation of the mesh held in mCPUMeshData
    mon_object* mpObject = nullptr;               // borrowed from the manager (mon_offline_object / mon_online_object)
    mon_offline* mpOffline = nullptr; mon_online* mpOnline = nullptr;      // the manager that owns the dataset (GetTwc)
};

}  // namespace nerf
