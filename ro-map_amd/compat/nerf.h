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

    int mId = -1, mClass = 0;
    Eigen::Matrix4f mObjTow = Eigen::Matrix4f::Identity();
    BoundingBox mBoundingBox;
    std::vector<FrameIdAndBbox> mFrameIdBbox;
    CPUMeshData mCPUMeshData; uint64_t mMeshGeneration = 0;   // generation of the mesh held in mCPUMeshData
    mon_object* mpObject = nullptr;               // borrowed from the manager (mon_offline_object / mon_online_object)
};

}  // namespace nerf
