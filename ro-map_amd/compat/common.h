// compat/common.h -- source-compatible replacement for CORE/include/common.h (PODs crossing the libMON boundary).
// Built only where Eigen and GLEW exist (the RO-MAP build tree); the GPU work goes through include/mon_core.h.
#pragma once
#include <Eigen/Core>
#include <Eigen/Dense>
#include <GL/glew.h>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace std;   // the reference headers export this; its consumers rely on it

namespace nerf {

struct FrameIdAndBbox { uint32_t FrameId; uint32_t x, y, h, w; };          // CORE/include/common.h:18-23 == mon_frame_bbox

struct BoundingBox {                                                        // common.h:25-30
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Eigen::Vector3f min = Eigen::Vector3f::Zero();
    Eigen::Vector3f max = Eigen::Vector3f::Zero();
};

struct CPUMeshData {                                                        // common.h:32-41 (read by DrawCPUMesh)
    std::vector<float> verts, normals;
    std::vector<uint8_t> colors;
    std::vector<uint32_t> indices;
    bool have_reslult = false;
    std::mutex mesh_mutex;
};

}  // namespace nerf
