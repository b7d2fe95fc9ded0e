// batch_device.h -- device body of GenerateRays shared by k_gen_candidates (kernels_batch.hip) and the fused backend's
// combined candidate + weight-fragment kernel (kernels_fused.hip).
#pragma once
#include "device_common.h"
#include "model.h"

namespace mon {

// One thread per candidate ray (GenerateRays, CORE/src/nerf_model.cu:369-446).  Candidate data is written un-compacted at
// index i; the validity bit goes into mask[i/64] via a wave ballot (wave64: one 64-bit word per wavefront, no atomics).
__device__ __forceinline__ void gen_candidate(const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, uint32_t nb, uint32_t iter, uint32_t i) {
    const uint32_t R = oc.R;
    bool ok = false;
    if (i < R) {
        const mon_frame_bbox box = b.boxes[i % nb];
        const float u0 = batch_rand(oc, kStreamXY, iter, 2u * i), u1 = batch_rand(oc, kStreamXY, iter, 2u * i + 1u);
        uint32_t x = box.x + (uint32_t)(u0 * (float)(int)box.w);          // :395
        uint32_t y = box.y + (uint32_t)(u1 * (float)(int)box.h);          // :396
        x = min(x, (uint32_t)ds.K.W - 1u); y = min(y, (uint32_t)ds.K.H - 1u);   // guard (reference reads out of bounds for boxes past the image)
        const size_t pix = ((size_t)box.FrameId * ds.K.H + y) * ds.K.W + x;
        const uint32_t rgba = ds.rgba[pix];
        const uint32_t inst = rgba >> 24;
        ok = !(inst != 0u && inst != oc.instance_id);                    // occlusion :398-401
        if (ok) {
            float o[3], d[3], dn, t0, t1;
            pixel_ray(ds.K, (float)x, (float)y, ds.poses + (size_t)box.FrameId * 16, oc.Tow.m, false, o, d, dn);
            ok = ray_intersect(oc.aabb, o, d, t0, t1);
            if (ok) {
                b.cand_o[3 * i] = o[0]; b.cand_o[3 * i + 1] = o[1]; b.cand_o[3 * i + 2] = o[2];
                b.cand_d[3 * i] = d[0]; b.cand_d[3 * i + 1] = d[1]; b.cand_d[3 * i + 2] = d[2];
                b.cand_dn[i] = dn; b.cand_t0[i] = fmaxf(t0, 0.0f); b.cand_t1[i] = t1;     // :423-424
                b.cand_rgba[i] = rgba;
                b.cand_depth[i] = (inst != 0u && ds.depth != nullptr && oc.use_depth) ? ds.depth[pix] * dn : 0.0f;   // :431-434
            }
        }
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && i < R) b.mask[i >> 6] = bal;
}

}  // namespace mon
