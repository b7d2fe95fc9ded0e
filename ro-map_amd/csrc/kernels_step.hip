// kernels_step.hip -- NeRF_Model::Step (CORE/src/nerf_model.cu:1504-1550), the reference's schedule with per-ray SAMPLE compaction.
// The reference marks it "unavailable, for reference only" and neither driver calls it; it is built here behind mon_set_option("step_variant", 1) on the
// layer-at-a-time kernels (backend 0) so that the row exists and can be checked against a CPU restatement (tests/test_step_variant.py):
//   1. inference of every sample with the training weights (:1509)                                   -> launch_encode + launch_mlp_forward (model.cpp)
//   2. VolumeRenderGradient (:957-1132), one thread per ray: composite until T < 1e-4 (numsteps), colour-only L2 loss, ONE background colour for all rays
//      (the kernel's by-value copy of the generator: every thread draws the same three floats, :1038 -- here the iteration's first three RandColors), the
//      numsteps positions and their dL/dO into a compacted batch.  Slots: the reference takes them with atomicAdd in arrival order; here an exclusive prefix
//      sum over the rays (k_step_count -> k_step_scan -> k_step_gradient), so the batch -- and every result -- is deterministic.
//   3. fill_rollover (:258-266) + fill_rollover_and_rescale (:269-279): the n compacted samples repeated cyclically up to the batch size B, the COPIES'
//      gradients scaled by n / B (the originals keep theirs: `i < n * stride` returns early)            -> k_step_rollover
//   4. forward + backward of the full-size compacted batch (:1545-1548) and the optimizer step           -> the backend-0 kernels (model.cpp)
#include "device_common.h"
#include "model.h"

namespace mon {

// first loop of the kernel (:996-1036): samples in front of the transmittance cut, the composited colour with the shared background
__global__ void __launch_bounds__(64) k_step_count(BatchPtrs b, ObjectConst oc, const DevState* __restrict__ st,
                                                   uint32_t* __restrict__ steps /* [R + 1], [0] = 0 */) {
    if (st->n_valid == 0u) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, R = oc.R, S = oc.S;
    if (i >= R) return;
    const half4_t* out = reinterpret_cast<const half4_t*>(b.O) + (size_t)i * S; const float* td = b.tdist + (size_t)i * S;
    float T = 1.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, last = 0.f; uint32_t n = 0;
    for (; n < S; ++n) {
        if (T < kTransmittanceEps) break;
        const half4_t v = out[n];
        const float c0 = logistic_f((float)v[0]), c1 = logistic_f((float)v[1]), c2 = logistic_f((float)v[2]);
        const float cur = td[n], dt = cur - last, sigma = __expf((float)v[3]);
        const float alpha = 1.f - __expf(-sigma * dt), w = alpha * T;
        r0 += w * c0; r1 += w * c1; r2 += w * c2; dep += w * cur; T *= (1.f - alpha); last = cur;      // (depth: |point - o| = t for a unit direction)
    }
    const float bg0 = batch_rand(oc, kStreamColor, st->iter, 0u), bg1 = batch_rand(oc, kStreamColor, st->iter, 1u),
            bg2 = batch_rand(oc, kStreamColor, st->iter, 2u);
    b.rgb_ray[3 * i] = r0 + T * bg0; b.rgb_ray[3 * i + 1] = r1 + T * bg1; b.rgb_ray[3 * i + 2] = r2 + T * bg2; b.depth_ray[i] = dep; b.mask_ray[i] = 1.f - T;
    steps[i + 1] = n; if (i == 0u) steps[0] = 0u;
}
// inclusive prefix over steps[1..R] in place (one block; R <= 16384): steps[j] = first compacted slot of ray j, steps[R] = compacted samples
__global__ void __launch_bounds__(1024) k_step_scan(uint32_t* __restrict__ steps, uint32_t R, const DevState* __restrict__ st) {
    if (st->n_valid == 0u) return;
    __shared__ uint32_t wsum[16]; __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < R; base += 1024u) {
        const uint32_t j = base + threadIdx.x; uint32_t v = j < R ? steps[j + 1] : 0u, inc = v;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, sh, 64); if ((int)(threadIdx.x & 63u) >= sh) inc += o; }
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = carry_s; for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
        if (j < R) steps[j + 1] = before + inc;
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = before + inc;
        __syncthreads();
    }
}
// second loop (:1048-1131): positions and dL/dO of the ray's numsteps samples at its slots of the compacted batch; colour-only L2 loss
__global__ void __launch_bounds__(64) k_step_gradient(BatchPtrs b, ObjectConst oc, DevState* __restrict__ st, const uint32_t* __restrict__ steps,
        float* __restrict__ pts_c, uint16_t* __restrict__ dO_c) {
    if (st->n_valid == 0u) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, R = oc.R, S = oc.S;
    float loss = 0.f;
    if (i < R) {
        const half4_t* out = reinterpret_cast<const half4_t*>(b.O) + (size_t)i * S; const float* td = b.tdist + (size_t)i * S;
        const uint32_t base = steps[i], ns = steps[i + 1] - base;
        const float rgb0 = b.rgb_ray[3 * i], rgb1 = b.rgb_ray[3 * i + 1], rgb2 = b.rgb_ray[3 * i + 2];
        const float e0 = rgb0 - b.target[3 * i], e1 = rgb1 - b.target[3 * i + 1], e2 = rgb2 - b.target[3 * i + 2];
        const float g0 = 2.f * e0, g1 = 2.f * e1, g2 = 2.f * e2;
        loss = (e0 * e0 + e1 * e1 + e2 * e2) / 3.f; b.loss_ray[i] = loss;
        const float ls = oc.loss_scale / (float)R;
        float T = 1.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, last = 0.f;
        half4_t* dout = reinterpret_cast<half4_t*>(dO_c) + base;
        for (uint32_t n = 0; n < ns; ++n) {
            if (T < kTransmittanceEps) break;
            const size_t s = (size_t)i * S + n, d = (size_t)base + n;
            pts_c[3 * d] = b.pts[3 * s]; pts_c[3 * d + 1] = b.pts[3 * s + 1]; pts_c[3 * d + 2] = b.pts[3 * s + 2];
            const half4_t v = out[n];
            const float c0 = logistic_f((float)v[0]), c1 = logistic_f((float)v[1]), c2 = logistic_f((float)v[2]);
            const float cur = td[n], dt = cur - last, sigma = __expf((float)v[3]);
            const float alpha = 1.f - __expf(-sigma * dt), w = alpha * T; last = cur;
            q0 += w * c0; q1 += w * c1; q2 += w * c2; T *= (1.f - alpha);
            half4_t dv;
            dv[0] = (half_t)(ls * ((w * g0) * (c0 * (1.f - c0))));
            dv[1] = (half_t)(ls * ((w * g1) * (c1 * (1.f - c1))));
            dv[2] = (half_t)(ls * ((w * g2) * (c2 * (1.f - c2))));
            float dot = 0.f; dot += g0 * (T * c0 - (rgb0 - q0)); dot += g1 * (T * c1 - (rgb1 - q1)); dot += g2 * (T * c2 - (rgb2 - q2));
            const float dsig = __expf(clamp_f((float)v[3], -15.f, 15.f));
            dv[3] = (half_t)(ls * (dsig * (dt * dot)));
            dout[n] = dv;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) loss += __shfl_down(loss, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&st->loss_sum, loss);
}
// fill_rollover (positions) + fill_rollover_and_rescale (gradients): slots n .. B - 1 repeat the compacted batch cyclically, the copies' gradients times n / B
__global__ void __launch_bounds__(256) k_step_rollover(const uint32_t* __restrict__ steps, uint32_t R, uint32_t B, float* __restrict__ pts_c,
        uint16_t* __restrict__ dO_c, DevState* __restrict__ st) {
    if (st->n_valid == 0u) return;
    const uint32_t n = steps[R], i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) st->n_scatter_now = n;                                  // (reported as the step's sample count: mon_object_info / tests)
    if (n == 0u || i < n || i >= B) return;
    const uint32_t src = i % n;
    pts_c[3 * (size_t)i] = pts_c[3 * (size_t)src]; pts_c[3 * (size_t)i + 1] = pts_c[3 * (size_t)src + 1]; pts_c[3 * (size_t)i + 2] = pts_c[3 * (size_t)src + 2];
    const half4_t v = reinterpret_cast<const half4_t*>(dO_c)[src]; half4_t o;
#pragma unroll
    for (int a = 0; a < 4; ++a) o[a] = (half_t)(((float)v[a] * (float)n) / (float)B);
    reinterpret_cast<half4_t*>(dO_c)[i] = o;
}

void launch_step_compaction(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st, uint32_t* steps, float* pts_c) {
    const uint32_t R = oc.R, B = R * oc.S;
    hipLaunchKernelGGL(k_step_count, dim3((R + 63) / 64), dim3(64), 0, s, b, oc, st, steps);
    hipLaunchKernelGGL(k_step_scan, dim3(1), dim3(1024), 0, s, steps, R, st);
    hipLaunchKernelGGL(k_step_gradient, dim3((R + 63) / 64), dim3(64), 0, s, b, oc, st, steps, pts_c, b.dO);
    hipLaunchKernelGGL(k_step_rollover, dim3((B + 255) / 256), dim3(256), 0, s, steps, R, B, pts_c, b.dO, st);
}

}  // namespace mon
