// kernels_composite.hip -- volumetric compositing and the hand-derived loss gradient (backend 0).
//   VolumeRender                        CORE/src/nerf_model.cu:735-815
//   VolumeRenderGradient_No_Compacted   :817-954   (+ the dloss_dout memset :1578, folded in)
//   SumLoss                             :1231-1253 (block reduce + one atomic)
//   VolumeRender_Render                 :1134-1229
// One thread per ray, like the reference; the fused backend does the same arithmetic with one
// wavefront lane per sample and wave scans (kernels_fused.hip).
#include "device_common.h"
#include "model.h"
#include "fused_device.h"      // the DPP wave scans

namespace mon {

// The same composite + loss gradient with one LANE per sample (S = 32: a half-wave per ray, two rays per wave), transmittance / colour / depth as DPP wave scans
// -- the arithmetic of k_fused_train's composite (kernels_fused.hip), as a kernel of its own for the layer-at-a-time path: one thread per ray left the chip to
// 16 workgroups and took 25 us of every step of the shapes outside the fused kernels.
__global__ void __launch_bounds__(1024) k_composite_grad_wave(BatchPtrs b, ObjectConst oc, DevState* __restrict__ st) {
    if (st->n_valid == 0u) return;
    const uint32_t R = oc.R, lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const bool ok = ray < R; const uint32_t i = ok ? ray : R - 1u, s_idx = i * 32u + n;
    const half4_t v = reinterpret_cast<const half4_t*>(b.O)[s_idx];
    const float t = b.tdist[s_idx];
    const float v0 = (float)v[0], v1 = (float)v[1], v2 = (float)v[2], v3 = (float)v[3];
    const float c0 = logistic_f(v0), c1 = logistic_f(v1), c2 = logistic_f(v2), sigma = __expf(v3);
    float tprev = lane_prev(t, 0.f); if (n == 0u) tprev = 0.f;                         // :770 last_distance = 0
    const float dt = t - tprev, alpha = 1.f - __expf(-sigma * dt), om = 1.f - alpha;
    const float tincl = scan_mul32(om);                                               // T after this sample
    float T = lane_prev(tincl, 1.f); if (n == 0u) T = 1.f;                             // T before this sample
    const bool active = T >= kTransmittanceEps;                                        // :774 early-out (T is non-increasing)
    const unsigned long long bal = __ballot(active);
    const int nact0 = __popc((uint32_t)bal), nact1 = __popc((uint32_t)(bal >> 32));   // sample 0 is always active: >= 1
    const float Tf0 = lane_bcast(tincl, nact0 - 1), Tf1 = lane_bcast(tincl, 32 + nact1 - 1), Tfin = h ? Tf1 : Tf0;
    const float wgt = active ? alpha * T : 0.f;
    const float p0 = scan_add32(wgt * c0), p1 = scan_add32(wgt * c1), p2 = scan_add32(wgt * c2), pd = scan_add32(wgt * t);
    const auto last = [&](float x) { const float a = lane_bcast(x, 31), c = lane_bcast(x, 63); return h ? c : a; };
    const float bg0 = b.bgcol[3 * i], bg1 = b.bgcol[3 * i + 1], bg2 = b.bgcol[3 * i + 2];
    const float rgb0 = last(p0) + Tfin * bg0, rgb1 = last(p1) + Tfin * bg1, rgb2 = last(p2) + Tfin * bg2, dep = last(pd), mask = 1.f - Tfin;
    const float e0 = rgb0 - b.target[3 * i], e1 = rgb1 - b.target[3 * i + 1], e2 = rgb2 - b.target[3 * i + 2];
    const float g0 = 2.f * e0, g1 = 2.f * e1, g2 = 2.f * e2, mean_loss = (e0 * e0 + e1 * e1 + e2 * e2) / 3.f;
    const float tdp = b.target_depth[i];
    float dl_dd = 0.f; if (tdp > 0.f) dl_dd = 0.5f * ((dep - tdp >= 0.f) ? 1.f : -1.f);
    const bool is_obj = b.ray_flag[i] == 1;
    const float loss = is_obj ? mean_loss + dl_dd * (dep - tdp) + (1.f - mask) : mean_loss + mask;
    const float ls = oc.loss_scale / (float)R;
    half4_t dv = { (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f };                // samples after the early-out keep zero gradient (:1578 memset)
    if (active) {
        const float Tn = tincl, s0 = rgb0 - p0, s1 = rgb1 - p1, s2 = rgb2 - p2;           // T after the update (:912), suffix (:915)
        dv[0] = (half_t)opaque_f32(ls * ((wgt * g0) * (c0 * (1.f - c0))));
        dv[1] = (half_t)opaque_f32(ls * ((wgt * g1) * (c1 * (1.f - c1))));
        dv[2] = (half_t)opaque_f32(ls * ((wgt * g2) * (c2 * (1.f - c2))));
        const float dsig = __expf(clamp_f(v3, -15.f, 15.f)), depth_sup = dl_dd * (Tn * t - (dep - pd)), dmask = 1.f - mask;
        float dl;
        if (is_obj) { const float dlm = 0.5f * (mask >= 1.f ? 1.f : -1.f), dot = g0 * (Tn * c0 - s0) + g1 * (Tn * c1 - s1) + g2 * (Tn * c2 - s2);
            dl = dsig * dt * (dot + depth_sup + dlm * dmask); }
        else { const float dlm = 0.5f * (mask >= 0.f ? 1.f : -1.f); dl = dsig * dt * dlm * dmask + dsig * 0.01f; }
        dv[3] = (half_t)opaque_f32(ls * dl);
    }
    if (ok) {
        reinterpret_cast<half4_t*>(b.dO)[s_idx] = dv;
        if (n == 0u) { b.rgb_ray[3 * i] = rgb0; b.rgb_ray[3 * i + 1] = rgb1; b.rgb_ray[3 * i + 2] = rgb2; b.depth_ray[i] = dep; b.mask_ray[i] = mask;
            b.loss_ray[i] = loss; }
    }
    // one atomic per WORKGROUP (its 32 rays' losses; one per wave was 2048 adds into one address: most of the kernel's 28 us)
    __shared__ float wsum[16];
    const float l0 = lane_bcast(ok && n == 0u ? loss : 0.f, 0), l1 = lane_bcast(ok && n == 0u ? loss : 0.f, 32);
    if (lane == 0u) wsum[threadIdx.x >> 6] = l0 + l1;
    __syncthreads();
    if (threadIdx.x == 0u) { float a = 0.f; for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) a += wsum[w]; atomicAdd(&st->loss_sum, a); }
}

__global__ void __launch_bounds__(64) k_composite_grad(BatchPtrs b, ObjectConst oc, DevState* __restrict__ st) {
    if (st->n_valid == 0u) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t R = oc.R, S = oc.S;
    float loss = 0.f;
    if (i < R) {
        const half4_t* out = reinterpret_cast<const half4_t*>(b.O) + (size_t)i * S;
        half4_t* dout = reinterpret_cast<half4_t*>(b.dO) + (size_t)i * S;
        const float* td = b.tdist + (size_t)i * S;
        // ---- forward composite :762-813
        float T = 1.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, last = 0.f;
        for (uint32_t n = 0; n < S; ++n) {
            if (T < kTransmittanceEps) break;
            const half4_t v = out[n];
            const float c0 = logistic_f((float)v[0]), c1 = logistic_f((float)v[1]), c2 = logistic_f((float)v[2]);
            const float cur = td[n], dt = cur - last, sigma = __expf((float)v[3]);
            const float alpha = 1.f - __expf(-sigma * dt), w = alpha * T;
            r0 += w * c0; r1 += w * c1; r2 += w * c2; dep += w * cur; T *= (1.f - alpha); last = cur;
        }
        const float rgb0 = r0 + T * b.bgcol[3 * i], rgb1 = r1 + T * b.bgcol[3 * i + 1], rgb2 = r2 + T * b.bgcol[3 * i + 2];
        const float mask = 1.f - T;
        b.rgb_ray[3 * i] = rgb0; b.rgb_ray[3 * i + 1] = rgb1; b.rgb_ray[3 * i + 2] = rgb2; b.depth_ray[i] = dep; b.mask_ray[i] = mask;
        // ---- loss + gradient :853-953
        const float e0 = rgb0 - b.target[3 * i], e1 = rgb1 - b.target[3 * i + 1], e2 = rgb2 - b.target[3 * i + 2];
        const float g0 = 2.f * e0, g1 = 2.f * e1, g2 = 2.f * e2;
        const float mean_loss = (e0 * e0 + e1 * e1 + e2 * e2) / 3.f;
        const float tdp = b.target_depth[i];
        float dl_dd = 0.f;
        if (tdp > 0.f) dl_dd = 0.5f * ((dep - tdp >= 0.f) ? 1.f : -1.f);
        const bool is_obj = b.ray_flag[i] == 1;
        loss = is_obj ? mean_loss + dl_dd * (dep - tdp) + (1.f - mask) : mean_loss + mask;
        b.loss_ray[i] = loss;
        const float ls = oc.loss_scale / (float)R;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, d2 = 0.f; T = 1.f; last = 0.f;
        uint32_t n = 0;
        for (; n < S; ++n) {
            if (T < kTransmittanceEps) break;
            const half4_t v = out[n];
            const float c0 = logistic_f((float)v[0]), c1 = logistic_f((float)v[1]), c2 = logistic_f((float)v[2]);
            const float cur = td[n], dt = cur - last, sigma = __expf((float)v[3]);
            const float alpha = 1.f - __expf(-sigma * dt), w = alpha * T;
            q0 += w * c0; q1 += w * c1; q2 += w * c2; d2 += w * cur; T *= (1.f - alpha);
            const float s0 = rgb0 - q0, s1 = rgb1 - q1, s2 = rgb2 - q2;
            half4_t dv;
            dv[0] = (half_t)opaque_f32(ls * ((w * g0) * (c0 * (1.f - c0))));
            dv[1] = (half_t)opaque_f32(ls * ((w * g1) * (c1 * (1.f - c1))));
            dv[2] = (half_t)opaque_f32(ls * ((w * g2) * (c2 * (1.f - c2))));
            const float dsig = __expf(clamp_f((float)v[3], -15.f, 15.f));
            const float depth_sup = dl_dd * (T * cur - (dep - d2));
            const float dmask = 1.f - mask;
            float dl;
            if (is_obj) {
                const float dlm = 0.5f * (mask >= 1.f ? 1.f : -1.f);
                const float dot = g0 * (T * c0 - s0) + g1 * (T * c1 - s1) + g2 * (T * c2 - s2);
                dl = dsig * dt * (dot + depth_sup + dlm * dmask);
            } else {
                const float dlm = 0.5f * (mask >= 0.f ? 1.f : -1.f);
                dl = dsig * dt * dlm * dmask + dsig * 0.01f;
            }
            dv[3] = (half_t)opaque_f32(ls * dl);
            dout[n] = dv; last = cur;
        }
        const half4_t z = { (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f };
        for (; n < S; ++n) dout[n] = z;         // samples after the early-out keep zero gradient (:1578 memset)
    }
    // wave reduce + one atomic per wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) loss += __shfl_down(loss, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&st->loss_sum, loss);
}

// One thread per pixel: rays that missed the box, or ended with opacity <= 0.5, become white / 0 / 0.
__global__ void __launch_bounds__(64) k_composite_render(BatchPtrs b, uint32_t S, uint32_t n_rays, float* __restrict__ rgb, float* __restrict__ depth,
        float* __restrict__ mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    float o0 = 1.f, o1 = 1.f, o2 = 1.f, od = 0.f, om = 0.f;
    if (b.ray_flag[i]) {
        const half4_t* out = reinterpret_cast<const half4_t*>(b.O) + (size_t)i * S;
        const float* td = b.tdist + (size_t)i * S;
        float T = 1.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, last = 0.f;
        for (uint32_t n = 0; n < S; ++n) {
            if (T < kTransmittanceEps) break;
            const half4_t v = out[n];
            const float c0 = logistic_f((float)v[0]), c1 = logistic_f((float)v[1]), c2 = logistic_f((float)v[2]);
            const float cur = td[n], dt = cur - last, sigma = __expf((float)v[3]);
            const float alpha = 1.f - __expf(-sigma * dt), w = alpha * T;
            r0 += w * c0; r1 += w * c1; r2 += w * c2; dep += w * cur; T *= (1.f - alpha); last = cur;
        }
        if (1.f - T > 0.5f) { o0 = r0 + T; o1 = r1 + T; o2 = r2 + T; od = dep / b.ray_dn[i]; om = 1.f; }     // :1213-1220, background 1.0
    }
    rgb[3 * i] = o0; rgb[3 * i + 1] = o1; rgb[3 * i + 2] = o2; depth[i] = od; mask[i] = om;
}

// output_half_to_float :311-317 (raw density channel)
__global__ void __launch_bounds__(256) k_extract_density(const uint16_t* __restrict__ O, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)reinterpret_cast<const half_t*>(O)[(size_t)i * kOut + 3];
}

void launch_composite_grad(hipStream_t s, const BatchPtrs& b, const ObjectConst& oc, DevState* st) {
    // (a lane per sample where a ray's samples are a half-wave; the one-thread-per-ray kernel otherwise)
    if (oc.S == 32u) hipLaunchKernelGGL(k_composite_grad_wave, dim3((oc.R * 32u + 1023u) / 1024u), dim3(1024), 0, s, b, oc, st);
    else hipLaunchKernelGGL(k_composite_grad, dim3((oc.R + 63) / 64), dim3(64), 0, s, b, oc, st);
}
void launch_composite_render(hipStream_t s, const BatchPtrs& b, uint32_t S, uint32_t n_rays, float* rgb, float* depth, float* mask) {
    hipLaunchKernelGGL(k_composite_render, dim3((n_rays + 63) / 64), dim3(64), 0, s, b, S, n_rays, rgb, depth, mask);
}
void launch_extract_density(hipStream_t s, const uint16_t* O, float* out, uint32_t n) {
    hipLaunchKernelGGL(k_extract_density, dim3((n + 255) / 256), dim3(256), 0, s, O, out, n);
}

}  // namespace mon
