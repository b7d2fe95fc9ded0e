// kernels_render.hip -- k_fused_render (NeRF_Model::Render: one wavefront per pixel ray, 2 x 32 samples) and the occupancy grid kernels of the opt-in
// forward-pass skipping; both evaluate the network with the tile_forward of fused_device.h (gathers + MFMA MLP).
#include "fused_device.h"

namespace mon {

// ------------------------------------------------------------------ fused render kernel
// One wavefront per pixel ray, 2S = 64 samples as two 32-sample tiles with a carried transmittance;
// rays that miss the box and tiles behind an opaque prefix are skipped (wave-uniform).
// GenerateRenderInputPoints :593-626 + inference + VolumeRender_Render :1134-1229.
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_fused_render(FusedArgs a, uint32_t n_rays, uint32_t idx_base, float* __restrict__ rgb, float* __restrict__ depth,
        float* __restrict__ mask) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    LevelLds* llt = reinterpret_cast<LevelLds*>(smem + S::FRAG_BYTES);
    build_fragments<EPAD, W, NH>(frags, llt, a, false);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31;
    const int L = a.nd.L; const uint32_t S2 = 2u * a.oc.S;      // 64
    const half2_t* table = reinterpret_cast<const half2_t*>(a.params + a.nd.n_mlp);
    // (from the argument segment: it ends up in the buffer descriptor, which must be scalar)
    const LevelRegs lregs = load_level_regs_uniform(a.lt, L, lane); const uint32_t table_bytes = a.lt.offset[L] * 4u;
    for (uint32_t ray = blockIdx.x * S::WAVES + wave; ray < n_rays; ray += gridDim.x * S::WAVES) {
        float o0 = 1.f, o1 = 1.f, o2 = 1.f, od = 0.f, om_ = 0.f;
        if (a.b.ray_flag[ray]) {
            const float t0 = a.b.ray_t0[ray], t1 = a.b.ray_t1[ray], dtr = (t1 - t0) / (float)S2;
            float Tc = 1.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, tlast = 0.f;
            for (uint32_t tile = 0; tile < 2u; ++tile) {
                if (Tc < kTransmittanceEps) break;
                const uint32_t k = tile * 32u + (uint32_t)n;
                const float t = fmaf(dtr, (float)k + render_rand(a.oc, idx_base + ray * S2 + k), t0);
                float x[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { const float p = fmaf(t, a.b.ray_d[3 * ray + d], a.b.ray_o[3 * ray + d]);
                    x[d] = (p - a.oc.aabb.mn[d]) / (a.oc.aabb.mx[d] - a.oc.aabb.mn[d]); }
                TileState<EPAD, W, NH> ts;
                tile_forward<EPAD, W, NH>(ts, frags, lregs, table, table_bytes, L, x, lane);
                const float c0 = logistic_f(ts.out4[0]), c1 = logistic_f(ts.out4[1]), c2 = logistic_f(ts.out4[2]), sigma = __expf(ts.out4[3]);
                float tprev = lane_prev(t, tlast); if (n == 0) tprev = tlast;
                const float alpha = 1.f - __expf(-sigma * (t - tprev)), omv = 1.f - alpha;
                const float tincl = scan_mul32(omv) * Tc;
                float T = lane_prev(tincl, Tc); if (n == 0) T = Tc;
                const bool active = T >= kTransmittanceEps;
                const int nact = __popc((uint32_t)__ballot(active));
                const float wgt = active ? alpha * T : 0.f;
                r0 += lane_bcast(scan_add32(wgt * c0), 31); r1 += lane_bcast(scan_add32(wgt * c1), 31); r2 += lane_bcast(scan_add32(wgt * c2), 31);
                dep += lane_bcast(scan_add32(wgt * t), 31);
                Tc = (nact > 0) ? lane_bcast(tincl, nact > 0 ? nact - 1 : 0) : Tc;      // all 64 lanes carry half-wave 0's state (uniform control flow)
                tlast = lane_bcast(t, 31);
            }
            if (1.f - Tc > 0.5f) { o0 = r0 + Tc; o1 = r1 + Tc; o2 = r2 + Tc; od = dep / a.b.ray_dn[ray]; om_ = 1.f; }      // :1213-1220
        }
        if (lane == 0) { rgb[3 * ray] = o0; rgb[3 * ray + 1] = o1; rgb[3 * ray + 2] = o2; depth[ray] = od; mask[ray] = om_; }
    }
}

// ------------------------------------------------------------------ occupancy grid (N1: forward-pass skipping, default off)
// BASELINE.json's north star names occupancy-grid skipping; the reference has none (it always takes 32 uniform samples inside the box,
// nerf_model.cu:536-566), so the feature is opt-in (mon_config::occupancy_skip) and the parity tests run without it.  A kOccRes^3 bit grid over
// the object's box is refreshed from the CURRENT training weights every kOccInterval iterations after a warm-up: one wavefront evaluates the
// network's raw density at the centres of 32 cells (the same tile_forward as training) and ballots "density above the threshold" into one
// word; a second pass dilates by one cell in every direction.  k_fused_train then skips the gathers of samples in empty cells.
template <int EPAD, int W, int NH>
__global__ void __launch_bounds__(256) k_occ_density(FusedArgs a, float raw_threshold, uint32_t* __restrict__ bits_out) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    LevelLds* llt = reinterpret_cast<LevelLds*>(smem + S::FRAG_BYTES);
    build_fragments<EPAD, W, NH>(frags, llt, a, false);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31;
    const half2_t* table = reinterpret_cast<const half2_t*>(a.params + a.nd.n_mlp);
    constexpr uint32_t n_words = kOccRes * kOccRes * kOccRes / 32;
    const LevelRegs lregs = load_level_regs_uniform(a.lt, a.nd.L, lane); const uint32_t table_bytes = a.lt.offset[a.nd.L] * 4u;
    for (uint32_t word = blockIdx.x * S::WAVES + wave; word < n_words; word += gridDim.x * S::WAVES) {
        const uint32_t cell = word * 32u + (uint32_t)n, cx = cell % kOccRes, cy = (cell / kOccRes) % kOccRes, cz = cell / (kOccRes * kOccRes);
        const float x[3] = { ((float)cx + 0.5f) / (float)kOccRes, ((float)cy + 0.5f) / (float)kOccRes, ((float)cz + 0.5f) / (float)kOccRes };
        TileState<EPAD, W, NH> ts;
        tile_forward<EPAD, W, NH>(ts, frags, lregs, table, table_bytes, a.nd.L, x, lane);
        // raw channel 3 = log density (network_to_density = exp, nerf_model.cu:49)
        const uint32_t occ = (uint32_t)__ballot(lane < 32 && ts.out4[3] > raw_threshold);
        if (lane == 0) bits_out[word] = occ;
    }
}
// a cell stays live if it or any of its 26 neighbours is occupied (the network is only sampled at cell centres)
__global__ void __launch_bounds__(256) k_occ_dilate(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    constexpr int WPR = kOccRes / 32;                                                   // words per x row
    const uint32_t word = blockIdx.x * blockDim.x + threadIdx.x;
    if (word >= (uint32_t)(kOccRes * kOccRes * WPR)) return;
    const int wx = (int)(word % WPR), cy = (int)((word / WPR) % kOccRes), cz = (int)(word / (WPR * kOccRes));
    uint32_t acc = 0u;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
        const int y = cy + dy, z = cz + dz; if (y < 0 || y >= kOccRes || z < 0 || z >= kOccRes) continue;
        const uint32_t* row = in + ((size_t)z * kOccRes + y) * WPR;
        const uint32_t w = row[wx], wl = wx > 0 ? row[wx - 1] : 0u, wr = wx + 1 < WPR ? row[wx + 1] : 0u;
        acc |= w | (w << 1) | (w >> 1) | (wl >> 31) | (wr << 31);
    }
    out[word] = acc;
}
template <int EPAD, int W, int NH>
static void occ_update_t(hipStream_t s, const FusedArgs& a, float raw_threshold, uint32_t* tmp, uint32_t* bits) {
    using S = FusedShape<EPAD, W, NH>;
    constexpr uint32_t n_words = kOccRes * kOccRes * kOccRes / 32;
    hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::F_WOT * 512 + 255) / 256), dim3(256), 0, s, a.params, a.nd.L,
            const_cast<uint16_t*>(a.frag_image), (const DevState*)nullptr);
    hipLaunchKernelGGL((k_occ_density<EPAD, W, NH>), dim3(n_words / S::WAVES), dim3(256), S::FRAG_BYTES + S::LT_BYTES, s, a, raw_threshold, tmp);
    hipLaunchKernelGGL(k_occ_dilate, dim3((n_words + 255) / 256), dim3(256), 0, s, tmp, bits);
}
template <int EPAD, int W, int NH>
static void fused_render_t(hipStream_t s, const FusedArgs& a, uint32_t n_rays, uint32_t idx_base, float* rgb, float* depth, float* mask) {
    using S = FusedShape<EPAD, W, NH>;
    const uint32_t smem = S::FRAG_BYTES + S::LT_BYTES;
    uint32_t grid = (n_rays + 3) / 4; if (grid > 2048u) grid = 2048u;
    // first chunk of a render call
    if (a.keep_zero & 1u) hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::F_WOT * 512 + 255) / 256), dim3(256), 0, s, a.params, a.nd.L,
            const_cast<uint16_t*>(a.frag_image), (const DevState*)nullptr);
    hipLaunchKernelGGL((k_fused_render<EPAD, W, NH>), dim3(grid), dim3(256), smem, s, a, n_rays, idx_base, rgb, depth, mask);
}


void launch_fused_render(hipStream_t s, const LevelFast& lt, const NetDims& nd, const uint16_t* params, const BatchPtrs& b, const ObjectConst& oc,
        uint32_t n_rays, uint32_t idx_base, float* rgb, float* depth, float* mask, uint16_t* frag_image, int build_image) {
    // `keep_zero` doubles as "build the fragment image first" on the host side of the render path
    FusedArgs a{ lt, nd, oc, b, params, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, frag_image, build_image ? 1u : 0u };
    MON_FUSED_DISPATCH(fused_render_t, s, a, n_rays, idx_base, rgb, depth, mask);
}


void launch_occupancy_update(hipStream_t s, const LevelFast& lt, const NetDims& nd, const uint16_t* params, const ObjectConst& oc, uint16_t* frag_image,
        float raw_threshold, uint32_t* tmp, uint32_t* bits) {
    FusedArgs a{}; a.lt = lt; a.nd = nd; a.oc = oc; a.params = params; a.frag_image = frag_image;
    MON_FUSED_DISPATCH(occ_update_t, s, a, raw_threshold, tmp, bits);
}

}  // namespace mon
