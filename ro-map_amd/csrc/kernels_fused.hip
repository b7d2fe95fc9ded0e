// kernels_fused.hip -- k_fused_train: forward + backward of one object NeRF for gfx950 (backend 1), its launcher, the weight-fragment image and the
// stand-alone candidate / fragment kernel.  Mapping and shared device code: fused_device.h.  The encoded features come from k_encode_tiles (PRE variant,
// kernels_encode.hip) or from the kernel's own gathers (tables beyond the LDS tiles, occupancy skipping, debug dump).
#include "fused_device.h"

namespace mon {

// ------------------------------------------------------------------ fused training kernel
// (PRE: the encode was done by k_encode_tiles -- features are loaded, not gathered)
template <int EPAD, int W, int NH, bool DUMP, bool ATOMIC_LEVELS, bool OCC = false, bool PRE = false>
// (two waves per SIMD; the widest networks -- two hidden layers of 64, one of 128: tcnn FullyFusedMLP's largest width -- hold their dW accumulators and
// activations in up to 512 registers at one wave per SIMD)
__global__ void __launch_bounds__(256, (((NH == 2 && W == 64) || W == 128) ? 1 : 2)) k_fused_train(FusedArgs a) {
    using S = FusedShape<EPAD, W, NH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* frags = reinterpret_cast<half_t*>(smem);
    LevelLds* llt = reinterpret_cast<LevelLds*>(smem + S::FRAG_BYTES);
    unsigned long long* cwords = reinterpret_cast<unsigned long long*>(smem + S::FRAG_BYTES + 512);      // [256]
    uint32_t* cprefix = reinterpret_cast<uint32_t*>(smem + S::FRAG_BYTES + 512 + 2048);                   // [257] exclusive prefix, [nwords] = total
    unsigned char* dyn = smem + S::FRAG_BYTES + S::LT_BYTES;
#ifdef MON_FUSED_TIMING
    TimingCtx tcx; for (float& v : tcx.acc) v = 0.f; tcx.last = clock64(); TimingCtx* tc = &tcx;
    // start time (100 MHz ticks, low 24 bits), HW_ID[15:0]
    tcx.acc[13] = (float)(uint32_t)(wall_clock64() & 0xffffffull); tcx.acc[15] = (float)(uint32_t)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
#else
    TimingCtx* tc = nullptr;
#endif
    // ---- prologue.  Everything a wave needs before its first gather is requested up front, and only the weight fragments (first read by the MLP) wait for
    //      the workgroup barrier: the iteration counter, the candidates' ballot words, the fragment image.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int L = a.nd.L, LPH = (L + 1) >> 1;
    const uint32_t R = a.oc.R, iter = a.st->iter;
    // ---- ray compaction (fill_rollover_rays :280-294 without a kernel of its own): training ray j is valid candidate number (j mod n_valid) in candidate
    //      order.  Up to 4096 candidates (64 ballot words) every WAVE keeps the words and their exclusive prefix in registers, one word per lane, and finds a
    //      ray's candidate with a ballot and two v_readlanes; larger batches go through a table in LDS built by wave 0.
    const uint32_t nwords = a.oc.R >> 6;                                           // <= 256 (fused_supported)
    const bool small = nwords <= 64u;                                              // uniform
    unsigned long long my_word = 0ull;
    // (uniform) the position pass left the compacted rays' records: no ballot words, no scan, no select
    const bool have_rec = PRE && a.b.ray_rec != nullptr;
    if (!have_rec && small && (uint32_t)lane < nwords) my_word = a.b.mask[lane];
    // (from the argument segment: it ends up in the buffer descriptor, which must be scalar)
    const LevelRegs lregs = load_level_regs_uniform(a.lt, L, lane); const uint32_t table_bytes = a.lt.offset[L] * 4u;
    build_fragments<EPAD, W, NH>(frags, llt, a, true);
    half_t* scr = reinterpret_cast<half_t*>(dyn + wave * S::SCR_BYTES);
    // pad feature rows stay zero; every other row is rewritten per ray before it is read
    for (int i = 2 * a.nd.L * 32 + lane; i < EPAD * 32; i += 64) scr[S::SCR_E + i] = (half_t)0.f;
    uint32_t my_excl = 0u, nvalid = 0u;
    if (have_rec) nvalid = a.st->n_valid_pre;
    else if (small) { const uint32_t c = __popcll(my_word), inc = scan_add64_u32(c); my_excl = inc - c;
        nvalid = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63); }
    else if (wave == 0) {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nwords; base += 64) {
            const unsigned long long wd = (base + lane < nwords) ? a.b.mask[base + lane] : 0ull;
            const uint32_t c = __popcll(wd); const uint32_t inc = scan_add64_u32(c);
            if (base + lane < nwords) { cwords[base + lane] = wd; cprefix[base + lane] = carry + inc - c; }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) cprefix[nwords] = carry;
    }
    // candidate number `kth` -> candidate index (wave-uniform)
    const auto select = [&](uint32_t kth) -> uint32_t {
        unsigned long long wd; uint32_t kk, lo;
        if (small) {
            lo = (uint32_t)__popcll(__ballot(my_excl <= kth)) - 1u;                  // the prefix is non-decreasing (lanes past the last word hold the total)
            wd = ((unsigned long long)lane_u((uint32_t)(my_word >> 32), (int)lo) << 32) | lane_u((uint32_t)my_word, (int)lo);
            kk = kth - lane_u(my_excl, (int)lo);
        } else {
            lo = 0; uint32_t hi = nwords - 1u;
            while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (cprefix[mid] <= kth) lo = mid; else hi = mid - 1u; }
            wd = cwords[lo]; kk = kth - cprefix[lo];
        }
        uint32_t pos = 0;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t c = __popcll(wd & ((1ull << sh) - 1ull)); if (kk >= c) { kk -= c; wd >>= sh; pos += sh; } }
        return (lo << 6) + pos;
    };
    // the candidate's record, one field per lane (rgba, t0, t1, d[3], o[3], depth): ONE load, requested a whole ray ahead of its use
    const char* rec_base; uint32_t rec_mul = 1u;
    {   const void* fb = lane == 0 ? (const void*)a.b.cand_rgba : lane == 1 ? (const void*)a.b.cand_t0 : lane == 2 ? (const void*)a.b.cand_t1
                       : lane < 6 ? (const void*)(a.b.cand_d + (lane - 3)) : lane < 9 ? (const void*)(a.b.cand_o + (lane - 6)) : (const void*)a.b.cand_depth;
        rec_base = reinterpret_cast<const char*>(fb); if (lane >= 3 && lane < 9) rec_mul = 3u; }
    const auto load_record = [&](uint32_t cand) -> uint32_t { uint32_t v = 0u;
        if (lane < 10) v = *reinterpret_cast<const uint32_t*>(rec_base + 4u * (size_t)(cand * rec_mul)); return v; };
    // (PRE with records: lane l < 10 takes field l of ray `r`'s 12-float record -- one coalesced 40-byte load, requested a ray ahead like the candidate record)
    const auto load_ray_rec = [&](uint32_t r) -> uint32_t { uint32_t v = 0u;
        if (lane < 12) v = reinterpret_cast<const uint32_t*>(a.b.ray_rec)[12u * (size_t)r + (uint32_t)lane]; return v; };
    const uint32_t ray0 = blockIdx.x * S::WAVES + wave;
    uint32_t cand = 0u, rec = 0u;
    if (have_rec) { if (nvalid != 0u && ray0 < R) rec = load_ray_rec(ray0); }
    else if (small && nvalid != 0u && ray0 < R) { cand = select(ray0 % nvalid); rec = load_record(cand); }
    __syncthreads();
    if (!have_rec && !small) { nvalid = cprefix[nwords]; if (nvalid != 0u && ray0 < R) { cand = select(ray0 % nvalid); rec = load_record(cand); } }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->n_valid = nvalid; a.st->loss_sum = 0.f; }
    if (nvalid == 0u) return;                                                        // batch skipped (uniform over the grid)

    // wave-uniform
    const uint32_t lds_level_mask = (ATOMIC_LEVELS && a.big_switch != 0u && big_levels_binned(a.st->n_scatter_last, a.big_switch)) ? 0xffffffffu
            : a.lds_level_mask;
    const half2_t* table = reinterpret_cast<const half2_t*>(a.params + a.nd.n_mlp);
    typedef __attribute__((address_space(1))) half2_t gh2;
    gh2* gtable = (gh2*)reinterpret_cast<half2_t*>(a.ggrid);
    const float ls = a.oc.loss_scale / (float)R;
    const uint32_t n_bins = a.n_bins;                                                // ray bins of the compacted gradient rows

    float16_t dW0[S::MB], dWo[S::MB], dW1[NH == 2 ? S::MB : 1][NH == 2 ? S::MB : 1];
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) { dW0[mb] = float16_t{ 0 }; dWo[mb] = float16_t{ 0 }; }
    if constexpr (NH == 2) {
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < S::MB; ++nb) dW1[mb][nb] = float16_t{ 0 };
    }
    float loss_acc = 0.f;
    // ---- phase stagger.  All waves of a CU share one texture-address path, and a ray's 64 gather instructions keep it busy for ~1.5 us; the waves start
    // together and their phases have equal lengths, so they would ALL gather, then ALL run the MLP / composite / backward with the address path idle.  Half of
    // the waves therefore start late by about one gather phase: from then on one group computes while the other gathers.
    if (a.stagger & 0xffffu) {
        const uint32_t mode = (a.stagger >> 16) & 3u;
        const uint32_t slot = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) ;      // HW_ID.wave_id: this wave's slot on its SIMD
        const bool late = mode == 0u ? (slot & 1u) != 0u : mode == 1u ? blockIdx.x >= (gridDim.x >> 1) : mode == 2u ? (wave & 1) != 0 : wave != 0;
        const uint32_t units = (a.stagger & 0xffffu) * (mode == 3u ? (uint32_t)wave : 1u);        // mode 3: graduated, wave w waits w units
        if (late) for (uint32_t i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    tstamp(tc, 0);

    // ---- ray loop.  The candidate record of ray k + 1 is requested while ray k is processed.  (Also tried: requesting the first EB level pairs of ray k + 1
    //      before ray k's MLP / composite / backward -- no gain, 48.4 vs 48.6 us without the stagger and 9 spilled registers: the gather phase is bound by the
    //      chip-wide L2 request rate, not by its start-up latency; an encode-only variant of this kernel took 39.5 us, profiles/r03_*.)
    struct RaySample { float t, x[3]; uint32_t rgba; float tdp, t0, t1; bool live, any; };
    // sample n of the ray described by record `recv` (GenerateInputPoints, nerf_model.cu:553-566)
    const auto ray_sample = [&](uint32_t recv, uint32_t ray) {
        RaySample q; q.rgba = lane_u(recv, 0); q.tdp = __builtin_bit_cast(float, lane_u(recv, 9));
        const float t0 = __builtin_bit_cast(float, lane_u(recv, 1)), t1 = __builtin_bit_cast(float, lane_u(recv, 2));
        const float rd[3] = { __builtin_bit_cast(float, lane_u(recv, 3)), __builtin_bit_cast(float, lane_u(recv, 4)),
                __builtin_bit_cast(float, lane_u(recv, 5)) };
        const float ro[3] = { __builtin_bit_cast(float, lane_u(recv, 6)), __builtin_bit_cast(float, lane_u(recv, 7)),
                __builtin_bit_cast(float, lane_u(recv, 8)) };
        const float dtr = (t1 - t0) / 32.0f; q.t0 = t0; q.t1 = t1;
        q.t = fmaf(dtr, (float)n + batch_rand(a.oc, kStreamDt, iter, ray * 32u + (uint32_t)n), t0);
#pragma unroll
        for (int d = 0; d < 3; ++d) { const float p = fmaf(q.t, rd[d], ro[d]); q.x[d] = (p - a.oc.aabb.mn[d]) / (a.oc.aabb.mx[d] - a.oc.aabb.mn[d]); }
        // occupancy-grid skipping (default off): a sample whose cell the grid marks empty is not evaluated -- no gathers, alpha = 0, no gradient
        q.live = true;
        // (level-tile chain: the position pass looked the cells up and left the ray's 32 live bits in word 11 of its record -- k_encode_tiles encoded exactly
        // those samples, and a grid refreshed since then takes effect with the next position pass)
        if constexpr (OCC && PRE) q.live = ((lane_u(recv, 11) >> (uint32_t)n) & 1u) != 0u;
        else if constexpr (OCC) {
            const uint32_t cx = (uint32_t)min(max((int)(q.x[0] * (float)kOccRes), 0), kOccRes - 1),
                    cy = (uint32_t)min(max((int)(q.x[1] * (float)kOccRes), 0), kOccRes - 1),
                    cz = (uint32_t)min(max((int)(q.x[2] * (float)kOccRes), 0), kOccRes - 1);
            q.live = ((a.occ_bits[((cz * kOccRes + cy) * kOccRes + cx) >> 5] >> (cx & 31u)) & 1u) != 0u;
        }
        q.any = !OCC || __ballot(q.live) != 0ull;                                         // false: the whole ray crosses empty cells only, nothing to evaluate
        return q;
    };
    const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(table, table_bytes);
    const uint32_t ray_stride = gridDim.x * S::WAVES;
    // PRE: lane (n, h) reads the features of the levels half-wave h owns, one dword (half2) per level at [level][ray * 32 + n] -- 128 contiguous bytes per
    // half-wave and level; a ray's eight loads are requested one ray ahead, like its candidate record
    uint32_t epre[S::LLV];
    const auto load_encoded = [&](uint32_t r) {
#pragma unroll
        for (int il = 0; il < S::LLV; ++il) {
            const int level = h * LPH + il; epre[il] = 0u;
            if (il < LPH && level < L) epre[il] = reinterpret_cast<const uint32_t*>(a.e_soa)[(size_t)level * (R * 32u) + r * 32u + (uint32_t)n];
        }
    };
    if constexpr (PRE) { if (ray0 < R) load_encoded(ray0); }
    for (uint32_t ray = ray0; ray < R; ray += ray_stride) {
        const RaySample cur = ray_sample(rec, ray); const uint32_t cand_this = have_rec ? lane_u(rec, 10) : cand;      // (only the debug dump reads it)
        const uint32_t kth = ray % nvalid, rgba = cur.rgba, s_idx = ray * 32u + (uint32_t)n;
        const float t = cur.t, tdp = cur.tdp, t0 = cur.t0, t1 = cur.t1; const bool live = cur.live, is_obj = (rgba >> 24) != 0u;
        const float x[3] = { cur.x[0], cur.x[1], cur.x[2] };
        tstamp(tc, 1);
        TileState<EPAD, W, NH> ts;
        if constexpr (PRE) {
#pragma unroll
            for (int il = 0; il < S::LLV; ++il) {
                // (a dead sample's slot was not encoded: zero features, like the gather chain's masked loads)
                const half2_t v = __builtin_bit_cast(half2_t, (!OCC || live) ? epre[il] : 0u); ts.ef[2 * il] = v.x; ts.ef[2 * il + 1] = v.y; }
        } else if (cur.any) { GatherWindow<EPAD, W, NH> gw; encode_begin<EPAD, W, NH, OCC>(gw, lregs, rsrc, x, lane, live);
            encode_finish<EPAD, W, NH, OCC>(ts, gw, lregs, rsrc, x, lane, L, live); }
        tstamp(tc, 2);
        // the next ray's candidate record is requested HERE, behind this ray's last gather: vmcnt retires in order, so a load issued before the gathers would
        // have to land before the first level pair can be consumed; now its latency runs under the MLP, composite and backward pass
        if (ray + ray_stride < R) { if (have_rec) rec = load_ray_rec(ray + ray_stride); else { cand = select((ray + ray_stride) % nvalid);
                rec = load_record(cand); } if constexpr (PRE) load_encoded(ray + ray_stride); }
        if (!OCC || __ballot(live) != 0ull) mlp_forward<EPAD, W, NH>(ts, frags, lane);
        else {
#pragma unroll
            for (int i = 0; i < EPAD / 2; ++i) ts.ef[i] = (half_t)0.f;
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) { ts.h0[mb][0] = half8_t{}; ts.h0[mb][1] = half8_t{}; if constexpr (NH == 2) { ts.h1[mb][0] = half8_t{};
                    ts.h1[mb][1] = half8_t{}; } }
#pragma unroll
            for (int c = 0; c < 4; ++c) ts.out4[c] = 0.f;
        }

        tstamp(tc, 4);
        // ---- composite (VolumeRender :762-813) as wave scans over lanes 0..31
        const float v0 = ts.out4[0], v1 = ts.out4[1], v2 = ts.out4[2], v3 = ts.out4[3];
        const float c0 = logistic_f(v0), c1 = logistic_f(v1), c2 = logistic_f(v2), sigma = __expf(v3);
        float tprev = lane_prev(t, 0.f); if (n == 0) tprev = 0.f;                         // :770 last_distance = 0
        const float dt = t - tprev;
        const float alpha = (!OCC || live) ? 1.f - __expf(-sigma * dt) : 0.f, om = 1.f - alpha;
        const float tincl = scan_mul32(om);                                               // T after this sample
        float T = lane_prev(tincl, 1.f); if (n == 0) T = 1.f;                             // T before this sample
        const bool active = T >= kTransmittanceEps;                                        // :774 early-out (T is non-increasing)
        const int nact = __popc((uint32_t)__ballot(active));                               // lanes 0..31 = the ray's samples
        const float Tfin = lane_bcast(tincl, nact - 1);                                    // broadcast from half-wave 0 (sample 0 is always active: nact >= 1)
        const float wgt = active ? alpha * T : 0.f;
        const float p0 = scan_add32(wgt * c0), p1 = scan_add32(wgt * c1), p2 = scan_add32(wgt * c2), pd = scan_add32(wgt * t);
        // :760, :438-441
        const float bg0 = batch_rand(a.oc, kStreamColor, iter, 3u * kth), bg1 = batch_rand(a.oc, kStreamColor, iter, 3u * kth + 1u),
                bg2 = batch_rand(a.oc, kStreamColor, iter, 3u * kth + 2u);
        const float rgb0 = lane_bcast(p0, 31) + Tfin * bg0, rgb1 = lane_bcast(p1, 31) + Tfin * bg1, rgb2 = lane_bcast(p2, 31) + Tfin * bg2;
        const float dep = lane_bcast(pd, 31), mask = 1.f - Tfin;
        // ---- loss + dL/dO (VolumeRenderGradient_No_Compacted :853-953)
        const float tg0 = is_obj ? (float)(rgba & 0xffu) / 255.0f : bg0, tg1 = is_obj ? (float)((rgba >> 8) & 0xffu) / 255.0f : bg1, tg2 = is_obj
                ? (float)((rgba >> 16) & 0xffu) / 255.0f : bg2;
        const float e0 = rgb0 - tg0, e1 = rgb1 - tg1, e2 = rgb2 - tg2;
        const float g0 = 2.f * e0, g1 = 2.f * e1, g2 = 2.f * e2;
        float dl_dd = 0.f; if (tdp > 0.f) dl_dd = 0.5f * ((dep - tdp >= 0.f) ? 1.f : -1.f);
        const float mean_loss = (e0 * e0 + e1 * e1 + e2 * e2) / 3.f;
        const float loss = is_obj ? mean_loss + dl_dd * (dep - tdp) + (1.f - mask) : mean_loss + mask;
        half8_t bdo;
#pragma unroll
        for (int j = 0; j < 8; ++j) bdo[j] = (half_t)0.f;
        if (active && h == 0 && (!OCC || live)) {
            const float Tn = tincl;                                                       // T after the update (:912)
            const float s0 = rgb0 - p0, s1 = rgb1 - p1, s2 = rgb2 - p2;                   // suffix :915
            bdo[0] = (half_t)(ls * ((wgt * g0) * (c0 * (1.f - c0))));
            bdo[1] = (half_t)(ls * ((wgt * g1) * (c1 * (1.f - c1))));
            bdo[2] = (half_t)(ls * ((wgt * g2) * (c2 * (1.f - c2))));
            const float dsig = __expf(clamp_f(v3, -15.f, 15.f));
            const float depth_sup = dl_dd * (Tn * t - (dep - pd));
            const float dmask = 1.f - mask;
            float dl;
            if (is_obj) {
                const float dlm = 0.5f * (mask >= 1.f ? 1.f : -1.f);
                const float dot = g0 * (Tn * c0 - s0) + g1 * (Tn * c1 - s1) + g2 * (Tn * c2 - s2);
                dl = dsig * dt * (dot + depth_sup + dlm * dmask);
            } else {
                const float dlm = 0.5f * (mask >= 0.f ? 1.f : -1.f);
                dl = dsig * dt * dlm * dmask + dsig * 0.01f;
            }
            bdo[3] = (half_t)(ls * dl);
        }
        // ---- which samples carry a gradient at all.  dL/dO is fp16 with a loss scale of 128/R: once empty space is learnt
        //      (sigma = exp(-15), alpha * T -> 0) it underflows to exact zeros, 94-98 % of the samples after ~200 steps
        //      (tools/zero_grad_fraction.py).  Zero rows contribute exact zeros to dW and dE, so a ray without any is done
        // here, and only the non-zero samples are handed to k_grid_scatter, compacted into ray bins (ray mod n_bins, up to 128: one returning atomic per ray on
        // 16 counters cost 7 us of contention).  Inside a
        //      bin the order is whatever the atomics give -- irrelevant, the scatter's integer accumulation is exact -- while
        //      bin membership, and with it every fp16-rounded partial table, is a fixed function of the ray: the result stays
        //      deterministic.  The slot reservation is issued now and consumed after the MFMAs.
        const uint2 bdo_bits = __builtin_bit_cast(uint2, half4_t{ bdo[0], bdo[1], bdo[2], bdo[3] });
        // option keep_zero_samples: no skipping (the exactness test's A/B)
        const uint32_t nz32 = a.keep_zero ? 0xffffffffu : (uint32_t)__ballot(h == 0 && ((bdo_bits.x | bdo_bits.y) & 0x7fff7fffu) != 0u);
        const uint32_t nz_cnt = __popc(nz32);
        uint32_t slot_base = 0u;
        if (nz_cnt != 0u && lane == 0 && lds_level_mask) slot_base = atomicAdd(&a.st->n_scatter[scatter_counter(iter, ray & (n_bins - 1u))], nz_cnt);
        if (lane == 0) {
            loss_acc += loss;
            a.b.rgb_ray[3 * ray] = rgb0; a.b.rgb_ray[3 * ray + 1] = rgb1; a.b.rgb_ray[3 * ray + 2] = rgb2;
            a.b.depth_ray[ray] = dep; a.b.mask_ray[ray] = mask; a.b.loss_ray[ray] = loss;
        }
        if (DUMP) {
            if (lane == 0) {
                for (int d = 0; d < 3; ++d) { a.b.ray_o[3 * ray + d] = a.b.cand_o[3 * cand_this + d]; a.b.ray_d[3 * ray + d] = a.b.cand_d[3 * cand_this + d]; }
                a.b.ray_t0[ray] = t0; a.b.ray_t1[ray] = t1; a.b.ray_dn[ray] = a.b.cand_dn[cand_this]; a.b.ray_flag[ray] = is_obj ? 1 : 0;
                a.b.target_depth[ray] = tdp;
                a.b.bgcol[3 * ray] = bg0; a.b.bgcol[3 * ray + 1] = bg1; a.b.bgcol[3 * ray + 2] = bg2; a.b.target[3 * ray] = tg0; a.b.target[3 * ray + 1] = tg1;
                a.b.target[3 * ray + 2] = tg2;
            }
            if (h == 0) {
                a.b.pts[3 * s_idx] = x[0]; a.b.pts[3 * s_idx + 1] = x[1]; a.b.pts[3 * s_idx + 2] = x[2]; a.b.tdist[s_idx] = t;
                half4_t o4 = { (half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3 }; reinterpret_cast<half4_t*>(a.b.O)[s_idx] = o4;
                half4_t d4 = { bdo[0], bdo[1], bdo[2], bdo[3] }; reinterpret_cast<half4_t*>(a.b.dO)[s_idx] = d4;
            }
            half_t* Eo = reinterpret_cast<half_t*>(a.b.E) + (size_t)s_idx * EPAD;
#pragma unroll
            for (int il = 0; il < S::LLV; ++il) { const int level = h * LPH + il; if (il < LPH && level < L) { Eo[2 * level] = ts.ef[2 * il];
                    Eo[2 * level + 1] = ts.ef[2 * il + 1]; } }
            half_t* Ho = reinterpret_cast<half_t*>(a.b.Hid) + (size_t)s_idx * W * NH;
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    Ho[32 * mb + rho(h, j)] = ts.h0[mb][0][j]; Ho[32 * mb + rho(h, 8 + j)] = ts.h0[mb][1][j];
                    if constexpr (NH == 2) { Ho[W + 32 * mb + rho(h, j)] = ts.h1[mb][0][j]; Ho[W + 32 * mb + rho(h, 8 + j)] = ts.h1[mb][1][j]; }
                }
        }

        if (nz_cnt != 0u || DUMP) {
        // ---- transposes needed by the weight gradients
        {
#pragma unroll
        for (int il = 0; il < S::LLV; ++il) {
            const int level = h * LPH + il;
            if (il < LPH && level < L) { scr[S::SCR_E + (2 * level) * 32 + n] = ts.ef[2 * il]; scr[S::SCR_E + (2 * level + 1) * 32 + n] = ts.ef[2 * il + 1]; }
        }
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb) {
            if constexpr (NH == 2) { scratch_store_units(scr + S::SCR_HB, mb, n, h, ts.h0[mb][0], ts.h0[mb][1]);
                scratch_store_units(scr + S::SCR_HA, mb, n, h, ts.h1[mb][0], ts.h1[mb][1]); }
            else scratch_store_units(scr + S::SCR_HA, mb, n, h, ts.h0[mb][0], ts.h0[mb][1]);
        }
        }

        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) scr[S::SCR_DO + c * 32 + n] = bdo[c];
        }
        tstamp(tc, 5);
        // ---- backward: dWout += H_last^T-side outer products (K = samples, via the LDS transposes)
        const int m = n;     // A-fragment row / B-fragment column of this lane
        {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8_t bcol;
#pragma unroll
            for (int j = 0; j < 8; ++j) bcol[j] = (half_t)0.f;
            if (m < kOut) bcol = *reinterpret_cast<const half8_t*>(scr + S::SCR_DO + m * 32 + 16 * s + 8 * h);
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) {
                const half8_t arow = *reinterpret_cast<const half8_t*>(scr + S::SCR_HA + (32 * mb + m) * 32 + 16 * s + 8 * h);
                dWo[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow, bcol, dWo[mb], 0, 0, 0);
            }
        }
        }
        // ---- dH of the last hidden layer = relu' * (Wout^T dO)
        half8_t dhl[S::MB][2];
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb) {
            float16_t acc = float16_t{ 0 };
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_WOT + mb, lane), bdo, acc, 0, 0, 0);
            if constexpr (NH == 2) mask_pack(acc, ts.h1[mb][0], ts.h1[mb][1], dhl[mb][0], dhl[mb][1]);
            else mask_pack(acc, ts.h0[mb][0], ts.h0[mb][1], dhl[mb][0], dhl[mb][1]);
            scratch_store_units(scr + S::SCR_HA, mb, n, h, dhl[mb][0], dhl[mb][1]);       // H_last no longer needed: reuse as dH_last
        }
        half8_t dh0[S::MB][2];
        if constexpr (NH == 2) {
            // dW1[u2][u1] += dH1[u2][n] * H0[n][u1]
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int nb = 0; nb < S::MB; ++nb) {
                    const half8_t bcol = *reinterpret_cast<const half8_t*>(scr + S::SCR_HB + (32 * nb + m) * 32 + 16 * s + 8 * h);
#pragma unroll
                    for (int mb = 0; mb < S::MB; ++mb) {
                        const half8_t arow = *reinterpret_cast<const half8_t*>(scr + S::SCR_HA + (32 * mb + m) * 32 + 16 * s + 8 * h);
                        dW1[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow, bcol, dW1[mb][nb], 0, 0, 0);
                    }
                }
            // dH0 = relu' * (W1^T dH1)
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) {
                float16_t acc = float16_t{ 0 };
#pragma unroll
                for (int s = 0; s < S::KSW; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W1T + mb * S::KSW + s, lane),
                        dhl[s >> 1][s & 1], acc, 0, 0, 0);
                mask_pack(acc, ts.h0[mb][0], ts.h0[mb][1], dh0[mb][0], dh0[mb][1]);
                scratch_store_units(scr + S::SCR_HB, mb, n, h, dh0[mb][0], dh0[mb][1]);   // H0 no longer needed: reuse as dH0
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb) { dh0[mb][0] = dhl[mb][0]; dh0[mb][1] = dhl[mb][1]; }
        }
        tstamp(tc, 6);
        // ---- dW0[u][f] += dH0[u][n] * E[n][f]
        {
            const half_t* dh0_scr = scr + (NH == 2 ? S::SCR_HB : S::SCR_HA);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8_t bcol = *reinterpret_cast<const half8_t*>(scr + S::SCR_E + (m & (EPAD - 1)) * 32 + 16 * s + 8 * h);
#pragma unroll
                for (int mb = 0; mb < S::MB; ++mb) {
                    const half8_t arow = *reinterpret_cast<const half8_t*>(dh0_scr + (32 * mb + m) * 32 + 16 * s + 8 * h);
                    dW0[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow, bcol, dW0[mb], 0, 0, 0);
                }
            }
        }
        // ---- dE = W0^T dH0, rows permuted so register r of half h is local feature r of that half
        float16_t de = float16_t{ 0 };
#pragma unroll
        for (int s = 0; s < S::KSW; ++s) de = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(frags, S::F_W0T + s, lane), dh0[s >> 1][s & 1], de, 0, 0, 0);
        if (DUMP) {
            half_t* dEo = reinterpret_cast<half_t*>(a.b.dE) + (size_t)s_idx * EPAD;
            half_t* dHo = reinterpret_cast<half_t*>(a.b.dHid) + (size_t)s_idx * W * NH;
#pragma unroll
            for (int il = 0; il < S::LLV; ++il) { const int level = h * LPH + il; if (il < LPH && level < L) { dEo[2 * level] = (half_t)de[2 * il];
                    dEo[2 * level + 1] = (half_t)de[2 * il + 1]; } }
#pragma unroll
            for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dHo[(NH - 1) * W + 32 * mb + rho(h, j)] = dhl[mb][0][j]; dHo[(NH - 1) * W + 32 * mb + rho(h, 8 + j)] = dhl[mb][1][j];
                    if constexpr (NH == 2) { dHo[32 * mb + rho(h, j)] = dh0[mb][0][j]; dHo[32 * mb + rho(h, 8 + j)] = dh0[mb][1][j]; }
                }
        }
        // ---- grid backward (tcnn kernel_grid_backward).  Levels small enough for an LDS tile hand dL/dE to
        //      k_grid_scatter (global packed-f16 atomics sustain only ~21 Gop/s on gfx950, ~12x below the gather
        //      rate: profiles/); larger levels scatter here with 8 global_atomic_pk_add_f16 per level.
        tstamp(tc, 7);
        const uint32_t Btot = R * 32u;
        const bool mine = ((nz32 >> n) & 1u) != 0u;                                        // this lane's sample is one of the non-zero ones
        const uint32_t bin_cap = Btot / n_bins, in_bin = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_base) + __popc(nz32 & ((1u << n) - 1u));
        const uint32_t slot = (ray & (n_bins - 1u)) * bin_cap + in_bin;                    // a bin holds the samples of R / n_bins rays at most
        const bool do_store = mine && in_bin < bin_cap;
        // one 16-byte store per sample
        if (lds_level_mask && h == 0 && do_store) reinterpret_cast<float4_t*>(a.x_soa)[slot] = float4_t{ x[0], x[1], x[2], 0.f };
#pragma unroll
        for (int il = 0; il < S::LLV; ++il) {
            const int level = h * LPH + il;
            if (il < LPH && level < L) {
                const half_t q0 = (half_t)de[2 * il], q1 = (half_t)de[2 * il + 1];
                if (!ATOMIC_LEVELS || ((lds_level_mask >> level) & 1u)) {
                    // (clamped to the fixed-point range of the exact LDS accumulation, LevelFast::fix_clamp: |dL/dE| * fix_scale stays inside int32)
                    if (do_store) a.de_soa[(size_t)level * Btot + slot] = half2_t{ (half_t)clamp_f((float)q0, -a.lt.fix_clamp, a.lt.fix_clamp),
                            (half_t)clamp_f((float)q1, -a.lt.fix_clamp, a.lt.fix_clamp) };
                } else if constexpr (ATOMIC_LEVELS) {
                    const float gq0 = (float)q0, gq1 = (float)q1;
                    if (gq0 != 0.f || gq1 != 0.f) {
                        gh2* gl = gtable + llt->offset[level];
                        level_corners(*llt, level, x, [&](int, uint32_t idx, float wgt2) {
                            __builtin_amdgcn_global_atomic_fadd_v2f16(gl + idx, half2_t{ (half_t)(wgt2 * gq0), (half_t)(wgt2 * gq1) });
                            if (a.touched_grid) a.touched_grid[(llt->offset[level] + idx) >> 2] = 1;
                        });
                    }
                }
            }
        }
        }
        tstamp(tc, 8);
    }

    // ---- reduce the weight-gradient accumulators over the workgroup's waves, write one fp32 partial row -- in ACCUMULATOR layout (frag_layout.h acc_param):
    //      each wave stores its registers to a private LDS copy as 16-byte pieces, consecutive lanes at consecutive addresses (no bank conflicts, no
    //      transposition), then all threads sum the four copies with 16-byte reads and store the row coalesced.  The summing kernel maps columns to parameters.
    __syncthreads();
    {
        float* cp = reinterpret_cast<float*>(dyn) + (size_t)wave * (S::ACC_COLS + 64);
#pragma unroll
        for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                *reinterpret_cast<float4_t*>(cp + ((mb * 4 + rq) * 64 + lane) * 4) = float4_t{ dW0[mb][4 * rq], dW0[mb][4 * rq + 1], dW0[mb][4 * rq + 2],
                        dW0[mb][4 * rq + 3] };
                if (n < kOut) *reinterpret_cast<float4_t*>(cp + S::ACC_WO + ((mb * 4 + rq) * 8 + h * 4 + n) * 4) = float4_t{ dWo[mb][4 * rq],
                        dWo[mb][4 * rq + 1], dWo[mb][4 * rq + 2], dWo[mb][4 * rq + 3] };
                if constexpr (NH == 2) {
#pragma unroll
                    for (int nb = 0; nb < S::MB; ++nb)
                        *reinterpret_cast<float4_t*>(cp + S::ACC_W1 + (((mb * S::MB + nb) * 4 + rq) * 64 + lane) * 4) = float4_t{ dW1[mb][nb][4 * rq],
                                dW1[mb][nb][4 * rq + 1], dW1[mb][nb][4 * rq + 2], dW1[mb][nb][4 * rq + 3] };
                }
            }
        if (lane == 0) cp[S::ACC_COLS] = loss_acc;
    }
    __syncthreads();
    {
        const float* r0 = reinterpret_cast<const float*>(dyn);
        float* dst = a.partials + (size_t)blockIdx.x * (S::ACC_COLS + 64);
        constexpr int ST = S::ACC_COLS + 64;
        for (int i = threadIdx.x * 4; i < S::ACC_COLS; i += blockDim.x * 4) {
            const float4_t v0 = *reinterpret_cast<const float4_t*>(r0 + i), v1 = *reinterpret_cast<const float4_t*>(r0 + ST + i);
            const float4_t v2 = *reinterpret_cast<const float4_t*>(r0 + 2 * ST + i), v3 = *reinterpret_cast<const float4_t*>(r0 + 3 * ST + i);
            *reinterpret_cast<float4_t*>(dst + i) = (v0 + v1) + (v2 + v3);
        }
        if (threadIdx.x == 0) dst[S::ACC_COLS] = (r0[S::ACC_COLS] + r0[ST + S::ACC_COLS]) + (r0[2 * ST + S::ACC_COLS] + r0[3 * ST + S::ACC_COLS]);
    }
#ifdef MON_FUSED_TIMING
    tstamp(tc, 9);
    tcx.acc[14] = (float)(uint32_t)(wall_clock64() & 0xffffffull);
    if (lane == 0) for (int k = 0; k < 16; ++k) a.b.tdist[(blockIdx.x * S::WAVES + wave) * 16 + k] = tcx.acc[k];
#endif
}

// ------------------------------------------------------------------ host side
bool fused_supported(const NetDims& nd, uint32_t S, uint32_t R) {
    return S == 32 && R <= 16384u && nd.L >= 1 && nd.L <= kMaxLevels && (nd.Epad == 16 || nd.Epad == 32) && (((nd.W == 32 || nd.W == 64) && (nd.NH == 1 || nd.NH == 2)) || (nd.W == 128 && nd.NH == 1));
}

uint32_t fused_train_grid(const NetDims&, uint32_t R) {
    const uint32_t want = (R + 3) / 4;            // one ray per wavefront when it fits
    // (two workgroups per CU; the dW partial rows are allocated for kMaxFusedGrid of them.  One ray per wave -- 768 / 1024 workgroups -- measured slower,
    // HISTORY 7.6)
    return want < kMaxFusedGrid ? want : kMaxFusedGrid;
}

template <int EPAD, int W, int NH>
static void fused_train_t(hipStream_t s, const FusedArgs& a, uint32_t grid, int dump) {
    using S = FusedShape<EPAD, W, NH>;
    static std::atomic<uint64_t> attr_devices{ 0 }; static std::mutex attr_mu;
    once_per_device(attr_devices, attr_mu, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                S::SMEM_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_train<EPAD, W, NH, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                S::SMEM_BYTES);
    });
    const bool all_lds = a.lds_level_mask != 0u && (a.lds_level_mask == ((a.nd.L >= 32) ? 0xffffffffu : ((1u << a.nd.L) - 1u)));
    // the benched chain's kernel with the debug dump compiled in (mon_object_set_debug_dump(obj, 2)): E / h / O / dO / dh / dE of every sample
    if (a.e_soa && all_lds && dump) {
        hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, true, false, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
        return;
    }
    // features from k_encode_tiles (with the occupancy grid: a dead sample's features are there too, its alpha and gradient are zero all the same)
    if (a.e_soa && all_lds && !dump) {
        if (a.occ_bits) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false, true, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
        else hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
        return;
    }
    // (the debug dump evaluates every sample)
    if (dump) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, true, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
    else if (a.occ_bits) { if (all_lds) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
                           else hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, true, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a); }
    else if (all_lds) hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, false>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
    else hipLaunchKernelGGL((k_fused_train<EPAD, W, NH, false, true>), dim3(grid), dim3(256), S::SMEM_BYTES, s, a);
}
template <int EPAD, int W, int NH>
static void candidates_frags_t(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st, const uint16_t* params,
        const NetDims& nd, uint16_t* image) {
    using S = FusedShape<EPAD, W, NH>;
    const uint32_t cand_blocks = (oc.R + 255) / 256, frag_blocks = (S::N_FRAGS * 512 + 255) / 256;
    hipLaunchKernelGGL((k_candidates_and_frags<EPAD, W, NH>), dim3(cand_blocks + frag_blocks), dim3(256), 0, s, b, ds, oc, st, cand_blocks, params, nd.L,
            image);
}

void launch_fused_train(hipStream_t s, const LevelFast& lt, const NetDims& nd, const ParamPtrs& p, const BatchPtrs& b, const ObjectConst& oc, DevState* st,
        float* dw_partials, int debug_dump,
                        uint16_t* de_soa, float* x_soa, uint32_t lds_level_mask, uint16_t* frag_image, uint32_t big_switch, uint8_t* touched,
                                const uint32_t* occ_bits, uint32_t n_bins, const uint16_t* e_soa) {
    const uint32_t keep_zero = options().keep_zero_samples != 0 ? 1u : 0u;
    uint32_t stagger = kDefaultStagger;
    if (oc.R < 2u * 4u * fused_train_grid(nd, oc.R)) stagger = 0u;      // a wave with one ray has no second phase to interleave: the delay would only be lost
    FusedArgs a{ lt, nd, oc, b, p.half, p.ggrid, dw_partials, st, reinterpret_cast<half2_t*>(de_soa), x_soa, lds_level_mask, frag_image, keep_zero, touched
            ? touched + (nd.n_mlp >> 3) : nullptr, big_switch, n_bins, stagger, occ_bits, reinterpret_cast<const half2_t*>(e_soa) };
    if (e_soa) a.stagger = 0u;          // (the stagger interleaves gather phases with compute phases; a pre-encoded batch has no gather phase)
    const uint32_t grid = fused_train_grid(nd, oc.R);
    MON_FUSED_DISPATCH(fused_train_t, s, a, grid, debug_dump);
}
template <int EPAD, int W, int NH>
static void build_frag_image_t(hipStream_t s, const uint16_t* params, const NetDims& nd, uint16_t* image) {
    using S = FusedShape<EPAD, W, NH>;
    hipLaunchKernelGGL((k_build_frag_image<EPAD, W, NH>), dim3((S::N_FRAGS * 512 + 255) / 256), dim3(256), 0, s, params, nd.L, image, (const DevState*)nullptr);
}
void launch_build_frag_image(hipStream_t s, const uint16_t* params, const NetDims& nd, uint16_t* image) {
    MON_FUSED_DISPATCH(build_frag_image_t, s, params, nd, image); }

void launch_candidates_and_frags(hipStream_t s, const BatchPtrs& b, const DatasetPtrs& ds, const ObjectConst& oc, const DevState* st, const uint16_t* params,
        const NetDims& nd, uint16_t* frag_image) {
    MON_FUSED_DISPATCH(candidates_frags_t, s, b, ds, oc, st, params, nd, frag_image);
}

}  // namespace mon
