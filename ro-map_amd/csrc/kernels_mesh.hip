// kernels_mesh.hip -- density lattice -> triangle mesh on gfx950.
// Reference behaviour: CORE/src/marching_cubes.cu gen_vertices :41-92, gen_faces :94-433, accumulate_1ring :435-476,
// MarchingCubes :478-509; CORE/src/nerf_model.cu compute_mesh_vertex_colors :2050-2069, trans_mesh_data :341-360.
//
// The reference numbers vertices and faces with global atomicAdd (run-dependent order) in two count+emit passes and sums
// normals with float atomics.  Here the numbering is a prefix sum over the lattice (vertices by (lattice index, axis),
// faces by cell index, table order inside a cell) and normals are GATHERED per vertex from the <= 4 cells around its edge
// in ascending face order: no atomics anywhere, output is reproducible bit for bit, vertex / triangle sets are the reference's.
// A 64^3 lattice is 1 MB of density: every pass is a single coalesced sweep, bound by launch latency (~5 us each).
#include "model.h"

namespace mon {

// Paul Bourke's public 256-case triangle table ("Polygonising a scalar field"; the table the reference cites at
// marching_cubes.cu:96-97), packed: 4 bits per edge id, 0xF = end, bits 60-63 = triangle count.
__device__ const uint64_t kTriTable[256] = {
    0x0fffffffffffffffull, 0x1ffffffffffff380ull, 0x1ffffffffffff910ull, 0x2fffffffff189381ull,
    0x1ffffffffffffa21ull, 0x2fffffffffa21380ull, 0x2fffffffff920a29ull, 0x3ffffff89a8a2382ull,
    0x1ffffffffffff2b3ull, 0x2fffffffff0b82b0ull, 0x2fffffffffb32091ull, 0x3ffffffb89b912b1ull,
    0x2fffffffff3ab1a3ull, 0x3ffffffab8a801a0ull, 0x3ffffff9ab9b3093ull, 0x2fffffffffb8aa89ull,
    0x1ffffffffffff874ull, 0x2fffffffff437034ull, 0x2fffffffff748910ull, 0x3ffffff137174914ull,
    0x2fffffffff748a21ull, 0x3ffffffa21403743ull, 0x3ffffff748209a29ull, 0x4fff4973727929a2ull,
    0x2fffffffff2b3748ull, 0x3ffffff40242b74bull, 0x3ffffffb32748109ull, 0x4fff1292b9b49b74ull,
    0x3ffffff487ab31a3ull, 0x4fff4b7401b41ab1ull, 0x4fff30bab9b09874ull, 0x3ffffffab99b4b74ull,
    0x1ffffffffffff459ull, 0x2fffffffff380459ull, 0x2fffffffff051450ull, 0x3ffffff513538458ull,
    0x2fffffffff459a21ull, 0x3ffffff594a21803ull, 0x3ffffff204245a25ull, 0x4fff8434535235a2ull,
    0x2fffffffffb32459ull, 0x3ffffff594b802b0ull, 0x3ffffffb32510450ull, 0x4fff584b82852512ull,
    0x3ffffff45931ab3aull, 0x4fffab81a8180594ull, 0x4fff30bab5b05045ull, 0x3ffffffb8aa85845ull,
    0x2fffffffff975879ull, 0x3ffffff375359039ull, 0x3ffffff751710870ull, 0x2fffffffff753351ull,
    0x3ffffff21a759879ull, 0x4fff37503505921aull, 0x4fff25a758528208ull, 0x3ffffff7533525a2ull,
    0x3ffffff2b3987597ull, 0x4fffb72029279759ull, 0x4fff751871810b32ull, 0x3ffffff51771b12bull,
    0x4fffb3a31a758859ull, 0x50aba010b7905075ull, 0x507570805a30b0abull, 0x2fffffffff5b75abull,
    0x1ffffffffffff56aull, 0x2fffffffff6a5380ull, 0x2fffffffff6a5109ull, 0x3ffffff6a5891381ull,
    0x2fffffffff162561ull, 0x3ffffff803621561ull, 0x3ffffff620609569ull, 0x4fff823625285895ull,
    0x2fffffffff56ab32ull, 0x3ffffff56a02b80bull, 0x3ffffff6a5b32910ull, 0x4fffb892b92916a5ull,
    0x3ffffff315356b36ull, 0x4fff6b51505b0b80ull, 0x4fff9505606306b3ull, 0x3ffffff89bb96956ull,
    0x2fffffffff8746a5ull, 0x3ffffffa56374034ull, 0x3ffffff7486a5091ull, 0x4fff49737179156aull,
    0x3ffffff874156216ull, 0x4fff743403625521ull, 0x4fff620560509748ull, 0x5962695923497937ull,
    0x3ffffff56a4872b3ull, 0x4fffb720242746a5ull, 0x4fff6a5b32874910ull, 0x56a54b7b492b9129ull,
    0x4fff6b51535b3748ull, 0x5b404b7b016b5b15ull, 0x574836b630560950ull, 0x4fff9b7974b96956ull,
    0x2fffffffffa4694aull, 0x3ffffff380a946a4ull, 0x3ffffff04606a10aull, 0x4fffa16468618138ull,
    0x3ffffff462421941ull, 0x4fff462942921803ull, 0x2fffffffff624420ull, 0x3ffffff624428238ull,
    0x3ffffff32b46a94aull, 0x4fff6a4a94b82280ull, 0x4fffa164606102b3ull, 0x51b8b12184a16146ull,
    0x4fff36b319639469ull, 0x514641916b0181b8ull, 0x3ffffff4600636b3ull, 0x2fffffffff86b846ull,
    0x3ffffffa98a876a7ull, 0x4fffa76a907a0370ull, 0x4fff0818717a176aull, 0x3ffffff37117a76aull,
    0x4fff768981861621ull, 0x5937390976192962ull, 0x3ffffff206607087ull, 0x2fffffffff276237ull,
    0x4fff76898a86ab32ull, 0x57a9a76790b72702ull, 0x5b32a767a1871081ull, 0x4fff17616a71b12bull,
    0x563136b619768698ull, 0x2fffffffff76b190ull, 0x4fff06b0b3607087ull, 0x1ffffffffffff6b7ull,
    0x1ffffffffffffb67ull, 0x2fffffffff67b803ull, 0x2fffffffff67b910ull, 0x3ffffff67b138918ull,
    0x2fffffffff7b621aull, 0x3ffffff7b6803a21ull, 0x3ffffff7b69a2092ull, 0x4fff89a38a3a27b6ull,
    0x2fffffffff726327ull, 0x3ffffff026067807ull, 0x3ffffff910732672ull, 0x4fff678891681261ull,
    0x3ffffff73171a67aull, 0x4fff801781a7167aull, 0x4fff7a69a0a70730ull, 0x3ffffff9a88a7a67ull,
    0x2fffffffff68b486ull, 0x3ffffff640603b63ull, 0x3ffffff109648b68ull, 0x4fff63b139369649ull,
    0x3ffffff1a28b6486ull, 0x4fff640b60b03a21ull, 0x4fff9a2920b648b4ull, 0x536463b34923a39aull,
    0x3ffffff264248328ull, 0x2fffffffff264240ull, 0x4fff834642432091ull, 0x3ffffff642241491ull,
    0x4fff1a6648168318ull, 0x3ffffff40660a01aull, 0x539a9303a6834364ull, 0x2fffffffff4a649aull,
    0x2fffffffffb67594ull, 0x3ffffff67b594380ull, 0x3ffffffb67045105ull, 0x4fff51345343867bull,
    0x3ffffffb6721a459ull, 0x4fff594380a217b6ull, 0x4fff204a24a45b67ull, 0x567b25a523453843ull,
    0x3ffffff945267327ull, 0x4fff786260680459ull, 0x4fff045051673263ull, 0x5851584812786826ull,
    0x4fff73167161a459ull, 0x5459078701671a61ull, 0x5a737a6a305a4a04ull, 0x4fffa84a458a7a67ull,
    0x3ffffff98b9b6596ull, 0x4fff590650360b63ull, 0x4fffb65510b508b0ull, 0x3ffffff1355363b6ull,
    0x4fff65b8b9b59a21ull, 0x5a21965690b603b0ull, 0x552025a50865b58bull, 0x4fff35a3a25363b6ull,
    0x4fff283265825985ull, 0x3ffffff260069659ull, 0x5826283865081851ull, 0x2fffffffff612651ull,
    0x5698965683a61631ull, 0x4fff06505960a01aull, 0x2fffffffffa65830ull, 0x1ffffffffffff65aull,
    0x2fffffffffb57a5bull, 0x3ffffff03857ba5bull, 0x3ffffff091ba57b5ull, 0x4fff1381897ba57aull,
    0x3ffffff15717b21bull, 0x4fffb27571721380ull, 0x4fff7b2209729579ull, 0x5289823295b27257ull,
    0x3ffffff573532a52ull, 0x4fff52a578258028ull, 0x4fff2a37353a5109ull, 0x525752a278129289ull,
    0x2fffffffff573531ull, 0x3ffffff571170780ull, 0x3ffffff735539309ull, 0x2fffffffff795789ull,
    0x3ffffff8ba8a5485ull, 0x4fff03bba50b5405ull, 0x4fff54aba8a48910ull, 0x541314943b54a4baull,
    0x4fff8548b2582152ull, 0x5b151b2b543b0b40ull, 0x558b8545b2950520ull, 0x2fffffffff3b2549ull,
    0x4fff483543253a52ull, 0x3ffffff0244252a5ull, 0x5910854583a532a3ull, 0x4fff2492914252a5ull,
    0x3ffffff153358548ull, 0x2fffffffff501540ull, 0x4fff530509358548ull, 0x1ffffffffffff549ull,
    0x3ffffffba9b947b4ull, 0x4fffba97b9794380ull, 0x4fffb470414b1ba1ull, 0x54bab474a1843413ull,
    0x4fff219b294b97b4ull, 0x53801b2b197b9479ull, 0x3ffffff04224b47bull, 0x4fff42343824b47bull,
    0x4fff947732972a92ull, 0x570207872a4797a9ull, 0x5a040a1a472a3a73ull, 0x2fffffffff4782a1ull,
    0x3ffffff317714194ull, 0x4fff178180714194ull, 0x2fffffffff347304ull, 0x1ffffffffffff784ull,
    0x2fffffffff8ba8a9ull, 0x3ffffffa9bb93903ull, 0x3ffffffba88a0a10ull, 0x2fffffffffa3ba13ull,
    0x3ffffff8b99b1b21ull, 0x4fff9b2921b93903ull, 0x2fffffffffb08b20ull, 0x1ffffffffffffb23ull,
    0x3ffffff98aa82832ull, 0x2fffffffff2902a9ull, 0x4fff8a1810a82832ull, 0x1ffffffffffff2a1ull,
    0x2fffffffff819831ull, 0x1ffffffffffff190ull, 0x1ffffffffffff830ull, 0x0fffffffffffffffull,
};

struct McGrid { int rx, ry, rz; uint32_t res1, res2, res3; float thresh; float sc[3], off[3]; };

__device__ inline int mc_cell_mask(const float* __restrict__ d, uint32_t idx, const McGrid& g) {   // corner order :390-400
    const float th = g.thresh; int mask = 0;
    if (d[idx] > th) mask |= 1;
    if (d[idx + 1] > th) mask |= 2;
    if (d[idx + 1 + g.res1] > th) mask |= 4;
    if (d[idx + g.res1] > th) mask |= 8;
    if (d[idx + g.res2] > th) mask |= 16;
    if (d[idx + g.res2 + 1] > th) mask |= 32;
    if (d[idx + g.res2 + 1 + g.res1] > th) mask |= 64;
    if (d[idx + g.res2 + g.res1] > th) mask |= 128;
    return mask;
}

// per lattice point: bit a of `cross` = the +a edge is cut; returns index count of the cell rooted here
__device__ inline uint32_t mc_point(const float* __restrict__ d, uint32_t idx, const McGrid& g, uint32_t& cross, int& mask) {
    const uint32_t x = idx % g.res1, y = (idx / g.res1) % (uint32_t)g.ry, z = idx / g.res2;
    const bool in0 = d[idx] > g.thresh; cross = 0; mask = 0;
    const bool lx = x + 1 < (uint32_t)g.rx, ly = y + 1 < (uint32_t)g.ry, lz = z + 1 < (uint32_t)g.rz;
    if (lx && in0 != (d[idx + 1] > g.thresh)) cross |= 1;
    if (ly && in0 != (d[idx + g.res1] > g.thresh)) cross |= 2;
    if (lz && in0 != (d[idx + g.res2] > g.thresh)) cross |= 4;
    if (!(lx && ly && lz)) return 0;
    mask = mc_cell_mask(d, idx, g);
    return 3u * (uint32_t)(kTriTable[mask] >> 60);
}

// block-wide exclusive scan of two counters packed in one u32 (verts in the low 10 bits is not enough: use two scans in u64)
__device__ inline uint64_t block_excl_scan(uint64_t v, uint64_t* lds, uint64_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint64_t inc = v;
    #pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const uint64_t s = lds[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// counters are packed as (indices << 32) | verts
__global__ void __launch_bounds__(256) k_mc_count(const float* __restrict__ d, McGrid g, uint64_t* __restrict__ block_sums) {
    __shared__ uint64_t lds[4];
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x; uint64_t v = 0;
    if (idx < g.res3) { uint32_t cross; int mask; const uint32_t ni = mc_point(d, idx, g, cross, mask); v = ((uint64_t)ni << 32) | (uint64_t)__popc(cross); }
    uint64_t total; block_excl_scan(v, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the block sums in place; totals -> block_sums[nb]
__global__ void __launch_bounds__(1024) k_mc_scan(uint64_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint64_t lds[16];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
        const uint32_t b = b0 + threadIdx.x; const uint64_t v = b < nb ? block_sums[b] : 0; uint64_t total;
        const uint64_t ex = block_excl_scan(v, lds, total);
        if (b < nb) block_sums[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

// gen_vertices :41-92 with prefix-sum numbering; vertidx [3][res3]: 0 = no vertex, else id + 1
__global__ void __launch_bounds__(256) k_mc_vertices(const float* __restrict__ d, McGrid g, const uint64_t* __restrict__ block_offs,
                                                     int32_t* __restrict__ vertidx, float* __restrict__ verts) {
    __shared__ uint64_t lds[4];
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x; uint32_t cross = 0; int mask;
    if (idx < g.res3) mc_point(d, idx, g, cross, mask);
    uint64_t total; uint32_t id = (uint32_t)block_offs[blockIdx.x] + (uint32_t)block_excl_scan((uint64_t)__popc(cross), lds, total);
    if (idx >= g.res3) return;
    const uint32_t x = idx % g.res1, y = (idx / g.res1) % (uint32_t)g.ry, z = idx / g.res2;
    const float f0 = d[idx]; const uint32_t step[3] = { 1u, g.res1, g.res2 };
    #pragma unroll
    for (int a = 0; a < 3; ++a) {
        int32_t vi = 0;
        if (cross & (1u << a)) {
            const float f1 = d[idx + step[a]]; const float dt = (g.thresh - f0) / (f1 - f0);
            float p[3] = { (float)x, (float)y, (float)z }; p[a] += dt;
            verts[3 * id + 0] = fmaf(p[0], g.sc[0], g.off[0]); verts[3 * id + 1] = fmaf(p[1], g.sc[1], g.off[1]);
            verts[3 * id + 2] = fmaf(p[2], g.sc[2], g.off[2]);
            vi = (int32_t)(++id);
        }
        vertidx[idx + g.res3 * (uint32_t)a] = vi;
    }
}

__device__ inline void mc_local_edges(const int32_t* __restrict__ vertidx, uint32_t idx, const McGrid& g, int32_t* le) {   // :406-421
    const uint32_t ix = idx, iy = idx + g.res3, iz = idx + 2 * g.res3;
    le[0] = vertidx[ix]; le[1] = vertidx[iy + 1]; le[2] = vertidx[ix + g.res1]; le[3] = vertidx[iy];
    le[4] = vertidx[ix + g.res2]; le[5] = vertidx[iy + 1 + g.res2]; le[6] = vertidx[ix + g.res1 + g.res2]; le[7] = vertidx[iy + g.res2];
    le[8] = vertidx[iz]; le[9] = vertidx[iz + 1]; le[10] = vertidx[iz + 1 + g.res1]; le[11] = vertidx[iz + g.res1];
}

// gen_faces :94-433 with prefix-sum numbering
__global__ void __launch_bounds__(256) k_mc_faces(const float* __restrict__ d, McGrid g, const uint64_t* __restrict__ block_offs,
                                                  const int32_t* __restrict__ vertidx, uint32_t* __restrict__ indices) {
    __shared__ uint64_t lds[4];
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x; uint32_t cross, ni = 0; int mask = 0;
    if (idx < g.res3) ni = mc_point(d, idx, g, cross, mask);
    uint64_t total; const uint32_t base = (uint32_t)(block_offs[blockIdx.x] >> 32) + (uint32_t)block_excl_scan((uint64_t)ni, lds, total);
    if (!ni) return;
    int32_t le[12]; mc_local_edges(vertidx, idx, g, le);
    const uint64_t t = kTriTable[mask];
    for (uint32_t i = 0; i < ni; ++i) {
        const int e = (int)((t >> (4 * i)) & 15); int32_t v = 0;
        #pragma unroll
        for (int k = 0; k < 12; ++k) v = (e == k) ? le[k] : v;          // keeps le[] in registers
        indices[base + i] = (uint32_t)(v - 1);
    }
}

// accumulate_1ring :435-476 (normals only) as a gather: the vertex on the +a edge of lattice point `idx` is shared by the
// <= 4 cells around that edge; visit them in ascending cell index and their triangles in table order (= face order).
__global__ void __launch_bounds__(256) k_mc_normals(const float* __restrict__ d, McGrid g, const int32_t* __restrict__ vertidx,
                                                    const float* __restrict__ verts, float* __restrict__ normals_raw) {
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.res3) return;
    const int x = (int)(idx % g.res1), y = (int)((idx / g.res1) % (uint32_t)g.ry), z = (int)(idx / g.res2);
    for (int a = 0; a < 3; ++a) {
        const int32_t vi = vertidx[idx + g.res3 * (uint32_t)a];
        if (!vi) continue;
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;
        for (int q = 0; q < 4; ++q) {                                   // (hi, lo) offsets, high axis first => ascending cell index
            const int ohi = 1 - (q >> 1), olo = 1 - (q & 1); int cx = x, cy = y, cz = z, e;
            if (a == 0) { cy -= olo; cz -= ohi; e = olo ? (ohi ? 6 : 2) : (ohi ? 4 : 0); }
            else if (a == 1) { cx -= olo; cz -= ohi; e = olo ? (ohi ? 5 : 1) : (ohi ? 7 : 3); }
            else { cx -= olo; cy -= ohi; e = olo ? (ohi ? 10 : 9) : (ohi ? 11 : 8); }
            if (cx < 0 || cy < 0 || cz < 0 || cx >= g.rx - 1 || cy >= g.ry - 1 || cz >= g.rz - 1) continue;
            const uint32_t cidx = (uint32_t)cx + (uint32_t)cy * g.res1 + (uint32_t)cz * g.res2;
            const int mask = mc_cell_mask(d, cidx, g);
            const uint64_t t = kTriTable[mask]; const int nt = (int)(t >> 60);
            for (int k = 0; k < nt; ++k) {
                const int e0 = (int)((t >> (12 * k)) & 15), e1 = (int)((t >> (12 * k + 4)) & 15), e2 = (int)((t >> (12 * k + 8)) & 15);
                if (e0 != e && e1 != e && e2 != e) continue;
                int32_t le[12]; mc_local_edges(vertidx, cidx, g, le);
                int32_t ia = 0, ib = 0, ic = 0;
                #pragma unroll
                for (int j = 0; j < 12; ++j) { ia = (e0 == j) ? le[j] : ia; ib = (e1 == j) ? le[j] : ib; ic = (e2 == j) ? le[j] : ic; }
                const float* pa = verts + 3 * (ia - 1); const float* pb = verts + 3 * (ib - 1); const float* pc = verts + 3 * (ic - 1);
                const float u0 = pb[0] - pa[0], u1 = pb[1] - pa[1], u2 = pb[2] - pa[2], v0 = pa[0] - pc[0], v1 = pa[1] - pc[1], v2 = pa[2] - pc[2];
                n0 += u1 * v2 - u2 * v1; n1 += u2 * v0 - u0 * v2; n2 += u0 * v1 - u1 * v0;
            }
        }
        normals_raw[3 * (vi - 1) + 0] = n0; normals_raw[3 * (vi - 1) + 1] = n1; normals_raw[3 * (vi - 1) + 2] = n2;
    }
}

// generate_nerf_network_inputs_from_positions nerf_model.cu:319-326 (WarpPoint :140-144)
__global__ void __launch_bounds__(256) k_mesh_warp(const float* __restrict__ verts, float* __restrict__ pts, uint32_t v0, uint32_t n, Aabb box) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    #pragma unroll
    for (int a = 0; a < 3; ++a) pts[3 * i + a] = (verts[3 * (v0 + i) + a] - box.mn[a]) / (box.mx[a] - box.mn[a]);
}

// extract_rgb_with_activation :328-339 + the colour half of trans_mesh_data :355-357
__global__ void __launch_bounds__(256) k_mesh_colors(const uint16_t* __restrict__ O, float* __restrict__ colf, uint8_t* __restrict__ col8, uint32_t v0,
        uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const half_t* o = reinterpret_cast<const half_t*>(O) + (size_t)i * kOut;
    #pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = logistic_f((float)o[c]); colf[3 * (v0 + i) + c] = v;
        col8[3 * (v0 + i) + c] = (uint8_t)clamp_f(v * 255.0f, 0.0f, 255.0f);
    }
}

// the normal half of trans_mesh_data :349-353 (Eigen normalized(): a zero vector stays zero)
__global__ void __launch_bounds__(256) k_mesh_normalize(const float* __restrict__ raw, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = raw[3 * i], b = raw[3 * i + 1], c = raw[3 * i + 2]; const float nn = fmaf(c, c, fmaf(b, b, a * a));
    if (nn > 0.0f) { const float s = sqrtf(nn); out[3 * i] = a / s; out[3 * i + 1] = b / s; out[3 * i + 2] = c / s; }
    else { out[3 * i] = a; out[3 * i + 1] = b; out[3 * i + 2] = c; }
}

static McGrid make_grid(int rx, int ry, int rz, float thresh, const float* amin, const float* amax) {
    McGrid g; g.rx = rx; g.ry = ry; g.rz = rz; g.res1 = (uint32_t)rx; g.res2 = (uint32_t)rx * ry; g.res3 = g.res2 * (uint32_t)rz; g.thresh = thresh;
    const int r[3] = { rx, ry, rz };
    for (int a = 0; a < 3; ++a) { g.sc[a] = (amax[a] - amin[a]) / (float)(r[a] - 1); g.off[a] = amin[a]; }
    return g;
}

void launch_mc_count(hipStream_t s, const float* density, int rx, int ry, int rz, float thresh, uint64_t* block_sums) {
    const float z3[3] = { 0, 0, 0 }, o3[3] = { 1, 1, 1 }; const McGrid g = make_grid(rx, ry, rz, thresh, z3, o3); const uint32_t nb = (g.res3 + 255) / 256;
    hipLaunchKernelGGL(k_mc_count, dim3(nb), dim3(256), 0, s, density, g, block_sums);
    hipLaunchKernelGGL(k_mc_scan, dim3(1), dim3(1024), 0, s, block_sums, nb);
}
void launch_mc_emit(hipStream_t s, const float* density, int rx, int ry, int rz, float thresh, const float* amin, const float* amax, const uint64_t* block_offs,
                    int32_t* vertidx, float* verts, uint32_t* indices, float* normals_raw, float* normals, uint32_t n_verts_real, uint32_t n_indices) {
    const McGrid g = make_grid(rx, ry, rz, thresh, amin, amax); const uint32_t nb = (g.res3 + 255) / 256;
    hipLaunchKernelGGL(k_mc_vertices, dim3(nb), dim3(256), 0, s, density, g, block_offs, vertidx, verts);
    if (n_indices) hipLaunchKernelGGL(k_mc_faces, dim3(nb), dim3(256), 0, s, density, g, block_offs, vertidx, indices);
    if (n_verts_real) {
        hipLaunchKernelGGL(k_mc_normals, dim3(nb), dim3(256), 0, s, density, g, vertidx, verts, normals_raw);
        hipLaunchKernelGGL(k_mesh_normalize, dim3((n_verts_real + 255) / 256), dim3(256), 0, s, normals_raw, normals, n_verts_real);
    }
}
void launch_mesh_warp(hipStream_t s, const float* verts, float* pts, uint32_t v0, uint32_t n, const Aabb& box) {
    hipLaunchKernelGGL(k_mesh_warp, dim3((n + 255) / 256), dim3(256), 0, s, verts, pts, v0, n, box);
}
void launch_mesh_colors(hipStream_t s, const uint16_t* O, float* colf, uint8_t* col8, uint32_t v0, uint32_t n) {
    hipLaunchKernelGGL(k_mesh_colors, dim3((n + 255) / 256), dim3(256), 0, s, O, colf, col8, v0, n);
}

}  // namespace mon
