/*
 * mon_mesh_oracle.c -- CPU restatement of the reference's mesh extraction.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as mon_oracle.c; PARITY UNPINNED: the reference ships no mesh fixtures).
 * Restates CORE/src/marching_cubes.cu (CORE = /root/reference/dependencies/Multi-Object-NeRF/Core):
 *   gen_vertices :41-92      one vertex per lattice edge whose end points straddle `thresh` ("inside" = f > thresh),
 *                            position = (lattice + dt) * (max-min)/(res-1) + min,  dt = (thresh-f0)/(f1-f0)
 *   gen_faces :94-433        8-bit corner mask (corner order :390-400), triangle list from the 256-case table, edge -> vertex
 *                            through the per-axis vertex-index lattice (:406-421)
 *   accumulate_1ring :435-476  un-normalised area-weighted normal (pb-pa) x (pa-pc) added to the three corners
 *   MarchingCubes :478-509   vertex count rounded up to a multiple of 128, padding vertices are all-zero
 * and the consumers in CORE/src/nerf_model.cu: compute_mesh_vertex_colors :2050-2069 (WarpPoint :140-144, logistic rgb),
 * trans_mesh_data :341-360 (normalise, colour -> u8 by truncation of clamp(c*255, 0, 255)).
 *
 * The reference numbers vertices and faces with atomicAdd, i.e. in a run-dependent order, and sums normals with float
 * atomics.  This restatement (and the HIP path) fixes the order: vertices by (lattice index, axis x<y<z), faces by cell
 * index then table order, normal contributions in face order.  Vertex / triangle SETS are the reference's.
 *
 * The 256-case triangle table is Paul Bourke's public "Polygonising a scalar field" table (the one the reference cites
 * at marching_cubes.cu:96-97), packed 4 bits per edge id, 0xF = end, bits 60-63 = triangle count.
 */
#include <stdint.h>
#include <string.h>
#include <math.h>

static const uint64_t kTriTable[256] = {
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

uint64_t orc_mc_case(int mask) { return kTriTable[mask & 255]; }

static inline int inside(float f, float thresh) { return f > thresh; }

/* corner mask of cell (x,y,z), marching_cubes.cu:390-400 */
static int cell_mask(const float* d, uint32_t idx, uint32_t res1, uint32_t res2, float th) {
    int mask = 0;
    if (d[idx] > th) mask |= 1;
    if (d[idx + 1] > th) mask |= 2;
    if (d[idx + 1 + res1] > th) mask |= 4;
    if (d[idx + res1] > th) mask |= 8;
    if (d[idx + res2] > th) mask |= 16;
    if (d[idx + res2 + 1] > th) mask |= 32;
    if (d[idx + res2 + 1 + res1] > th) mask |= 64;
    if (d[idx + res2 + res1] > th) mask |= 128;
    return mask;
}

/* pass 1 (count only, :489-491): real vertex count and index count */
void orc_mc_count(const float* d, int rx, int ry, int rz, float th, uint32_t* n_verts, uint32_t* n_indices) {
    uint32_t nv = 0, ni = 0; const uint32_t res1 = (uint32_t)rx, res2 = (uint32_t)rx * ry;
    for (int z = 0; z < rz; ++z) for (int y = 0; y < ry; ++y) for (int x = 0; x < rx; ++x) {
        const uint32_t idx = (uint32_t)x + (uint32_t)y * res1 + (uint32_t)z * res2; const int in0 = inside(d[idx], th);
        if (x < rx - 1 && in0 != inside(d[idx + 1], th)) ++nv;
        if (y < ry - 1 && in0 != inside(d[idx + res1], th)) ++nv;
        if (z < rz - 1 && in0 != inside(d[idx + res2], th)) ++nv;
        if (x < rx - 1 && y < ry - 1 && z < rz - 1) ni += 3u * (uint32_t)(kTriTable[cell_mask(d, idx, res1, res2, th)] >> 60);
    }
    *n_verts = nv; *n_indices = ni;
}

/* pass 2: verts [3 * padded], vertidx [3 * res^3] (0 = none, else id+1), indices [n_indices], normals_raw [3 * padded] */
void orc_mc_extract(const float* d, int rx, int ry, int rz, float th, const float* amin, const float* amax,
                    float* verts, int32_t* vertidx, uint32_t* indices, float* normals_raw, uint32_t n_verts_padded) {
    const uint32_t res1 = (uint32_t)rx, res2 = (uint32_t)rx * ry, res3 = res2 * (uint32_t)rz;
    const float sc[3] = { (amax[0] - amin[0]) / (float)(rx - 1), (amax[1] - amin[1]) / (float)(ry - 1), (amax[2] - amin[2]) / (float)(rz - 1) };
    memset(verts, 0, sizeof(float) * 3 * n_verts_padded); memset(normals_raw, 0, sizeof(float) * 3 * n_verts_padded);
    memset(vertidx, 0, sizeof(int32_t) * 3 * (size_t)res3);
    uint32_t nv = 0;
    for (int z = 0; z < rz; ++z) for (int y = 0; y < ry; ++y) for (int x = 0; x < rx; ++x) {
        const uint32_t idx = (uint32_t)x + (uint32_t)y * res1 + (uint32_t)z * res2; const float f0 = d[idx]; const int in0 = inside(f0, th);
        const int lim[3] = { x < rx - 1, y < ry - 1, z < rz - 1 }; const uint32_t step[3] = { 1u, res1, res2 };
        for (int a = 0; a < 3; ++a) {
            if (!lim[a]) continue;
            const float f1 = d[idx + step[a]];
            if (in0 == inside(f1, th)) continue;
            const float dt = (th - f0) / (f1 - f0); float p[3] = { (float)x, (float)y, (float)z }; p[a] += dt;
            for (int c = 0; c < 3; ++c) verts[3 * nv + c] = fmaf(p[c], sc[c], amin[c]);
            vertidx[idx + res3 * (uint32_t)a] = (int32_t)(nv + 1); ++nv;
        }
    }
    uint32_t ni = 0;
    for (int z = 0; z < rz - 1; ++z) for (int y = 0; y < ry - 1; ++y) for (int x = 0; x < rx - 1; ++x) {
        const uint32_t idx = (uint32_t)x + (uint32_t)y * res1 + (uint32_t)z * res2;
        const int mask = cell_mask(d, idx, res1, res2, th);
        if (!mask || mask == 255) continue;
        const uint32_t ix = idx, iy = idx + res3, iz = idx + 2 * res3;
        const int32_t le[12] = { vertidx[ix], vertidx[iy + 1], vertidx[ix + res1], vertidx[iy],
                                 vertidx[ix + res2], vertidx[iy + 1 + res2], vertidx[ix + res1 + res2], vertidx[iy + res2],
                                 vertidx[iz], vertidx[iz + 1], vertidx[iz + 1 + res1], vertidx[iz + res1] };
        const uint64_t t = kTriTable[mask]; const int n = (int)(t >> 60) * 3;
        for (int i = 0; i < n; ++i) indices[ni + i] = (uint32_t)(le[(t >> (4 * i)) & 15] - 1);
        ni += (uint32_t)n;
    }
    for (uint32_t f = 0; f < ni; f += 3) {                     /* accumulate_1ring :435-476, normals only */
        const uint32_t ia = indices[f], ib = indices[f + 1], ic = indices[f + 2];
        const float* pa = verts + 3 * ia; const float* pb = verts + 3 * ib; const float* pc = verts + 3 * ic;
        const float u[3] = { pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2] }, v[3] = { pa[0] - pc[0], pa[1] - pc[1], pa[2] - pc[2] };
        const float n[3] = { u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0] };
        for (int c = 0; c < 3; ++c) { normals_raw[3 * ia + c] += n[c]; normals_raw[3 * ib + c] += n[c]; normals_raw[3 * ic + c] += n[c]; }
    }
}

/* trans_mesh_data nerf_model.cu:341-360: Eigen normalized() leaves a zero vector unchanged */
void orc_mesh_to_cpu(const float* normals_raw, const float* colors, uint32_t n, float* normals, uint8_t* colors8) {
    for (uint32_t i = 0; i < n; ++i) {
        const float* v = normals_raw + 3 * i; const float nn = fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0]));
        if (nn > 0.0f) { const float s = sqrtf(nn); for (int c = 0; c < 3; ++c) normals[3 * i + c] = v[c] / s; }
        else for (int c = 0; c < 3; ++c) normals[3 * i + c] = v[c];
        for (int c = 0; c < 3; ++c) { float q = colors[3 * i + c] * 255.0f; q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q); colors8[3 * i + c] = (uint8_t)q; }
    }
}
