"""Known-answer tests of the marching-cubes restatement (oracle/mon_mesh_oracle.c): intrinsic properties that pin the
256-case table and the edge numbering without the reference (closed 2-manifold, orientation, geometry of a sphere)."""
import numpy as np
import pytest


def _sphere(res, r=0.7, centre=(0.05, -0.03, 0.02)):
    ax = np.linspace(-1.0, 1.0, res, dtype=np.float64)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")                # x fastest
    return (r - np.sqrt((x - centre[0]) ** 2 + (y - centre[1]) ** 2 + (z - centre[2]) ** 2)).astype(np.float32).reshape(-1)


def _edge_use(idx):
    tri = idx.reshape(-1, 3).astype(np.int64)
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])
    return tri, e


def test_case_table_is_consistent(orc):
    """Every case lists only edges whose end points differ in the mask; complementary masks cut the same edge set."""
    ends = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    for mask in range(256):
        t = orc.lib().orc_mc_case(mask); n = t >> 60
        used = {(t >> (4 * i)) & 15 for i in range(3 * n)}
        cut = {k for k, (a, b) in enumerate(ends) if ((mask >> a) & 1) != ((mask >> b) & 1)}
        assert used == cut or (n == 0 and not cut), mask
        assert all(((t >> (4 * i)) & 15) == 15 for i in range(3 * n, 15))
        t2 = orc.lib().orc_mc_case(255 - mask)
        assert {(t2 >> (4 * i)) & 15 for i in range(3 * (t2 >> 60))} == used


@pytest.mark.parametrize("res", [16, 33])
def test_sphere_is_closed_oriented_manifold(orc, res):
    d = _sphere(res)
    m = orc.marching_cubes(d, (res, res, res), 0.0, [-1, -1, -1], [1, 1, 1])
    nv = m["n_verts_real"]; v = m["verts"]; tri, e = _edge_use(m["indices"])
    assert v.shape[0] % 128 == 0 and v.shape[0] - nv < 128 and np.all(v[nv:] == 0)          # MarchingCubes :496
    assert tri.max() == nv - 1 and np.unique(tri).size == nv
    # closed + consistently oriented: every directed edge appears once and its reverse once
    key = e[:, 0] * (nv + 1) + e[:, 1]; rkey = e[:, 1] * (nv + 1) + e[:, 0]
    assert np.unique(key).size == key.size and np.array_equal(np.sort(key), np.sort(rkey))
    assert nv - e.shape[0] // 2 + tri.shape[0] == 2                                         # Euler characteristic of a sphere
    # geometry: vertices lie on the sphere up to the linear-interpolation error, area ~ 4 pi r^2
    c = np.array([0.05, -0.03, 0.02]); h = 2.0 / (res - 1)
    assert np.abs(np.linalg.norm(v[:nv] - c, axis=1) - 0.7).max() < 0.25 * h * h / 0.7 + 1e-5
    pa, pb, pc = v[tri[:, 0]].astype(np.float64), v[tri[:, 1]].astype(np.float64), v[tri[:, 2]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(pb - pa, pc - pa), axis=1).sum()
    assert abs(area / (4 * np.pi * 0.49) - 1.0) < 0.02
    # accumulate_1ring's (pb-pa)x(pa-pc) with the table's winding points OUT of the "inside" (f > thresh) region, i.e. away
    # from the object; the ply writer reverses the index order (marching_cubes.cu:609) so its faces wind the same way
    nrm, _ = orc.mesh_to_cpu(m["normals_raw"], np.zeros_like(v))
    radial = (v[:nv] - c) / np.linalg.norm(v[:nv] - c, axis=1, keepdims=True)
    assert ((nrm[:nv] * radial).sum(1) > 0.9).all()
    assert np.allclose(np.linalg.norm(nrm[:nv], axis=1), 1.0, atol=1e-5) and np.all(nrm[nv:] == 0)


def test_vertex_positions_and_ragged_grid(orc):
    """Non-cubic lattice, anisotropic box: interpolated position formula (gen_vertices :58-63)."""
    rx, ry, rz = 5, 4, 3
    d = np.zeros((rz, ry, rx), np.float32); d[1, 2, 3] = 3.0                      # one lattice point inside -> octahedron
    m = orc.marching_cubes(d.reshape(-1), (rx, ry, rz), 2.0, [-1, 0, 2], [1, 3, 4])
    assert m["n_verts_real"] == 6 and m["indices"].size == 8 * 3
    sc = np.array([2 / 4, 3 / 3, 2 / 2]); off = np.array([-1, 0, 2.0]); base = np.array([3, 2, 1.0])
    want = []
    for a in range(3):
        for s, dt in ((-1, 2.0 / 3.0), (0, 1.0 / 3.0)):                            # edge from the lower neighbour / to the upper neighbour
            p = base.copy(); p[a] += s + dt; want.append(p * sc + off)
    got = m["verts"][:6]
    assert np.allclose(np.sort(got.view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=["x", "y", "z"]).view("f4").reshape(-1, 3),
                       np.array(sorted(map(tuple, want)), np.float32), atol=1e-6)
    # u8 colour conversion truncates after the clamp (trans_mesh_data :355-357)
    _, c8 = orc.mesh_to_cpu(np.zeros((2, 3), np.float32), np.array([[0.999, 0.5, -0.2], [1.2, 0.00391, 0.00393]], np.float32))
    assert c8.tolist() == [[254, 127, 0], [255, 0, 1]]


def test_empty_and_full_fields(orc):
    for val in (0.0, 5.0):
        m = orc.marching_cubes(np.full(8 * 8 * 8, val, np.float32), (8, 8, 8), 2.0, [-1, -1, -1], [1, 1, 1])
        assert m["n_verts_real"] == 0 and m["verts"].shape[0] == 0 and m["indices"].size == 0
