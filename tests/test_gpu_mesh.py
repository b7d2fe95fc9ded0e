"""GPU parity tests of mesh extraction: the HIP marching cubes + normals + vertex colours, called through
the C ABI, against oracle/mon_mesh_oracle.c -- bit-exact vertices, indices and normals on the same lattice."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import C1, ROOT

pytestmark = pytest.mark.gpu


def _sphere(res3, r=0.7):
    rx, ry, rz = res3
    z, y, x = np.meshgrid(np.linspace(-1, 1, rz), np.linspace(-1, 1, ry), np.linspace(-1, 1, rx), indexing="ij")
    return (r - np.sqrt((x - 0.05) ** 2 + (y + 0.03) ** 2 + (z - 0.02) ** 2) + 0.05 * np.sin(7 * x) * np.cos(5 * y)).astype(np.float32).reshape(-1)


def _same_mesh(got, want):
    assert got["n_verts_real"] == want["n_verts_real"] and got["verts"].shape == want["verts"].shape
    assert np.array_equal(got["indices"], want["indices"])                                    # index work: bit-exact
    assert np.array_equal(got["verts"].view(np.uint32), want["verts"].view(np.uint32))        # same fmaf chain: bit-exact
    assert np.array_equal(got["normals_raw"].view(np.uint32), want["normals_raw"].view(np.uint32))


@pytest.mark.parametrize("res3", [(16, 16, 16), (64, 64, 64), (37, 21, 50), (2, 2, 2), (130, 3, 5)])
def test_marching_cubes_matches_oracle_on_analytic_fields(pkg, orc, res3):
    assert pkg.device_count() >= 1
    d = _sphere(res3)
    got = pkg.marching_cubes(d, res3, 0.0, [-1.0, -0.5, 0.25], [1.0, 0.75, 2.0])
    want = orc.marching_cubes(d, res3, 0.0, [-1.0, -0.5, 0.25], [1.0, 0.75, 2.0])
    _same_mesh(got, want)
    # random field: every one of the 256 cases and many cells sharing vertices
    rs = np.random.RandomState(3); d = rs.uniform(-1, 1, res3[0] * res3[1] * res3[2]).astype(np.float32)
    _same_mesh(pkg.marching_cubes(d, res3, 0.1, [-1, -1, -1], [1, 1, 1]), orc.marching_cubes(d, res3, 0.1, [-1, -1, -1], [1, 1, 1]))


def test_marching_cubes_edge_cases(pkg):
    for val in (0.0, 5.0):                                                      # nothing / everything inside: empty mesh
        m = pkg.marching_cubes(np.full(512, val, np.float32), (8, 8, 8), 2.0, [-1, -1, -1], [1, 1, 1])
        assert m["verts"].shape[0] == 0 and m["indices"].size == 0 and m["n_verts_real"] == 0
    with pytest.raises(pkg.MonError):
        pkg.marching_cubes(np.zeros(4, np.float32), (1, 2, 2), 0.0, [-1, -1, -1], [1, 1, 1])


def test_object_mesh_matches_oracle(pkg, orc, ss, tmp_path):
    """GenerateMesh on a trained object: density lattice, marching cubes, 1-ring normals, vertex colours, CPUMeshData, ply."""
    sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)
    ds, obj = ge.make_problem(pkg, sc, C1); ref = ge.make_oracle(orc, sc, C1)
    assert obj.train(300) < 0.05
    with pytest.raises(pkg.MonError):
        obj.get_mesh()                                                          # nothing generated yet
    ref.set_params(obj.get_params(0)); ema = obj.get_params(2)
    nv, ni = obj.generate_mesh(32, 2.0)
    got = obj.get_mesh(raw=True)
    assert got["verts"].shape[0] == nv and nv % 128 == 0 and got["indices"].size == ni and ni > 300
    dens = obj.density_grid(32, 32, 32)
    # (1) marching cubes / normals are bit-exact given the device's own density lattice
    want = orc.marching_cubes(dens, (32, 32, 32), 2.0, ref._amin, ref._amax)
    _same_mesh(got, want)
    nrm, _ = orc.mesh_to_cpu(want["normals_raw"], np.zeros_like(want["verts"]))
    assert np.array_equal(got["normals"].view(np.uint32), nrm.view(np.uint32))
    # (2) the lattice itself against the restatement run on the same inference (EMA) weights: fp16 outputs, same fmaf chains
    ref.set_ema(ema)
    rd = ref.density_grid(32, 32, 32, use_ema=True)
    assert (dens == rd).mean() > 0.999 and np.abs(dens - rd).max() < 0.05
    # (3) colours: logistic of the fp16 network output at the warped vertices
    col = ref.mesh_colors(got["verts"])
    assert np.abs(got["colors_f32"] - col).max() < 2e-3
    assert np.abs(got["colors"].astype(int) - np.clip(col * 255.0, 0, 255).astype(np.uint8).astype(int)).max() <= 1
    # (4) geometry sanity: the surface is the ellipsoid the scene was rendered from
    ob = sc.objects[0]; v = got["verts"][:got["n_verts_real"]]
    rad = np.linalg.norm(v / ob["radii"], axis=1)
    assert 0.8 < np.median(rad) < 1.2, np.median(rad)
    # (5) ply writer: header counts, vertex lines, reversed winding
    path = str(tmp_path / "obj.ply"); obj.save_mesh(path)
    lines = open(path).read().split("\n")
    hdr_end = lines.index("end_header")
    assert lines[0] == "ply" and "element vertex %d" % nv in lines and "element face %d" % (ni // 3) in lines
    first = lines[hdr_end + 1].split(); assert len(first) == 9
    assert np.allclose([float(q) for q in first[:3]], got["verts"][0], atol=1e-5) and [int(q) for q in first[6:]] == got["colors"][0].tolist()
    f0 = [int(q) for q in lines[hdr_end + 1 + nv].split()]
    assert f0 == [3, int(got["indices"][2]), int(got["indices"][1]), int(got["indices"][0])]
    obj.save_mesh(str(tmp_path / "obj.obj")); assert open(str(tmp_path / "obj.obj")).readline().startswith("v ")
    # determinism: regenerating gives identical bytes (the reference's atomics order does not)
    obj.generate_mesh(32, 2.0); again = obj.get_mesh(raw=True)
    assert all(np.array_equal(again[k], got[k]) for k in ("verts", "normals", "colors", "indices"))
    obj.close(); ds.close(); ref.close()


def test_mesh_matches_golden_fixture(pkg, ss):
    """No live oracle: marching cubes of the fixed analytic field and GenerateMesh of a c1 object with the pattern parameters
    against tests/golden/mesh.npz (bit-exact geometry; colours within one 8-bit level: logistic through __expf)."""
    from parity import CFGS, SCENE, load_golden
    from make_golden import MC_BOX, MC_RES, mc_field
    g = load_golden("mesh")
    m = pkg.marching_cubes(mc_field(MC_RES), MC_RES, 0.0, *MC_BOX)
    assert m["n_verts_real"] == int(g["mc_n_real"]) and np.array_equal(m["indices"], g["mc_indices"])
    assert np.array_equal(m["verts"].view(np.uint32), g["mc_verts"].view(np.uint32)) and np.array_equal(m["normals_raw"].view(np.uint32),
            g["mc_normals_raw"].view(np.uint32))
    sc = ss.make_scene(**SCENE); ds, obj = ge.make_problem(pkg, sc, CFGS["c1"])
    # parity.pattern_params
    k = np.arange(obj.info().n_grid_params, dtype=np.float64); p = obj.get_params(0); p[obj.info().n_mlp_params:] = (0.5 * np.sin(0.37 * k)).astype(np.float32)
    obj.set_params(p)
    # (a) the layer-at-a-time network (fp32 sums in the oracle's order): the lattice equals the oracle's bit for bit, and so does the geometry
    old = pkg.get_option("tile_render"); pkg.set_option("tile_render", 0)
    try:
        obj.generate_mesh(16, 0.0); o = obj.get_mesh()
    finally:
        pkg.set_option("tile_render", old)
    assert o["n_verts_real"] == int(g["obj_n_real"]) and np.array_equal(o["indices"], g["obj_indices"])
    assert np.array_equal(o["verts"].view(np.uint32), g["obj_verts"].view(np.uint32)) and np.array_equal(o["normals"].view(np.uint32),
            g["obj_normals"].view(np.uint32))
    assert np.abs(o["colors"].astype(int) - g["obj_colors"].astype(int)).max() <= 1
    # (b) the default since round 4: level tiles + the MFMA network (kernels_tilerender.hip).  Its fp16 outputs are those of the renderer -- within one fp16
    # ulp of the oracle's on a fraction of a percent of the points (MFMA summation order) --, so the surface has the same topology and its vertices move by
    # a 1e-3 of a cell at most
    obj.generate_mesh(16, 0.0); t = obj.get_mesh()
    assert t["n_verts_real"] == int(g["obj_n_real"]) and np.array_equal(t["indices"], g["obj_indices"])
    cell = float(np.max(np.ptp(g["obj_verts"][:int(g["obj_n_real"])], axis=0))) / 16.0
    assert np.abs(t["verts"] - g["obj_verts"]).max() < 2e-2 * cell and (t["verts"] == g["obj_verts"]).mean() > 0.9
    assert np.abs(t["normals"] - g["obj_normals"]).max() < 5e-2
    assert np.abs(t["colors"].astype(int) - g["obj_colors"].astype(int)).max() <= 1
    obj.close(); ds.close()
