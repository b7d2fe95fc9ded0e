"""GPU tests (-m gpu) of the inference side on feature-planar level tiles (ro-map_amd/csrc/kernels_tilerender.hip): Render /
RenderVideo (CORE/src/nerf_model.cu:1702-1830, 1832-1991), GetDensityOnGrid (:2007-2048) and the mesh's vertex colours.
The bar: images BIT-identical to the gather render (option tile_render = 0), which the parity tests tie to the oracle and the
golden fixtures; the lattice of raw densities against the oracle like tests/test_gpu_mesh.py."""
import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import C1
from parity import CFGS, SCENE, load_golden, pattern_params

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tile_option(pkg):
    old = pkg.get_option("tile_render")
    yield lambda v: pkg.set_option("tile_render", v)
    pkg.set_option("tile_render", old)


def _render_both(pkg, obj, box, pose, tile_option, **kw):
    tile_option(0); a = obj.render(box, pose, **kw)
    tile_option(2); b = obj.render(box, pose, **kw)
    return a, b


@pytest.mark.parametrize("name", sorted(CFGS))
def test_tile_render_is_bit_identical_to_the_gather_render(pkg, orc, ss, name, tile_option):
    """Every network shape of the fixtures, pattern parameters: crops around the chunk boundaries (a single pixel, 16 383 / 16 385
    pixels, a whole image in which most rays miss the box), world-frame and object-frame poses."""
    sc = ss.make_scene(**SCENE); kw = CFGS[name]; g = load_golden(name)
    ds, obj = ge.make_problem(pkg, sc, kw); obj.set_backend(1)
    ref = ge.make_oracle(orc, sc, kw); p = pattern_params(ref); ref.close(); obj.set_params(p)
    ob = sc.objects[0]["boxes"][2]; v, cx, cy = int(ob[0]), int(ob[1] + ob[4] // 2), int(ob[2] + ob[3] // 2)
    pose = ss.colmajor(sc.Twc[v])
    boxes = [g["render_box"], (v, cx, cy, 1, 1), (v, 0, 0, 127, 129), (v, 0, 0, 113, 145), (v, 0, 0, sc.H, sc.W), (v, sc.W - 3, sc.H - 2, 2, 3)]
    for bx in boxes:
        box = np.array(bx, np.uint32); pose = ss.colmajor(sc.Twc[int(box[0])])
        (rgb0, d0, m0), (rgb1, d1, m1) = _render_both(pkg, obj, box, pose, tile_option)
        assert rgb1.shape == (int(box[3]), int(box[4]), 3)
        assert np.array_equal(m0, m1) and np.array_equal(rgb0.view(np.uint32), rgb1.view(np.uint32)) and np.array_equal(d0.view(np.uint32),
                d1.view(np.uint32)), bx
    box = np.array(g["render_box"], np.uint32); Toc = ss.colmajor(sc.objects[0]["Tow"] @ sc.Twc[int(box[0])])
    (rgb0, d0, m0), (rgb1, d1, m1) = _render_both(pkg, obj, box, Toc, tile_option, pose_is_Toc=True)
    assert np.array_equal(m0, m1) and np.array_equal(rgb0, rgb1) and np.array_equal(d0, d1) and m1.mean() > 0.02
    # golden fixture directly on the tile path
    tile_option(2); rgb, depth, mask = obj.render(box, ss.colmajor(sc.Twc[int(box[0])]))
    gsame = mask.astype(np.uint8) == g["render_mask"]
    assert gsame.mean() > 0.995 and np.abs(rgb - g["render_rgb"].astype(np.float32))[gsame].max() < 5e-3
    obj.close(); ds.close()


@pytest.mark.parametrize("kw", [dict(log2_hashmap_size=14, n_levels=8, base_resolution=8, per_level_scale=1.5, n_neurons=64, n_hidden_layers=1),
                                dict(log2_hashmap_size=15, n_levels=12, base_resolution=12, per_level_scale=1.7, n_neurons=32, n_hidden_layers=2),
                                dict(log2_hashmap_size=16, n_levels=16, base_resolution=16, per_level_scale=2.0, n_neurons=64, n_hidden_layers=2),
                                # tcnn FullyFusedMLP's widest network on the fused kernels (round 5), both encoder paddings
                                dict(log2_hashmap_size=15, n_levels=16, n_neurons=128, n_hidden_layers=1),
                                dict(log2_hashmap_size=15, n_levels=6, n_neurons=128, n_hidden_layers=1)],
                         ids=["T14-L8", "T15-L12-2x32", "T16-scale2-2x64", "T15-1x128", "T15-L6-1x128"])
def test_tile_render_matches_the_gather_render_on_other_level_tables(pkg, ss, tile_option, kw):
    """Level tables the fixtures do not have -- small hashed levels (T = 2^14, 2^15), a per-level scale of 2 (the finest levels' resolutions pass the table
    size, the res = 65 536 level indexes by x alone), other base resolutions -- on a briefly trained object: bit-identical images on both paths, the
    density lattice likewise."""
    sc = ss.make_scene(**SCENE)
    ds, obj = ge.make_problem(pkg, sc, dict(rays_per_batch=1024, **kw)); obj.set_backend(1)
    obj.train(60)
    ob = sc.objects[0]["boxes"][1]; v = int(ob[0])
    for bx in [(v, 0, 0, sc.H, sc.W), (v, int(ob[1]), int(ob[2]), int(ob[3]), int(ob[4]))]:
        box = np.array(bx, np.uint32)
        (rgb0, d0, m0), (rgb1, d1, m1) = _render_both(pkg, obj, box, ss.colmajor(sc.Twc[v]), tile_option)
        assert np.array_equal(m0, m1) and np.array_equal(rgb0.view(np.uint32), rgb1.view(np.uint32)) and np.array_equal(d0.view(np.uint32),
                d1.view(np.uint32)), (kw, bx)
        assert m1.mean() > 0.01
    tile_option(0); g0 = obj.density_grid(24, 20, 28)
    tile_option(2); g1 = obj.density_grid(24, 20, 28)
    assert np.allclose(g0, g1, rtol=2e-3, atol=1e-4) and np.isfinite(g1).all()
    obj.close(); ds.close()


def test_tile_render_of_a_trained_object_all_entry_points(pkg, ss, tile_option):
    """base.json network trained 300 steps (EMA weights, opaque surfaces: the opaque-prefix skip is exercised): the owner's render,
    the snapshot render on the inference stream, a second object's render in between (the per-device workspace changes hands and
    its tile image is rebuilt), all bit-identical to the gather render; repeated renders reuse the image."""
    sc = ss.make_scene(n_views=16, H=240, W=320, f=260.0, n_objects=2, seed=3)
    ds, a = ge.make_problem(pkg, sc, dict(sample_seed=5)); a.set_backend(1)
    _, b = ge.make_problem(pkg, sc, dict(sample_seed=6), obj_index=1, dataset=ds); b.set_backend(1)
    a.train(300); b.train(100)
    for obj, oi in ((a, 0), (b, 1), (a, 0)):
        ob = sc.objects[oi]["boxes"][1]; v = int(ob[0]); pose = ss.colmajor(sc.Twc[v])
        box = np.array([v, max(0, int(ob[1]) - 20), max(0, int(ob[2]) - 20), min(sc.H, int(ob[3]) + 40), min(sc.W, int(ob[4]) + 40)], np.uint32)
        box[3] = min(int(box[3]), sc.H - int(box[2])); box[4] = min(int(box[4]), sc.W - int(box[1]))
        assert int(box[3]) * int(box[4]) >= 4096
        (rgb0, d0, m0), (rgb1, d1, m1) = _render_both(pkg, obj, box, pose, tile_option)
        assert m1.mean() > 0.05 and np.array_equal(m0, m1) and np.array_equal(rgb0, rgb1) and np.array_equal(d0, d1)
        tile_option(1); rgb2, d2, m2 = obj.render(box, pose)              # default: a crop of this size takes the tile path
        assert np.array_equal(rgb2, rgb1) and np.array_equal(d2, d1)
        tile_option(0); s0 = obj.render_snapshot(box, pose)
        tile_option(2); s1 = obj.render_snapshot(box, pose)
        for u, w in zip(s0[:3], s1[:3]):
            assert np.array_equal(u, w)
        assert np.array_equal(s1[0], rgb1)                                  # the snapshot published at the end of train() holds the same weights
    a.close(); b.close(); ds.close()


def test_tile_render_xorwow_stream(pkg, ss, tile_option):
    """'Same inputs' mode: the render's jitter comes from a fresh XORWOW generator per Render (nerf_model.cu:1725-1728, :1781)."""
    sc = ss.make_scene(**SCENE)
    ds, obj = ge.make_problem(pkg, sc, dict(CFGS["c2s"], rng_flags=1)); obj.set_backend(1)
    obj.train(50)
    box = np.array([0, 0, 0, sc.H, sc.W], np.uint32); pose = ss.colmajor(sc.Twc[0])
    (rgb0, d0, m0), (rgb1, d1, m1) = _render_both(pkg, obj, box, pose, tile_option)
    assert np.array_equal(m0, m1) and np.array_equal(rgb0, rgb1) and np.array_equal(d0, d1)
    obj.close(); ds.close()


def test_density_lattice_and_mesh_on_level_tiles(pkg, orc, ss, tile_option):
    """GetDensityOnGrid and the vertex colours through k_encode_feat + the MFMA MLP against the oracle on the same inference weights
    (the bars of test_object_mesh_matches_oracle), and against the layer-at-a-time kernels they replace."""
    sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)
    ds, obj = ge.make_problem(pkg, sc, C1); ref = ge.make_oracle(orc, sc, C1); obj.set_backend(1)
    assert obj.train(300) < 0.05
    ref.set_params(obj.get_params(0)); ref.set_ema(obj.get_params(2))
    tile_option(0); d_old = obj.density_grid(32, 32, 32); obj.generate_mesh(32, 2.0); m_old = obj.get_mesh(raw=True)
    tile_option(1); d_new = obj.density_grid(32, 32, 32); nv, ni = obj.generate_mesh(32, 2.0); m_new = obj.get_mesh(raw=True)
    rd = ref.density_grid(32, 32, 32, use_ema=True)
    assert (d_new == rd).mean() > 0.999 and np.abs(d_new - rd).max() < 0.05
    assert (d_new == d_old).mean() > 0.999 and np.abs(d_new - d_old).max() < 0.05
    col = ref.mesh_colors(m_new["verts"])
    assert np.abs(m_new["colors_f32"] - col).max() < 2e-3
    if np.array_equal(d_new, d_old):
        assert np.array_equal(m_new["verts"], m_old["verts"]) and np.array_equal(m_new["indices"], m_old["indices"])
    want = orc.marching_cubes(d_new, (32, 32, 32), 2.0, ref._amin, ref._amax)
    assert np.array_equal(m_new["indices"], want["indices"]) and np.array_equal(m_new["verts"].view(np.uint32), want["verts"].view(np.uint32))
    # a lattice whose point count is not a multiple of anything convenient, and one beyond a chunk (2 ^ 20 samples)
    g1 = obj.density_grid(7, 5, 3); tile_option(0); g0 = obj.density_grid(7, 5, 3)
    assert g1.shape == g0.shape and np.abs(g1 - g0).max() < 0.05
    tile_option(1); big = obj.density_grid(128, 96, 96); assert big.size == 128 * 96 * 96 and np.isfinite(big).all()
    sub = obj.density_grid(2, 2, 2)
    assert np.array_equal(big.reshape(-1)[[0, 127]], sub.reshape(-1)[[0, 1]])          # corners of the cube are lattice points of both
    obj.close(); ds.close(); ref.close()
