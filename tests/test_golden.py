"""CPU: the oracle reproduces the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import numpy as np
import pytest

import __graft_entry__ as ge
from parity import CFGS, SCENE, close_f32, grid_probe_indices, load_golden, pattern_params


@pytest.mark.parametrize("name", sorted(CFGS))
def test_oracle_reproduces_golden(orc, ss, name):
    g = load_golden(name); sc = ss.make_scene(**SCENE)
    m = ge.make_oracle(orc, sc, CFGS[name]); m.set_params(pattern_params(m))
    m.generate_batch(); m.forward_backward()
    assert m.n_valid == int(g["n_valid"])
    for b in ("ray_o", "ray_d", "ray_t0", "ray_t1", "target", "bgcol", "ray_flag", "pts", "tdist", "E", "O", "dO", "dE", "rgb_ray", "depth_ray", "mask_ray",
            "loss_ray"):
        assert np.array_equal(m.buffer(b)[:g[b].size], g[b]), b
    close_f32(m.buffer("gmlp"), g["gmlp"], "gmlp", 1e-6, 1e-5)            # OpenMP partial sums: order may differ
    gi = grid_probe_indices(m.n_params - m.n_mlp)
    assert np.array_equal(m.buffer("ggrid")[gi], g["ggrid_probe"])
    m.train_step()
    close_f32(m.buffer("master")[:m.n_mlp], g["master_mlp_after"], "mlp params after one step", 2e-6)
    assert np.array_equal(m.buffer("master")[m.n_mlp:][gi], g["master_grid_probe_after"])
    m2 = ge.make_oracle(orc, sc, CFGS[name]); m2.set_params(pattern_params(m2))
    rgb, depth, mask = m2.render(g["render_box"], ss.colmajor(sc.Twc[int(g["render_box"][0])]), use_ema=False)
    assert np.array_equal(mask.astype(np.uint8), g["render_mask"]) and np.array_equal(depth, g["render_depth"])
    assert np.abs(rgb - g["render_rgb"].astype(np.float32)).max() < 1e-3
    assert np.array_equal(m2.density_grid(9, 9, 9, use_ema=False), g["density_probe"])
    m.close(); m2.close()


def test_oracle_reproduces_mesh_golden(orc, ss):
    from make_golden import MC_BOX, MC_RES, mc_field
    g = load_golden("mesh")
    m = orc.marching_cubes(mc_field(MC_RES), MC_RES, 0.0, *MC_BOX)
    assert m["n_verts_real"] == int(g["mc_n_real"]) and np.array_equal(m["indices"], g["mc_indices"])
    assert np.array_equal(m["verts"], g["mc_verts"]) and np.array_equal(m["normals_raw"], g["mc_normals_raw"])
    sc = ss.make_scene(**SCENE); mo = ge.make_oracle(orc, sc, CFGS["c1"]); mo.set_params(pattern_params(mo))
    o = mo.generate_mesh(16, 0.0, use_ema=False)
    assert o["n_verts_real"] == int(g["obj_n_real"]) and np.array_equal(o["indices"], g["obj_indices"]) and np.array_equal(o["verts"], g["obj_verts"])
    assert np.array_equal(o["normals"], g["obj_normals"]) and np.array_equal(o["colors"], g["obj_colors"])
    mo.close()
