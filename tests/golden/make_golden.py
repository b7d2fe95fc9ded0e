"""Generates the committed golden fixtures from the CPU oracle (oracle/mon_oracle.c).

The reference itself cannot run here (CUDA-only, tiny-cuda-nn submodule absent) and ships no vectors, so these
fixtures pin the ORACLE's behaviour (regression pin + a GPU-box check that needs no live oracle); the oracle in
turn is pinned by the closed-form / autograd KATs in tests/test_oracle_kat.py.
Run:  python tests/golden/make_golden.py     (writes tests/golden/*.npz; inputs are regenerated from seeds)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

SCENE = dict(n_views=12, H=120, W=160, f=130.0, seed=0)
CFGS = {"c1": dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2),
        "c2s": dict(rays_per_batch=256, n_levels=16, n_neurons=64, n_hidden_layers=1)}      # base.json network, small batch


def pattern_params(m):
    """Deterministic non-trivial parameters: init MLP weights, grid = 0.5 sin(0.37 k) (k = flat grid index)."""
    p = m.buffer("master").copy()
    k = np.arange(m.n_params - m.n_mlp, dtype=np.float64)
    p[m.n_mlp:] = (0.5 * np.sin(0.37 * k)).astype(np.float32)
    return p


def grid_probe_indices(n_grid):
    return (np.arange(2048, dtype=np.int64) * 7919) % n_grid


def mc_field(res3):
    """Fixed analytic density lattice (x fastest): a wobbly sphere, threshold 0."""
    rx, ry, rz = res3
    z, y, x = np.meshgrid(np.linspace(-1, 1, rz), np.linspace(-1, 1, ry), np.linspace(-1, 1, rx), indexing="ij")
    return (0.7 - np.sqrt((x - 0.05) ** 2 + (y + 0.03) ** 2 + (z - 0.02) ** 2) + 0.05 * np.sin(7 * x) * np.cos(5 * y)).astype(np.float32).reshape(-1)


MC_RES = (21, 17, 19); MC_BOX = ([-1.0, -0.5, 0.25], [1.0, 0.75, 2.0])


def make_mesh_golden(orc, ss):
    """Marching cubes + normals of the analytic field, and GenerateMesh of a c1 object with the pattern parameters."""
    out = {}
    m = orc.marching_cubes(mc_field(MC_RES), MC_RES, 0.0, *MC_BOX)
    out["mc_verts"] = m["verts"]; out["mc_indices"] = m["indices"]; out["mc_normals_raw"] = m["normals_raw"]; out["mc_n_real"] = np.uint32(m["n_verts_real"])
    sc = ss.make_scene(**SCENE); mo = ge.make_oracle(orc, sc, CFGS["c1"]); mo.set_params(pattern_params(mo))
    g = mo.generate_mesh(16, 0.0, use_ema=False)                 # the pattern grid is oscillatory: threshold 0 gives a large surface
    for k in ("verts", "indices", "normals", "colors"):
        out["obj_" + k] = g[k]
    out["obj_n_real"] = np.uint32(g["n_verts_real"]); mo.close()
    np.savez_compressed(os.path.join(HERE, "mesh.npz"), **out)
    print("mesh: mc", m["n_verts_real"], "verts", m["indices"].size // 3, "faces; object", g["n_verts_real"], "verts", g["indices"].size // 3, "faces;",
          os.path.getsize(os.path.join(HERE, "mesh.npz")), "bytes")


def main():
    orc = ge.load_oracle(); ss = ge.load_tools()
    sc = ss.make_scene(**SCENE)
    for name, kw in CFGS.items():
        m = ge.make_oracle(orc, sc, kw)
        m.set_params(pattern_params(m))
        m.generate_batch(); m.forward_backward()
        R, S, B = m.R, m.S, m.R * m.S
        nr, ns = min(R, 64), min(B, 512)
        gi = grid_probe_indices(m.n_params - m.n_mlp)
        out = dict(n_valid=np.uint32(m.n_valid), loss=np.float32(m.loss))
        for b, cnt, width in [("ray_o", nr, 3), ("ray_d", nr, 3), ("ray_t0", nr, 1), ("ray_t1", nr, 1), ("target", nr, 3), ("bgcol", nr, 3),
                              ("ray_flag", nr, 1), ("pts", ns, 3), ("tdist", ns, 1), ("E", ns, m.Epad), ("O", ns, 4), ("dO", ns, 4), ("dE", ns, m.Epad),
                              ("rgb_ray", nr, 3), ("depth_ray", nr, 1), ("mask_ray", nr, 1), ("loss_ray", nr, 1)]:
            out[b] = m.buffer(b)[:cnt * width]
        out["gmlp"] = m.buffer("gmlp")
        out["ggrid_probe"] = m.buffer("ggrid")[gi]; out["ggrid_abs_probe"] = m.buffer("ggrid_abs")[gi]
        # one full optimizer step from the same state
        m.train_step()
        out["master_mlp_after"] = m.buffer("master")[:m.n_mlp]
        out["master_grid_probe_after"] = m.buffer("master")[m.n_mlp:][gi]
        # deterministic render with the pattern parameters (EMA not used: fresh model)
        m2 = ge.make_oracle(orc, sc, kw); m2.set_params(pattern_params(m2))
        box = sc.objects[0]["boxes"][0]
        rgb, depth, mask = m2.render(box, ss.colmajor(sc.Twc[box[0]]), use_ema=False)
        out["render_box"] = np.asarray(box, np.uint32); out["render_rgb"] = rgb.astype(np.float16); out["render_depth"] = depth.astype(np.float32)
        out["render_mask"] = mask.astype(np.uint8)
        out["density_probe"] = m2.density_grid(9, 9, 9, use_ema=False)
        np.savez_compressed(os.path.join(HERE, "%s.npz" % name), **out)
        print(name, "n_valid", m.n_valid, "loss", m.loss, os.path.getsize(os.path.join(HERE, "%s.npz" % name)), "bytes")
        m.close(); m2.close()


if __name__ == "__main__":
    main()
    make_mesh_golden(ge.load_oracle(), ge.load_tools())
