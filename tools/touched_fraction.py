import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
ds, obj = ge.make_problem(pkg, sc, {"sample_seed": 2024})
for steps in (5, 200, 800, 3000):
    obj.train(steps - (0 if steps == 5 else {200: 5, 800: 200, 3000: 800}[steps]))
    obj.train_stages(1 | 2)
    g = obj.buffer("ggrid_h").view(np.float16).astype(np.float32)
    nz = g != 0
    ent = nz.reshape(-1, 2).any(1); ch = nz.reshape(-1, 8).any(1)
    st = obj.buffer("state")
    print("step %5d: samples with gradient %6d | params nonzero %.3f, entries %.3f, 8-param chunks %.3f" % (steps, int(st[8]), nz.mean(), ent.mean(),
            ch.mean()), flush=True)
    obj.train_stages(4)
