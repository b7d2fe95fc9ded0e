"""Fraction of ray-samples whose dL/dE is exactly zero (samples behind the termination point of their ray) in the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
ds, obj = ge.make_problem(pkg, sc, {}); obj.set_backend(1)
done = 0
for target in (1, 50, 200, 500, 1000, 2000, 5000):
    obj.set_debug_dump(False); obj.train(target - done - 1); obj.set_debug_dump(True); obj.train(1); done = target
    dE = obj.buffer("dE").reshape(-1, 32); flag = obj.buffer("ray_flag")
    zero = (dE == 0).all(1).reshape(-1, 32)                       # [rays, samples]
    nact = (~zero).sum(1)
    print("step %5d: zero-gradient samples %.1f%%; object rays %.1f%% (mean active samples %.1f), background rays (mean active %.1f)"
          % (target, 100 * zero.mean(), 100 * flag.mean(), nact[flag > 0].mean(), nact[flag == 0].mean()), flush=True)
