#!/usr/bin/env python
"""hipGraph replay (option use_graph: a captured PAIR of iterations, 8 kernel nodes, replayed n / 2 times) against plain stream launches, one base.json object:
microseconds per step for train() calls of 20, 20, 200, 1000, 1000 steps, each setting twice.   python tools/graph_ab.py"""
import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
for g in (0, 1, 0, 1):
    pkg.set_option("use_graph", g)
    ds, obj = ge.make_problem(pkg, sc, dict(sample_seed=2024)); obj.set_backend(1)
    obj.train(6); pkg.lib().mon_device_synchronize(0)
    res = []
    for n in (20, 20, 200, 1000, 1000):
        t0 = time.perf_counter(); obj.train(n); pkg.lib().mon_device_synchronize(0); res.append(round(1e6 * (time.perf_counter() - t0) / n, 2))
    print("use_graph", g, "us/step for calls of 20, 20, 200, 1000, 1000 steps:", res, flush=True)
    obj.close(); ds.close()
