#!/usr/bin/env python
"""Step time of one network shape on the bench scene (steps 5..25 and 805..825), for rocprofv3 --kernel-trace --stats:  python tools/shape_times.py W NH [L]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
W, NH = int(sys.argv[1]), int(sys.argv[2]); L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
ds, obj = ge.make_problem(pkg, sc, dict(sample_seed=2024, n_neurons=W, n_hidden_layers=NH, n_levels=L))
out = {"shape": "%dx%d L%d" % (W, NH, L), "backend": int(obj.info().backend)}
for name, extra in (("dense", 0), ("late", 780)):
    obj.train(extra + 5); pkg.lib().mon_device_synchronize(0)
    t0 = time.perf_counter(); obj.train(20); pkg.lib().mon_device_synchronize(0); out[name + "_us"] = round(1e6 * (time.perf_counter() - t0) / 20, 1)
print(json.dumps(out), flush=True)
