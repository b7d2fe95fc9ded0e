"""Device memory of one object NeRF (hipMemGetInfo deltas): base.json and the T = 2^22 stress table, and of the per-device tile render workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=16, H=240, W=320, f=260.0, seed=2)
ds, first = ge.make_problem(pkg, sc, dict(sample_seed=1)); first.close()
for name, kw in (("base.json", {}), ("T = 2^22", dict(log2_hashmap_size=22))):
    f0, _ = pkg.device_mem_info(0)
    objs = [ge.make_problem(pkg, sc, dict(sample_seed=2 + k, **kw), dataset=ds)[1] for k in range(4)]
    for o in objs:
        o.train(2)
    f1, _ = pkg.device_mem_info(0)
    box = np.array([0, 0, 0, sc.H, sc.W], np.uint32); objs[0].render(box, ss.colmajor(sc.Twc[0]))
    f2, _ = pkg.device_mem_info(0)
    print("%-10s %6.1f MB per object (mean of 4), + %d MB once per device for a full-frame render (workspace / output buffers)" % (name,
            (f0 - f1) / 4 / 2 ** 20, (f1 - f2) >> 20), flush=True)
    for o in objs:
        o.close()
