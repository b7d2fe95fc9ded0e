"""Cost of training in short slices (the online manager trains in slices of 16 / n iterations so that callers get in between): us per object-step for one
object and for four objects trained concurrently (thread + stream each), slice lengths 4 / 16 / 200.   python tools/slice_bench.py [repo root]"""
import os, sys, time, threading
root = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, root); os.chdir(root)
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=16, H=240, W=320, f=260.0, seed=2)
ds = None; objs = []
for k in range(4):
    ds, o = ge.make_problem(pkg, sc, dict(sample_seed=700 + k), dataset=ds); objs.append(o)
for o in objs: o.train(600)
def run(o, n, sl):
    for _ in range(n): o.train(sl)
for sl in (4, 16, 200):
    n = 800 // sl
    pkg.lib().mon_device_synchronize(0); t0 = time.perf_counter(); run(objs[0], n, sl); pkg.lib().mon_device_synchronize(0); t1 = time.perf_counter() - t0
    th = [threading.Thread(target=run, args=(o, n, sl)) for o in objs]
    pkg.lib().mon_device_synchronize(0); t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; pkg.lib().mon_device_synchronize(0)
    t4 = time.perf_counter() - t0
    print("%s slice %3d: 1 object %.1f us/step; 4 objects concurrently %.1f us per object-step" % (root[-5:], sl, 1e6 * t1 / (n * sl), 1e6 * t4 / (4 * n * sl)))
