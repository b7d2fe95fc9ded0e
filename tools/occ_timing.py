"""Late-training step time with and without occupancy skipping (mon_config::occupancy_skip), per forward chain and refresh interval: the bench's
`late_training_with_occupancy_skipping` line taken apart.   python tools/occ_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
ds = None
def run(tag, cfg_kw, opts):
    global ds
    for k, v in opts.items(): pkg.set_option(k, v)
    d, o = ge.make_problem(pkg, sc, dict(sample_seed=2024, **cfg_kw), dataset=ds); ds = d
    o.train(800); pkg.lib().mon_device_synchronize(0)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o.train(200); pkg.lib().mon_device_synchronize(0); ts.append((time.perf_counter() - t0) / 200)
    print("%-52s %s  median %.2f us/step" % (tag, " ".join("%.2f" % (1e6 * t) for t in ts), 1e6 * sorted(ts)[2]), flush=True)
    o.close()
    for k in opts: pkg.set_option(k, {"lds_encode": 1}[k])
run("default (no occupancy grid)", {}, {})
run("occupancy skipping (level tiles, live-sample lists)", dict(occupancy_skip=1), {})
run("occupancy skipping, gather chain only", dict(occupancy_skip=1), dict(lds_encode=0))
run("default again", {}, {})
