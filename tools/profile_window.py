#!/usr/bin/env python
"""The driver's bench window and nothing else, for rocprofv3: base.json object on the bench scene, W warm-up steps and K steps
from init, then exit (no torch import, no render, no CPU baseline), so that a kernel trace / PMC pass of this command holds
exactly the dispatches `bench.py --gpus 1 --steps K --warmup W` times.  tools/rocpd_window.py then averages dispatch
numbers [W, W+K) of every kernel.

   python tools/profile_window.py [--warmup 5] [--steps 20] [--extra 0] [--log2-hashmap-size 0]
   --extra N: N further steps before the window (e.g. 800 for the late-training regime)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--extra", type=int, default=0)
    ap.add_argument("--log2-hashmap-size", type=int, default=0)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--occupancy", action="store_true", help="mon_config::occupancy_skip = 1 (opt-in occupancy-grid skipping)")
    ap.add_argument("--objects", type=int, default=1,
            help="K objects trained concurrently, one host thread each (dispatch numbers [K W, K (W + K_steps)) of a kernel are then the window)")
    a = ap.parse_args()
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=a.views, H=480, W=640, f=525.0, seed=0)
    kw = dict(sample_seed=2024)
    if a.log2_hashmap_size:
        kw["log2_hashmap_size"] = a.log2_hashmap_size
    if a.occupancy:
        kw["occupancy_skip"] = 1
    import threading
    ds, obj = ge.make_problem(pkg, sc, kw)
    objs = [obj] + [ge.make_problem(pkg, sc, dict(kw, sample_seed=2024 + k), dataset=ds)[1] for k in range(1, a.objects)]
    def all_train(n):
        if len(objs) == 1:
            return obj.train(n)
        th = [threading.Thread(target=o.train, args=(n,)) for o in objs]
        [t.start() for t in th]; [t.join() for t in th]
    if a.extra:
        all_train(a.extra)
    all_train(a.warmup)
    pkg.lib().mon_device_synchronize(0)
    t0 = time.perf_counter(); all_train(a.steps); pkg.lib().mon_device_synchronize(0); dt = time.perf_counter() - t0
    B = obj.cfg.rays_per_batch * obj.cfg.n_samples
    print("window: %d object(s), steps %d..%d, %.4f ms/step per object-step, %.1f M ray-samples/s aggregate, scattered samples in the last step %d" %
          (len(objs), a.extra + a.warmup, a.extra + a.warmup + a.steps, 1e3 * dt / a.steps / len(objs), len(objs) * a.steps * B / dt / 1e6,
                  int(obj.buffer("state")[24])))
    for o in objs[1:]:
        o.close()
    obj.close(); ds.close()


if __name__ == "__main__":
    main()
