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
    a = ap.parse_args()
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=a.views, H=480, W=640, f=525.0, seed=0)
    kw = dict(sample_seed=2024)
    if a.log2_hashmap_size:
        kw["log2_hashmap_size"] = a.log2_hashmap_size
    ds, obj = ge.make_problem(pkg, sc, kw)
    if a.extra:
        obj.train(a.extra)
    obj.train(a.warmup)
    pkg.lib().mon_device_synchronize(0)
    t0 = time.perf_counter(); obj.train(a.steps); pkg.lib().mon_device_synchronize(0); dt = time.perf_counter() - t0
    B = obj.cfg.rays_per_batch * obj.cfg.n_samples
    print("window: steps %d..%d, %.4f ms/step, %.1f M ray-samples/s, scattered samples in the last step %d" %
          (a.extra + a.warmup, a.extra + a.warmup + a.steps, 1e3 * dt / a.steps, a.steps * B / dt / 1e6, int(obj.buffer("state")[24])))
    obj.close(); ds.close()


if __name__ == "__main__":
    main()
