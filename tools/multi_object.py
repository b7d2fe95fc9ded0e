#!/usr/bin/env python
"""Aggregate training rate of K base.json objects on one GPU (thread + stream per object), per option setting:
   python tools/multi_object.py [K ...]            MON_OPTIONS=train_lanes=0 python tools/multi_object.py 1 4 8
   Every object trains `warm` steps alone first, then all K train `steps` steps concurrently; prints aggregate G ray-samples/s."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    warm, steps = int(os.environ.get("MON_MO_WARM", "5")), int(os.environ.get("MON_MO_STEPS", "100"))
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    for K in ks:
        ds = None; objs = []
        for k in range(K):
            ds, o = ge.make_problem(pkg, sc, dict(json.loads(os.environ.get("MON_MO_CFG", "{}")), sample_seed=3000 + k), dataset=ds); objs.append(o)
        th = [threading.Thread(target=o.train, args=(warm,)) for o in objs]
        [t.start() for t in th]; [t.join() for t in th]
        pkg.lib().mon_device_synchronize(0); t0 = time.perf_counter()
        lanes, chunk = int(os.environ.get("MON_MO_LANES", "0")), int(os.environ.get("MON_MO_CHUNK", "25"))
        sem = threading.Semaphore(lanes) if lanes > 0 else None

        def run(o):                                   # host-side prototype of a per-device scheduler: at most `lanes` objects have work in flight
            if sem is None:
                o.train(steps); return
            for _ in range(steps // chunk):
                with sem:
                    o.train(chunk)
        th = [threading.Thread(target=run, args=(o,)) for o in objs]
        [t.start() for t in th]; [t.join() for t in th]
        pkg.lib().mon_device_synchronize(0); dt = time.perf_counter() - t0
        print(json.dumps({"objects": K, "steps": [warm, warm + steps], "aggregate_G": round(K * steps * 4096 * 32 / dt / 1e9, 3),
                "us_per_object_step": round(1e6 * dt / steps / K, 2)}), flush=True)
        for o in objs:
            o.close()
        ds.close()


if __name__ == "__main__":
    main()
