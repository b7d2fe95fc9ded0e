#!/usr/bin/env python
"""Object churn on one GPU: T host threads each create an object, train it in slices of random length (sometimes rendering, reading parameters or adding boxes
in between), destroy it and start over, for `seconds` -- the per-device training lanes, stream pool and snapshot side see objects come and go while others
train.
   python tools/churn_stress.py [threads] [seconds]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=16, H=240, W=320, f=260.0, seed=2)
    ds, first = ge.make_problem(pkg, sc, dict(sample_seed=1)); first.close()
    free0, _ = pkg.device_mem_info(0)
    stop = time.perf_counter() + seconds; stats = [0] * T; steps = [0] * T; errs = []

    def worker(k):
        rng = np.random.default_rng(100 + k)
        try:
            while time.perf_counter() < stop:
                _, o = ge.make_problem(pkg, sc, dict(sample_seed=int(rng.integers(1, 1 << 30))), dataset=ds)
                l0 = o.train(1)
                for _ in range(int(rng.integers(1, 12))):
                    n = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 64, 150])); l = o.train(n); steps[k] += n
                    assert np.isfinite(l), l
                    r = rng.random()
                    box = sc.objects[0]["boxes"][int(rng.integers(0, len(sc.objects[0]["boxes"])))]; pose = ss.colmajor(sc.Twc[int(box[0])])
                    if r < 0.15:
                        rgb, _, _ = o.render(box, pose); assert np.isfinite(rgb).all()
                    elif r < 0.3:
                        rgb, _, _, _ = o.render_snapshot(box, pose); assert np.isfinite(rgb).all()
                    elif r < 0.4:
                        assert np.isfinite(o.get_params(0)).all()
                    elif r < 0.5:
                        o.add_boxes(sc.objects[0]["boxes"][:2])
                assert o.train(1) < max(l0, 1.0) * 2 and o.info().skipped_batches == 0
                o.close(); stats[k] += 1
        except Exception as e:                                   # pragma: no cover
            errs.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    flip = threading.Event()

    def flipper():                                              # MON_CHURN_FLIP=1: the lanes switched off / to 2 / to 3 every millisecond under everything else
        i = 0
        while not flip.is_set():
            pkg.set_option("train_lanes", (0, 2, 3)[i % 3]); i += 1; time.sleep(0.001)
    ft = threading.Thread(target=flipper) if os.environ.get("MON_CHURN_FLIP") else None
    t0 = time.perf_counter(); [t.start() for t in th]
    if ft:
        ft.start()
    [t.join() for t in th]; dt = time.perf_counter() - t0
    if ft:
        flip.set(); ft.join(); pkg.set_option("train_lanes", 2)
    free1, _ = pkg.device_mem_info(0)
    if free0 - free1 > (256 << 20):                              # every object gone: device memory is back (to within what the runtime's pools keep)
        errs.append(("leak", "%d MB of device memory not returned" % ((free0 - free1) >> 20)))
    print("churn: %d threads, %.1f s: %d objects created and destroyed, %d training steps (%.2f G ray-samples/s aggregate), device memory %+d MB, errors: %s" % (T, dt, sum(stats), sum(steps), sum(steps) * 131072 / dt / 1e9, (free1 - free0) >> 20, errs))
    ds.close()
    sys.exit(1 if errs else 0)


if __name__ == "__main__":
    main()
