#!/usr/bin/env python
"""Per-kernel HIP-event durations of the bench workload (base.json object, bench scene) in the two windows the profiles quote --
dense = steps 5..25 from init (every sample carries a gradient; what `bench.py --steps 20 --warmup 5` times), sparse = steps
805..825 -- plus the un-instrumented step time of both windows and a CRC of the trained parameters (A/B checks of kernel variants:
MON_CORE_LIB=ro-map_amd/build_<tag>/libmon_core.so python tools/kernel_times.py).  One line of JSON."""
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def window(pkg, sc, extra, kw):
    ds, obj = ge.make_problem(pkg, sc, kw)
    obj.train(extra + 5); pkg.lib().mon_device_synchronize(0)
    t0 = time.perf_counter(); obj.train(20); pkg.lib().mon_device_synchronize(0); plain = (time.perf_counter() - t0) / 20
    crc = zlib.crc32(obj.get_params(0).tobytes()); n_grad = int(obj.buffer("state")[24])      # gradient-carrying samples of the window's last iteration
    live = None
    if kw.get("occupancy_skip"):
        try:
            c = obj.buffer("live_cnt").reshape(2, 64, 16)[:, :, 0]; live = [int(c[0].sum()), int(c[1].sum()), int(c.max())]      # per parity set; largest partition
        except Exception:
            live = None
    obj.close()
    _, obj = ge.make_problem(pkg, sc, kw, dataset=ds)
    obj.train(extra + 5); obj.set_profiling(True); obj.profile(reset=True); obj.train(20); p = obj.profile(reset=True)
    obj.close(); ds.close()
    avg = lambda k: round(1e3 * p["ms"][k] / max(1, p["launches"][k]), 2)
    return dict(step_us=round(1e6 * plain, 2), points_us=avg(7), encode_us=avg(6), fused_us=avg(1), scatter_us=avg(4), optim_us=avg(2), crc="%08x" % crc, n_grad=n_grad, live=live)


def main():
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    kw = json.loads(os.environ.get("MON_KT_CFG", "{}")); kw.setdefault("sample_seed", 2024)
    out = dict(lib=os.environ.get("MON_CORE_LIB", "default"), dense=window(pkg, sc, 0, kw))
    if not os.environ.get("MON_KT_DENSE_ONLY"):
        out["sparse"] = window(pkg, sc, 800, kw)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
