#!/usr/bin/env python
"""Sweeps one library option over a list of values and prints tools/kernel_times.py's per-kernel durations for each:
   python tools/option_sweep.py big_switch 0 4096 16384 65536 [--sparse]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
import kernel_times as kt  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name, values = args[0], [int(v, 0) for v in args[1:]]
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    kw = json.loads(os.environ.get("MON_KT_CFG", "{}")); kw.setdefault("sample_seed", 2024)
    for v in values:
        pkg.set_option(name, v)
        out = {name: v, "dense": kt.window(pkg, sc, 0, kw)}
        if "--sparse" in sys.argv:
            out["sparse"] = kt.window(pkg, sc, 800, kw)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
