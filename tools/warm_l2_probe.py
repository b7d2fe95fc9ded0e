#!/usr/bin/env python
"""k_fused_train with and without an optimizer step between launches: the grid tables (3.7 MB at base.json) are rewritten by every k_optimizer, so each of
the 8 XCD L2s has to fetch them again at the start of the next k_fused_train.  Repeating stage 2 alone (same weights, same iteration, same samples) shows the
kernel with those lines already in L2 -- the gap is what the cold fills cost."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    ds, obj = ge.make_problem(pkg, sc, {"sample_seed": 2024})
    extra = int(os.environ.get("MON_EXTRA", "0"))
    obj.train(extra + 5); obj.set_profiling(True); obj.profile(reset=True)
    obj.train(20); p = obj.profile(reset=True); normal = 1e3 * p["ms"][1] / p["launches"][1]
    obj.train_stages(1)
    for _ in range(3): obj.train_stages(2)
    obj.profile(reset=True)
    for _ in range(20): obj.train_stages(2)
    p = obj.profile(reset=True); warm = 1e3 * p["ms"][1] / p["launches"][1]
    print(json.dumps(dict(extra=extra, fused_us_between_optimizer_steps=round(normal, 2), fused_us_repeated_same_weights=round(warm, 2))))


if __name__ == "__main__":
    main()
