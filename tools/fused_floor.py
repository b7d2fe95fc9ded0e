#!/usr/bin/env python
"""What bounds k_fused_train: the same kernel with parts switched off (option fused_ablate; results are wrong, only the time is used), HIP-event
durations over the bench window (steps 5..25 from init, base.json object, bench scene).

  full                 the product kernel
  encode only          fused_ablate = 96: prologue + the hash-grid gathers and interpolation of every ray, nothing else (no MLP, composite, backward, stores, dW reduction)
  (round 2 also ran the encode-only variant on 256 and on 64 workgroups through an option `fused_grid`, retired in round 4 with the experiment closed:
   profiles/r02_fused_floor.md keeps those rows)
  no rays              fused_ablate = 40: prologue only (launch, fragment image, ray select)

Prints one markdown table (committed as profiles/r02_fused_floor.md)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
import kernel_times as kt  # noqa: E402


def main():
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0); kw = {"sample_seed": 2024}
    rows = [("full (stagger on)", {}), ("full, stagger off", {"fused_stagger": 0}), ("encode only (fused_ablate 96)", {"fused_ablate": 96, "fused_stagger": 0}),
            ("no rays (fused_ablate 40): prologue only", {"fused_ablate": 40, "fused_stagger": 0})]
    defaults = {"fused_ablate": 0, "fused_stagger": -1}
    out = {}
    print("| k_fused_train variant | mean launch, us (HIP events, steps 5..25) |\n|---|---|")
    for name, opts in rows:
        for k, v in defaults.items():
            pkg.set_option(k, opts.get(k, v))
        w = kt.window(pkg, sc, 0, kw); out[name] = w["fused_us"]
        print("| %s | %.2f |" % (name, w["fused_us"]), flush=True)
    for k, v in defaults.items():
        pkg.set_option(k, v)
    print("\n```\n%s\n```" % json.dumps(out))


if __name__ == "__main__":
    main()
