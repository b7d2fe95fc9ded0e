#!/usr/bin/env python
"""Per-phase cycle breakdown of k_grid_scatter (GPU box; build first: tools/variant_build.sh sctime -DMON_SCATTER_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MON_CORE_LIB"] = os.path.join(ROOT, "ro-map_amd", "build_sctime", "libmon_core.so")
import __graft_entry__ as ge  # noqa: E402

PH = ["setup (counters, level constants)", "tile clear", "barrier after clear", "walk (+ the dW row loads behind it)", "barrier (wait for the slowest wave)",
        "tile write-out", "wave start (low 24 bits of the clock)"]


def main():
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    ds, obj = ge.make_problem(pkg, sc, dict(sample_seed=2024))
    L = C.CDLL(os.environ["MON_CORE_LIB"])
    for steps, name in ((10, "dense (step 10)"), (800, "sparse (step 810)")):
        obj.train(steps)
        buf = np.zeros((256, 16, 8), np.float32); L.mon_debug_scatter_timing(buf.ctypes.data_as(C.c_void_p))
        lv = buf[:, 0, 7].astype(int)
        print("\n== %s: mean cycles per wave; workgroups by level" % name)
        print("| level | " + " | ".join(PH) + " | total |"); print("|---|" + "---|" * (len(PH) + 1))
        for l in sorted(set(lv)):
            m = buf[lv == l][:, :, :7].mean((0, 1)); print("| %d | " % l + " | ".join("%.0f" % v for v in m) + " | %.0f |" % m.sum())
        w = buf[:, :, 3]; print("walk cycles per wave: min %.0f mean %.0f max %.0f" % (w.min(), w.mean(), w.max()))
        wg = int(np.where(lv == 7)[0][0]); print("workgroup %d (level 7), per wave:" % wg)
        print(np.array2string(buf[wg, :, :7], precision=0, suppress_small=True, max_line_width=200))
        st = buf[:, :, 6]
        print("wave start clock (low bits): spread inside a workgroup max %.0f; over the grid %.0f" % ((st.max(1) - st.min(1)).max(), st.max() - st.min()))
    obj.close(); ds.close()


if __name__ == "__main__":
    main()
