#!/usr/bin/env python
"""Per-phase cycle breakdown of k_fused_train (runs on the GPU box).

Builds a -DMON_FUSED_TIMING variant of the library into ro-map_amd/build_timing/ (every phase boundary drains the memory
counters and reads the shader clock, so a phase owns the latency it waits for; the sum is therefore an upper bound of the
un-instrumented wave time), runs a few training steps of the bench workload and prints the mean cycles per wave and phase."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PH = ["prologue (frag image, compaction table)", "ray select + position", "encode (16 levels of gathers)", "MLP forward (MFMA)", "LDS transposes for dW",
      "composite + loss + dL/dO", "dWo, dH (MFMA)", "dW0, dE (MFMA)", "dE / x stores", "epilogue (dW reduce + store)"]


def main():
    out = os.path.join(ROOT, "ro-map_amd", "build_timing")
    if not os.path.exists(os.path.join(out, "libmon_core.so")) or os.environ.get("MON_TIMING_REBUILD"):
        subprocess.check_call([os.path.join(ROOT, "tools", "variant_build.sh"), "timing", "-DMON_FUSED_TIMING"])
    lib = os.path.join(out, "libmon_core.so")
    if os.environ.get("MON_TIMING_BUILD_ONLY"):
        return
    import __graft_entry__ as ge
    pkg = ge.load_package(); ss = ge.load_tools()
    import importlib; binding = importlib.import_module(pkg.__name__ + ".binding")
    binding.lib_path = lambda: lib; binding._lib = None
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    grid = 512          # (k_fused_train's grid: two workgroups per CU)
    ds, obj = ge.make_problem(pkg, sc, {}); obj.set_backend(1)
    obj.train(int(os.environ.get("MON_TIMING_STEPS", "50")))
    t = obj.buffer("tdist")[:grid * 4 * 16].reshape(grid * 4, 16)
    print("grid %d workgroups: %d rays per wave" % (grid, 4096 // (grid * 4)))
    tot = t[:, :10].sum(1).mean()
    print("| phase | mean cycles / wave | share |\n|---|---|---|")
    for k, name in enumerate(PH):
        print("| %s | %.0f | %.1f%% |" % (name, t[:, k].mean(), 100 * t[:, k].mean() / tot))
    print("| total | %.0f | |" % tot)
    tw = t[:, :10].sum(1); enc = t[:, 2]
    print("per-wave total: p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f;   encode: p10 %.0f p50 %.0f p90 %.0f max %.0f" % (*np.percentile(tw, [10, 50, 90,
            99, 100]), *np.percentile(enc, [10, 50, 90, 100])))
    hw = t[:, 15].astype(np.int64); start = t[:, 13]; end = t[:, 14]
    t0 = start.min(); dur = (np.where(end < start, end + 2 ** 24, end) - t0) / 100.0; st_us = (start - t0) / 100.0
    print("wall clock: wave start p50 %.2f p99 %.2f max %.2f us after the first, wave end p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (*np.percentile(st_us, [50,
            99, 100]), *np.percentile(dur, [10, 50, 90, 100])))
    cu = (hw >> 8) & 0xff; se = (hw >> 13) & 7; key = cu  # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    ids = np.unique(hw >> 8)
    per = np.array([dur[(hw >> 8) == i].max() for i in ids]); cnt = np.array([((hw >> 8) == i).sum() for i in ids])
    print("distinct (se, sh, cu) ids per XCD slice seen: %d; waves per id: min %d max %d; slowest-wave end per id: p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (len(ids), cnt.min(), cnt.max(), *np.percentile(per, [10, 50, 90, 100])))
    wgt = tw.reshape(-1, 4).max(1)
    print("per-workgroup (slowest wave, without the epilogue wait): p50 %.0f p90 %.0f max %.0f;  by XCD (blockIdx %% 8) mean: %s" % (*np.percentile((tw - t[:,
            9]).reshape(-1, 4).max(1), [50, 90, 100]), " ".join("%.0f" % wgt[x::8].mean() for x in range(8))))


if __name__ == "__main__":
    main()
