#!/usr/bin/env python
"""profiles/<round>_window_T22.md (round = argv[1], default r04) from gpurun_out/<round>_T22_{init,late,init8,late8} -- sections whose directory is missing are left out -- (tools/gpu_profile_window.sh with WINDOW_ARGS="--log2-hashmap-size 22 [--objects 8]")."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from update_profiles_render import counters, table  # noqa: E402


STEP_BYTES = {}      # tag -> counter traffic of one step's kernels, bytes (sum over kernels of MB per launch x launches per step of the 20-step window)


def section(tag, title, L):
    d = os.path.join(ROOT, "gpurun_out", tag)
    if not os.path.exists(os.path.join(d, "kernel_window.md")):
        return
    t = table(os.path.join(d, "kernel_window.md")); c = counters(os.path.join(d, "pmc_window.md")) if os.path.exists(os.path.join(d, "pmc_window.md")) else {}
    head = [l for l in open(os.path.join(d, "trace.log")) if l.startswith("window:")]
    L.append("## %s\n" % title)
    if head:
        L.append("`%s`\n" % head[-1].strip())
    L.append("| kernel | dispatches | avg us | min us | max us | traffic beyond L2, MB / launch | rate TB/s | L2 requests | L2 hit share |\n|---|---|---|---|---|---|---|---|---|")
    tot = 0.0
    for k, r in sorted(t.items(), key=lambda kv: -kv[1]["total"]):
        cc = c.get(k, {}); g = lambda n: cc.get(n, (0, 0.0))[1]
        mb = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) / 1e3
        hit = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        L.append("| %s | %d | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (k, r["n"], r["avg"], r["mn"], r["mx"], ("%.0f" % mb) if cc else "-",
                ("%.2f" % (mb / r["avg"])) if cc and r["avg"] else "-",
                                                                 ("%.2f M" % (g("TCC_REQ_sum") / 1e6)) if cc else "-", ("%.2f" % hit) if cc else "-"))
        tot += r["total"] / max(1, r["n"])
        if cc:
            STEP_BYTES[tag] = STEP_BYTES.get(tag, 0.0) + mb * 1e6 * r["n"] / 20.0
    L.append("\nsum of the kernels' average durations per step: %.0f us\n" % tot)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    L = ["# %s: the stress" % rnd + " configuration (BASELINE configs[4]: hash T = 2^22, 105 M parameters = 211 MB fp16 per object) -- rocprofv3 windows on one MI355X\n",
         "`tools/gpu_profile_window.sh <tag> <extra>` with `WINDOW_ARGS=\"--log2-hashmap-size 22 [--objects 8]\"`: kernel trace of steps extra+5 .. extra+25 of",
         "`tools/profile_window.py`, then one `--pmc` pass per counter set.  \"traffic beyond L2\" = (2 x FETCH_SIZE + WRITE_SIZE) KB of the committed pass: what the L2s",
         "fetched from / wrote to the fabric -- Infinity Cache (256 MB, memory side) or HBM; the counters do not tell the two apart, the RATE does: a kernel whose traffic",
         "moves faster than HBM streams (6-7.3 TB/s for a two-stream copy, `profiles/r02_copybench.md`; 8 TB/s peak) is served partly by the Infinity Cache.\n"]
    section(rnd + "_T22_init", "one object, steps 20..40 from init (60 % of the 13.2 M parameter chunks carry a gradient)", L)
    section(rnd + "_T22_late", "one object, steps 800..820 (a few thousand gradient-carrying samples per step)", L)
    section(rnd + "_T22_init8", "eight objects on one GPU (a thread and a stream each), steps 20..40 of every object", L)
    section(rnd + "_T22_late8", "eight objects on one GPU, steps 800..820 (kernel trace only)", L)
    notes = os.path.join(ROOT, "profiles", rnd + "_window_T22_notes.md")
    if os.path.exists(notes):
        L.append(open(notes).read())
    open(os.path.join(ROOT, "profiles", rnd + "_window_T22.md"), "w").write("\n".join(L) + "\n")
    # the step's counter traffic for bench.py's stress_T22 line, with the fingerprint of the kernel sources it was measured on (tools/fingerprint.py)
    if (rnd + "_T22_init") in STEP_BYTES and (rnd + "_T22_late") in STEP_BYTES:
        import json
        from fingerprint import kernel_sources_sha16
        pj_path = os.path.join(ROOT, "profiles", "pmc_traffic.json"); pj = json.load(open(pj_path))
        pj["stress_T22"] = {"source": "profiles/%s_window_T22.md (gpurun %s_T22_init / %s_T22_late)" % (rnd, rnd, rnd), "kernel_sources_sha16": kernel_sources_sha16(),
                            "step_bytes_beyond_l2_steps_20_40": int(STEP_BYTES[rnd + "_T22_init"]), "step_bytes_beyond_l2_steps_800_820": int(STEP_BYTES[rnd + "_T22_late"])}
        json.dump(pj, open(pj_path, "w"), indent=1)
    print("\n".join(L))


if __name__ == "__main__":
    main()
