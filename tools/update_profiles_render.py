#!/usr/bin/env python
"""profiles/r04_window_render.md from the before / after runs of tools/gpu_render_window.sh (gpurun_out/<before>, gpurun_out/<after>):
   python tools/update_profiles_render.py r04_render_before r04_render_after"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path):
    rows = {}
    for l in open(path):
        c = [x.strip() for x in l.strip().strip("|").split("|")]
        if len(c) >= 6 and re.match(r"\d+ \(\d+\)", c[1]):
            rows[c[0]] = dict(n=int(c[1].split()[0]), avg=float(c[2]), mn=float(c[3]), mx=float(c[4]), total=float(c[5]), rest=c[6:])
    return rows


def counters(path):
    out = {}
    for l in open(path):
        c = [x.strip() for x in l.strip().strip("|").split("|")]
        if len(c) == 4 and c[2].isdigit():
            out.setdefault(c[0], {})[c[1]] = (int(c[2]), float(c[3]))
    return out


def main():
    before, after = sys.argv[1], sys.argv[2]
    B, A = (os.path.join(ROOT, "gpurun_out", d) for d in (before, after))
    tb, ta = table(os.path.join(B, "kernel_window.md")), table(os.path.join(A, "kernel_window.md"))
    cb, ca = counters(os.path.join(B, "pmc_window.md")), counters(os.path.join(A, "pmc_window.md"))
    crops = 71                                   # 11 crop renders (1 + 10) + 60 orbit views in the traced command
    rk_b = [k for k in tb if k.startswith("k_fused_render") or k == "k_render_rays"]
    rk_a = [k for k in ta if k.startswith("k_tile_render") or k in ("k_encode_feat", "k_render_points", "k_render_rays_jobs")]
    sum_b = sum(tb[k]["total"] for k in rk_b); sum_a = sum(ta[k]["total"] for k in rk_a)
    L = []
    L.append("# r04: the render half of the metric -- rocprofv3 window of `tools/render_window.py` (313x229 crop x 11, the 60-view 240x320 orbit, 64^3 mesh x 4) on one MI355X\n")
    L.append("Command: `tools/gpu_render_window.sh <tag>` = one `rocprofv3 --kernel-trace --stats` run + one `--pmc` pass per counter set (kernel-trace only) of")
    L.append("`python tools/render_window.py --train 300 --crops 10 --orbit 60 --meshes 3`; base.json object on the bench scene.  **before** = the tree at the start of")
    L.append("round 4 (`k_fused_render`: 128 four-byte gathers per sample from the L2-resident table), **after** = level tiles in LDS (`kernels_tilerender.hip`).")
    L.append("Images are bit-identical (`crc rgb 4f9972c4 depth 9d47789c mask 04b4aae6` in both runs; `tests/test_tile_render.py`).\n")
    L.append("## wall clock of the untraced run (`plain.log`)\n")
    L.append("| | before | after |\n|---|---|---|")
    pb, pa = open(os.path.join(B, "plain.log")).read().splitlines(), open(os.path.join(A, "plain.log")).read().splitlines()
    for lb, la in zip(pb, pa):
        if ":" in lb and not lb.startswith("crc"):
            L.append("| %s | %s | %s |" % (lb.split(":")[0], lb.split(":", 1)[1].strip(), la.split(":", 1)[1].strip()))
    L.append("")
    L.append("## render kernels per crop (71 crops of ~72-77 k pixels in the traced command)\n")
    L.append("| | kernels | dispatches | total us | us per crop |\n|---|---|---|---|---|")
    L.append("| before | %s | %d | %.0f | **%.1f** |" % (" + ".join(rk_b), sum(tb[k]["n"] for k in rk_b), sum_b, sum_b / crops))
    L.append("| after | %s | %d | %.0f | **%.1f** |" % (" + ".join(rk_a), sum(ta[k]["n"] for k in rk_a), sum_a, sum_a / crops))
    L.append("\nratio before / after: **%.2fx** of render-kernel time per crop.\n" % (sum_b / sum_a))
    for name, t in (("before", tb), ("after", ta)):
        L.append("### %s: every kernel of the window\n" % name)
        L.append("| kernel | dispatches | avg us | min us | max us | total us | arch VGPR | LDS B | workgroups x threads |\n|---|---|---|---|---|---|---|---|---|")
        for k, r in sorted(t.items(), key=lambda kv: -kv[1]["total"]):
            if k.startswith("k_encode_tiles") or k.startswith("k_copy_from_host"):
                continue                          # the 300 training steps before the renders / the dataset upload
            L.append("| %s | %d | %.2f | %.2f | %.2f | %.1f | %s | %s | %s |" % (k, r["n"], r["avg"], r["mn"], r["mx"], r["total"], r["rest"][0], r["rest"][2],
                    r["rest"][3]))
        L.append("")
    L.append("## counters (means per dispatch; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB on gfx950)\n")
    L.append("| kernel | HBM MB / dispatch | L2 requests | L2 hit share | VALU wave-insts | LDS wave-insts | MFMA wave-insts | LDS conflict share | wave time waiting |\n|---|---|---|---|---|---|---|---|---|")
    for src, names in ((cb, [k for k in cb if k.startswith("k_fused_render")]), (ca,
            ["k_encode_feat"] + [k for k in ca if k.startswith("k_tile_render")] + ["k_render_points", "k_render_rays_jobs"])):
        for k in names:
            c = src.get(k, {})
            g = lambda n: c.get(n, (0, 0.0))[1]
            hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) / 1e3
            hit = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
            conf = g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))
            wait = g("SQ_WAIT_INST_ANY") / max(1.0, g("SQ_WAVE_CYCLES"))
            L.append("| %s | %.1f | %.2f M | %.2f | %.2f M | %.2f M | %.3f M | %.2f | %.2f |" % (k, hbm, g("TCC_REQ_sum") / 1e6, hit, g("SQ_INSTS_VALU") / 1e6,
                    g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_MFMA") / 1e6, conf, wait))
    L.append("")
    L.append(open(os.path.join(ROOT, "profiles", "r04_window_render_notes.md")).read() if os.path.exists(os.path.join(ROOT, "profiles",
            "r04_window_render_notes.md")) else "")
    open(os.path.join(ROOT, "profiles", "r04_window_render.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L[:40]))


if __name__ == "__main__":
    main()
