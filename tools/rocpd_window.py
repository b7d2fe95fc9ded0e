#!/usr/bin/env python
"""Per-kernel summary of rocprofv3 rocpd databases over a WINDOW of dispatches: for every kernel name, its dispatches are put in
dispatch order and numbers [skip, skip + take) are averaged -- with tools/profile_window.py as the profiled command that is exactly
the window `bench.py --steps K --warmup W` times.

   python tools/rocpd_window.py <dir containing *_results.db (searched recursively)> [--skip 5] [--take 20] [--only k_fused,k_grid,k_opt]
Prints a markdown table: kernel-trace durations (if the db holds a kernel trace) and every PMC counter found."""
import argparse
import glob
import os
import re
import sqlite3


def short(name):
    s = re.sub(r"\(.*", "", name); s = re.sub(r"^void ", "", s); s = re.sub(r"^mon::", "", s)
    if s.startswith("_Z"):                      # still mangled: keep the kernel's own identifier
        m = re.search(r"\d+(k_[a-z0-9_]+?)(?=E|I)", s)
        if m:
            s = m.group(1)
    # k_fused_train's instantiations take turns within one run (with the occupancy grid: <.., OCC = false, ..> during the warm-up, <.., true, ..> after it):
    # one dispatch sequence, so that window [skip, skip + take) means the same iterations as for the other kernels
    if s.startswith("k_fused_train"):
        s = "k_fused_train"
    # (likewise k_optimizer<.., LIVE>: the position blocks compact live samples once the grid is in use)
    m = re.match(r"k_optimizer<(true|false), (true|false), (true|false), (true|false)>", s)
    if m:
        s = "k_optimizer<%s, %s, %s>" % m.group(1, 2, 3)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root"); ap.add_argument("--skip", type=int, default=5); ap.add_argument("--take", type=int, default=20)
    ap.add_argument("--only", default="k_encode_tiles,k_sample_points,k_fused_train,k_grid_scatter,k_optimizer,k_big,k_reduce,k_cand,k_step")
    a = ap.parse_args()
    only = [s for s in a.only.split(",") if s]
    keep = lambda n: (not only) or any(o in n for o in only)
    dbs = sorted(glob.glob(os.path.join(a.root, "**", "*_results.db"), recursive=True))
    if os.path.isfile(a.root):
        dbs = [a.root]
    dur = {}; res = {}; ctr = {}
    for p in dbs:
        db = sqlite3.connect(p); cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "kernels" in tabs:
            per = {}
            for name, did, s, e, vg, ag, lds, gx, wx in cur.execute("select name, dispatch_id, start, end, vgpr_count, accum_vgpr_count, lds_size, grid_x, workgroup_x from kernels order by dispatch_id"):
                n = short(name)
                if keep(n):
                    per.setdefault(n, []).append((e - s) / 1e3); res[n] = (vg, ag, lds, gx // max(1, wx), wx)
            has_ctr = "counters_collection" in tabs and cur.execute("select count(*) from counters_collection").fetchone()[0] > 0
            if not has_ctr:                      # durations of a counter pass are perturbed: only the plain kernel trace counts
                for n, v in per.items():
                    w = v[a.skip:a.skip + a.take]
                    if w:
                        dur[n] = (len(w), sum(w) / len(w), min(w), max(w), len(v))
        if "counters_collection" in tabs:
            per = {}
            for name, did, cn, val in cur.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection order by dispatch_id"):
                n = short(name)
                if keep(n):
                    per.setdefault((n, cn), {}).setdefault(did, 0.0)
                    per[(n, cn)][did] += float(val)          # one row per XCD / instance dimension: summed per dispatch
            for (n, cn), d in per.items():
                v = [d[k] for k in sorted(d)][a.skip:a.skip + a.take]
                if v:
                    ctr[(n, cn)] = (len(v), sum(v) / len(v))
    print("window: dispatches [%d, %d) of every kernel\n" % (a.skip, a.skip + a.take))
    if dur:
        # (rocprofv3's VGPR columns count in units of two registers on gfx950: 104 here = 207 in the code object's .vgpr_count)
        print("| kernel | dispatches in window (of) | avg us | min us | max us | total us | arch VGPR | acc VGPR | LDS B | workgroups x threads |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        for n, (c, avg, mn, mx, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
            vg, ag, lds, g, w = res[n]
            print("| %s | %d (%d) | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %d x %d |" % (n[:70], c, tot, avg, mn, mx, avg * c, vg, ag, lds, g, w))
        print()
    if ctr:
        print("| kernel | counter | dispatches | mean per dispatch |")
        print("|---|---|---|---|")
        for (n, cn), (c, m) in sorted(ctr.items()):
            print("| %s | %s | %d | %.1f |" % (n[:70], cn, c, m))


if __name__ == "__main__":
    main()
