#!/usr/bin/env python
"""Per-kernel averages of the PMC counters collected by tools/gpu_profile_round.sh / tools/gpu_pmc_one.sh (rocprofv3 rocpd SQLite output)."""
import glob
import os
import re
import sqlite3
import sys


def main():
    root = sys.argv[1]
    rows = {}
    for db_path in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        db = sqlite3.connect(db_path); cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            continue
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        ncol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
        for k, n, v in cur.execute("select %s, %s, %s from counters_collection" % (kcol, ncol, vcol)):
            k = re.sub(r"\(.*", "", k).replace("void ", "")
            a = rows.setdefault((k, n), [0, 0.0]); a[0] += 1; a[1] += float(v)
    print("| kernel | counter | dispatches | mean per dispatch |")
    print("|---|---|---|---|")
    for (k, n), a in sorted(rows.items()):
        print("| %s | %s | %d | %.1f |" % (k[:60], n, a[0], a[1] / a[0]))


if __name__ == "__main__":
    main()
