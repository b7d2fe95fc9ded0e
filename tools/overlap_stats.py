#!/usr/bin/env python
"""Overlap of kernels in a rocprofv3 --kernel-trace database (rocpd): per kernel the mean duration, and for the whole trace window the busy time
(union of kernel intervals), the sum of kernel durations and their ratio (1.0 = nothing ever overlapped).   python tools/overlap_stats.py <results.db>
[skip_first_ms]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = [(re.sub(r"\(.*", "", n).replace("void ", "")[:44], s, e) for n, s, e in rows if "k_fused_train" in n or "k_grid_scatter" in n or "k_optimizer" in n]
t0 = rows[0][1] + skip; rows = [r for r in rows if r[1] >= t0]
per = {}
for n, s, e in rows:
    per.setdefault(n, []).append((e - s) / 1e3)
busy = 0.0; cs, ce = rows[0][1], rows[0][2]
for _, s, e in rows[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for _, s, e in rows)
for n, v in sorted(per.items()):
    print("%-46s calls %6d   mean %7.2f us" % (n, len(v), sum(v) / len(v)))
print("window %.2f ms   busy (union) %.2f ms   sum of kernel durations %.2f ms   overlap factor %.2f   idle %.1f %%" % ((rows[-1][2] - rows[0][1]) / 1e6,
        busy / 1e6, tot / 1e6, tot / busy, 100.0 * (1.0 - busy / (rows[-1][2] - rows[0][1]))))
