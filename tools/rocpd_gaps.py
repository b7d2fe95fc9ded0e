#!/usr/bin/env python
"""Idle time between consecutive kernels of the training loop in a rocprofv3 rocpd trace:
   python tools/rocpd_gaps.py <trace_results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted(cur.execute("select %s, start, end from kernels" % name_col).fetchall(), key=lambda r: r[1])
short = lambda n: re.sub(r"^void ", "", re.sub(r"[<(].*", "", n)).replace("_ZN3mon14", "")[:24]
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    k = (short(n0), short(n1)); g = (s1 - e0) / 1e3
    if g < 200:                      # skip host-side pauses
        a = gaps.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += g
print("| after | before | count | mean gap us |\n|---|---|---|---|")
for k, a in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:8]:
    print("| %s | %s | %d | %.2f |" % (k[0], k[1], a[0], a[1] / a[0]))
