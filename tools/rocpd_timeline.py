#!/usr/bin/env python
"""Start / end of every dispatch of a few consecutive training steps in a rocprofv3 --kernel-trace database (rocpd), relative to the first one:
shows whether kernels on different streams ran side by side.   python tools/rocpd_timeline.py <dir-or-db> [first_step] [n_steps]"""
import glob
import os
import re
import sqlite3
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cur = sqlite3.connect(p).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sc = "stream_id" if "stream_id" in cols else "0"; qc = "queue_id" if "queue_id" in cols else "0"
rows = sorted(cur.execute("select name, start, end, %s, %s from kernels" % (sc, qc)).fetchall(), key=lambda r: r[1])
short = lambda s: re.sub(r"^void (mon::)?", "", re.sub(r"[<(].*", "", s))[:20]
opt = [i for i, r in enumerate(rows) if "k_fused_train" in r[0]]
i0, i1 = opt[first], opt[first + n]
t0 = rows[i0][1]
print("| kernel | stream | queue | start us | end us | dur us |\n|---|---|---|---|---|---|")
for nme, s, e, st, q in rows[i0:i1]:
    print("| %s | %s | %s | %.2f | %.2f | %.2f |" % (short(nme), st, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print("\n%d steps in %.2f us = %.2f us per step" % (n, (rows[i1][1] - t0) / 1e3, (rows[i1][1] - t0) / 1e3 / n))
