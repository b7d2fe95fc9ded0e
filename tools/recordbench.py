#!/usr/bin/env python
"""Record traffic of the large-table optimizer without its arithmetic (microbench mode 33): read + write of every touched 128-byte chunk record, a lane per
record (the shipped shape) against 8 / 4 lanes per record, at T = 2^22's 13.2 M records for several touched fractions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = (3072 + 2 * 52727808) // 8
for pct in (60, 25, 5, 100):
    for variant, name in ((0, "lane per record"), (1, "8 lanes per record"), (2, "4 lanes per record")):
        ns = pkg.microbench(33, pct, variant, N)          # mode 33 returns best_ms * 1e6 / touched records = nanoseconds per record
        print("touched %3d %%  %-20s %.4f ns per record  = %.2f TB/s of record traffic (256 B per record)" % (pct, name, ns, 256.0 / ns / 1e3), flush=True)
