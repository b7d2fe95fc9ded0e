"""Scatter/gather micro-benchmarks on the GPU box (16.8 M operations = one base.json training step)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = 954368; OPS = 16 * 1024 * 1024
names = {0: "pk_f16 atomic, shared table", 1: "pk_f16 atomic, per-XCD private tables", 2: "f32 atomic, shared table", 3: "half2 gather", 4: "pk_f16 atomic, line-local bursts"}
for pattern in (0, 1):
    for mode in (3, 0, 1, 2, 4):
        ms = pkg.microbench(mode, pattern, N, OPS)
        print("pattern %d  mode %d  %-40s %8.3f ms  %7.1f Gop/s" % (pattern, mode, names[mode], ms, OPS / ms / 1e6), flush=True)
for n in (4096, 65536, 1 << 20, 1 << 24):
    for mode in (3, 0, 1):
        ms = pkg.microbench(mode, 0, n, OPS)
        print("uniform over %9d entries  mode %d  %-40s %8.3f ms  %7.1f Gop/s" % (n, mode, names[mode], ms, OPS / ms / 1e6), flush=True)
