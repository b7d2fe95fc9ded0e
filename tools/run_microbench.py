"""Scatter/gather micro-benchmarks on the GPU box (16.8 M operations = one base.json training step)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = 954368; OPS = 16 * 1024 * 1024
names = {5: "16-byte quad gather", 6: "8-byte pair gather", 0: "pk_f16 atomic, shared table", 1: "pk_f16 atomic, per-XCD private tables",
        2: "f32 atomic, shared table", 3: "half2 gather", 4: "pk_f16 atomic, line-local bursts"}
for pattern in (0, 1):
    for mode in (3, 6, 5, 0, 1, 2, 4):
        ms = pkg.microbench(mode, pattern, N, OPS)
        print("pattern %d  mode %d  %-40s %8.3f ms  %7.1f Gop/s" % (pattern, mode, names[mode], ms, OPS / ms / 1e6), flush=True)
for n in (4096, 65536, 1 << 20, 1 << 24):
    for mode in (3, 0, 1):
        ms = pkg.microbench(mode, 0, n, OPS)
        print("uniform over %9d entries  mode %d  %-40s %8.3f ms  %7.1f Gop/s" % (n, mode, names[mode], ms, OPS / ms / 1e6), flush=True)

lds = {16: "ds_add_u64 random", 10: "ds_pk_add_f16 random", 11: "ds_add_f32 random", 12: "ds_add_u32 random", 13: "ds_pk_add_f16 lane pairs share",
        15: "ds_pk_add_f16 4 lanes share", 14: "ds_write_b32 random"}
for mode, nm in lds.items():
    ms = pkg.microbench(mode, 0, N, OPS)
    print("LDS 128KB tile x 256 WGs  mode %d  %-34s %8.3f ms  %8.1f Gop/s (chip)  %6.2f op/clk/CU @2.1GHz" % (mode, nm, ms, OPS / ms / 1e6,
            OPS / ms / 1e6 / 256 / 2.1), flush=True)
