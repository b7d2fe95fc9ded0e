"""Gather line-sharing probes on the GPU box (16.8 M half2 gathers = one base.json training step's forward)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = 954368; OPS = 16 * 1024 * 1024
names = {3: "every lane its own line", 9: "same, buffer_load", 7: "one lane: 2 consecutive loads share a line", 8: "lanes l, l^32 share a line",
        20: "4 lanes (l, l^16, l^32, l^48) share a line", 21: "adjacent lanes l, l^1 share a line", 22: "lanes l, l^16 share a line",
        23: "4 adjacent lanes share a line", 6: "8-byte pair gather", 5: "16-byte quad gather"}
for pattern in (0, 1):
    for mode in (3, 9, 7, 8, 21, 22, 20, 23, 6, 5):
        ms = pkg.microbench(mode, pattern, N, OPS)
        print("pattern %d  mode %2d  %-48s %8.3f ms  %7.1f G gathers/s  %5.2f lanes/clk/CU @2.4GHz" % (pattern, mode, names[mode], ms, OPS / ms / 1e6,
                OPS / ms / 1e6 / 256 / 2.4), flush=True)

for mode, nm in {17: "ds_read_b32 random (LCG)", 19: "ds_read_b64 random (LCG)", 18: "ds_add_u32 random (LCG)", 12: "ds_add_u32 random (hash index)",
        14: "ds_write_b32 random (hash index)"}.items():
    for ops in (OPS, 8 * OPS):
        ms = pkg.microbench(mode, 0, N, ops)
        print("LDS 128KB tile x 256 WGs x 1024 thr  mode %d  %-34s %9d ops %8.3f ms  %6.2f lanes/clk/CU @2.4GHz" % (mode, nm, ops, ms,
                ops / ms / 1e6 / 256 / 2.4), flush=True)
