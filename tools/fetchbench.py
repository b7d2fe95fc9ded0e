#!/usr/bin/env python
"""Fetch granularity of random 4-byte reads under every cache policy gfx950 offers (microbench modes 80-82): 2^26 reads of pseudo-random
words of a table beyond the L2s (256 MB: about the T = 2^22 table, inside the Infinity Cache; 1 GiB: beyond it).   python tools/fetchbench.py [--quick]
A rocprofv3 --pmc pass of `--quick` (one table size) lists the instantiations as separate kernels: TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum per dispatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
OPS = 1 << 26
variants = [(80, 0, "buffer_load_dword"), (80, 1, "buffer_load_dword sc0"), (80, 2, "buffer_load_dword nt"), (80, 3, "buffer_load_dword sc0 nt"),
            (80, 16, "buffer_load_dword sc1"), (80, 17, "buffer_load_dword sc0 sc1"), (80, 18, "buffer_load_dword nt sc1"), (80, 19, "buffer_load_dword sc0 nt sc1"),
            (81, 0, "buffer_load_dword lds"), (81, 2, "buffer_load_dword lds nt"), (81, 16, "buffer_load_dword lds sc1"), (81, 17, "buffer_load_dword lds sc0 sc1"),
            (82, 0, "global_load_dword"), (82, 2, "global_load_dword nt")]
sizes = [(1 << 26, "256 MB")] if "--quick" in sys.argv else [(1 << 26, "256 MB"), (1 << 28, "1 GiB"), (1 << 22, "16 MB")]
for n, nm in sizes:
    for mode, pat, name in variants:
        ms = pkg.microbench(mode, pat, n, OPS)
        print("table %-7s %-30s %8.3f ms  %6.1f G reads/s  (= %5.2f TB/s at 64 B, %5.2f TB/s at 128 B per read)" % (nm, name, ms, OPS / ms / 1e6, OPS * 64 / ms / 1e9,
                OPS * 128 / ms / 1e9), flush=True)
