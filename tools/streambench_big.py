"""The optimizer's memory streams without its arithmetic (microbench mode 31) at the T = 2^22 stress size: what a dense sweep of 105 M parameters' state can reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = 3072 + 2 * 52727808
for units in (1, 2, 4, 8):
    for plain in (0, 1):
        for blocks in (1024, 2048, 4096, 8192):
            ms = pkg.microbench(31, blocks, plain | (units << 4), n)
            mb = n * (24 + 8 + 4) / 1e6
            print("units %d  %s stores  %5d blocks: %.1f us  %.0f MB  %.2f TB/s" % (units, "plain" if plain else "nt   ", blocks, 1e3 * ms, mb, mb / ms / 1e6),
                    flush=True)
