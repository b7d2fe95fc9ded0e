"""What the optimizer's memory streams cost without its arithmetic (microbench modes 31 / 32, ro-map_amd/csrc/microbench.hip)."""
import sys; sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
n = 1911808          # base.json: 3072 + 2 * 954368 parameters
for parts in (0, 4):
    for mode, per in ((31, 8), (32, 4)):
        for units in (1, 2, 4, 8):
            for plain in (0, 1):
                best = None
                for blocks in (256, 512, 1024, 2048):
                    if blocks * 256 * units * per < n // 2: continue          # less than half of the array covered per pass: more passes, same thing
                    ms = pkg.microbench(mode, blocks, plain | (units << 4) | (parts << 8), n)
                    if best is None or ms < best[0]: best = (ms, blocks)
                mb = n * (24 + 8 + 4 + 2 * parts) / 1e6
                print("mode %d  units %d  %s stores  partial tables %d: %.1f us (%d blocks)  %.0f MB  %.2f TB/s" % (mode, units, "plain" if plain else "nt   ",
                        parts, 1e3 * best[0], best[1], mb, mb / best[0] / 1e6), flush=True)
