import sys; sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
for mb in (20, 40, 80):
    for blocks in (512, 1024, 2048, 4096):
        ms = pkg.microbench(30, blocks, 1, mb * 1000000)
        print("copy %3d MB (traffic %3d MB)  %4d blocks  %.1f us  %.2f TB/s" % (mb, 2 * mb, blocks, 1e3 * ms, 2 * mb * 1e6 / (ms * 1e-3) / 1e12), flush=True)
