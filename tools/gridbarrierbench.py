"""What a persistent single-object step would trade: grid barriers inside one resident grid of 256 x 1024 threads with the CU's whole LDS
against back-to-back launches of the same grid (microbench modes 70 / 71)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
for wgs in (256, 128):
    for n in (100, 1000):
        a = min(pkg.microbench(70, wgs, 1024, n) for _ in range(3)); b = min(pkg.microbench(71, wgs, 1024, n) for _ in range(3))
        c = min(pkg.microbench(72, wgs, 1024, n) for _ in range(3))
        print("%3d workgroups x 1024 threads, 160 KB LDS each: %4d phases -- resident grid + flat grid barriers %.2f us per phase, XCD-hierarchical barriers %.2f us per phase, separate launches %.2f us per phase" % (wgs, n, 1e3 * a / n, 1e3 * c / n, 1e3 * b / n), flush=True)

# VERDICT r05 item 4's probe: a per-LEVEL barrier (16 workgroups of one XCD) with a 64 KB write -> read exchange per workgroup and phase (modes 73 / 74);
# negative = the run is invalid (-1: a reader saw stale data, -2: the dispatcher did not give every XCD 32 workgroups)
for n in (100, 1000):
    a = [pkg.microbench(73, 0, 1024, n) for _ in range(3)]; b = [pkg.microbench(74, 0, 1024, n) for _ in range(3)]
    print("level barrier, 16 groups of 16 workgroups, 64 KB out + 64 KB in per workgroup and phase, %4d phases: agent-scope semantics %s us per phase, same-XCD semantics "
          "(relaxed arrive, L1 invalidate only) %s us per phase" % (n, " ".join("%.2f" % (1e3 * t / n) for t in a), " ".join("%.2f" % (1e3 * t / n) for t in b)), flush=True)
