"""One fingerprint of the kernel sources (ro-map_amd/csrc/*.hip, *.h, *.cpp + build.sh): written into profiles/pmc_traffic.json when a profile window is
committed, recomputed by bench.py at run time -- a kernel change without a re-profile then shows up as `traffic_stale` in the bench line instead of silently
keeping the old traffic and rocprofv3 durations."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16():
    d = os.path.join(ROOT, "ro-map_amd", "csrc")
    # (diagnostics and micro-benchmarks are not on the product path)
    names = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp")) and f not in ("microbench.hip", "diag.cpp", "diag_kernels.hip"))
    h = hashlib.sha256()
    for n in names + ["../build.sh"]:
        h.update(n.encode()); h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sources_sha16())
