"""One fingerprint of the kernel sources (ro-map_amd/csrc/*.hip, *.h, *.cpp + build.sh) with comments and white space removed: written into
profiles/pmc_traffic.json when a profile window is committed, recomputed by bench.py at run time -- a kernel change without a re-profile then shows up as
`traffic_stale` in the bench line instead of silently keeping the old traffic and rocprofv3 durations (editing a comment does not)."""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("microbench.hip", "diag.cpp", "diag_kernels.hip")          # diagnostics and micro-benchmarks are not on the product path

_TOKEN = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)


def strip(text):
    """C / C++ source without comments (string and character literals kept), white space collapsed."""
    out = _TOKEN.sub(lambda m: " " if m.group(0).startswith("/") else m.group(0), text)
    return re.sub(r"\s+", " ", out).strip()


def _hash(read):
    d = os.path.join(ROOT, "ro-map_amd", "csrc")
    names = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp")) and f not in SKIP)
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode()); h.update(strip(read("ro-map_amd/csrc/" + n)).encode())
    # build.sh: its compiler flags are part of what the numbers were measured on ('#' comments dropped)
    h.update(re.sub(r"\s+", " ", re.sub(r"(?m)^\s*#.*$", "", read("ro-map_amd/build.sh"))).encode())
    return h.hexdigest()[:16]


def kernel_sources_sha16():
    return _hash(lambda rel: open(os.path.join(ROOT, rel)).read())


def kernel_sources_sha16_at(rev):
    """The same fingerprint of a committed tree (checks that only comments changed since a profile was taken)."""
    return _hash(lambda rel: subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (rev, rel)], capture_output=True, text=True, check=True).stdout)


if __name__ == "__main__":
    print(kernel_sources_sha16_at(sys.argv[1]) if len(sys.argv) > 1 else kernel_sources_sha16())
