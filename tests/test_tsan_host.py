"""ThreadSanitizer over the host side: every source of libmon_core.so compiled --offload-host-only with -fsanitize=thread, linked against a
stand-in HIP runtime (tests/tsan/hip_stub.cpp: device memory = host memory, streams complete at once, launches do nothing) and driven through the C ABI by
tests/tsan/tsan_driver.cpp -- five objects trained from their own threads through the training lanes while the lane count flips, a viewer rendering from the
published snapshots, objects created and destroyed meanwhile, then the online manager's whole protocol with a viewer.  No GPU.  The run must finish and TSAN
must stay silent (the first run of this build found the unguarded object list of the online manager and the plain `long` options)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_side_is_clean_under_threadsanitizer(tmp_path):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "tsan", "build_and_run.sh"), str(tmp_path / "tsan")], capture_output=True, text=True, timeout=900)
    log = (tmp_path / "tsan" / "tsan.log").read_text() if (tmp_path / "tsan" / "tsan.log").exists() else ""
    assert r.returncode == 0 and "tsan driver finished" in r.stdout and "tsan reports: 0" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:] + log[-3000:]
