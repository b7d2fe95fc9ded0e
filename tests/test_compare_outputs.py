"""tools/compare_with_reference_outputs.py (the diff of a CUDA-reference run's output directory against ours) on synthetic directories in
the reference's layout (nerf.cu:255-349): identical runs pass, a slightly noisy run passes, a visibly different one and a missing image do not."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _write(root, oid, rng, noise, shift=0.0, drop=None):
    from PIL import Image
    d = os.path.join(root, str(oid))
    for sub in ("test_img", "test_depth", "test_mask"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    yy, xx = np.mgrid[0:48, 0:64]
    for k in range(4):
        st = "%.6f" % (k * 0.1)
        if drop == k:
            continue
        m = ((xx - 32 - shift) ** 2 + (yy - 24) ** 2) < (14 + k) ** 2
        img = np.where(m[..., None], np.stack([0.3 + 0.01 * xx, 0.5 + 0 * xx, 0.8 - 0.01 * yy], -1), 1.0) + noise * rng.standard_normal((48, 64, 3))
        dep = np.where(m, 1.5 + 0.002 * xx, 0.0) * (1.0 + noise * 0.1)
        Image.fromarray((np.clip(img, 0, 1) * 255).astype(np.uint8)).save(os.path.join(d, "test_img", st + ".png"))
        Image.fromarray((dep * 20000).astype(np.uint16)).save(os.path.join(d, "test_depth", st + ".png"))
        Image.fromarray((m * 255).astype(np.uint8)).save(os.path.join(d, "test_mask", st + ".png"))
    v = rng.uniform(-0.2, 0.2, (300, 3)) if noise == 0 else None
    return d, v


def _ply(path, v):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face 0\n"
                "property list uchar int vertex_indices\nend_header\n" % (len(v) + 2))
        for p in v:
            f.write("%0.5f %0.5f %0.5f\n" % tuple(p))
        f.write("0.00000 0.00000 0.00000\n0.00000 0.00000 0.00000\n")              # the reference's zero padding


def test_compare_script_on_synthetic_directories(tmp_path):
    pytest.importorskip("PIL")
    rs = np.random.RandomState(0)
    ref = str(tmp_path / "ref"); same = str(tmp_path / "same"); close = str(tmp_path / "close"); far = str(tmp_path / "far"); short = str(tmp_path / "short")
    d, v = _write(ref, 0, rs, 0.0); _ply(os.path.join(d, "obj.ply"), v)
    d2, _ = _write(same, 0, rs, 0.0); _ply(os.path.join(d2, "obj.ply"), v)
    d3, _ = _write(close, 0, rs, 0.004); _ply(os.path.join(d3, "obj.ply"), v + rs.normal(0, 5e-4, v.shape))
    d4, _ = _write(far, 0, rs, 0.0, shift=3.0); _ply(os.path.join(d4, "obj.ply"), v * 1.2)
    _write(short, 0, rs, 0.0, drop=2)
    run = lambda a, b: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_with_reference_outputs.py"), a, b], capture_output=True, text=True)
    r = run(ref, same); assert r.returncode == 0 and "inside tolerance" in r.stdout, r.stdout + r.stderr
    r = run(ref, close); assert r.returncode == 0, r.stdout + r.stderr
    r = run(ref, far); assert r.returncode == 1 and "OUTSIDE" in r.stdout, r.stdout + r.stderr
    r = run(ref, short); assert r.returncode == 1 and "missing" in r.stdout, r.stdout + r.stderr
    r = run(ref, str(tmp_path)); assert r.returncode == 2
