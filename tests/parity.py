"""Shared comparison helpers for the parity tests (tolerances are stated here once).

Bars (restated): integer / index work bit-exact (n_valid, flags, hash indices); geometry
fp32 exact up to 1e-6 abs (explicit fmaf chains on both sides); fp16 activations within 2 fp16 ulp of the
value scale (MFMA / libm summation order); compositing 2e-4 abs (expf implementations differ); grid gradient
within the fp16-accumulation bound 2^-9 * sum|contributions|; parameters after one step: fraction of entries
off by more than 1e-4 below 0.5 % (Adam's first step is +-lr * sign(g), so only near-zero gradients differ)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CFGS, SCENE, grid_probe_indices, pattern_params  # noqa: E402,F401


def h2f(a):
    return np.asarray(a, np.uint16).view(np.float16).astype(np.float32)


def load_golden(name):
    return dict(np.load(os.path.join(HERE, "golden", name + ".npz")))


def close_half(got_u16, want_u16, what, ulps=2.0, scale=None, frac_ok=1.0):
    g, w = h2f(got_u16).astype(np.float64), h2f(want_u16).astype(np.float64)
    sc = max(np.abs(w).max(), 1e-12) if scale is None else scale
    tol = ulps * 2.0 ** -10 * np.maximum(np.abs(w), sc * 2.0 ** -6) + 1e-7
    bad = np.abs(g - w) > tol
    assert bad.mean() <= 1.0 - frac_ok, "%s: %.4f%% outside tolerance, max err %.3e (scale %.3e)" % (what, 100 * bad.mean(), np.abs(g - w).max(), sc)
    return float((g == w).mean())


def close_f32(got, want, what, atol, rtol=0.0):
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(g - w); tol = atol + rtol * np.abs(w)
    assert (err <= tol).all(), "%s: max err %.3e at %d (tol %.3e)" % (what, err.max(), int(err.argmax()),
            float(np.ravel(tol)[err.argmax()] if np.ndim(tol) else tol))
    return float((g == w).mean())


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)
