"""The XORWOW sample stream of the "same inputs" mode (orc_config.rng_flags / mon_config.rng_flags): the oracle's own implementation (seeding, next, the 2^67
subsequence jump by matrix squaring, host-API ordering) against known answers from rocRAND's engine (tests/golden/xorwow_rocrand.json, generator next to it),
and the structure of the host ordering.  CPU only; the GPU side (rocRAND's HOST generator on the device, the HIP path's fill kernel) is in
tests/test_xorwow_gpu.py."""
import ctypes as C
import json
import os

import numpy as np

from conftest import ROOT

FIX = os.path.join(ROOT, "tests", "golden", "xorwow_rocrand.json")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_engine_matches_rocrand_known_answers(orc):
    cases = json.load(open(FIX))["cases"]
    assert len(cases) >= 20
    for c in cases:
        out = np.zeros(8, np.uint32)
        orc.lib().orc_xorwow_lane_draws(C.c_uint64(c["seed"]), 1, c["subsequence"], 8, _p(out))
        assert out.tolist() == c["draws"], (c["seed"], c["subsequence"])
        # the (0, 1] map of the rocRAND flavour: 2^-32 + v * 2^-32 in fp32
        u = np.float32(2.3283064e-10) + out[:4].astype(np.float32) * np.float32(2.3283064e-10)
        assert np.array_equal(u, np.array(c["uniform"], np.float32))


def test_host_ordering_and_state_carry_over(orc):
    """One generate call: value j comes from lane j mod LANES, draw floor(j / LANES); a second call continues every lane where the first left it."""
    lanes, n1, n2 = 4096, 8192, 12288
    first = np.zeros(n1, np.float32); orc.lib().orc_xorwow_generate(C.c_uint64(0), 1, lanes, 0, n1, _p(first))
    second = np.zeros(n2, np.float32); orc.lib().orc_xorwow_generate(C.c_uint64(0), 1, lanes, n1, n2, _p(second))
    both = np.zeros(n1 + n2, np.float32); orc.lib().orc_xorwow_generate(C.c_uint64(0), 1, lanes, 0, n1 + n2, _p(both))
    assert np.array_equal(first, both[:n1]) and np.array_equal(second, both[n1:])          # two calls = one call of the total size ...
    odd = np.zeros(100 + 5000, np.float32); sz = np.array([100, 5000], np.uint32)
    orc.lib().orc_xorwow_generate_calls(C.c_uint64(0), 1, lanes, 2, _p(sz), _p(odd))
    # ... whatever the sizes: the offset of the ordering rule runs across calls
    assert np.array_equal(odd, both[:5100])
    for lane in (0, 1, 4095):
        d = np.zeros(5, np.uint32); orc.lib().orc_xorwow_lane_draws(C.c_uint64(0), 1, lane, 5, _p(d))
        u = np.float32(2.3283064e-10) + d.astype(np.float32) * np.float32(2.3283064e-10)
        assert np.array_equal(both[lane::lanes][:5], u)
    assert (both > 0).all() and (both <= 1).all() and abs(both.mean() - 0.5) < 0.01


def test_curand_flavour_differs_only_in_seeding_and_the_half_ulp(orc):
    """The cuRAND flavour (CURAND-A1/A2: seed scramble constants and the +2^-33 of _curand_uniform, from the published header) shares transition and jump with
    the rocRAND flavour: same statistics, different stream."""
    a = np.zeros(4096 * 4, np.float32); b = np.zeros_like(a)
    orc.lib().orc_xorwow_generate(C.c_uint64(0), 0, 4096, 0, a.size, _p(a)); orc.lib().orc_xorwow_generate(C.c_uint64(0), 1, 4096, 0, b.size, _p(b))
    assert not np.array_equal(a, b) and (a > 0).all() and (a <= 1).all() and abs(a.mean() - 0.5) < 0.01 and abs(a.var() - 1 / 12) < 0.005


def test_oracle_trains_in_same_inputs_mode(orc, small_scene):
    """XORWOW stream + tcnn's init order: a different but equally valid training run (it learns), repeatable, and `advance_iter` consumes the skipped draws."""
    import __graft_entry__ as ge
    from conftest import C1
    kw = dict(C1, xorwow=1, tcnn_init_order=1)
    a = ge.make_oracle(orc, small_scene, kw); b = ge.make_oracle(orc, small_scene, kw); c = ge.make_oracle(orc, small_scene, C1)
    assert not np.array_equal(a.buffer("master"), c.buffer("master"))                       # the init order permutes the draws ...
    # ... within a tensor: same multiset
    assert np.array_equal(np.sort(a.buffer("master")[:a.n_mlp][:a.W * a.Epad]), np.sort(c.buffer("master")[:a.n_mlp][:a.W * a.Epad]))
    l0 = a.train(1); l1 = a.train(150); b.train(151)
    assert l1 < 0.25 * l0 and np.array_equal(a.buffer("master"), b.buffer("master"))
    a.generate_batch(); pa = a.buffer("pts").copy()
    d = ge.make_oracle(orc, small_scene, kw); d.set_params(a.buffer("master"))
    for _ in range(151):
        d.advance_iter()
    d.generate_batch()
    assert np.array_equal(pa, d.buffer("pts"))
    for m in (a, b, c, d):
        m.close()
