"""Occupancy-grid skipping in the forward pass (BASELINE.json north star; mon_config::occupancy_skip).  The reference has no such grid -- it
evaluates all 32 samples of every ray (nerf_model.cu:536-566) -- so the switch is off by default and every parity test runs without it.
Here: it is opt-in, does nothing during its warm-up (bit-identical parameters), and afterwards trades a bounded loss of quality for fewer
table gathers."""
import time
import zlib

import numpy as np
import pytest

import __graft_entry__ as ge
from parity import psnr

pytestmark = pytest.mark.gpu


def _score(obj, sc, ss):
    ps, ious = [], []
    for box in sc.objects[0]["boxes"][::4]:
        v, x, y, h, w = (int(q) for q in box); rgb, depth, mask = obj.render(box, ss.colmajor(sc.Twc[v]))
        gm = sc.instance[v, y:y + h, x:x + w] > 0; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
        ps.append(psnr(rgb, gt)); ious.append((mask.astype(bool) & gm).sum() / max(1, (mask.astype(bool) | gm).sum()))
    return float(np.mean(ps)), float(np.mean(ious))


def test_occupancy_skipping_is_opt_in_inert_during_warmup_and_close_afterwards(pkg, ss):
    assert pkg.device_count() >= 1
    assert pkg.default_config().occupancy_skip == 0                                  # off unless asked for
    sc = ss.make_scene(n_views=16, H=240, W=320, f=260.0, seed=2)
    ds, exact = ge.make_problem(pkg, sc, dict(sample_seed=77))
    _, skip = ge.make_problem(pkg, sc, dict(sample_seed=77, occupancy_skip=1), dataset=ds)
    exact.train(256); skip.train(256)                                                # warm-up: every cell counts as occupied
    assert zlib.crc32(exact.get_params(0).tobytes()) == zlib.crc32(skip.get_params(0).tobytes())
    exact.train(544); skip.train(544)
    t = []
    for o in (exact, skip):
        pkg.lib().mon_device_synchronize(0); t0 = time.perf_counter(); o.train(200); pkg.lib().mon_device_synchronize(0)
        t.append((time.perf_counter() - t0) / 200)
    (p_e, iou_e), (p_s, iou_s) = _score(exact, sc, ss), _score(skip, sc, ss)
    print("exact %.2f dB IoU %.3f, %.1f us/step; occupancy skipping %.2f dB IoU %.3f, %.1f us/step" % (p_e, iou_e, 1e6 * t[0], p_s, iou_s, 1e6 * t[1]))
    assert zlib.crc32(exact.get_params(0).tobytes()) != zlib.crc32(skip.get_params(0).tobytes())      # it really skipped something
    assert p_s > p_e - 1.5 and iou_s > 0.9 and iou_s > iou_e - 0.03
    assert t[1] < 1.05 * t[0]                                                        # never slower (the refresh costs ~60 us every 32 iterations)
    for o in (exact, skip):
        assert np.isfinite(o.get_params(0)).all()
        o.close()
    ds.close()


def test_occupancy_grid_keeps_refreshing_under_hipgraph_replay_after_an_odd_iteration_count(pkg, ss):
    """the hipGraph path asks for a refresh only at the start of a captured pair of iterations; after an odd number of iterations an exact
    `iter % 32 == 0` test was never true again.  The refresh is due at the first asked-for iteration at or after the next multiple of the interval."""
    assert pkg.device_count() >= 1
    sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)
    pkg.set_option("use_graph", 1)
    try:
        ds, obj = ge.make_problem(pkg, sc, dict(rays_per_batch=256, occupancy_skip=1))
        obj.train(257)                                                               # odd: pairs now start at odd iterations
        last, due = obj.occupancy_state(); assert last == 256 and due == 288, (last, due)
        obj.train(64)                                                                # pairs at 257, 259, ...: 289 is the first at or after 288
        last, due = obj.occupancy_state(); assert last >= 288 and due == (last // 32 + 1) * 32 and due > 257 + 64 - 32, (last, due)
        obj.train(63)
        last2, _ = obj.occupancy_state(); assert last2 > last
        assert np.isfinite(obj.get_params(0)).all()
        obj.close(); ds.close()
    finally:
        pkg.set_option("use_graph", 0)


def test_occupancy_skipping_trains_identically_on_either_forward_chain_and_across_the_switch(pkg, ss):
    """With the grid in use the level-tile chain encodes the LIVE samples only: the position pass looks every sample's cell up, leaves the ray's live bits in its
    record and compacts the live samples of each encode partition into a list (k_encode_tiles<LIVE> walks it, k_fused_train<PRE, OCC> takes the bits from the
    record); the gather chain (option lds_encode = 0) masks the dead samples' loads instead.  Both must leave the same parameters bit for bit -- also across
    grid refreshes, after which the positions already sampled for the next iteration are sampled again -- and so must hipGraph replay: five calls of 160 steps."""
    assert pkg.device_count() >= 1
    sc = ss.make_scene(n_views=16, H=240, W=320, f=260.0, seed=2)
    crcs = []
    for lds, graph in ((1, 0), (0, 0), (1, 1)):                  # (the third run: hipGraph replay, whose cached graph must follow the change of chain)
        pkg.set_option("lds_encode", lds); pkg.set_option("use_graph", graph)
        try:
            ds, obj = ge.make_problem(pkg, sc, dict(sample_seed=78, occupancy_skip=1))
            c = []
            for _ in range(5):
                obj.train(160); c.append(zlib.crc32(obj.get_params(0).tobytes()))
            last, due = obj.occupancy_state(); assert last > 0                      # the grid was refreshed and is in use
            # ... on the level tiles to the end (round 6: k_encode_tiles walks the live samples only and wins in every regime): a further call launches it
            # every iteration (kernel class 6)
            if lds and not graph:
                obj.set_profiling(True); obj.profile(reset=True); obj.train(8); prof = obj.profile(reset=True); obj.set_profiling(False)
                assert prof["launches"][6] == 8 and prof["launches"][1] == 8, prof["launches"]
            crcs.append(c); obj.close(); ds.close()
        finally:
            pkg.set_option("lds_encode", 1); pkg.set_option("use_graph", 0)
    assert crcs[0] == crcs[1] == crcs[2], crcs
