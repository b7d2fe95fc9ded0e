"""XORWOW "same inputs" mode on the GPU box (-m gpu).
  1. The oracle's host-generator model (lane k = subsequence k of 2^67 steps, value n of a call from lane n mod LANES, lanes keep their state between calls)
     against the real thing: librocrand's HOST generator (ROCRAND_RNG_PSEUDO_XORWOW, default seed) running on the device -- rocRAND uses LANES = 131072
     (tools/xorwow_probe.py; cuRAND documents 4096), the reference's per-iteration call sizes and awkward ones.
  2. The HIP path in XORWOW mode (k_xorwow_fill + the consumers) against the oracle in the same mode: batch generation bit for bit, training, render.
cuRAND itself is not in the image: its flavour differs from the pinned one in four seeding constants and the 2^-33 of _curand_uniform (CURAND-A1/A2, DESIGN.md
1)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import C1
from parity import close_f32

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_rocrand_host_generator_matches_the_oracle_stream(pkg, orc):
    assert pkg.device_count() >= 1, "no HIP device visible: the GPU tests must run on the MI355X box"
    # (plain HIP allocations: torch cannot always initialise its own context once the product library holds the device)
    rr = C.CDLL("librocrand.so"); hip = C.CDLL("libamdhip64.so")
    # two base.json iterations (2R, 3R, S R with R = 4096), then sizes that are no multiple of anything
    for sizes in ([8192, 12288, 131072, 8192, 12288, 131072], [2048, 3072, 32768, 5, 131073, 262144 + 7, 1]):
        g = C.c_void_p(); assert rr.rocrand_create_generator(C.byref(g), 401) == 0                     # ROCRAND_RNG_PSEUDO_XORWOW, seed 0 by default
        got = []
        for n in sizes:
            d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(4 * n)) == 0
            assert rr.rocrand_generate_uniform(g, d, C.c_size_t(n)) == 0 and hip.hipDeviceSynchronize() == 0
            a = np.zeros(n, np.float32); assert hip.hipMemcpy(_p(a), d, C.c_size_t(4 * n), 2) == 0; hip.hipFree(d); got.append(a)
        rr.rocrand_destroy_generator(g)
        want = np.zeros(sum(sizes), np.float32); sz = np.array(sizes, np.uint32)
        orc.lib().orc_xorwow_generate_calls(C.c_uint64(0), 1, 131072, len(sizes), _p(sz), _p(want))
        off = 0
        for i, (n, a) in enumerate(zip(sizes, got)):
            assert np.array_equal(a, want[off:off + n]), "call %d of %d values (after %s)" % (i, n, sizes[:i]); off += n


@pytest.mark.parametrize("backend", [0, 1])
@pytest.mark.parametrize("mode", [dict(xorwow=2, xorwow_lanes=131072), dict(xorwow=1), dict(xorwow=1, tcnn_init_order=1)], ids=["rocrand131072", "curand4096",
        "curand4096_tcnn_init"])
def test_hip_path_in_xorwow_mode_matches_the_oracle(pkg, orc, ss, small_scene, backend, mode):
    assert pkg.device_count() >= 1
    kw = dict(C1, **mode)
    # the tile chain also at this small batch (default: from 3072 rays up), so that its position / candidate pipeline runs on the XORWOW arrays
    pkg.set_option("lds_encode", 2)
    try:
        ds, obj = ge.make_problem(pkg, small_scene, kw); obj.set_backend(backend)
    finally:
        pkg.set_option("lds_encode", 1)
    ref = ge.make_oracle(orc, small_scene, kw)
    assert np.array_equal(obj.get_params(0), ref.buffer("master"))                     # same initial weights, also in tcnn's element order
    if backend == 1:
        obj.set_debug_dump(True)
    # stage-wise: batch generation bit for bit, three iterations of the stream
    for it in range(3):
        obj.train_stages(1 | 2); ref.generate_batch(); ref.forward_backward()
        assert int(obj.buffer("state")[2]) == ref.n_valid and ref.n_valid > 0
        for b in ("ray_o", "ray_d", "ray_t0", "ray_t1", "target", "bgcol", "pts", "tdist"):
            close_f32(obj.buffer(b), ref.buffer(b), "%s (iteration %d)" % (b, it), 1e-6)
        assert np.array_equal(obj.buffer("E"), ref.buffer("E")), "hash-grid encode must be bit-exact (iteration %d)" % it
        obj.train_stages(4); ref.train_step()
        # (one Adam step leaves ~0.3 % of the weights a learning rate apart: the stream is what is compared here)
        ref.set_params(obj.get_params(0))
    if backend == 1:
        # the default fused path: level-tile encode, positions prepared one iteration ahead
        obj.set_debug_dump(False)
    l_hip = obj.train(60); l_ref = ref.train(60)
    assert np.isfinite(l_hip) and abs(l_hip - l_ref) < max(0.5 * l_ref, 0.02), (l_hip, l_ref)
    box = small_scene.objects[0]["boxes"][2]; pose = ss.colmajor(small_scene.Twc[int(box[0])])
    rgb, depth, mask = obj.render(box, pose); rr, rd, rm = ref.render(box, pose)        # a fresh generator per Render on both sides
    assert (mask.astype(bool) != rm.astype(bool)).mean() < 0.03 and -10 * np.log10(np.mean((rgb - rr) ** 2) + 1e-12) > 25.0
    rgb2, _, _ = obj.render(box, pose); assert np.array_equal(rgb, rgb2)                # ... so a second render repeats the first exactly
    obj.close(); ds.close(); ref.close()
