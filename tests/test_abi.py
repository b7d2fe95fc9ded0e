"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mon_core.h
declares, parses the reference's config schema, and fails loudly (no CPU fallback) without a device."""
import numpy as np
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols(name="mon_core.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mon_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    L = ctypes.CDLL(pkg.lib_path())
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libmon_core.so does not export %s" % s
    assert sorted(pkg.exported_symbols()) == syms, "binding table and header disagree"


def test_default_config_is_base_json(pkg):
    c = pkg.default_config()
    j = pkg.config_from_json(os.path.join(ROOT, "ro-map_amd", "configs", "base.json"))
    for f, _ in pkg.MonConfig._fields_:
        assert getattr(c, f) == getattr(j, f), f
    assert (c.n_levels, c.n_features, c.log2_hashmap_size, c.base_resolution) == (16, 2, 16, 16)
    assert (c.n_neurons, c.n_hidden_layers, c.rays_per_batch, c.n_samples) == (64, 1, 4096, 32)
    assert c.per_level_scale == 2.0 and c.loss_scale == 128.0 and abs(c.learning_rate - 1e-2) < 1e-9
    assert abs(c.ema_decay - 0.95) < 1e-7 and c.decay_start == 20000 and c.decay_interval == 10000 and abs(c.decay_base - 0.33) < 1e-7
    assert abs(c.epsilon - 1e-15) < 1e-20 and abs(c.l2_reg - 1e-6) < 1e-12 and c.param_seed == 1337


def test_c1_config_parses(pkg):
    j = pkg.config_from_json(os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"))
    assert (j.n_levels, j.n_neurons, j.n_hidden_layers, j.rays_per_batch) == (4, 32, 2, 1024)


def test_config_errors(pkg, tmp_path):
    with pytest.raises(pkg.MonError) as e:
        pkg.config_from_json(str(tmp_path / "missing.json"))
    assert e.value.code == 4 and "config file error" in str(e.value)
    bad = tmp_path / "bad.json"; bad.write_text("{ \"encoding\": { \"otype\": \"Frequency\" } }")
    with pytest.raises(pkg.MonError):
        pkg.config_from_json(str(bad))
    broken = tmp_path / "broken.json"; broken.write_text("{ \"encoding\": ")
    with pytest.raises(pkg.MonError):
        pkg.config_from_json(str(broken))


def test_config_reader_survives_damaged_files(pkg, tmp_path):
    """Truncations, byte flips and pathological nesting of base.json: a clean error or a config, never a crash; values outside what the
    kernels support are rejected when the level table is built (mon_debug_fast_index goes through the same check)."""
    import numpy as np
    good = open(os.path.join(ROOT, "ro-map_amd", "configs", "base.json"), "rb").read()
    rs = np.random.RandomState(0); f = tmp_path / "m.json"
    for cut in range(0, len(good), 37):
        f.write_bytes(good[:cut])
        try:
            pkg.config_from_json(str(f))
        except pkg.MonError:
            pass
    for _ in range(300):
        b = bytearray(good)
        for pos in rs.randint(0, len(b), rs.randint(1, 4)):
            b[pos] = rs.randint(32, 127)
        f.write_bytes(bytes(b))
        try:
            pkg.config_from_json(str(f))
        except pkg.MonError:
            pass
    f.write_text("[" * 100000)
    with pytest.raises(pkg.MonError):
        pkg.config_from_json(str(f))
    f.write_text('{"encoding": {"otype": "HashGrid", "n_levels": 99, "n_features_per_level": 2, "log2_hashmap_size": 40}, '
                 '"network": {"n_neurons": 64, "n_hidden_layers": 1}}')
    try:
        c = pkg.config_from_json(str(f))
    except pkg.MonError:
        c = None
    if c is not None:
        with pytest.raises(pkg.MonError):
            pkg.fast_index(c, 0, 0, 0, 0)


def test_no_silent_cpu_fallback(pkg):
    """Without a HIP device every compute entry point must fail with MON_ERR_NO_DEVICE, not run elsewhere."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.MonError) as e:
        pkg.Dataset(0, 48, 64, 50.0, 50.0, 31.5, 23.5, 4)
    assert e.value.code == 2
    with pytest.raises(pkg.MonError):
        import numpy as np
        pkg.selftest_mfma(np.zeros((32, 16), np.uint16), np.zeros((16, 32), np.uint16))


def test_product_does_not_reference_the_oracle():
    """The product tree must not include, link or call anything under oracle/."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "ro-map_amd")):
        for f in fs:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            t = open(os.path.join(dp, f), errors="ignore").read()
            if re.search(r"mon_oracle|oracle_binding|orc_[a-z]+\(|libmon_oracle", t):
                bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = os.popen("ldd %s" % os.path.join(ROOT, "ro-map_amd", "libmon_core.so")).read()
    assert "oracle" not in out


def test_closed_form_corner_index_matches_tcnn_loop(pkg, orc):
    """The fused kernels' per-level index constants (built on the host) against the oracle's restatement of tcnn's
    grid_index loop, for every level of several configurations -- including the uint32 stride wrap-around at res = 2^16."""
    import ctypes as C
    import numpy as np
    rs = np.random.RandomState(0)
    for kw in (dict(), dict(log2_hashmap_size=19), dict(log2_hashmap_size=22), dict(base_resolution=20, n_levels=8), dict(n_levels=4),
               dict(base_resolution=24, per_level_scale=1.5, n_levels=12, log2_hashmap_size=15)):
        cfg = pkg.default_config(**kw); ocfg = orc.default_config(**kw)
        off = np.zeros(33, np.uint32); sc = np.zeros(32, np.float32); res = np.zeros(32, np.uint32)
        orc.lib().orc_level_table(C.byref(ocfg), off.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p))
        for l in range(cfg.n_levels):
            size = int(off[l + 1] - off[l]); r = int(res[l])
            pts = [tuple(int(v) for v in rs.randint(0, r + 1, 3)) for _ in range(200)] + [(0, 0, 0), (r, r, r), (r, 0, 0), (0, r, 0), (0, 0, r), (r - 1, r, r)]
            for (x, y, z) in pts:
                got, n = pkg.fast_index(cfg, l, x, y, z)
                assert n == size and got == orc.lib().orc_grid_index(size, r, x, y, z), (kw, l, r, size, x, y, z)


@pytest.mark.parametrize("shape", [(32, 64, 1, 16), (16, 32, 2, 4), (32, 64, 2, 13), (16, 64, 1, 7), (32, 32, 2, 16), (32, 128, 1, 16), (16, 128, 1, 6)])
def test_fragment_layout_maps_are_inverse(pkg, shape):
    """frag_layout.h: every image element names the parameter whose slot list contains it, and the other way round
    (the optimizer writes the image through frag_slots, k_build_frag_image reads through frag_source)."""
    epad, W, NH, L = shape
    src, slots = pkg.frag_layout(epad, W, NH, L)
    used = np.nonzero(src >= 0)[0]
    assert used.size > 0 and src.max() < slots.shape[0]
    hit = (slots[src[used], 0] == used) | (slots[src[used], 1] == used)
    assert hit.all()
    ps = np.repeat(np.arange(slots.shape[0]), 2); flat = slots.reshape(-1); ok = flat >= 0
    assert (src[flat[ok]] == ps[ok]).all()
    assert np.unique(flat[ok]).size == ok.sum() == used.size           # a bijection between used image elements and slots
    # every real weight is present: W0 columns of real features, W1, the 4 real output rows
    real = np.zeros(slots.shape[0], bool)
    real[: W * epad] = (np.arange(W * epad) % epad) < 2 * L
    if NH == 2:
        real[W * epad: W * epad + W * W] = True
    off_wo = W * epad + (NH - 1) * W * W; real[off_wo: off_wo + 4 * W] = True
    assert ((slots[:, 0] >= 0) == real).all() and ((slots[:, 1] >= 0) == real).all()


@pytest.mark.parametrize("shape", [(32, 64, 1, 16), (16, 32, 2, 4), (32, 64, 2, 13), (16, 64, 1, 7), (32, 32, 2, 16), (32, 128, 1, 16), (16, 128, 1, 6)])
def test_accumulator_layout_of_the_dw_partial_rows(pkg, shape):
    """frag_layout.h acc_param: k_fused_train writes its weight-gradient partial rows in MFMA accumulator order and the summing kernels map columns to
    parameters.  The map must hit every parameter that can carry a gradient exactly once and nothing else."""
    epad, W, NH, L = shape
    prm = pkg.acc_layout(epad, W, NH, L)
    n_mlp = W * epad + (NH - 1) * W * W + 16 * W
    hit = prm[prm >= 0]
    assert hit.max() < n_mlp and np.unique(hit).size == hit.size                  # no parameter is summed from two columns
    expect = np.zeros(n_mlp, bool)
    expect[: W * epad] = True                                                     # all of W0 (pad-feature columns receive exact zeros: their inputs are zero)
    if NH == 2:
        expect[W * epad: W * epad + W * W] = True
    off_wo = W * epad + (NH - 1) * W * W; expect[off_wo: off_wo + 4 * W] = True    # the 4 real output rows; rows 4..15 of the padded layer never get a gradient
    got = np.zeros(n_mlp, bool); got[hit] = True
    assert (got == expect).all()
    assert prm.size == (W // 32) * 1024 + (NH - 1) * (W // 32) ** 2 * 1024 + (W // 32) * 128
    assert ((prm < 0).sum() == 0) if epad == 32 else ((prm < 0).sum() == (W // 32) * 512)   # a 16-wide encoding leaves half of the dW0 tile columns unused


def test_diagnostics_live_in_their_own_library(pkg):
    """libmon_core.so is the product only: the micro-benchmarks, the MFMA self-test, the debug read-back and the layout hooks are exported by
    libmon_core_diag.so (include/mon_core_diag.h), which links against the product library -- never the other way round."""
    import subprocess
    diag = header_symbols("mon_core_diag.h")
    assert len(diag) >= 6 and sorted(pkg.diag_symbols()) == diag and not set(diag) & set(header_symbols())
    assert os.path.exists(pkg.diag_lib_path()), "run __graft_entry__.build() first"
    core = ctypes.CDLL(pkg.lib_path()); dl = pkg.diag_lib()
    for s in diag:
        assert hasattr(dl, s), "libmon_core_diag.so does not export " + s
        assert not hasattr(core, s), "diagnostic symbol %s is still exported by the product library" % s
    defined = subprocess.run(["nm", "-D", "--defined-only", pkg.lib_path()], capture_output=True, text=True).stdout
    undefined = subprocess.run(["nm", "-D", "--undefined-only", pkg.lib_path()], capture_output=True, text=True).stdout
    assert "microbench" not in defined and "selftest" not in defined and "getenv" not in undefined      # no environment knobs on the product path
    assert pkg.yaml_number("%YAML:1.0\n# Camera.fx: 1.0\nCamera.Height_mm: 9999\n  Camera.H : 480\nCamera.fx: 525.5\n", "Camera.H") == 480.0
    assert pkg.yaml_number("# Camera.fx: 1.0\nCamera.fx: 525.5\n", "Camera.fx") == 525.5
    with pytest.raises(pkg.MonError):
        pkg.yaml_number("Camera.Height: 3\nCamera.fx:\n", "Camera.H")


def test_options_are_an_explicit_interface(pkg):
    """Test and tuning switches go through mon_set_option / mon_get_option; the product library reads no environment variables."""
    assert pkg.get_option("big_switch") == 16384 and pkg.get_option("backend") == -1 and pkg.get_option("train_lanes") == 2
    pkg.set_option("keep_zero_samples", 1); assert pkg.get_option("keep_zero_samples") == 1; pkg.set_option("keep_zero_samples", 0)
    with pytest.raises(pkg.MonError):
        pkg.set_option("no_such_switch", 1)
    # round 6: nine names, each documented in include/mon_core.h; the A/B switches whose losing setting only a measurement wanted are variant builds
    names = ("backend", "use_graph", "big_switch", "lds_encode", "tile_render", "step_variant", "keep_zero_samples", "train_lanes", "roctx")
    hdr = open(os.path.join(ROOT, "include", "mon_core.h")).read()
    for n in names:
        pkg.get_option(n); assert '"%s"' % n in hdr, n
    for retired in ("steps16", "state_records", "touched_flags", "lazy_ema", "lane_chunk", "online_slice_min", "offline_outer", "offline_inner"):
        with pytest.raises(pkg.MonError):
            pkg.get_option(retired)
    # the offline schedule is an API call of its own (reference: 10 x 500, nerf_manager.cu:89)
    pkg.set_offline_schedule(3, 40); pkg.set_offline_schedule(10, 500)
    with pytest.raises(pkg.MonError):
        pkg.set_offline_schedule(0, 500)
