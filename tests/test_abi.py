"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mon_core.h
declares, parses the reference's config schema, and fails loudly (no CPU fallback) without a device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mon_core.h")).read()
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
