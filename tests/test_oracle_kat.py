"""Known-answer tests that pin the CPU oracle (oracle/mon_oracle.c).

The reference ships no golden vectors for this path (parity unpinned), so the oracle is pinned by
closed forms and independent re-derivations: hash-index KATs, fp64 NumPy trilinear encode, closed-form
compositing, torch-autograd of the equivalent scalar loss for the hand-derived gradient
(nerf_model.cu:817-954), fp64 MLP backward, closed-form first Adam/EMA step."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import C1


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ level table / hash index
def test_level_table_base_json(orc):
    cfg = orc.default_config()
    off = np.zeros(17, np.uint32); sc = np.zeros(16, np.float32); res = np.zeros(16, np.uint32)
    epad = orc.lib().orc_level_table(C.byref(cfg), _p(off), _p(sc), _p(res))
    assert epad == 32
    assert list(res[:4]) == [16, 32, 64, 128] and list(sc[:3]) == [15.0, 31.0, 63.0]
    sizes = np.diff(off.astype(np.int64))
    assert sizes[0] == 4096 and sizes[1] == 32768 and (sizes[2:] == 65536).all()
    assert off[16] == 954368                       # 954 368 entries = 1 908 736 params


def test_level_table_c1_and_stress(orc):
    cfg = orc.default_config(**C1)
    off = np.zeros(17, np.uint32); sc = np.zeros(16, np.float32); res = np.zeros(16, np.uint32)
    assert orc.lib().orc_level_table(C.byref(cfg), _p(off), _p(sc), _p(res)) == 16        # 8 features padded to 16
    assert off[4] == 4096 + 32768 + 65536 + 65536                                           # 167 936 entries
    cfg = orc.default_config(log2_hashmap_size=22)
    orc.lib().orc_level_table(C.byref(cfg), _p(off), _p(sc), _p(res))
    assert off[16] == 52727808


def test_grid_index_kats(orc):
    gi = orc.lib().orc_grid_index
    # dense level 0 of base.json: res 16, 4096 entries: x + y*16 + z*256, the +1 corner aliases (mod size)
    assert gi(4096, 16, 3, 2, 1) == 3 + 32 + 256
    assert gi(4096, 16, 16, 15, 15) == (16 + 15 * 16 + 15 * 256) % 4096
    # dense level 1: res 32, 32768 entries
    assert gi(32768, 32, 31, 31, 31) == 31 + 31 * 32 + 31 * 1024
    # hashed level: res 64 (64^3 > 65536): x ^ y*2654435761 ^ z*805459861 (mod 2^32) mod 65536
    for (x, y, z) in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (17, 33, 63), (64, 64, 64)]:
        want = ((x * 1) ^ ((y * 2654435761) & 0xffffffff) ^ ((z * 805459861) & 0xffffffff)) % 65536
        assert gi(65536, 64, x, y, z) == want
    assert gi(65536, 64, 0, 1, 0) == 2654435761 % 65536 == 31153


def test_rng_and_half(orc):
    L = orc.lib()
    u = np.array([L.orc_rand01(2024, s, 7, i) for s in range(4) for i in range(2000)])
    assert (u >= 0).all() and (u < 1).all() and abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert L.orc_rand01(2024, 0, 0, 0) != L.orc_rand01(2024, 0, 1, 0) != L.orc_rand01(2025, 0, 0, 0)
    xs = np.concatenate([np.random.RandomState(0).randn(2000).astype(np.float32) * s for s in (1e-6, 1e-3, 1.0, 300.0, 70000.0)])
    got = np.array([L.orc_f2h(float(x)) for x in xs], np.uint16)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    assert (got == want).all()
    hs = np.arange(0, 65536, 7, dtype=np.uint16)
    back = np.array([L.orc_h2f(int(h)) for h in hs], np.float32)
    ref = hs.view(np.float16).astype(np.float32)
    assert np.array_equal(back[~np.isnan(ref)], ref[~np.isnan(ref)])


# ------------------------------------------------------------------ encode vs fp64 NumPy
def _numpy_corners(cfg, x):
    """tcnn's grid walk re-derived in NumPy, independent of the oracle's level_corners: yields (level, corner k, global entry index [n], weight [n] fp64)
    for positions x [n, 3] -- fractional position from the kernel's fp32 `scale * x + 0.5`, everything after that in fp64 / exact integers."""
    L = cfg.n_levels
    off = np.zeros(17, np.uint32); sc = np.zeros(16, np.float32); res = np.zeros(16, np.uint32)
    from oracle_binding import lib
    lib().orc_level_table(C.byref(cfg), _p(off), _p(sc), _p(res))
    for l in range(L):
        size = int(off[l + 1] - off[l]); r = int(res[l])
        pos = np.float32(sc[l]) * x.astype(np.float32) + np.float32(0.5)        # fp32 like the kernel; fmaf vs mul+add differ < 1 ulp
        pos = pos.astype(np.float64)
        g = np.floor(pos); fr = pos - g; g = g.astype(np.int64)
        for k in range(8):
            w = np.ones(x.shape[0]); q = []
            for d in range(3):
                if k & (1 << d):
                    w = w * fr[:, d]; q.append(g[:, d] + 1)
                else:
                    w = w * (1 - fr[:, d]); q.append(g[:, d])
            qx, qy, qz = (np.asarray(v, np.uint64) & 0xffffffff for v in q)
            # tcnn grid_index: linear index while the running stride (uint32) fits the table, otherwise the prime hash
            stride, dense = 1, np.zeros(x.shape[0], np.uint64)
            for coord in (qx, qy, qz):
                if stride <= size:
                    dense = (dense + coord * stride) & 0xffffffff; stride = (stride * r) & 0xffffffff
            if size < stride:
                idx = ((qx ^ (qy * 2654435761 & 0xffffffff) ^ (qz * 805459861 & 0xffffffff)) & 0xffffffff) % size
            else:
                idx = dense % size
            yield l, k, idx.astype(np.int64) + int(off[l]), w


def _numpy_encode(cfg, table_f64, x):
    out = np.zeros((x.shape[0], 2 * cfg.n_levels))
    for l, _, idx, w in _numpy_corners(cfg, x):
        out[:, 2 * l] += w * table_f64[idx, 0]; out[:, 2 * l + 1] += w * table_f64[idx, 1]
    return out


@pytest.mark.parametrize("kw", [C1, dict(n_levels=8, log2_hashmap_size=14)])
def test_encode_matches_fp64_numpy(orc, kw):
    cfg = orc.default_config(**kw)
    m = orc.OracleModel(cfg)
    rs = np.random.RandomState(1)
    master = m.buffer("master"); master[m.n_mlp:] = rs.uniform(-1, 1, m.n_params - m.n_mlp).astype(np.float32)
    m.set_params(master)
    half = m.buffer("half")
    x = rs.uniform(0, 1, (3000, 3)).astype(np.float32)
    x[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.999999, 0.25, 0.75], [1e-7, 1e-7, 1e-7]]
    E = np.zeros((x.shape[0], m.Epad), np.uint16)
    orc.lib().orc_encode(m.h, _p(half), _p(x), x.shape[0], _p(E))
    table = orc.h2f(half[m.n_mlp:]).astype(np.float64).reshape(-1, 2)
    want = _numpy_encode(cfg, table, x)
    got = orc.h2f(E)[:, :2 * cfg.n_levels]
    assert np.abs(got - want).max() < 2e-3          # fp16 output rounding of values in [-1, 1]
    assert (E[:, 2 * cfg.n_levels:] == 0).all()
    m.close()


# ------------------------------------------------------------------ compositing closed form
def test_composite_constant_density_slab(orc):
    S = 32; sigma = 3.0; tmin, tmax = 0.8, 1.6
    t = (tmin + (tmax - tmin) * (np.arange(S) + 0.5) / S).astype(np.float32)
    out = np.zeros((S, 4), np.float32); out[:, 0] = 0.3; out[:, 1] = -0.2; out[:, 2] = 1.1; out[:, 3] = math.log(sigma)
    out_h = orc.f2h(out)
    bg = np.array([0.2, 0.4, 0.6], np.float32); rgb = np.zeros(3, np.float32); depth = np.zeros(1, np.float32); mask = np.zeros(1, np.float32)
    orc.lib().orc_composite(_p(out_h), _p(t), S, _p(bg), _p(rgb), _p(depth), _p(mask))
    sig_h = math.exp(float(orc.h2f(out_h[:, 3])[0]))
    # quirk a13: the first interval is measured from the ray origin, so the optical depth is sigma * t_last
    T = math.exp(-sig_h * float(t[-1]))
    assert abs(mask[0] - (1 - T)) < 2e-5
    c = 1 / (1 + np.exp(-orc.h2f(out_h[0, :3]).astype(np.float64)))
    assert np.abs(rgb - ((1 - T) * c + T * bg)).max() < 2e-5


def test_composite_early_out_zeroes_gradient(orc):
    S = 32; t = np.linspace(0.5, 1.5, S).astype(np.float32)
    out = np.zeros((S, 4), np.float32); out[:, 3] = 6.0            # sigma = e^6: opaque after a couple of samples
    out_h = orc.f2h(out); bg = np.zeros(3, np.float32)
    rgb = np.zeros(3, np.float32); depth = np.zeros(1, np.float32); mask = np.zeros(1, np.float32)
    orc.lib().orc_composite(_p(out_h), _p(t), S, _p(bg), _p(rgb), _p(depth), _p(mask))
    dO = np.ones((S, 4), np.uint16)
    tgt = np.array([0.1, 0.2, 0.3], np.float32)
    orc.lib().orc_gradient(_p(out_h), _p(t), S, 64, 128.0, 1, _p(tgt), 0.0, _p(rgb), float(depth[0]), float(mask[0]), _p(dO))
    nz = np.nonzero(np.any(dO != 0, axis=1))[0]
    assert nz.max() < 4 and (dO[4:] == 0).all()                    # samples after T < 1e-4 keep the memset zero (:1578)


# ------------------------------------------------------------------ hand-derived gradient vs autograd
def _torch_loss(v, t, tgt, tdepth, bg, is_obj):
    import torch
    c = torch.sigmoid(v[:, :3]); sig = torch.exp(v[:, 3])
    dt = t - torch.cat([torch.zeros(1, dtype=t.dtype), t[:-1]])
    def comp(sigma):
        alpha = 1 - torch.exp(-sigma * dt)
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=t.dtype), 1 - alpha]), 0)
        w = alpha * T[:-1]
        return (w[:, None] * c).sum(0) + T[-1] * bg, (w * t).sum(), 1 - T[-1]
    rgb, depth, mask = comp(sig)
    if is_obj:
        loss = ((rgb - tgt) ** 2).sum() + 0.5 * (1 - mask)
        if tdepth > 0:
            loss = loss + 0.5 * torch.abs(depth - tdepth)
        return loss
    # background rays: colour gradient only through the colours (density detached), opacity penalty + 0.01 * sum(sigma)
    rgb_c, _, _ = comp(sig.detach())
    return ((rgb_c - tgt) ** 2).sum() + 0.5 * mask + 0.01 * sig.sum()


@pytest.mark.parametrize("is_obj,tdepth", [(1, 0.0), (1, 1.05), (0, 0.0)])
def test_hand_gradient_matches_autograd(orc, is_obj, tdepth):
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(5 + is_obj); S = 32; nR = 1024; ls = 128.0
    t = np.sort(rs.uniform(0.7, 1.6, S)).astype(np.float32)
    out = rs.normal(0, 1.0, (S, 4)).astype(np.float32); out[:, 3] = rs.normal(0.3, 0.8, S)
    out_h = orc.f2h(out); v = orc.h2f(out_h).astype(np.float64)
    bg = rs.uniform(0, 1, 3).astype(np.float32); tgt = bg.copy() if not is_obj else rs.uniform(0, 1, 3).astype(np.float32)
    rgb = np.zeros(3, np.float32); depth = np.zeros(1, np.float32); mask = np.zeros(1, np.float32)
    orc.lib().orc_composite(_p(out_h), _p(t), S, _p(bg), _p(rgb), _p(depth), _p(mask))
    dO = np.zeros((S, 4), np.uint16)
    orc.lib().orc_gradient(_p(out_h), _p(t), S, nR, ls, is_obj, _p(tgt), float(tdepth), _p(rgb), float(depth[0]), float(mask[0]), _p(dO))
    vt = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    loss = _torch_loss(vt, torch.tensor(t, dtype=torch.float64), torch.tensor(tgt, dtype=torch.float64), tdepth, torch.tensor(bg, dtype=torch.float64), is_obj)
    loss.backward()
    want = vt.grad.numpy() * (ls / nR)
    got = orc.h2f(dO).astype(np.float64)
    assert mask[0] < 1 - 1e-4                       # the early-out did not trigger in this case
    err = np.abs(got - want)
    assert (err <= 2e-3 * np.abs(want) + 1e-7).all(), err.max()      # fp16 rounding of the stored gradient


# ------------------------------------------------------------------ MLP forward/backward vs fp64
@pytest.mark.parametrize("kw", [C1, dict(rays_per_batch=64, n_levels=16, n_neurons=64, n_hidden_layers=1)])
def test_mlp_forward_backward_vs_fp64(orc, small_scene, kw):
    import __graft_entry__ as ge
    kw = dict(kw); kw["rays_per_batch"] = 64
    m = ge.make_oracle(orc, small_scene, kw)
    rs = np.random.RandomState(2)
    master = m.buffer("master"); master[m.n_mlp:] = rs.uniform(-0.5, 0.5, m.n_params - m.n_mlp).astype(np.float32)
    m.set_params(master)
    m.generate_batch(); assert m.n_valid > 0
    m.forward_backward()
    W, NH, Ep, B = m.W, m.NH, m.Epad, m.R * m.S
    w = orc.h2f(m.buffer("half")[:m.n_mlp]).astype(np.float64)
    E = orc.h2f(m.buffer("E")).astype(np.float64).reshape(B, Ep)
    mats = []; o = 0
    for layer in range(NH + 1):
        rows = 16 if layer == NH else W; cols = Ep if layer == 0 else W
        mats.append(w[o:o + rows * cols].reshape(rows, cols)); o += rows * cols
    acts = [E]; a = E
    for layer in range(NH):
        a = np.maximum(a @ mats[layer].T, 0); a = a.astype(np.float16).astype(np.float64); acts.append(a)
    out = (a @ mats[NH].T)[:, :4]
    got_out = orc.h2f(m.buffer("O")).reshape(B, 4)
    assert np.abs(got_out - out).max() < 4e-3 * max(1.0, np.abs(out).max())
    got_h = orc.h2f(m.buffer("Hid")).reshape(B, NH, W)
    assert np.abs(got_h[:, NH - 1] - acts[NH]).max() < 4e-3 * max(1.0, np.abs(acts[NH]).max())
    # backward with the oracle's own dO (checked separately against autograd)
    dO = orc.h2f(m.buffer("dO")).astype(np.float64).reshape(B, 4)
    d = (dO @ mats[NH][:4]) * (got_h[:, NH - 1] > 0)
    dWout = dO.T @ got_h[:, NH - 1].astype(np.float64)
    gm = m.buffer("gmlp").astype(np.float64)
    o_out = W * Ep + (NH - 1) * W * W
    gout = gm[o_out:o_out + 16 * W].reshape(16, W)
    scale = np.abs(dWout).max() + 1e-12
    assert np.abs(gout[:4] - dWout).max() < 2e-3 * scale and (gout[4:] == 0).all()
    dh = orc.h2f(m.buffer("dHid")).astype(np.float64).reshape(B, NH, W)
    assert np.abs(dh[:, NH - 1] - d).max() < 3e-3 * (np.abs(d).max() + 1e-12)
    dcur = dh[:, NH - 1]
    for layer in range(NH - 1, 0, -1):
        dprev = (dcur @ mats[layer]) * (got_h[:, layer - 1] > 0)
        assert np.abs(dh[:, layer - 1] - dprev).max() < 3e-3 * (np.abs(dprev).max() + 1e-12)
        dcur = dh[:, layer - 1]
    dE = dcur @ mats[0]
    got_dE = orc.h2f(m.buffer("dE")).astype(np.float64).reshape(B, Ep)
    assert np.abs(got_dE - dE).max() < 3e-3 * (np.abs(dE).max() + 1e-12)
    dW0 = dcur.T @ E
    g0 = gm[:W * Ep].reshape(W, Ep)
    assert np.abs(g0 - dW0).max() < 2e-3 * (np.abs(dW0).max() + 1e-12)
    # grid gradient: sum of contributions equals the scatter of dE with trilinear weights (conservation per level:
    # the weights of the 8 corners sum to 1, so per level sum_entries g == sum_samples dE up to fp16 rounding of contributions)
    gg = m.buffer("ggrid").astype(np.float64).reshape(-1, 2); ga = m.buffer("ggrid_abs").astype(np.float64).reshape(-1, 2)
    cfg = m.cfg
    off = np.zeros(17, np.uint32); sc = np.zeros(16, np.float32); res = np.zeros(16, np.uint32)
    orc.lib().orc_level_table(C.byref(cfg), _p(off), _p(sc), _p(res))
    for l in range(cfg.n_levels):
        s_entries = gg[off[l]:off[l + 1]].sum(0); s_samples = got_dE[:, 2 * l:2 * l + 2].sum(0)
        tol = 1e-3 * ga[off[l]:off[l + 1]].sum(0) + 1e-6
        assert (np.abs(s_entries - s_samples) <= tol).all(), (l, s_entries, s_samples)
    # ... and PER ENTRY (conservation alone would accept a wrong corner -> entry map that preserves sums): an independent fp64 scatter
    # (np.add.at over the NumPy re-derivation of tcnn's walk) of w * dE, against the oracle's fp32 sum of the fp16-rounded contributions h(w * dE)
    pts = m.buffer("pts").reshape(B, 3)
    want = np.zeros_like(gg); cnt = np.zeros(gg.shape[0])
    for l, _, idx, w in _numpy_corners(cfg, pts):
        np.add.at(want[:, 0], idx, w * got_dE[:, 2 * l]); np.add.at(want[:, 1], idx, w * got_dE[:, 2 * l + 1]); np.add.at(cnt, idx, 1)
    # per contribution: half an fp16 ulp, or half the smallest subnormal (2^-24) where it underflows
    tol_e = 2.0 ** -10 * ga + cnt[:, None] * 2.0 ** -25 + 1e-9
    bad = np.abs(gg - want) > tol_e
    assert not bad.any(), "grid gradient: %d entries off, worst %.3e at entry %d" % (bad.sum(), np.abs(gg - want).max(), int(np.abs(gg - want).max(1).argmax()))
    assert (want != 0).any(1).sum() > 1000 and ((gg != 0).any(1) == (np.abs(want) > 0).any(1)).mean() > 0.999      # the same entries are touched
    m.close()


# ------------------------------------------------------------------ the whole backward pass vs autograd
@pytest.mark.parametrize("kw", [dict(n_levels=4, n_neurons=32, n_hidden_layers=2), dict(n_levels=16, n_neurons=64, n_hidden_layers=1)], ids=["c1net", "c2net"])
@pytest.mark.parametrize("use_depth", [False, True])
def test_end_to_end_gradients_match_torch_autograd(orc, small_scene, kw, use_depth):
    """encode -> MLP -> composite -> loss on a 64-ray batch as ONE fp64 torch graph (the table and the matrices are leaves; corner indices and weights from the
    NumPy re-derivation of tcnn's walk), its autograd gradients against the oracle's hand-written backward chain: gmlp (dW of every layer) and the grid
    gradient.
    The oracle also rounds dL/dO, dh, dE and every scatter contribution to fp16 (tcnn's network precision): agreement is to a few 1e-3 of the gradient scale."""
    torch = pytest.importorskip("torch")
    import __graft_entry__ as ge
    kw = dict(kw, rays_per_batch=64)
    m = ge.make_oracle(orc, small_scene, kw, use_depth=use_depth)
    rs = np.random.RandomState(4)
    master = m.buffer("master"); master[m.n_mlp:] = rs.uniform(-0.5, 0.5, m.n_params - m.n_mlp).astype(np.float32)
    m.set_params(master)
    m.generate_batch(); assert m.n_valid > 0
    m.forward_backward()
    W, NH, Ep, R, S = m.W, m.NH, m.Epad, m.R, m.S; B = R * S; L = m.cfg.n_levels
    half = orc.h2f(m.buffer("half")).astype(np.float64)
    table = torch.tensor(half[m.n_mlp:].reshape(-1, 2), dtype=torch.float64, requires_grad=True)
    mats = []; o = 0
    for layer in range(NH + 1):
        rows = 16 if layer == NH else W; cols = Ep if layer == 0 else W
        mats.append(torch.tensor(half[o:o + rows * cols].reshape(rows, cols), dtype=torch.float64, requires_grad=True)); o += rows * cols
    pts = m.buffer("pts").reshape(B, 3)
    feats = [torch.zeros(B, 2, dtype=torch.float64) for _ in range(L)]
    for l, _, idx, w in _numpy_corners(m.cfg, pts):
        feats[l] = feats[l] + torch.tensor(w)[:, None] * table[torch.tensor(idx)]
    # forward values rounded to fp16 where the oracle (tcnn's network precision) rounds them, with a straight-through derivative: without it ~0.02 % of the ReLU
    # units sit on the other side of zero in fp64 and the comparison measures those flips (a 2-3 % norm error), not the backward chain
    h16 = lambda v: v + (v.detach().to(torch.float16).to(torch.float64) - v.detach())
    a = h16(torch.cat(feats + [torch.zeros(B, Ep - 2 * L, dtype=torch.float64)], 1))
    for layer in range(NH):
        a = h16(torch.relu(a @ mats[layer].T))
    out = h16((a @ mats[NH].T)[:, :4]).reshape(R, S, 4)
    t = torch.tensor(m.buffer("tdist").reshape(R, S).astype(np.float64))
    tgt = torch.tensor(m.buffer("target").reshape(R, 3).astype(np.float64)); bg = torch.tensor(m.buffer("bgcol").reshape(R, 3).astype(np.float64))
    flag = m.buffer("ray_flag"); tdep = m.buffer("target_depth").astype(np.float64)
    assert use_depth == bool((tdep > 0).any())
    total = sum(_torch_loss(out[r], t[r], tgt[r], float(tdep[r]), bg[r], int(flag[r])) for r in range(R)) * (m.cfg.loss_scale / R)
    total.backward()
    # MLP matrices: the oracle's gmlp is the loss-scaled fp32 dW, layer by layer (pad rows / columns are zero)
    gm = m.buffer("gmlp").astype(np.float64); o = 0
    for layer in range(NH + 1):
        rows = 16 if layer == NH else W; cols = Ep if layer == 0 else W
        got = gm[o:o + rows * cols].reshape(rows, cols); o += rows * cols; want = mats[layer].grad.numpy()
        if layer == NH:
            want = want.copy(); assert (got[4:] == 0).all(); want[4:] = 0
        rel = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)
        assert rel < 3e-3, "dW of layer %d: relative error %.3e" % (layer, rel)
    # grid: per-entry against the fp16-contribution bound, and in the norm
    gg = m.buffer("ggrid").astype(np.float64).reshape(-1, 2); ga = m.buffer("ggrid_abs").astype(np.float64).reshape(-1, 2); want = table.grad.numpy()
    rel = np.linalg.norm(gg - want) / (np.linalg.norm(want) + 1e-30)
    assert rel < 3e-3, "grid gradient: relative error %.3e" % rel
    bad = np.abs(gg - want) > 4e-3 * ga + 1e-4 * np.abs(want).max()
    assert bad.mean() < 1e-3, "grid gradient: %.3f%% of the entries outside the bound" % (100 * bad.mean())
    m.close()


# ------------------------------------------------------------------ Adam + EMA first step, closed form
def test_first_optimizer_step_closed_form(orc):
    cfg = orc.default_config(**C1)
    m = orc.OracleModel(cfg)
    n, nm = m.n_params, m.n_mlp
    w0 = m.buffer("master").astype(np.float64)
    rs = np.random.RandomState(3)
    gm = (rs.normal(0, 1, nm) * 128).astype(np.float32); gm[::5] = 0.0
    gg = np.zeros(n - nm, np.float32); idx = rs.choice(n - nm, 5000, replace=False); gg[idx] = rs.normal(0, 0.5, 5000)
    gg_h = orc.f2h(gg)
    orc.lib().orc_optimizer_step_with(m.h, _p(gm), _p(gg_h))
    w1 = m.buffer("master").astype(np.float64); steps = m.buffer("steps")
    lr, b1, b2, eps, l2, ls = 1e-2, 0.9, 0.99, 1e-15, 1e-6, 128.0
    g = np.concatenate([gm.astype(np.float64) / ls + l2 * w0[:nm], orc.h2f(gg_h).astype(np.float64) / ls])
    touched = np.ones(n, bool); touched[nm:] = orc.h2f(gg_h) != 0
    step = lr * math.sqrt(1 - b2) / (1 - b1) * ((1 - b1) * g) / (np.sqrt((1 - b2) * g * g) + eps)
    want = np.where(touched, w0 - step, w0)
    assert np.abs(w1 - want).max() < 2e-6
    assert (steps[touched] == 1).all() and (steps[~touched] == 0).all()
    assert np.abs(np.abs(w1 - w0)[touched & (np.abs(g) > 1e-12)] - lr).max() < 1e-5        # first Adam step has magnitude lr
    # EMA after the first step equals the new fp16 weights (debias_old = 0, debias_new = 1/(1-d))
    ema = orc.h2f(m.buffer("ema")); half = orc.h2f(m.buffer("half"))
    assert np.abs(ema - half).max() <= 1e-3 * np.abs(half).max() + 1e-7
    assert m.step == 1
    m.close()


def test_lr_decay_schedule(orc):
    cfg = orc.default_config(**C1); cfg.decay_start = 3; cfg.decay_interval = 2; cfg.decay_base = 0.5
    m = orc.OracleModel(cfg); nm = m.n_mlp
    gm = np.ones(nm, np.float32); gg_h = np.zeros(m.n_params - nm, np.uint16)
    w_prev = m.buffer("master")[:nm].astype(np.float64); deltas = []
    for _ in range(8):
        orc.lib().orc_optimizer_step_with(m.h, _p(gm), _p(gg_h))
        w = m.buffer("master")[:nm].astype(np.float64); deltas.append(np.abs(w - w_prev).mean()); w_prev = w
    # lr halves after steps 3, 5, 7 -> steps 4-5 use lr/2, 6-7 lr/4, 8 lr/8 (constant gradient => |delta| ~ lr)
    r = np.array(deltas) / deltas[0]
    assert np.allclose(r, [1, 1, 1, 0.5, 0.5, 0.25, 0.25, 0.125], rtol=0.05)
    m.close()


# ------------------------------------------------------------------ end-to-end: the restatement learns the scene
def test_oracle_learns_synthetic_object(orc, ss, small_scene):
    import __graft_entry__ as ge
    m = ge.make_oracle(orc, small_scene, C1)
    l0 = m.train(1); l1 = m.train(300)
    assert l1 < 0.1 * l0
    box = small_scene.objects[0]["boxes"][0]
    rgb, depth, mask = m.render(box, ss.colmajor(small_scene.Twc[box[0]]))
    v, x, y, h, w = (int(q) for q in box)
    gt = small_scene.rgb[v, y:y + h, x:x + w] / 255.0; gm = small_scene.instance[v, y:y + h, x:x + w] > 0
    gtw = np.where(gm[..., None], gt, 1.0)
    psnr = -10 * np.log10(((rgb - gtw) ** 2).mean())
    iou = (mask.astype(bool) & gm).sum() / max(1, (mask.astype(bool) | gm).sum())
    assert psnr > 24 and iou > 0.95, (psnr, iou)
    gz = small_scene.depth[v, y:y + h, x:x + w]
    both = mask.astype(bool) & gm
    assert np.abs(depth[both] - gz[both]).mean() < 0.05          # z-depth (depth / d_norm, :1218) in metres
    m.close()
