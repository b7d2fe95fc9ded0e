"""ctypes wrapper around oracle/libmon_oracle.so (TEST INFRASTRUCTURE ONLY -- see mon_oracle.c header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcConfig(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_features", C.c_int32), ("log2_hashmap_size", C.c_int32), ("base_resolution", C.c_int32),
                ("per_level_scale", C.c_float), ("n_neurons", C.c_int32), ("n_hidden_layers", C.c_int32), ("rays_per_batch", C.c_int32),
                ("n_samples", C.c_int32), ("loss_scale", C.c_float), ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("l2_reg", C.c_float), ("ema_decay", C.c_float), ("decay_start", C.c_int32), ("decay_interval", C.c_int32),
                ("decay_base", C.c_float), ("param_seed", C.c_uint32), ("rng_flags", C.c_uint32), ("sample_seed", C.c_uint64),
                ("use_depth", C.c_int32), ("numerics_flags", C.c_int32)]


NUM_GRID_HALF, NUM_TCNN_HALF = 1, 2


class OrcBBox(C.Structure):
    _fields_ = [("FrameId", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32), ("h", C.c_uint32), ("w", C.c_uint32)]


def default_config(**kw):
    c = OrcConfig(n_levels=16, n_features=2, log2_hashmap_size=16, base_resolution=16, per_level_scale=2.0, n_neurons=64, n_hidden_layers=1,
                  rays_per_batch=4096, n_samples=32, loss_scale=128.0, learning_rate=1e-2, beta1=0.9, beta2=0.99, epsilon=1e-15, l2_reg=1e-6,
                  ema_decay=0.95, decay_start=20000, decay_interval=10000, decay_base=0.33, param_seed=1337, sample_seed=2024, use_depth=0,
                  numerics_flags=0)
    flags = 0
    if kw.pop("grid_grad_half_accum", 0):          # ORC_NUM_GRID_HALF: grid gradients accumulated sequentially in fp16 (tcnn: atomicAdd(__half2))
        flags |= NUM_GRID_HALF
    if kw.pop("tcnn_half_accum", 0):               # ORC_NUM_TCNN_HALF: model of tiny-cuda-nn's own fp16 accumulation (encode, MLP fwd/bwd, dW)
        flags |= NUM_TCNN_HALF
    # "same inputs" mode (mon_oracle.c orc_config.rng_flags): xorwow = 0 counter RNG | 1 cuRAND flavour | 2 rocRAND flavour, xorwow_lanes (multiple of 1024,
    # default 4096), tcnn_init_order
    rng = int(kw.pop("xorwow", 0)) | (int(bool(kw.pop("tcnn_init_order", 0))) << 4) | ((int(kw.pop("xorwow_lanes", 0)) // 1024) << 16)
    for k, v in kw.items():
        setattr(c, k, v)
    c.numerics_flags |= flags; c.rng_flags |= rng
    return c


def build(force=False):
    so = os.path.join(_HERE, "libmon_oracle.so")
    src = os.path.join(_HERE, "mon_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


_lib = None

# buffer ids of orc_buffer()
BUF = dict(master=0, half=1, ema=2, m1=3, m2=4, steps=5, gmlp=6, ggrid=7, ggrid_abs=8, ggrid_h=9, pts=10, tdist=11, E=12, Hid=13, O=14, dO=15,
           dHid=16, dE=17, rgb_ray=18, depth_ray=19, mask_ray=20, loss_ray=21, ray_o=22, ray_d=23, ray_t0=24, ray_t1=25, target=26,
           target_depth=27, bgcol=28, ray_flag=29, sel=30, ray_dn=31)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p; L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ("orc_n_params", "orc_n_mlp_params", "orc_step", "orc_n_valid"):
            getattr(L, f).restype = C.c_uint32; getattr(L, f).argtypes = [C.c_void_p]
        L.orc_loss.restype = C.c_float; L.orc_loss.argtypes = [C.c_void_p]
        L.orc_epad.restype = C.c_int; L.orc_epad.argtypes = [C.c_void_p]
        L.orc_buffer.restype = C.c_void_p; L.orc_buffer.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_params.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_dataset.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_object.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_add_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_train_step.restype = C.c_uint32; L.orc_train_step.argtypes = [C.c_void_p]
        L.orc_train.restype = C.c_float; L.orc_train.argtypes = [C.c_void_p, C.c_int]
        L.orc_generate_batch.argtypes = [C.c_void_p]; L.orc_forward_backward.argtypes = [C.c_void_p]
        L.orc_optimizer_step_with.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_optimizer_step_with_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, OrcBBox, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_density_grid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_mlp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_composite.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_step_variant.argtypes = [C.c_void_p, C.c_int]; L.orc_n_compacted.restype = C.c_uint32; L.orc_n_compacted.argtypes = [C.c_void_p]
        L.orc_xorwow_lane_draws.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_xorwow_generate.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_xorwow_generate_calls.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_gradient.restype = C.c_float
        L.orc_gradient.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float,
                C.c_void_p]
        L.orc_level_table.restype = C.c_int; L.orc_level_table.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_grid_index.restype = C.c_uint32; L.orc_grid_index.argtypes = [C.c_uint32] * 5
        L.orc_rand01.restype = C.c_float; L.orc_rand01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_f2h.restype = C.c_uint16; L.orc_f2h.argtypes = [C.c_float]
        L.orc_h2f.restype = C.c_float; L.orc_h2f.argtypes = [C.c_uint16]
        L.orc_set_threads.argtypes = [C.c_int]; L.orc_max_threads.restype = C.c_int
        L.orc_set_ema.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_mc_case.restype = C.c_uint64; L.orc_mc_case.argtypes = [C.c_int]
        L.orc_mc_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_mc_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                C.c_uint32]
        L.orc_mesh_to_cpu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_mesh_colors.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_set_parallel_scatter.argtypes = [C.c_int]; L.orc_advance_iter.argtypes = [C.c_void_p]
        # the checker's loops are small: a modest team beats one thread per hardware thread on a 256-thread host
        L.orc_set_threads(int(os.environ.get("MON_ORACLE_THREADS", min(16, os.cpu_count() or 1))))
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleModel:
    """One object NeRF on the CPU (mirror of ro-map_amd binding.ObjectNeRF for the checker)."""

    def __init__(self, cfg):
        self.L = lib(); self.cfg = cfg
        self.h = self.L.orc_create(C.byref(cfg))
        self._keep = []
        self.R, self.S = cfg.rays_per_batch, cfg.n_samples
        self.n_params = self.L.orc_n_params(self.h); self.n_mlp = self.L.orc_n_mlp_params(self.h); self.Epad = self.L.orc_epad(self.h)
        self.W, self.NH = cfg.n_neurons, cfg.n_hidden_layers

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h); self.h = None

    def set_dataset(self, H, W, fx, fy, cx, cy, rgba, depth, poses):
        rgba = np.ascontiguousarray(rgba, np.uint8); poses = np.ascontiguousarray(poses, np.float32)
        depth = None if depth is None else np.ascontiguousarray(depth, np.float32)
        self._keep += [rgba, depth, poses]
        self.L.orc_set_dataset(self.h, H, W, rgba.shape[0], fx, fy, cx, cy, _p(rgba), _p(depth), _p(poses))

    def set_ema(self, ema_u16):
        e = np.ascontiguousarray(ema_u16, np.uint16); self.L.orc_set_ema(self.h, _p(e))

    def set_object(self, Tow16, amin, amax, inst):
        a, b, c = (np.ascontiguousarray(v, np.float32) for v in (Tow16, amin, amax))
        self.L.orc_set_object(self.h, _p(a), _p(b), _p(c), int(inst)); self._amin, self._amax = b.copy(), c.copy()

    def add_boxes(self, boxes):
        b = np.ascontiguousarray(boxes, np.uint32).reshape(-1, 5)
        self.L.orc_add_boxes(self.h, _p(b), b.shape[0])

    def set_params(self, master):
        m = np.ascontiguousarray(master, np.float32); assert m.size == self.n_params
        self.L.orc_set_params(self.h, _p(m))

    def train(self, iters):
        return self.L.orc_train(self.h, iters)

    def optimizer_step_with(self, gmlp, ggrid_f32):
        """Trainer::optimizer_step on externally supplied gradients (fp32 MLP gradient, fp32 grid gradient), still loss-scaled."""
        a = np.ascontiguousarray(gmlp, np.float32); b = np.ascontiguousarray(ggrid_f32, np.float32)
        assert a.size == self.n_mlp and b.size == self.n_params - self.n_mlp
        self.L.orc_optimizer_step_with_f32(self.h, _p(a), _p(b))

    def set_step_variant(self, on):
        self.L.orc_set_step_variant(self.h, int(on))

    @property
    def n_compacted(self):
        return self.L.orc_n_compacted(self.h)

    def advance_iter(self):
        self.L.orc_advance_iter(self.h)

    def generate_batch(self):
        self.L.orc_generate_batch(self.h)

    def forward_backward(self):
        self.L.orc_forward_backward(self.h)

    def train_step(self):
        return self.L.orc_train_step(self.h)

    @property
    def n_valid(self):
        return self.L.orc_n_valid(self.h)

    @property
    def loss(self):
        return self.L.orc_loss(self.h)

    @property
    def step(self):
        return self.L.orc_step(self.h)

    def buffer(self, name):
        R, B, n = self.R, self.R * self.S, self.n_params
        n_grid = n - self.n_mlp
        shapes = dict(master=(np.float32, n), half=(np.uint16, n), ema=(np.uint16, n), m1=(np.float32, n), m2=(np.float32, n), steps=(np.uint32, n),
                      gmlp=(np.float32, self.n_mlp), ggrid=(np.float32, n_grid), ggrid_abs=(np.float32, n_grid), ggrid_h=(np.uint16, n_grid),
                      pts=(np.float32, B * 3), tdist=(np.float32, B), E=(np.uint16, B * self.Epad), Hid=(np.uint16, B * self.W * self.NH),
                      O=(np.uint16, B * 4), dO=(np.uint16, B * 4), dHid=(np.uint16, B * self.W * self.NH), dE=(np.uint16, B * self.Epad),
                      rgb_ray=(np.float32, R * 3), depth_ray=(np.float32, R), mask_ray=(np.float32, R), loss_ray=(np.float32, R),
                      ray_o=(np.float32, R * 3), ray_d=(np.float32, R * 3), ray_t0=(np.float32, R), ray_t1=(np.float32, R), target=(np.float32, R * 3),
                      target_depth=(np.float32, R), bgcol=(np.float32, R * 3), ray_flag=(np.uint8, R), sel=(np.uint32, R), ray_dn=(np.float32, R))
        dt, cnt = shapes[name]
        ptr = self.L.orc_buffer(self.h, BUF[name])
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()

    def render(self, box, pose16, pose_is_Toc=False, use_ema=True):
        FrameId, x, y, h, w = (int(v) for v in box)
        rgb = np.empty((h, w, 3), np.float32); depth = np.empty((h, w), np.float32); mask = np.empty((h, w), np.float32)
        pose = np.ascontiguousarray(pose16, np.float32)
        self.L.orc_render(self.h, OrcBBox(FrameId, x, y, h, w), _p(pose), int(pose_is_Toc), int(use_ema), _p(rgb), _p(depth), _p(mask))
        return rgb, depth, mask

    def density_grid(self, rx, ry, rz, use_ema=True):
        out = np.empty(rx * ry * rz, np.float32)
        self.L.orc_density_grid(self.h, rx, ry, rz, int(use_ema), _p(out))
        return out

    def mesh_colors(self, verts, use_ema=True):
        v = np.ascontiguousarray(verts, np.float32); col = np.empty_like(v)
        self.L.orc_mesh_colors(self.h, _p(v), v.shape[0], int(use_ema), _p(col)); return col

    def generate_mesh(self, res=64, thresh=2.0, use_ema=True):
        """GenerateMesh + TransCPUMesh (nerf_model.cu:1993-2095)."""
        dens = self.density_grid(res, res, res, use_ema)
        mesh = marching_cubes(dens, (res, res, res), thresh, self._amin, self._amax)
        n = mesh["verts"].shape[0]; col = np.empty((n, 3), np.float32)
        self.L.orc_mesh_colors(self.h, _p(mesh["verts"]), n, int(use_ema), _p(col))
        mesh["colors_f32"] = col; mesh["density"] = dens
        mesh["normals"], mesh["colors"] = mesh_to_cpu(mesh["normals_raw"], col)
        return mesh


def marching_cubes(density, res3, thresh, amin, amax):
    """MarchingCubes + compute_mesh_1ring (marching_cubes.cu:478-509, 655-665) in the fixed order of mon_mesh_oracle.c."""
    L = lib(); rx, ry, rz = (int(v) for v in res3)
    d = np.ascontiguousarray(density, np.float32).reshape(-1); assert d.size == rx * ry * rz
    nv = C.c_uint32(0); ni = C.c_uint32(0)
    L.orc_mc_count(_p(d), rx, ry, rz, float(thresh), C.byref(nv), C.byref(ni))
    npad = (nv.value + 127) & ~127
    verts = np.zeros((npad, 3), np.float32); nraw = np.zeros((npad, 3), np.float32); idx = np.zeros(ni.value, np.uint32)
    vgrid = np.zeros(3 * d.size, np.int32)
    a0 = np.ascontiguousarray(amin, np.float32); a1 = np.ascontiguousarray(amax, np.float32)
    L.orc_mc_extract(_p(d), rx, ry, rz, float(thresh), _p(a0), _p(a1), _p(verts), _p(vgrid), _p(idx), _p(nraw), npad)
    return dict(verts=verts, normals_raw=nraw, indices=idx, n_verts_real=nv.value, vertidx=vgrid)


def mesh_to_cpu(normals_raw, colors):
    n = normals_raw.shape[0]; nrm = np.empty((n, 3), np.float32); c8 = np.empty((n, 3), np.uint8)
    lib().orc_mesh_to_cpu(_p(np.ascontiguousarray(normals_raw, np.float32)), _p(np.ascontiguousarray(colors, np.float32)), n, _p(nrm), _p(c8))
    return nrm, c8


def h2f(a):
    return np.asarray(a, np.uint16).view(np.float16).astype(np.float32)


def f2h(a):
    return np.asarray(a, np.float32).astype(np.float16).view(np.uint16)
