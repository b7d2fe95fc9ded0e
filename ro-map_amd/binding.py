"""ctypes binding of libmon_core.so.  Mirrors the reference's manager/object interface
(CORE/include/nerf_manager.h, nerf.h) one call per C-ABI entry point; no compute happens in Python and
there is no CPU fallback: every compute call raises MonError when the HIP library or a device is missing."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class MonError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mon_core error %d: %s" % (code, msg)); self.code = code


class MonConfig(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_features", C.c_int32), ("log2_hashmap_size", C.c_int32), ("base_resolution", C.c_int32),
                ("per_level_scale", C.c_float), ("n_neurons", C.c_int32), ("n_hidden_layers", C.c_int32), ("rays_per_batch", C.c_int32),
                ("n_samples", C.c_int32), ("loss_scale", C.c_float), ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("l2_reg", C.c_float), ("ema_decay", C.c_float), ("decay_start", C.c_int32), ("decay_interval", C.c_int32),
                ("decay_base", C.c_float), ("param_seed", C.c_uint32), ("rng_flags", C.c_uint32), ("sample_seed", C.c_uint64),
                ("use_depth", C.c_int32), ("occupancy_skip", C.c_int32)]


class MonBBox(C.Structure):
    _fields_ = [("FrameId", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32), ("h", C.c_uint32), ("w", C.c_uint32)]


class MonInfo(C.Structure):
    _fields_ = [("n_params", C.c_uint32), ("n_mlp_params", C.c_uint32), ("n_grid_params", C.c_uint32), ("encoded_width", C.c_uint32),
                ("train_step", C.c_uint32), ("n_boxes", C.c_uint32), ("last_n_valid", C.c_uint32), ("device", C.c_int32),
                ("last_loss", C.c_float), ("learning_rate", C.c_float), ("backend", C.c_int32), ("skipped_batches", C.c_uint32)]


class MonProfile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8)]


K_BATCH, K_FWDBWD, K_OPTIM, K_RENDER = 0, 1, 2, 3

# every symbol include/mon_core.h declares (checked by tests/test_abi.py against the header text)
_SIGS = {
    "mon_last_error": (C.c_char_p, []),
    "mon_version": (C.c_int, []),
    "mon_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mon_set_logical_devices": (C.c_int, [C.c_int]),
    "mon_offline_set_schedule": (C.c_int, [C.c_int, C.c_int]),
    "mon_set_option": (C.c_int, [C.c_char_p, C.c_long]),
    "mon_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_long)]),
    "mon_config_default": (C.c_int, [C.POINTER(MonConfig)]),
    "mon_config_from_json": (C.c_int, [C.c_char_p, C.POINTER(MonConfig)]),
    "mon_dataset_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "mon_dataset_add_frame": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_dataset_n_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "mon_dataset_destroy": (C.c_int, [C.c_void_p]),
    "mon_object_create": (C.c_int, [C.c_void_p, C.POINTER(MonConfig), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mon_object_add_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mon_object_train": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "mon_object_render": (C.c_int, [C.c_void_p, MonBBox, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mon_object_render_snapshot": (C.c_int, [C.c_void_p, MonBBox, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "mon_object_density_grid": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mon_object_info_get": (C.c_int, [C.c_void_p, C.POINTER(MonInfo)]),
    "mon_object_get_params": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "mon_object_set_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mon_object_train_stages": (C.c_int, [C.c_void_p, C.c_int]),
    "mon_object_set_backend": (C.c_int, [C.c_void_p, C.c_int]),
    "mon_object_set_debug_dump": (C.c_int, [C.c_void_p, C.c_int]),
    "mon_object_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mon_object_get_profile": (C.c_int, [C.c_void_p, C.POINTER(MonProfile), C.c_int]),
    "mon_object_destroy": (C.c_int, [C.c_void_p]),
    "mon_offline_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mon_offline_init": (C.c_int, [C.c_void_p]),
    "mon_offline_read_dataset": (C.c_int, [C.c_void_p]),
    "mon_offline_create_nerf": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mon_offline_wait_threads_end": (C.c_int, [C.c_void_p]),
    "mon_offline_n_objects": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mon_offline_object_loss": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mon_offline_render_test": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "mon_offline_save_mesh": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p]),
    "mon_offline_destroy": (C.c_int, [C.c_void_p]),
    "mon_object_generate_mesh": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mon_object_mesh_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mon_object_get_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mon_device_mem_info": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "mon_object_mesh_generation": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mon_object_get_config": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mon_object_copy_mesh": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int]),
    "mon_object_get_mesh_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_object_save_mesh": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mon_marching_cubes": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mon_offline_get_intrinsics": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_float)] * 4 + [C.POINTER(C.c_int)] * 2),
    "mon_offline_get_poses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mon_offline_object_meta": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
            C.POINTER(C.c_size_t)]),
    "mon_offline_set_output_dir": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mon_offline_object": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mon_online_object": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mon_online_render_nerfs_test": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "mon_generate_toc": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mon_online_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mon_online_init": (C.c_int, [C.c_void_p]),
    "mon_online_dataset_init": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_size_t]),
    "mon_online_new_frame": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_online_create_nerf": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    "mon_online_update_nerf_bbox": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]),
    "mon_online_get_frame_idx": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "mon_online_update_dataset": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "mon_online_get_pose": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "mon_online_wait_threads_end": (C.c_int, [C.c_void_p]),
    "mon_online_object_info": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "mon_online_render": (C.c_int, [C.c_void_p, C.c_size_t, MonBBox, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_online_destroy": (C.c_int, [C.c_void_p]),
    "mon_png_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    "mon_png_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mon_write_render_pngs": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_physical_device": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "mon_offline_object_stamp": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_char_p, C.c_size_t]),
    "mon_device_synchronize": (C.c_int, [C.c_int]),
}


# every symbol include/mon_core_diag.h declares (libmon_core_diag.so: diagnostics and test scaffolding, not the product)
_DIAG_SIGS = {
    "mon_object_debug_read": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "mon_dataset_debug_read": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_microbench": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "mon_debug_frag_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mon_debug_acc_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "mon_debug_fast_index": (C.c_int, [C.POINTER(MonConfig), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mon_selftest_mfma": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_debug_yaml_number": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_double)]),
    "mon_debug_render_jobs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]),
    "mon_debug_occupancy_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
}


def exported_symbols():
    return sorted(_SIGS)


# libmon_core_rccl.so (include/mon_core_rccl.h): the in-process gather-to-root over RCCL
_RCCL_SIGS = {
    "mon_gather_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mon_gather_destroy": (C.c_int, [C.c_void_p]),
    "mon_gather_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mon_gather_renders": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mon_gather_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "mon_gather_transport_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "mon_gather_set_transport": (C.c_int, [C.c_void_p, C.c_int]),
    "mon_gather_last_error": (C.c_char_p, []),
    "mon_offline_render_test_gathered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]),
}
_rccl_lib = None


def rccl_symbols():
    return sorted(_RCCL_SIGS)


def rccl_lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmon_core_rccl.so")


def rccl_lib():
    """libmon_core_rccl.so next to libmon_core.so (the core is loaded first: the gather library resolves its symbols against it)."""
    global _rccl_lib
    if _rccl_lib is None:
        lib()
        L = C.CDLL(rccl_lib_path(), mode=C.RTLD_GLOBAL)
        for name, (res, args) in _RCCL_SIGS.items():
            f = getattr(L, name); f.restype = res; f.argtypes = args
        _rccl_lib = L
    return _rccl_lib


def gather_plan(object_device, n_pix, n_devices):
    d = np.ascontiguousarray(object_device, np.int32); p = np.ascontiguousarray(n_pix, np.uint32)
    per = np.zeros(n_devices, np.uint64); off = np.zeros(len(d), np.uint64)
    rc = rccl_lib().mon_gather_plan(_p(d), _p(p), len(d), int(n_devices), _p(per), _p(off))
    if rc:
        raise MonError(rc, "mon_gather_plan failed")
    return per, off


def _gather_error():
    return (rccl_lib().mon_gather_last_error() or b"").decode("utf-8", "replace")


class Gather:
    """mon_gather: single-process RCCL communicator over the visible devices + the gather-to-root of rendered crops."""

    def __init__(self, root_device=0):
        self.h = C.c_void_p()
        rc = rccl_lib().mon_gather_create(int(root_device), C.byref(self.h))
        if rc:
            raise MonError(rc, "mon_gather_create: " + _gather_error())

    AUTO, RCCL, PEER_COPY = 0, 1, 2

    def set_transport(self, transport):
        rc = rccl_lib().mon_gather_set_transport(self.h, int(transport))
        if rc:
            raise MonError(rc, "mon_gather_set_transport: " + _gather_error())

    def renders(self, objects, boxes, poses16, pose_is_Toc=False):
        n = len(objects); b = np.ascontiguousarray(boxes, np.uint32).reshape(n, 5); T = np.ascontiguousarray(poses16, np.float32).reshape(n, 16)
        oh = (C.c_void_p * n)(*[o.h for o in objects])
        rgb = [np.empty((int(q[3]), int(q[4]), 3), np.float32) for q in b]; dep = [np.empty((int(q[3]), int(q[4])), np.float32) for q in b]
        msk = [np.empty_like(d) for d in dep]
        pr = (C.c_void_p * n)(*[a.ctypes.data for a in rgb]); pd = (C.c_void_p * n)(*[a.ctypes.data for a in dep])
        pm = (C.c_void_p * n)(*[a.ctypes.data for a in msk])
        rc = rccl_lib().mon_gather_renders(self.h, oh, _p(b), _p(T), int(pose_is_Toc), n, pr, pd, pm)
        if rc:
            raise MonError(rc, "mon_gather_renders: " + _gather_error())
        return list(zip(rgb, dep, msk))

    def stats(self):
        a = C.c_uint64(0); b = C.c_uint64(0); s = C.c_int(0); ms = C.c_double(0)
        rccl_lib().mon_gather_stats(self.h, C.byref(a), C.byref(b), C.byref(s), C.byref(ms))
        nr = C.c_int(0); br = C.c_uint64(0); mr = C.c_int(0); bc = C.c_uint64(0); mc = C.c_int(0)
        rccl_lib().mon_gather_transport_stats(self.h, C.byref(nr), C.byref(br), C.byref(mr), C.byref(bc), C.byref(mc))
        return dict(bytes_over_links=a.value, bytes_on_root=b.value, sending_devices=s.value, transfer_ms=ms.value, n_ranks=nr.value, bytes_rccl=br.value,
                messages_rccl=mr.value, bytes_peer_copy=bc.value, messages_peer_copy=mc.value)

    def offline_render_test(self, manager, out_dir, max_views=0):
        rc = rccl_lib().mon_offline_render_test_gathered(self.h, manager.h, out_dir.encode(), int(max_views))
        if rc:
            raise MonError(rc, "mon_offline_render_test_gathered: " + _gather_error())

    def close(self):
        if self.h:
            rccl_lib().mon_gather_destroy(self.h); self.h = C.c_void_p()


def diag_symbols():
    return sorted(_DIAG_SIGS)


def lib_path():
    """In-tree library; MON_CORE_LIB points tooling at an instrumented / experimental build of the same sources (tools/variant_build.sh)."""
    return os.environ.get("MON_CORE_LIB") or os.path.join(_HERE, "libmon_core.so")


_lib = None


def lib():
    """Loads libmon_core.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise MonError(-1, "libmon_core.so not built (run python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(p)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
        _lib = L
        # harness convenience (tools / tests that run in a subprocess): MON_OPTIONS="name=value,..." -> mon_set_option calls
        for kv in filter(None, os.environ.get("MON_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            if k.strip() == "offline_schedule":          # "offline_schedule=OUTERxINNER"
                o, i = v.lower().split("x"); rc = L.mon_offline_set_schedule(int(o), int(i))
            else:
                rc = L.mon_set_option(k.strip().encode(), int(v))
            if rc != 0:
                raise MonError(rc, L.mon_last_error().decode("utf-8", "replace"))
    return _lib


_diag = None


def diag_lib_path():
    return os.path.join(os.path.dirname(lib_path()), "libmon_core_diag.so")


def diag_lib():
    """Loads libmon_core_diag.so (after libmon_core.so, which it links against)."""
    global _diag
    if _diag is None:
        lib()
        p = diag_lib_path()
        if not os.path.exists(p):
            raise MonError(-1, "libmon_core_diag.so not built")
        L = C.CDLL(p)
        for name, (res, args) in _DIAG_SIGS.items():
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
        _diag = L
    return _diag


def set_option(name, value):
    """Process-wide test / tuning switch (include/mon_core.h: mon_set_option)."""
    _check(lib().mon_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_long(0); _check(lib().mon_get_option(name.encode(), C.byref(v))); return v.value


def yaml_number(text, key):
    v = C.c_double(0); _check(diag_lib().mon_debug_yaml_number(text.encode(), key.encode(), C.byref(v))); return v.value


def _check(rc):
    if rc != 0:
        raise MonError(rc, lib().mon_last_error().decode("utf-8", "replace"))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    rc = lib().mon_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_offline_schedule(outer=10, inner=500):
    """NerfManagerOffline's outer x inner training iterations per object (reference: 10 x 500), read by Offline.init."""
    _check(lib().mon_offline_set_schedule(int(outer), int(inner)))


def set_logical_devices(n):
    """n logical devices mapped round-robin onto the physical GPUs (0 = the physical devices themselves)."""
    _check(lib().mon_set_logical_devices(int(n)))


def device_mem_info(device=0):
    """(free, total) bytes of device memory (hipMemGetInfo)."""
    f = C.c_size_t(0); t = C.c_size_t(0); _check(lib().mon_device_mem_info(int(device), C.byref(f), C.byref(t))); return f.value, t.value


def default_config(**kw):
    c = MonConfig(); _check(lib().mon_config_default(C.byref(c)))
    kw = dict(kw)
    # "same inputs" mode (mon_config.rng_flags): xorwow = 0 counter RNG | 1 cuRAND flavour | 2 rocRAND flavour, xorwow_lanes (multiple of 1024, default 4096),
    # tcnn_init_order
    rng = int(kw.pop("xorwow", 0)) | (int(bool(kw.pop("tcnn_init_order", 0))) << 4) | ((int(kw.pop("xorwow_lanes", 0)) // 1024) << 16)
    for k, v in kw.items():
        setattr(c, k, v)
    c.rng_flags |= rng
    return c


def config_from_json(path):
    c = MonConfig(); _check(lib().mon_config_from_json(path.encode(), C.byref(c))); return c


def selftest_mfma(A_h, B_h, device=0):
    A = np.ascontiguousarray(A_h, np.uint16); B = np.ascontiguousarray(B_h, np.uint16); D = np.empty((32, 32), np.float32)
    _check(diag_lib().mon_selftest_mfma(device, _p(A), _p(B), _p(D))); return D


def fast_index(cfg, level, x, y, z):
    i = C.c_uint32(0); n = C.c_uint32(0); _check(diag_lib().mon_debug_fast_index(C.byref(cfg), level, x, y, z, C.byref(i), C.byref(n))); return i.value, n.value


def microbench(mode, pattern, n_entries, n_ops, device=0):
    ms = C.c_float(0); _check(diag_lib().mon_microbench(device, mode, pattern, n_entries, n_ops, C.byref(ms))); return ms.value


class Dataset:
    """nerf::NeRF_Dataset on one device (frames resident in HBM)."""

    def __init__(self, device, H, W, fx, fy, cx, cy, max_frames, use_depth=False):
        self.h = C.c_void_p(); self.H, self.W = H, W
        _check(lib().mon_dataset_create(device, H, W, fx, fy, cx, cy, max_frames, int(use_depth), C.byref(self.h)))

    def add_frame(self, frame_id, rgb_u8, instance_u8, Twc16, depth=None, is_bgr=False):
        rgb = np.ascontiguousarray(rgb_u8, np.uint8); inst = np.ascontiguousarray(instance_u8, np.uint8)
        pose = np.ascontiguousarray(Twc16, np.float32); d = None if depth is None else np.ascontiguousarray(depth, np.float32)
        assert rgb.shape[:2] == (self.H, self.W) and inst.shape == (self.H, self.W) and pose.size == 16
        _check(lib().mon_dataset_add_frame(self.h, frame_id, _p(rgb), rgb.shape[2], int(is_bgr), _p(inst), _p(d), _p(pose)))

    @property
    def n_frames(self):
        n = C.c_uint32(0); _check(lib().mon_dataset_n_frames(self.h, C.byref(n))); return n.value

    def debug_read(self, frame_id, with_depth=False):
        """(diagnostics library) what the device holds for one frame: packed rgba | instance << 24, depth or None, pose (Twc, column-major)."""
        rgba = np.empty((self.H, self.W), np.uint32); pose = np.empty(16, np.float32); dep = np.empty((self.H, self.W), np.float32) if with_depth else None
        _check(diag_lib().mon_dataset_debug_read(self.h, frame_id, _p(rgba), _p(dep), _p(pose)))
        return rgba, dep, pose

    def close(self):
        if self.h:
            lib().mon_dataset_destroy(self.h); self.h = None


# debug buffer ids (ro-map_amd/csrc/model.h MON_BUF_*): name -> (id, dtype, elements as f(R, B, info))
BUF = dict(master=0, half=1, ema=2, m1=3, m2=4, steps=5, gmlp=6, ggrid_h=9, pts=10, tdist=11, E=12, Hid=13, O=14, dO=15, dHid=16, dE=17,
           rgb_ray=18, depth_ray=19, mask_ray=20, loss_ray=21, ray_o=22, ray_d=23, ray_t0=24, ray_t1=25, target=26, target_depth=27, bgcol=28,
           ray_flag=29, ray_dn=31, mask=32, state=33, frag_train=34, frag_ref=35, x_all=36, e_soa=37, half_tiles=38, ggrid_f32=39, live_cnt=40)


class ObjectNeRF:
    """nerf::NeRF + nerf::NeRF_Model for one object."""

    def __init__(self, dataset, cfg, class_id, Tow16, aabb_min, aabb_max):
        self.h = C.c_void_p(); self.cfg = cfg; self.ds = dataset
        a, b, c = (np.ascontiguousarray(v, np.float32) for v in (Tow16, aabb_min, aabb_max))
        _check(lib().mon_object_create(dataset.h, C.byref(cfg), int(class_id), _p(a), _p(b), _p(c), C.byref(self.h)))
        self.R, self.S = cfg.rays_per_batch, cfg.n_samples

    def close(self):
        if self.h:
            lib().mon_object_destroy(self.h); self.h = None

    def add_boxes(self, boxes):
        b = np.ascontiguousarray(boxes, np.uint32).reshape(-1, 5)
        _check(lib().mon_object_add_boxes(self.h, _p(b), b.shape[0]))

    def train(self, iters):
        loss = C.c_float(0); _check(lib().mon_object_train(self.h, iters, C.byref(loss))); return loss.value

    def train_stages(self, bits):
        _check(lib().mon_object_train_stages(self.h, bits))

    def set_backend(self, backend):
        _check(lib().mon_object_set_backend(self.h, backend))

    def info(self):
        i = MonInfo(); _check(lib().mon_object_info_get(self.h, C.byref(i))); return i

    def render(self, box, pose16, pose_is_Toc=False):
        FrameId, x, y, h, w = (int(v) for v in box)
        rgb = np.empty((h, w, 3), np.float32); depth = np.empty((h, w), np.float32); mask = np.empty((h, w), np.float32)
        pose = np.ascontiguousarray(pose16, np.float32)
        _check(lib().mon_object_render(self.h, MonBBox(FrameId, x, y, h, w), _p(pose), int(pose_is_Toc), _p(rgb), _p(depth), _p(mask), 0))
        return rgb, depth, mask

    def render_snapshot(self, box, pose16, pose_is_Toc=False):
        """Viewer-side render from the last published inference weights on the inference stream (safe while another thread trains this object);
        returns (rgb, depth, mask, optimizer steps of the weights)."""
        FrameId, x, y, h, w = (int(v) for v in box)
        rgb = np.empty((h, w, 3), np.float32); depth = np.empty((h, w), np.float32); mask = np.empty((h, w), np.float32); st = C.c_uint32(0)
        pose = np.ascontiguousarray(pose16, np.float32)
        _check(lib().mon_object_render_snapshot(self.h, MonBBox(FrameId, x, y, h, w), _p(pose), int(pose_is_Toc), _p(rgb), _p(depth), _p(mask), C.byref(st)))
        return rgb, depth, mask, st.value

    def render_into(self, box, pose16, rgb_ptr, depth_ptr, mask_ptr, on_device, pose_is_Toc=False):
        """NeRF_Model::Render straight into caller-owned buffers given as raw addresses (3hw + hw + hw float32); `on_device`: they are
        HBM addresses of this object's device (e.g. a torch tensor's data_ptr() -- the final-render gather sends them over RCCL as they are)."""
        FrameId, x, y, h, w = (int(v) for v in box)
        pose = np.ascontiguousarray(pose16, np.float32)
        _check(lib().mon_object_render(self.h, MonBBox(FrameId, x, y, h, w), _p(pose), int(pose_is_Toc), C.c_void_p(int(rgb_ptr)), C.c_void_p(int(depth_ptr)),
                C.c_void_p(int(mask_ptr)), int(bool(on_device))))

    def generate_mesh(self, res=64, thresh=2.0):
        """GenerateMesh + TransCPUMesh (nerf_model.cu:1993-2095); returns (n_verts incl. padding, n_indices)."""
        nv = C.c_uint32(0); ni = C.c_uint32(0)
        _check(lib().mon_object_generate_mesh(self.h, int(res), float(thresh), C.byref(nv), C.byref(ni))); return nv.value, ni.value

    def get_mesh(self, try_lock=False, raw=False):
        """CPUMeshData (common.h:32-41) as a dict of numpy arrays.  Counts and data come from one hold of the mesh mutex
        (mon_object_copy_mesh), so this is safe from a viewer thread while the object's thread republishes the mesh."""
        nv = C.c_uint32(0); nr = C.c_uint32(0); ni = C.c_uint32(0)
        _check(lib().mon_object_mesh_counts(self.h, C.byref(nv), C.byref(nr), C.byref(ni)))
        for _ in range(8):
            cv, ci = nv.value, ni.value
            out = dict(verts=np.empty((cv, 3), np.float32), normals=np.empty((cv, 3), np.float32), colors=np.empty((cv, 3), np.uint8), indices=np.empty(ci,
                    np.uint32))
            rc = lib().mon_object_copy_mesh(self.h, cv, ci, _p(out["verts"]), _p(out["normals"]), _p(out["colors"]), _p(out["indices"]),
                                            C.byref(nv), C.byref(nr), C.byref(ni), int(try_lock))
            if rc == 1 and (nv.value > cv or ni.value > ci):
                continue                                            # the mesh grew in between: retry with the reported counts
            _check(rc); break
        else:
            raise MonError(1, "get_mesh: mesh kept growing")
        out = {k: (v[:nv.value] if k != "indices" else v[:ni.value]) for k, v in out.items()}; out["n_verts_real"] = nr.value
        if raw:
            out["normals_raw"] = np.empty((nv.value, 3), np.float32); out["colors_f32"] = np.empty((nv.value, 3), np.float32)
            _check(lib().mon_object_get_mesh_raw(self.h, _p(out["normals_raw"]), _p(out["colors_f32"])))
        return out

    def mesh_generation(self):
        g = C.c_uint64(0); _check(lib().mon_object_mesh_generation(self.h, C.byref(g))); return g.value

    def save_mesh(self, path):
        _check(lib().mon_object_save_mesh(self.h, path.encode()))

    def density_grid(self, rx, ry, rz):
        out = np.empty(rx * ry * rz, np.float32); _check(lib().mon_object_density_grid(self.h, rx, ry, rz, _p(out))); return out

    def get_params(self, which=0):
        n = self.info().n_params
        out = np.empty(n, np.float32 if which == 0 else np.uint16)
        _check(lib().mon_object_get_params(self.h, which, _p(out), out.nbytes)); return out

    def set_params(self, master):
        m = np.ascontiguousarray(master, np.float32); _check(lib().mon_object_set_params(self.h, _p(m), m.size))

    def buffer(self, name):
        i = self.info(); R, B, n = self.R, self.R * self.S, i.n_params
        W, NH, Ep = self.cfg.n_neurons, self.cfg.n_hidden_layers, i.encoded_width
        shapes = dict(master=(np.float32, n), half=(np.uint16, n), ema=(np.uint16, n), m1=(np.float32, n), m2=(np.float32, n), steps=(np.uint32, n),
                      gmlp=(np.float32, i.n_mlp_params), ggrid_h=(np.uint16, i.n_grid_params), ggrid_f32=(np.float32, i.n_grid_params), pts=(np.float32, B * 3), tdist=(np.float32, B),
                      E=(np.uint16, B * Ep), Hid=(np.uint16, B * W * NH), O=(np.uint16, B * 4), dO=(np.uint16, B * 4), dHid=(np.uint16, B * W * NH),
                      dE=(np.uint16, B * Ep), rgb_ray=(np.float32, R * 3), depth_ray=(np.float32, R), mask_ray=(np.float32, R), loss_ray=(np.float32, R),
                      ray_o=(np.float32, R * 3), ray_d=(np.float32, R * 3), ray_t0=(np.float32, R), ray_t1=(np.float32, R), target=(np.float32, R * 3),
                      target_depth=(np.float32, R), bgcol=(np.float32, R * 3), ray_flag=(np.uint8, R), ray_dn=(np.float32, R), mask=(np.uint64, R // 64),
                      state=(np.uint32, 28 + 2 * 128 * 16), frag_train=(np.uint16, 64 * 512), frag_ref=(np.uint16, 64 * 512),
                      live_cnt=(np.uint32, 2 * 64 * 16), x_all=(np.float32, B * 4), e_soa=(np.uint16, B * 2 * self.cfg.n_levels), half_tiles=(np.uint16, i.n_grid_params))
        dt, cnt = shapes[name]; out = np.empty(cnt, dt)
        _check(diag_lib().mon_object_debug_read(self.h, BUF[name], _p(out), out.nbytes)); return out

    def occupancy_state(self):
        out = (C.c_uint32 * 2)(); _check(diag_lib().mon_debug_occupancy_state(self.h, out)); return int(out[0]), int(out[1])

    def render_jobs(self, side=0):
        """Jobs (rays that hit the box) of the last crop the tile render evaluated on this object's device (side 0: train stream, 1: inference stream)."""
        n = C.c_uint32(0); _check(diag_lib().mon_debug_render_jobs(self.h, int(side), C.byref(n))); return n.value

    def set_debug_dump(self, on):
        _check(lib().mon_object_set_debug_dump(self.h, int(on)))

    def set_profiling(self, on):
        _check(lib().mon_object_set_profiling(self.h, int(on)))

    def profile(self, reset=True):
        p = MonProfile(); _check(lib().mon_object_get_profile(self.h, C.byref(p), int(reset)))
        return {"ms": list(p.ms), "launches": list(p.launches)}


def png_read(path):
    w, h, c, d = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _check(lib().mon_png_read(path.encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(d), None, 0))
    buf = np.empty(w.value * h.value * c.value * d.value // 8, np.uint8)
    _check(lib().mon_png_read(path.encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(d), _p(buf), buf.nbytes))
    if d.value == 16:
        return buf.view(">u2").astype(np.uint16).reshape(h.value, w.value, c.value)
    return buf.reshape(h.value, w.value, c.value)


def png_write(path, arr):
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[..., None]
    data = np.ascontiguousarray(a.astype(">u2") if a.dtype == np.uint16 else a.astype(np.uint8))
    _check(lib().mon_png_write(path.encode(), a.shape[1], a.shape[0], a.shape[2], 16 if arr.dtype == np.uint16 else 8, data.ctypes.data_as(C.c_void_p)))


class OfflineManager:
    """nerf::NerfManagerOffline: same call sequence as OfflineNeRF's main() (MON/main.cpp:322-340) minus the viewer."""

    def __init__(self, dataset_path, config_path, use_dense_depth=False):
        self.h = C.c_void_p()
        _check(lib().mon_offline_create(dataset_path.encode(), config_path.encode(), int(use_dense_depth), C.byref(self.h)))

    def init(self):
        _check(lib().mon_offline_init(self.h))

    def read_dataset(self):
        _check(lib().mon_offline_read_dataset(self.h))

    def create_nerf(self, object_file):
        _check(lib().mon_offline_create_nerf(self.h, object_file.encode()))

    def wait_threads_end(self):
        _check(lib().mon_offline_wait_threads_end(self.h))

    def n_objects(self):
        n = C.c_int(0); _check(lib().mon_offline_n_objects(self.h, C.byref(n))); return n.value

    def object_loss(self, idx):
        l = C.c_float(0); d = C.c_int(0); _check(lib().mon_offline_object_loss(self.h, idx, C.byref(l), C.byref(d))); return l.value, d.value

    def set_output_dir(self, path):
        _check(lib().mon_offline_set_output_dir(self.h, path.encode()))

    def intrinsics(self):
        f = [C.c_float(0) for _ in range(4)]; hw = [C.c_int(0), C.c_int(0)]
        _check(lib().mon_offline_get_intrinsics(self.h, *[C.byref(v) for v in f + hw])); return tuple(v.value for v in f + hw)

    def poses(self):
        n = C.c_size_t(0); _check(lib().mon_offline_get_poses(self.h, None, 0, C.byref(n)))
        T = np.empty((n.value, 16), np.float32); _check(lib().mon_offline_get_poses(self.h, _p(T), n.value, C.byref(n))); return T

    def object_meta(self, idx):
        n = C.c_size_t(0); cls = C.c_int(0); Tow = np.empty(16, np.float32); a0 = np.empty(3, np.float32); a1 = np.empty(3, np.float32)
        _check(lib().mon_offline_object_meta(self.h, idx, C.byref(cls), _p(Tow), _p(a0), _p(a1), None, 0, C.byref(n)))
        boxes = np.empty((n.value, 5), np.uint32)
        _check(lib().mon_offline_object_meta(self.h, idx, None, None, None, None, _p(boxes), n.value, C.byref(n)))
        return dict(cls=cls.value, Tow=Tow, aabb_min=a0, aabb_max=a1, boxes=boxes)

    def object(self, idx):
        """Borrowed handle of object idx (GetAllNeRF()[idx]); owned by the manager."""
        h = C.c_void_p(); _check(lib().mon_offline_object(self.h, idx, C.byref(h))); return _borrowed_object(h)

    def render_test(self, idx, out_dir, max_views=0):
        _check(lib().mon_offline_render_test(self.h, idx, out_dir.encode(), max_views))

    def close(self):
        if self.h:
            lib().mon_offline_destroy(self.h); self.h = None


class OnlineManager:
    """nerf::NerfManagerOnline as the SLAM frontend drives it (REF/src/System.cc:120-138, LocalMapping.cc:1122-1270)."""

    def __init__(self, config_path, use_sparse_depth=False, train_step_iterations=500):
        self.h = C.c_void_p()
        _check(lib().mon_online_create(config_path.encode(), int(use_sparse_depth), int(train_step_iterations), C.byref(self.h)))

    def init(self):
        _check(lib().mon_online_init(self.h))

    def dataset_init(self, fx, fy, cx, cy, H, W, imgs):
        _check(lib().mon_online_dataset_init(self.h, fx, fy, cx, cy, H, W, imgs))

    def new_frame(self, img_id, stamp, bgr_u8, instance_u8, Twc16, depth=None):
        bgr = np.ascontiguousarray(bgr_u8, np.uint8); inst = np.ascontiguousarray(instance_u8, np.uint8); pose = np.ascontiguousarray(Twc16, np.float32)
        d = None if depth is None else np.ascontiguousarray(depth, np.float32)
        _check(lib().mon_online_new_frame(self.h, img_id, stamp.encode(), _p(bgr), bgr.shape[2], _p(inst), _p(d), _p(pose)))

    def create_nerf(self, cls, Tow16, aabb_min, aabb_max):
        a, b, c = (np.ascontiguousarray(v, np.float32) for v in (Tow16, aabb_min, aabb_max)); idx = C.c_size_t(0)
        _check(lib().mon_online_create_nerf(self.h, int(cls), _p(a), _p(b), _p(c), C.byref(idx))); return idx.value

    def update_nerf_bbox(self, idx, boxes, train_step):
        b = np.ascontiguousarray(boxes, np.uint32).reshape(-1, 5)
        _check(lib().mon_online_update_nerf_bbox(self.h, idx, _p(b), b.shape[0], int(train_step)))

    def update_dataset(self, cur_id, Twc16s):
        """UpdateDataset: poses of the len(Twc16s) frames before cur_id replaced on every device."""
        T = np.ascontiguousarray(Twc16s, np.float32).reshape(-1, 16); _check(lib().mon_online_update_dataset(self.h, int(cur_id), T.shape[0], _p(T)))

    def get_pose(self, frame_id):
        T = np.empty(16, np.float32); _check(lib().mon_online_get_pose(self.h, int(frame_id), _p(T))); return T

    def get_frame_idx(self, stamp):
        i = C.c_int(0); _check(lib().mon_online_get_frame_idx(self.h, stamp.encode(), C.byref(i))); return i.value

    def wait_threads_end(self):
        _check(lib().mon_online_wait_threads_end(self.h))

    def object(self, idx):
        h = C.c_void_p(); _check(lib().mon_online_object(self.h, idx, C.byref(h))); return _borrowed_object(h)

    def render_nerfs_test(self, out_path, idx, stamps, boxes, Twcs16, radius):
        b = np.ascontiguousarray(boxes, np.uint32).reshape(-1, 5); T = np.ascontiguousarray(Twcs16, np.float32).reshape(-1, 16)
        arr = (C.c_char_p * len(stamps))(*[s.encode() for s in stamps])
        _check(lib().mon_online_render_nerfs_test(self.h, out_path.encode(), idx, arr, _p(b), _p(T), len(stamps), float(radius)))

    def object_info(self, idx):
        l = C.c_float(0); t = C.c_int(0); d = C.c_int(0); n = C.c_uint32(0)
        _check(lib().mon_online_object_info(self.h, idx, C.byref(l), C.byref(t), C.byref(d), C.byref(n)))
        return dict(loss=l.value, train_calls=t.value, device=d.value, n_boxes=n.value)

    def render(self, idx, box, Twc16):
        FrameId, x, y, h, w = (int(v) for v in box)
        rgb = np.empty((h, w, 3), np.float32); depth = np.empty((h, w), np.float32); mask = np.empty((h, w), np.float32)
        pose = np.ascontiguousarray(Twc16, np.float32)
        _check(lib().mon_online_render(self.h, idx, MonBBox(FrameId, x, y, h, w), _p(pose), _p(rgb), _p(depth), _p(mask))); return rgb, depth, mask

    def close(self):
        if self.h:
            lib().mon_online_destroy(self.h); self.h = None


def _borrowed_object(handle):
    o = ObjectNeRF.__new__(ObjectNeRF); o.h = handle; o.ds = None
    o.cfg = MonConfig(); _check(lib().mon_object_get_config(handle, C.byref(o.cfg))); o.R, o.S = o.cfg.rays_per_batch, o.cfg.n_samples
    o.close = lambda: None                     # the manager owns it
    return o


def marching_cubes(density, res3, thresh, aabb_min, aabb_max, device=0):
    """MarchingCubes + compute_mesh_1ring (marching_cubes.cu:478-509, 655-665) on a caller-supplied lattice (x fastest)."""
    rx, ry, rz = (int(v) for v in res3)
    d = np.ascontiguousarray(density, np.float32).reshape(-1); assert d.size == rx * ry * rz
    a0 = np.ascontiguousarray(aabb_min, np.float32); a1 = np.ascontiguousarray(aabb_max, np.float32)
    nv = C.c_uint32(0); nr = C.c_uint32(0); ni = C.c_uint32(0)
    _check(lib().mon_marching_cubes(device, _p(d), rx, ry, rz, float(thresh), _p(a0), _p(a1), None, None, None, 0, 0, C.byref(nv), C.byref(nr), C.byref(ni)))
    verts = np.empty((nv.value, 3), np.float32); nraw = np.empty((nv.value, 3), np.float32); idx = np.empty(ni.value, np.uint32)
    _check(lib().mon_marching_cubes(device, _p(d), rx, ry, rz, float(thresh), _p(a0), _p(a1), _p(verts), _p(nraw), _p(idx), nv.value, ni.value, C.byref(nv),
            C.byref(nr), C.byref(ni)))
    return dict(verts=verts, normals_raw=nraw, indices=idx, n_verts_real=nr.value)


def generate_toc(theta_deg, phi_deg, radius):
    T = np.empty(16, np.float32); _check(lib().mon_generate_toc(theta_deg, phi_deg, radius, _p(T))); return T


def acc_layout(epad, W, NH, L):
    """param[n_cols]: the MLP parameter each column of k_fused_train's dW partial rows sums into (frag_layout.h acc_param; -1 = pad column)."""
    nc = C.c_int(0)
    _check(diag_lib().mon_debug_acc_layout(epad, W, NH, L, None, C.byref(nc)))
    prm = np.empty(nc.value, np.int32)
    _check(diag_lib().mon_debug_acc_layout(epad, W, NH, L, _p(prm), C.byref(nc))); return prm


def frag_layout(epad, W, NH, L):
    """(source[n_image], slots[n_mlp, 2]) of the fused kernels' A-fragment image (frag_layout.h)."""
    ni = C.c_int(0); nm = C.c_int(0)
    _check(diag_lib().mon_debug_frag_layout(epad, W, NH, L, None, None, C.byref(ni), C.byref(nm)))
    src = np.empty(ni.value, np.int32); sl = np.empty((nm.value, 2), np.int32)
    _check(diag_lib().mon_debug_frag_layout(epad, W, NH, L, _p(src), _p(sl), C.byref(ni), C.byref(nm))); return src, sl
