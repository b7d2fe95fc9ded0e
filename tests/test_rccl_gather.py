"""libmon_core_rccl.so (include/mon_core_rccl.h): the gather-to-root of final renders over RCCL inside ONE process whose objects sit on several devices
(single-process communicator, grouped ncclSend / ncclRecv of the true sizes).  CPU: the exported surface and the message bookkeeping.
GPU: crops equal to mon_object_render's bit for bit, and OfflineNeRF's test images AND meshes through the gather equal to the per-object writer's byte for
byte.  A rank of the gather is a LOGICAL device: with mon_set_logical_devices(2 / 4) on the 1-GPU box the `rank != root` branch -- a message per rank, the
receive offsets, the unpack of the peers' messages, an empty rank, either root -- runs through the peer-copy transport (RCCL has one rank per GPU); RCCL
transfers between GPUs need an N > 1 node (the driver's), where the same bookkeeping carries them."""
import ctypes
import filecmp
import os
import re

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import ROOT


def _header_symbols(name):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mon_[a-z0-9_]+)\s*\(", text)))


def test_gather_library_exports_its_header_and_keeps_rccl_out_of_the_core(pkg):
    import subprocess
    syms = _header_symbols("mon_core_rccl.h")
    assert sorted(pkg.rccl_symbols()) == syms and len(syms) >= 5
    assert os.path.exists(pkg.rccl_lib_path()), "run __graft_entry__.build() first"
    L = pkg.rccl_lib()
    for s in syms:
        assert hasattr(L, s), "libmon_core_rccl.so does not export " + s
    core = ctypes.CDLL(pkg.lib_path())
    assert not any(hasattr(core, s) for s in syms)
    needed = subprocess.run(["readelf", "-d", pkg.lib_path()], capture_output=True, text=True).stdout
    assert "rccl" not in needed                                                     # the product library does not depend on RCCL
    needed = subprocess.run(["readelf", "-d", pkg.rccl_lib_path()], capture_output=True, text=True).stdout
    assert "librccl" in needed and "libmon_core.so" in needed


def test_gather_plan_packs_one_message_per_device(pkg):
    """Eight objects placed k mod 4 (CORE/src/nerf.cu:27-33) with crops of different sizes: one message per device, crops back to back in object order."""
    dev = [k % 4 for k in range(8)]; npx = [64 * 64, 100, 313 * 229, 1, 7, 640 * 480, 50, 33]
    per, off = pkg.gather_plan(dev, npx, 4)
    assert per.tolist() == [5 * (npx[0] + npx[4]), 5 * (npx[1] + npx[5]), 5 * (npx[2] + npx[6]), 5 * (npx[3] + npx[7])]
    assert off.tolist() == [0, 0, 0, 0, 5 * npx[0], 5 * npx[1], 5 * npx[2], 5 * npx[3]]
    per, off = pkg.gather_plan([], [], 3); assert per.tolist() == [0, 0, 0] and off.size == 0
    with pytest.raises(pkg.MonError):
        pkg.gather_plan([0, 5], [1, 1], 4)                                           # a device the communicator does not have


@pytest.mark.gpu
def test_gathered_crops_equal_the_objects_own_renders(pkg, ss):
    assert pkg.device_count() >= 1
    sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, n_objects=3, seed=4)
    ds, o0 = ge.make_problem(pkg, sc, dict(rays_per_batch=1024)); objs = [o0]
    for k in (1, 2):
        objs.append(ge.make_problem(pkg, sc, dict(rays_per_batch=1024, sample_seed=50 + k), obj_index=k, dataset=ds)[1])
    for o in objs:
        o.train(120)
    g = pkg.Gather(0)
    # one whole frame among the crops
    boxes = [sc.objects[k]["boxes"][2] for k in range(3)]; boxes[1] = np.array([int(boxes[1][0]), 0, 0, sc.H, sc.W], np.uint32)
    poses = [ss.colmajor(sc.Twc[int(b[0])]) for b in boxes]
    got = g.renders(objs, boxes, poses)
    for o, b, T, (rgb, dep, msk) in zip(objs, boxes, poses, got):
        r2, d2, m2 = o.render(b, T)
        assert np.array_equal(rgb, r2) and np.array_equal(dep, d2) and np.array_equal(msk, m2) and m2.mean() > 0.01
    st = g.stats()
    n_dev = pkg.device_count()
    assert st["bytes_over_links"] + st["bytes_on_root"] == 20 * sum(int(b[3]) * int(b[4]) for b in boxes)
    if n_dev == 1:
        assert st["bytes_over_links"] == 0 and st["sending_devices"] == 0
    got2 = g.renders(objs[:1], boxes[:1], poses[:1])                                 # a smaller call after a larger one (buffers are grow-only)
    assert np.array_equal(got2[0][0], got[0][0])
    g.close()
    for o in objs:
        o.close()
    ds.close()


@pytest.mark.gpu
def test_offline_test_images_through_the_gather_are_the_same_files(pkg, ss, tmp_path):
    """mon_offline_render_test_gathered against mon_offline_render_test on a 3-object sequence: same directory tree, same PNG and mesh bytes."""
    assert pkg.device_count() >= 1
    sc = ss.make_scene(n_views=8, H=120, W=160, f=130.0, n_objects=3, seed=6)
    seq = str(tmp_path / "seq"); ss.write_sequence(sc, seq)
    pkg.set_offline_schedule(2, 50)       # (a mesh exists from the 2nd outer step on, nerf.cu:138-145)
    try:
        m = pkg.OfflineManager(seq, os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json")); m.init(); m.read_dataset()
    finally:
        pkg.set_offline_schedule(10, 500)
    m.set_output_dir("")
    for k in range(3):
        m.create_nerf(os.path.join(seq, "obj_offline", "%d.txt" % k))
    m.wait_threads_end()
    a, b = str(tmp_path / "per_object"), str(tmp_path / "gathered")
    for k in range(3):
        m.render_test(k, a, 3)
    g = pkg.Gather(0); g.offline_render_test(m, b, 3); g.close()
    n = 0
    for k in range(3):
        for sub in ("test_img", "test_depth", "test_mask"):
            fa = sorted(os.listdir(os.path.join(a, str(k), sub))); fb = sorted(os.listdir(os.path.join(b, str(k), sub)))
            assert fa == fb and len(fa) == 3
            for f in fa:
                assert filecmp.cmp(os.path.join(a, str(k), sub, f), os.path.join(b, str(k), sub, f), shallow=False), (k, sub, f); n += 1
    assert n == 27
    for k in range(3):                                                              # "Save Object Mesh" (nerf.cu:397-403): the gathered tree has it too
        pa, pb = os.path.join(a, str(k), "obj.ply"), os.path.join(b, str(k), "obj.ply")
        assert os.path.exists(pa) and os.path.exists(pb) and filecmp.cmp(pa, pb, shallow=False), k
    assert sorted(os.listdir(os.path.join(a, "0"))) == sorted(os.listdir(os.path.join(b, "0")))
    m.close()


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            out[os.path.relpath(os.path.join(d, f), root)] = open(os.path.join(d, f), "rb").read()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_logical,root", [(2, 0), (2, 1), (4, 2)])
def test_gather_between_logical_devices_moves_and_unpacks_peer_messages(pkg, ss, n_logical, root):
    """Objects k -> logical device k mod n (CORE/src/nerf.cu:27-33), crops of different sizes, one rank left without an object when n = 4: every rank but the
    root sends its message (peer copy on a shared GPU), the root unpacks them from the receive offsets -- crops bit-equal to the objects' own renders."""
    assert pkg.device_count() >= 1
    n_phys = pkg.device_count()
    sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, n_objects=3, seed=4)
    pkg.set_logical_devices(n_logical)
    try:
        assert pkg.device_count() == n_logical
        n_obj = 5 if n_logical == 2 else 3                                         # n = 4: logical device 3 holds nothing
        dss = {}; objs = []
        for k in range(n_obj):
            d = k % n_logical
            if d not in dss:
                dss[d] = pkg.Dataset(d, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, sc.n_views)
                for v in range(sc.n_views):
                    dss[d].add_frame(v, sc.rgb[v], sc.instance[v], ss.colmajor(sc.Twc[v]))
            objs.append(ge.make_problem(pkg, sc, dict(rays_per_batch=1024, sample_seed=70 + k), obj_index=k % 3, dataset=dss[d])[1])
            assert int(objs[-1].info().device) == d
        for o in objs:
            o.train(100)
        boxes = [np.array(sc.objects[k % 3]["boxes"][1 + k], np.uint32) for k in range(n_obj)]
        boxes[1] = np.array([int(boxes[1][0]), 0, 0, sc.H, sc.W], np.uint32)       # a whole frame among the crops: messages of very different sizes
        poses = [ss.colmajor(sc.Twc[int(b[0])]) for b in boxes]
        g = pkg.Gather(root)
        for transport in (pkg.Gather.AUTO, pkg.Gather.PEER_COPY):
            g.set_transport(transport)
            got = g.renders(objs, boxes, poses)
            for o, b, T, (rgb, dep, msk) in zip(objs, boxes, poses, got):
                r2, d2, m2 = o.render(b, T)
                assert np.array_equal(rgb, r2) and np.array_equal(dep, d2) and np.array_equal(msk, m2) and m2.mean() > 0.01
            st = g.stats(); npx = [int(b[3]) * int(b[4]) for b in boxes]
            on_root = 20 * sum(p for k, p in enumerate(npx) if k % n_logical == root)
            assert st["n_ranks"] == n_logical and st["bytes_on_root"] == on_root and st["bytes_over_links"] == 20 * sum(npx) - on_root
            senders = len({k % n_logical for k in range(n_obj)} - {root})
            assert st["sending_devices"] == senders and st["messages_rccl"] + st["messages_peer_copy"] == senders and senders >= 1
            assert st["bytes_rccl"] + st["bytes_peer_copy"] == st["bytes_over_links"]
            if n_phys == 1 or transport == pkg.Gather.PEER_COPY:
                assert st["messages_rccl"] == 0 and st["bytes_peer_copy"] == st["bytes_over_links"]
        # a gather with objects of ONE non-root rank only (every other message empty), then only the root's
        other = next(k for k in range(n_obj) if k % n_logical != root)
        got1 = g.renders([objs[other]], [boxes[other]], [poses[other]]); assert np.array_equal(got1[0][0], got[other][0]) and g.stats()["sending_devices"] == 1
        mine = next(k for k in range(n_obj) if k % n_logical == root)
        got2 = g.renders([objs[mine]], [boxes[mine]], [poses[mine]]); assert np.array_equal(got2[0][0], got[mine][0]) and g.stats()["bytes_over_links"] == 0
        g.close()
        for o in objs:
            o.close()
        for d in dss.values():
            d.close()
    finally:
        pkg.set_logical_devices(0)


@pytest.mark.gpu
def test_offline_output_tree_through_the_gather_on_logical_devices(pkg, ss, tmp_path):
    """The whole OfflineNeRF output of a 3-object sequence placed k mod 2 -- PNGs and obj.ply -- through the gather (root = logical device 1) against the
    per-object writer: the same files with the same bytes; the gathered call also works when the caller has NOT waited for the training threads."""
    assert pkg.device_count() >= 1
    sc = ss.make_scene(n_views=8, H=120, W=160, f=130.0, n_objects=3, seed=6)
    seq = str(tmp_path / "seq"); ss.write_sequence(sc, seq)
    pkg.set_offline_schedule(2, 60); pkg.set_logical_devices(2)
    try:
        m = pkg.OfflineManager(seq, os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json")); m.init(); m.read_dataset(); m.set_output_dir("")
        for k in range(3):
            m.create_nerf(os.path.join(seq, "obj_offline", "%d.txt" % k))
        a, b = str(tmp_path / "per_object"), str(tmp_path / "gathered")
        g = pkg.Gather(1); g.offline_render_test(m, b, 3)                          # joins the training threads itself
        st = g.stats(); assert st["n_ranks"] == 2 and st["sending_devices"] == 1 and st["bytes_over_links"] > 0
        g.close()
        m.wait_threads_end()                                                        # a second wait after the threads were joined: the objects' results again
        for k in range(3):
            m.render_test(k, a, 3)
        ta, tb = _tree(a), _tree(b)
        assert sorted(ta) == sorted(tb) and len(ta) == 3 * (9 + 1), sorted(ta)
        assert all(ta[f] == tb[f] for f in ta), [f for f in ta if ta[f] != tb[f]]
        m.close()
    finally:
        pkg.set_logical_devices(0); pkg.set_offline_schedule(10, 500)
