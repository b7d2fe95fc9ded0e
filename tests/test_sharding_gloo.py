"""CPU, world_size 2, gloo: the N > 1 path of bench.py (object sharding + final-render gather + max-time)."""
import importlib
import os
import socket
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _sharding():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    return importlib.import_module("ro_map_amd.sharding")


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    sh = _sharding()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_objects = 5
    mine = sh.objects_of_rank(n_objects, world, rank)
    crops = []
    for k in mine:                                   # variable crop sizes, content encodes the object id
        h, w = 3 + k, 4 + 2 * k
        rgb = np.full((h, w, 3), k + 0.25, np.float32); depth = np.full((h, w), k + 0.5, np.float32); mask = np.full((h, w), float(k % 2), np.float32)
        c = sh.pack_crop(rgb, depth, mask)
        crops.append(torch.from_numpy(c) if k % 2 else c)          # both input kinds: host arrays and tensors already on the collective's device
    got = sh.gather_crops(dist, torch, crops, "cpu", root=0)                       # gather-to-root: the root holds every rank's crops, the others nothing
    tmax = sh.max_over_ranks(dist, torch, 1.0 + rank, "cpu")
    ok = (got is None) == (rank != 0)
    for r in range(world if rank == 0 else 0):
        ks = sh.objects_of_rank(n_objects, world, r)
        ok &= len(got[r]) == len(ks)
        for (rgb, depth, mask), k in zip(got[r], ks):
            ok &= rgb.shape == (3 + k, 4 + 2 * k, 3) and float(rgb[0, 0, 0]) == k + 0.25 and float(depth[-1, -1]) == k + 0.5 and float(mask[0,
                    0]) == float(k % 2)
    # a second gather with another root and an empty contribution from this rank's side when it has nothing to send
    got2 = sh.gather_crops(dist, torch, crops if rank == 0 else [], "cpu", root=1)
    ok &= (got2 is None) == (rank != 1)
    if rank == 1:
        ok &= len(got2[1]) == 0 and len(got2[0]) == len(sh.objects_of_rank(n_objects, world, 0))
    # (sharding.loopback_crop -- a message to oneself -- is an RCCL-only check, tests/test_rccl_gpu.py: gloo has no pair to itself)
    q.put((rank, mine, bool(ok), tmax))
    dist.destroy_process_group()


def test_gather_to_root_with_four_ranks():
    """BASELINE configs[2]'s shape in small: more ranks than two, objects spread unevenly (5 objects over 4 ranks), both gathers of _worker (root 0, root 1)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port(); world = 4
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in ps:
        p.join(timeout=60); assert p.exitcode == 0
    assert [r[1] for r in res] == [[0, 4], [1], [2], [3]]
    assert all(r[2] for r in res) and all(r[3] == 4.0 for r in res)


def test_round_robin_map_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port(); world = 2
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(timeout=60); assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]              # object k -> rank k mod world (nerf.cu:27-33)
    assert all(r[2] for r in res) and all(r[3] == 2.0 for r in res)


def test_every_object_has_exactly_one_owner():
    sh = _sharding()
    for world in (1, 2, 4, 8):
        for n in (0, 1, 4, 8, 64):
            seen = sorted(k for r in range(world) for k in sh.objects_of_rank(n, world, r))
            assert seen == list(range(n))
            assert all(sh.owner_of_object(k, world) == k % world for k in range(n))
