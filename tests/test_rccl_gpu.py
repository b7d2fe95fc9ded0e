"""RCCL on the GPU box (-m gpu): the final-render gather of ro-map_amd/sharding.py with torch.distributed's "nccl" backend (= RCCL on ROCm).
The box has ONE GPU and RCCL refuses two ranks on one device, so world_size is 1 here; what runs is still the real thing: an RCCL communicator is
created on cuda:0, the size gather and the max-time all_reduce are RCCL collectives on device buffers, and a device-resident crop rendered by
mon_object_render(dst_on_device = 1) goes through RCCL's point-to-point transport (a grouped send + receive to the own rank -- the path a peer's
message to the root takes).  The world_size-2 semantics of the same functions are covered on CPU by tests/test_sharding_gloo.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MON_ROOT"])
import numpy as np, torch, torch.distributed as dist, importlib
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools(); sh = importlib.import_module("ro_map_amd.sharding")
assert pkg.device_count() >= 1
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)
ds, obj = ge.make_problem(pkg, sc, dict(rays_per_batch=256)); obj.train(40)
box = sc.objects[0]["boxes"][1]; v = int(box[0]); pose = ss.colmajor(sc.Twc[v])
rgb, depth, mask = obj.render(box, pose)                                   # host copy: the expected content
packed = sh.render_packed(obj, box, pose, torch, dev)                      # the same crop, rendered straight into a tensor in HBM
assert packed.is_cuda
got = sh.gather_crops(dist, torch, [packed], dev, root=0)                  # RCCL gather of the sizes; the root's own message stays on the device until unpacked
(r2, d2, m2), = got[0]
ok_gather = bool(np.array_equal(r2, rgb) and np.array_equal(d2, depth) and np.array_equal(m2, mask))
back = sh.loopback_crop(dist, torch, packed)                               # ncclSend + ncclRecv of the device-resident crop (grouped), rank 0 -> rank 0
torch.cuda.synchronize()
ok_p2p = bool(back.is_cuda and torch.equal(back, packed))
tmax = sh.max_over_ranks(dist, torch, 1.25, dev)                           # RCCL all_reduce(MAX) on a device buffer
libs = sorted({l.split()[-1].split("/")[-1] for l in open("/proc/self/maps") if "rccl" in l or "libmon_core.so" in l})
print(json.dumps(dict(ok_gather=ok_gather, ok_p2p=ok_p2p, tmax=tmax, backend=dist.get_backend(), libs=libs, mask_px=int(mask.sum()))))
obj.close(); ds.close(); dist.destroy_process_group()
'''


def test_rccl_communicator_moves_a_device_resident_crop():
    import __graft_entry__ as ge
    assert ge.load_package().device_count() >= 1, "no HIP device visible: the GPU tests must run on the MI355X box"
    env = dict(os.environ, MON_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY",
            "0"))
    r = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=env, timeout=420)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["backend"] == "nccl" and j["ok_gather"] and j["ok_p2p"] and j["tmax"] == 1.25 and j["mask_px"] > 0, j
    assert any("rccl" in l for l in j["libs"]) and "libmon_core.so" in j["libs"], j["libs"]          # librccl really is mapped next to the product library
