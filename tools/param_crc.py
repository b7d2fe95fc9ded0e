"""CRC of the trained fp32 parameters after a fixed schedule on the bench scene (determinism / A-B checks of kernel variants):
   python tools/param_crc.py [steps ...]      e.g.  MON_OPTIONS=keep_zero_samples=1 python tools/param_crc.py 5 60 200
   MON_CRC_CFG='{"log2_hashmap_size": 20}' overrides network-configuration fields."""
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=24, H=240, W=320, f=260.0, seed=1)
ds, obj = ge.make_problem(pkg, sc, json.loads(os.environ.get("MON_CRC_CFG", "{}"))); obj.set_backend(1)
for k in [int(a) for a in sys.argv[1:]] or [1, 10, 100, 300]:
    obj.train(k); st = obj.buffer("state")
    print("steps+%d crc %08x scattered_samples %d" % (k, zlib.crc32(obj.get_params(0).tobytes()), int(st[24])), flush=True)
