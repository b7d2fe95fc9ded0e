"""End-to-end OfflineNeRF job on a synthetic sequence in the reference's on-disk layout (runs on the GPU box): writes the sequence,
runs the headless driver (tools/offline_nerf.cpp: 10 x 500 iterations per object, a mesh every 2nd outer step, <id>.ply, test images)
and reports wall time and the PSNR of the written test images."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from PIL import Image
ss = ge.load_tools(); ROOT = ge.ROOT
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, n_objects=n_obj, seed=5)
tmp = tempfile.mkdtemp(); seq = os.path.join(tmp, "seq"); out = os.path.join(tmp, "out"); ss.write_sequence(sc, seq)
exe = os.path.join(ROOT, "ro-map_amd", "offline_nerf"); cfg = os.path.join(ROOT, "ro-map_amd", "configs", "base.json")
t0 = time.perf_counter()
r = subprocess.run([exe, cfg, seq, "0", str(n_obj), out], capture_output=True, text=True, timeout=1200)
dt = time.perf_counter() - t0
print("\n".join(l for l in r.stdout.splitlines() if not l.startswith("Id:"))); assert r.returncode == 0, r.stderr
for k, ob in enumerate(sc.objects):
    v, x, y, h, w = (int(q) for q in ob["boxes"][0]); stamp = "%.6f" % (v * 0.1)
    img = np.asarray(Image.open(os.path.join(out, str(k), "test_img", stamp + ".png"))).astype(np.float64) / 255.0
    gm = sc.instance[v, y:y + h, x:x + w] == ob["cls"]; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
    ply = os.path.join(out, "%d.ply" % k)
    print("object %d: PSNR %.2f dB, mesh %s (%d bytes)" % (k, -10 * np.log10(np.mean((img - gt) ** 2)), os.path.basename(ply), os.path.getsize(ply)))
print("OfflineNeRF job, %d objects x 5000 iterations (R=4096, S=32) incl. sequence read, meshes, 4 test views each: %.2f s wall" % (n_obj, dt))
