"""Viewer-side render latency while N objects train on the same GPU (online manager, base.json, bench-like scene): 60 crop renders of object 0 under load,
then 20 on the idle device; sorted times in ms.   python tools/viewer_latency.py <n_objects>"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools(); ROOT = ge.ROOT
n_obj = int(sys.argv[1]); n_kf = 40
sc = ss.make_scene(n_views=n_kf, H=480, W=640, f=525.0, n_objects=n_obj, seed=11)
m = pkg.OnlineManager(os.path.join(ROOT, "ro-map_amd", "configs", "base.json"), False, 500)
m.init(); m.dataset_init(sc.fx, sc.fy, sc.cx, sc.cy, sc.H, sc.W, sc.n_views)
ids = {}
for v in range(sc.n_views):
    m.new_frame(v, "%.6f" % (v * 0.1), sc.rgb[v][..., ::-1], sc.instance[v], ss.colmajor(sc.Twc[v]))
    for k, ob in enumerate(sc.objects):
        if k not in ids: ids[k] = m.create_nerf(ob["cls"], ss.colmajor(ob["Tow"]), -ob["half"] / 1.1, ob["half"] / 1.1)
        m.update_nerf_bbox(ids[k], ob["boxes"][ob["boxes"][:, 0] == v], 1)
time.sleep(0.5)
bx = sc.objects[0]["boxes"][3]; tr = []
for i in range(60):
    t0 = time.perf_counter(); m.render(ids[0], bx, ss.colmajor(sc.Twc[int(bx[0])])); tr.append(1e3 * (time.perf_counter() - t0)); time.sleep(0.03)
print(n_obj, "objects, crop", bx[3:], "render ms sorted:", " ".join("%.2f" % t for t in sorted(tr)))
m.wait_threads_end()
tr = []
for i in range(20):
    t0 = time.perf_counter(); m.render(ids[0], bx, ss.colmajor(sc.Twc[int(bx[0])])); tr.append(1e3 * (time.perf_counter() - t0))
print("idle GPU render ms sorted:", " ".join("%.2f" % t for t in sorted(tr)))
m.close()
