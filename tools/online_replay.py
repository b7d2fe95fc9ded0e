"""Replay of the SLAM side's call sequence through the online manager (C4 of SURVEY 8d: REF/src/LocalMapping.cc:1122-1270 at
TrainStepIterations=500): keyframes arrive every `period` ms, every object gets its 2-D box with train_step=1, a viewer renders a crop of
object 0 every keyframe.  Reports how long NewFrameToDataset / UpdateNeRFBbox / a viewer render block the caller while the object threads
train, the training done, and the final quality.      python tools/online_replay.py [n_objects] [n_keyframes] [period_ms]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package(); ss = ge.load_tools(); ROOT = ge.ROOT
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_kf = int(sys.argv[2]) if len(sys.argv) > 2 else 60
period = float(sys.argv[3]) / 1e3 if len(sys.argv) > 3 else 0.05
sc = ss.make_scene(n_views=n_kf, H=480, W=640, f=525.0, n_objects=n_obj, seed=11)
m = pkg.OnlineManager(os.path.join(ROOT, "ro-map_amd", "configs", "base.json"), False, 500)
m.init(); m.dataset_init(sc.fx, sc.fy, sc.cx, sc.cy, sc.H, sc.W, sc.n_views)
ids = {}; t_frame, t_box, t_render = [], [], []
t_start = time.perf_counter()
for v in range(sc.n_views):
    t_next = t_start + (v + 1) * period
    t0 = time.perf_counter(); m.new_frame(v, "%.6f" % (v * 0.1), sc.rgb[v][..., ::-1], sc.instance[v], ss.colmajor(sc.Twc[v]))
    t_frame.append(time.perf_counter() - t0)
    for k, ob in enumerate(sc.objects):
        if k not in ids:
            ids[k] = m.create_nerf(ob["cls"], ss.colmajor(ob["Tow"]), -ob["half"] / 1.1, ob["half"] / 1.1)
        b = ob["boxes"][ob["boxes"][:, 0] == v]
        t0 = time.perf_counter(); m.update_nerf_bbox(ids[k], b, 1); t_box.append(time.perf_counter() - t0)
    if v > 12:
        bx = sc.objects[0]["boxes"][3]
        t0 = time.perf_counter(); m.render(ids[0], bx, ss.colmajor(sc.Twc[int(bx[0])])); t_render.append(time.perf_counter() - t0)
    time.sleep(max(0.0, t_next - time.perf_counter()))
t_feed = time.perf_counter() - t_start
t0 = time.perf_counter(); m.wait_threads_end(); t_wait = time.perf_counter() - t0
ms = lambda a: "mean %.2f / p99 %.2f / max %.2f ms" % (1e3 * np.mean(a), 1e3 * np.percentile(a, 99), 1e3 * np.max(a))
print("online replay: %d objects, %d keyframes every %.0f ms, TrainStepIterations 500" % (n_obj, n_kf, 1e3 * period))
print("  NewFrameToDataset blocks   " + ms(t_frame))
print("  UpdateNeRFBbox blocks      " + ms(t_box))
print("  viewer render (crop) takes " + ms(t_render))
calls = [m.object_info(i)["train_calls"] for i in ids.values()]
print("  training done while feeding + at WaitThreadsEnd: %s Train_Step_Online calls per object (x 500 iterations); feed %.2f s, WaitThreadsEnd %.2f s" % (calls, t_feed, t_wait))
for k, i in ids.items():
    ob = sc.objects[k]; v, x, y, h, w = (int(q) for q in ob["boxes"][3])
    rgb, depth, mask = m.render(i, ob["boxes"][3], ss.colmajor(sc.Twc[v]))
    gm = sc.instance[v, y:y + h, x:x + w] == ob["cls"]; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
    print("  object %d: loss %.5f, PSNR of a training view %.2f dB" % (k, m.object_info(i)["loss"], -10 * np.log10(np.mean((rgb - gt) ** 2))))
m.close()
