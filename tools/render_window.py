#!/usr/bin/env python
"""The render half of the metric and nothing else, for rocprofv3: a base.json object on the bench scene trained `--train` steps,
then (a) `--crops` renders of the first training box (NeRF_Model::Render, nerf_model.cu:1702-1830), (b) the 60-view orbit of
RenderVideo (:1832-1991: object-frame poses, central half of the image), (c) `--meshes` GenerateMesh calls at 64^3
(GetDensityOnGrid :2007-2048 + marching cubes) and (d) one density_grid query.  Wall times per phase are printed;
tools/rocpd_window.py --only k_render,k_fused_render,... summarises the kernels of a trace of this command.

   python tools/render_window.py [--train 1000] [--crops 20] [--orbit 60] [--meshes 5] [--options a=1,b=2]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=1000)
    ap.add_argument("--crops", type=int, default=20)
    ap.add_argument("--orbit", type=int, default=60)
    ap.add_argument("--meshes", type=int, default=5)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--options", default="")
    ap.add_argument("--log2-hashmap-size", type=int, default=0)
    a = ap.parse_args()
    pkg = ge.load_package(); ss = ge.load_tools()
    for kv in [s for s in a.options.split(",") if s]:
        k, v = kv.split("="); pkg.set_option(k, int(v))
    sc = ss.make_scene(n_views=a.views, H=480, W=640, f=525.0, seed=0)
    kw = dict(sample_seed=2024)
    if a.log2_hashmap_size:
        kw["log2_hashmap_size"] = a.log2_hashmap_size
    ds, obj = ge.make_problem(pkg, sc, kw)
    obj.train(a.train)
    sync = lambda: pkg.lib().mon_device_synchronize(0)
    sync()
    box = sc.objects[0]["boxes"][0]; v, x, y, h, w = (int(q) for q in box)
    pose = ss.colmajor(sc.Twc[v])
    S2 = 2 * obj.cfg.n_samples
    rgb, depth, mask = obj.render(box, pose)           # (first call: buffers grow)
    t0 = time.perf_counter()
    for _ in range(a.crops):
        rgb, depth, mask = obj.render(box, pose)
    sync(); tc = (time.perf_counter() - t0) / max(1, a.crops)
    print("crop %dx%d: %.3f ms per render incl. D2H, %.2f G nominal ray-samples/s, mask mean %.3f" %
          (h, w, 1e3 * tc, h * w * S2 / tc / 1e9, float(mask.mean())))
    # the orbit of RenderVideo: 6-degree steps at 30 degrees elevation, radius as mon_online_render_nerfs_test uses it
    import numpy as np
    obox = np.array([0, sc.W // 4, sc.H // 4, sc.H // 2, sc.W // 2], np.uint32)
    radius = 3.0 * float(np.linalg.norm(sc.objects[0]["half"]))
    t0 = time.perf_counter(); cur = 0.0; hit = 0.0
    for i in range(a.orbit):
        cur += 360.0 / max(1, a.orbit)
        r2, d2, m2 = obj.render(obox, pkg.generate_toc(cur, 30.0, radius), pose_is_Toc=True); hit += float(m2.mean())
    sync(); to = (time.perf_counter() - t0) / max(1, a.orbit)
    print("orbit %d views %dx%d: %.3f ms per view incl. D2H, %.2f G nominal ray-samples/s, mask mean %.3f" %
          (a.orbit, int(obox[3]), int(obox[4]), 1e3 * to, int(obox[3]) * int(obox[4]) * S2 / to / 1e9, hit / max(1, a.orbit)))
    nv = ni = 0
    obj.generate_mesh(64, 2.0)
    t0 = time.perf_counter()
    for _ in range(a.meshes):
        nv, ni = obj.generate_mesh(64, 2.0)
    sync(); tm = (time.perf_counter() - t0) / max(1, a.meshes)
    print("GenerateMesh 64^3: %.3f ms per call, %d vertices, %d indices" % (1e3 * tm, nv, ni))
    t0 = time.perf_counter(); g = obj.density_grid(64, 64, 64); td = time.perf_counter() - t0
    print("density_grid 64^3: %.3f ms, %d points above 2.0" % (1e3 * td, int((g > 2.0).sum())))
    import zlib
    print("crc rgb %08x depth %08x mask %08x" % (zlib.crc32(rgb.tobytes()), zlib.crc32(depth.tobytes()), zlib.crc32(mask.tobytes())))
    obj.close(); ds.close()


if __name__ == "__main__":
    main()
