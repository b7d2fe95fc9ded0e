#!/usr/bin/env python
"""Diffs the outputs of a run of the CUDA reference against a run of this library on the same inputs ("same inputs" mode: mon_config::rng_flags = XORWOW,
cuRAND flavour, tcnn init order).  Both sides write, per object id (NeRF::RenderTestImg, CORE/src/nerf.cu:255-349 / mon_online_render_nerfs_test,
mon_offline_*):   <dir>/<id>/test_img/<stamp>.png    8-bit colour (x 255)
                  <dir>/<id>/test_depth/<stamp>.png  16-bit z-depth (x 20000, nerf.cu:344)
                  <dir>/<id>/test_mask/<stamp>.png   8-bit mask (x 255)
                  <dir>/<id>/obj.ply                 ASCII mesh (marching_cubes.cu:573-605)
   python tools/compare_with_reference_outputs.py REFERENCE_DIR OUR_DIR [--json]
Stated tolerance (the two numeric models -- this repo's fp32 accumulation and tiny-cuda-nn's fp16 accumulation -- end a training run as far apart
as one run is from itself started one fp16 ulp away; tests/golden/numerics_study.json): per image mutual PSNR >= 30 dB, mask IoU >= 0.97, mean |depth
difference| over pixels both masks cover <= 1 % of the mean depth; per object mean mutual PSNR >= 34 dB; mesh vertex-set distance (symmetric mean nearest
neighbour) <= 1.5 % of the mesh's bounding-box diagonal.  Exit status 0 when every object is inside, 1 otherwise, 2 when the directories do not match up."""
import argparse
import json
import os
import sys

import numpy as np

TOL = dict(psnr_min_db=30.0, psnr_mean_db=34.0, mask_iou_min=0.97, depth_rel_max=0.01, mesh_rel_max=0.015)


def read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def read_ply_vertices(path):
    n = 0; rows = []
    with open(path) as f:
        for line in f:
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line.strip() == "end_header":
                break
        for _ in range(n):
            rows.append([float(v) for v in f.readline().split()[:3]])
    v = np.asarray(rows, np.float64).reshape(-1, 3)
    return v[np.any(v != 0, 1)]                       # (the reference pads the vertex list to a multiple of 128 with zero vertices, marching_cubes.cu:496)


def nn_mean(a, b):
    """mean distance from every point of a to its nearest neighbour in b (k-d tree when SciPy is there, chunked brute force otherwise)"""
    try:
        from scipy.spatial import cKDTree
        return float(cKDTree(b).query(a)[0].mean())
    except Exception:
        out = []
        for i in range(0, len(a), 512):
            out.append(np.sqrt(((a[i:i + 512, None, :] - b[None, :, :]) ** 2).sum(-1)).min(1))
        return float(np.concatenate(out).mean())


def compare_object(ref_dir, our_dir):
    res = dict(images=[], missing=[])
    stamps = sorted(f for f in os.listdir(os.path.join(ref_dir, "test_img")) if f.endswith(".png"))
    for st in stamps:
        paths = [os.path.join(d, sub, st) for d in (ref_dir, our_dir) for sub in ("test_img", "test_depth", "test_mask")]
        if not all(os.path.exists(p) for p in paths):
            res["missing"].append(st); continue
        ri, rd, rm, oi, od, om = (read_png(p) for p in paths)
        if ri.shape != oi.shape:
            res["missing"].append(st + " (size)"); continue
        mse = float(np.mean((ri.astype(np.float64) / 255.0 - oi.astype(np.float64) / 255.0) ** 2))
        a, b = rm > 127, om > 127
        iou = float((a & b).sum() / max(1, (a | b).sum()))
        both = a & b
        dr, do = rd.astype(np.float64) / 20000.0, od.astype(np.float64) / 20000.0
        drel = float(np.abs(dr - do)[both].mean() / max(1e-9, dr[both].mean())) if both.any() else 0.0
        res["images"].append(dict(stamp=st[:-4], psnr_db=99.0 if mse == 0 else -10 * np.log10(mse), mask_iou=iou, depth_rel=drel))
    rp, op = os.path.join(ref_dir, "obj.ply"), os.path.join(our_dir, "obj.ply")
    if os.path.exists(rp) and os.path.exists(op):
        va, vb = read_ply_vertices(rp), read_ply_vertices(op)
        if len(va) and len(vb):
            diag = float(np.linalg.norm(va.max(0) - va.min(0)))
            res["mesh"] = dict(ref_vertices=len(va), our_vertices=len(vb), rel_distance=0.5 * (nn_mean(va, vb) + nn_mean(vb, va)) / max(diag, 1e-12))
    im = res["images"]
    ok = bool(im) and not res["missing"]
    if im:
        res["psnr_min_db"] = min(i["psnr_db"] for i in im); res["psnr_mean_db"] = float(np.mean([i["psnr_db"] for i in im]))
        res["mask_iou_min"] = min(i["mask_iou"] for i in im); res["depth_rel_max"] = max(i["depth_rel"] for i in im)
        ok &= res["psnr_min_db"] >= TOL["psnr_min_db"] and res["psnr_mean_db"] >= TOL["psnr_mean_db"] and res["mask_iou_min"] >= TOL["mask_iou_min"] and res["depth_rel_max"] <= TOL["depth_rel_max"]
    if "mesh" in res:
        ok &= res["mesh"]["rel_distance"] <= TOL["mesh_rel_max"]
    res["inside_tolerance"] = bool(ok)
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("reference_dir"); ap.add_argument("our_dir"); ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    ids = sorted(d for d in os.listdir(a.reference_dir) if os.path.isdir(os.path.join(a.reference_dir, d, "test_img")))
    if not ids or any(not os.path.isdir(os.path.join(a.our_dir, d, "test_img")) for d in ids):
        print("object directories do not match up: %s vs %s" % (ids, sorted(os.listdir(a.our_dir)))); return 2
    out = {d: compare_object(os.path.join(a.reference_dir, d), os.path.join(a.our_dir, d)) for d in ids}
    if a.json:
        print(json.dumps(dict(tolerance=TOL, objects=out)))
    else:
        for d, r in out.items():
            print("object %s: %d images, PSNR min %.2f mean %.2f dB, mask IoU min %.3f, depth rel max %.4f%s -> %s" % (
                d, len(r["images"]), r.get("psnr_min_db", float("nan")), r.get("psnr_mean_db", float("nan")), r.get("mask_iou_min", float("nan")),
                        r.get("depth_rel_max", float("nan")),
                (", mesh %d / %d vertices, rel distance %.4f" % (r["mesh"]["ref_vertices"], r["mesh"]["our_vertices"],
                        r["mesh"]["rel_distance"])) if "mesh" in r else "",
                "inside tolerance" if r["inside_tolerance"] else "OUTSIDE tolerance" + (" (missing: %s)" % r["missing"] if r["missing"] else "")))
    return 0 if all(r["inside_tolerance"] for r in out.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
