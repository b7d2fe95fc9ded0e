"""Synthetic 'room'-like scenes in the reference's data model (test / bench input generator).

The real sequences are Google-Drive downloads (REF/README.md:61-66) and are not available offline, so
inputs are synthesised analytically:  textured ellipsoids (one per object) seen from an orbit of
pinhole cameras.  Conventions follow the reference:
  * pose = Twc, 4x4 column-major float32 (CORE/src/nerf_data.cu:95-108), camera looks along +z,
    x right, y down (pixel ray ((x-cx)/fx, (y-cy)/fy, 1), CORE/src/nerf_model.cu:403-405);
  * rgb 8-bit, instance 8-bit (0 = background, class id otherwise; nerf.cu:75), depth = z-depth in
    metres (nerf_model.cu:432 multiplies by the ray norm);
  * object file content: class, Two (tx ty tz qx qy qz qw), half extents a1 a2 a3, then
    `stamp x y h w` 2-D boxes (nerf.cu:58-118).
`write_sequence()` emits the on-disk layout of nerf_data.cu:27-121 / nerf.cu:58-118.
"""
import math
import os

import numpy as np


def _look_at(cam_pos, target, up=(0.0, 0.0, 1.0)):
    z = np.asarray(target, np.float64) - np.asarray(cam_pos, np.float64)
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, cam_pos
    return T


def _quat_from_R(R):
    w = math.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = math.copysign(math.sqrt(max(0.0, 1.0 + R[0, 0] - R[1, 1] - R[2, 2])) / 2.0, R[2, 1] - R[1, 2])
    y = math.copysign(math.sqrt(max(0.0, 1.0 - R[0, 0] + R[1, 1] - R[2, 2])) / 2.0, R[0, 2] - R[2, 0])
    z = math.copysign(math.sqrt(max(0.0, 1.0 - R[0, 0] - R[1, 1] + R[2, 2])) / 2.0, R[1, 0] - R[0, 1])
    return x, y, z, w


def _texture(p_obj, radii, phase):
    """Smooth colour field on the surface, values in [0.1, 0.9]."""
    q = p_obj / radii
    r = 0.5 + 0.4 * np.sin(5.0 * q[..., 0] + phase)
    g = 0.5 + 0.4 * np.sin(4.0 * q[..., 1] + 1.3 + phase)
    b = 0.5 + 0.4 * np.sin(6.0 * q[..., 2] + 2.1 + phase)
    return np.stack([r, g, b], -1)


class Scene:
    pass


def make_scene(n_views=24, H=120, W=160, f=130.0, n_objects=1, radius=1.1, elev_deg=30.0, crop=None, seed=0, fill=0.85):
    """Returns a Scene with numpy arrays.  crop=(h, w) crops every 2-D box to that size around its centre."""
    rng = np.random.RandomState(seed)
    sc = Scene()
    sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy = H, W, f, f, (W - 1) / 2.0, (H - 1) / 2.0
    sc.n_views = n_views
    # objects on a small ring around the world origin
    sc.objects = []
    for k in range(n_objects):
        ang = 2.0 * math.pi * k / max(1, n_objects)
        centre = np.array([0.0, 0.0, 0.0]) if n_objects == 1 else np.array([0.45 * math.cos(ang), 0.45 * math.sin(ang), 0.0])
        yaw = 0.3 * k
        Two = np.eye(4)
        Two[:3, :3] = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
        Two[:3, 3] = centre
        half = np.array([0.20, 0.16, 0.24]) * (1.0 if n_objects == 1 else 0.6) * (1.0 + 0.1 * rng.rand(3))
        sc.objects.append(dict(cls=k + 1, Two=Two, Tow=np.linalg.inv(Two), half=half, radii=half * fill, phase=0.7 * k))
    poses = []
    for v in range(n_views):
        th = 2.0 * math.pi * v / n_views
        el = math.radians(elev_deg + 8.0 * math.sin(3.0 * th))
        cam = radius * np.array([math.cos(el) * math.cos(th), math.cos(el) * math.sin(th), math.sin(el)])
        poses.append(_look_at(cam, (0.0, 0.0, 0.0)))
    sc.Twc = np.stack(poses)
    # ---- ray-cast
    ys, xs = np.mgrid[0:H, 0:W]
    dc = np.stack([(xs - sc.cx) / sc.fx, (ys - sc.cy) / sc.fy, np.ones_like(xs, np.float64)], -1)
    rgb = np.zeros((n_views, H, W, 3), np.uint8)
    inst = np.zeros((n_views, H, W), np.uint8)
    depth = np.zeros((n_views, H, W), np.float32)
    for v in range(n_views):
        Rwc, twc = sc.Twc[v][:3, :3], sc.Twc[v][:3, 3]
        dw = dc @ Rwc.T
        zbuf = np.full((H, W), np.inf)
        img = np.empty((H, W, 3))
        img[..., 0] = 0.35 + 0.1 * ys / H; img[..., 1] = 0.33; img[..., 2] = 0.30 + 0.1 * xs / W     # dull room background
        for ob in sc.objects:
            Row, tow = ob["Tow"][:3, :3], ob["Tow"][:3, 3]
            o = Row @ twc + tow
            d = dw @ Row.T
            # ellipsoid |p/radii| = 1 with p = o + t d (t is z-depth because dc.z == 1)
            on, dn = o / ob["radii"], d / ob["radii"]
            a = (dn * dn).sum(-1); b = 2.0 * (dn * on).sum(-1); c = (on * on).sum() - 1.0
            disc = b * b - 4 * a * c
            hit = disc > 0
            t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
            hit &= (t > 0) & (t < zbuf)
            p = o + t[..., None] * d
            col = _texture(np.where(hit[..., None], p, 0.0), ob["radii"], ob["phase"])
            img[hit] = col[hit]; zbuf[hit] = t[hit]; inst[v][hit] = ob["cls"]
        rgb[v] = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
        depth[v] = np.where(np.isfinite(zbuf), zbuf, 0.0).astype(np.float32)
    sc.rgb, sc.instance, sc.depth = rgb, inst, depth
    # ---- 2-D boxes: projection of the 8 corners of each 3-D box
    for ob in sc.objects:
        boxes = []
        corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64) * ob["half"]
        for v in range(n_views):
            Tcw = np.linalg.inv(sc.Twc[v])
            pc = (Tcw[:3, :3] @ (ob["Two"][:3, :3] @ corners.T + ob["Two"][:3, 3:4]) + Tcw[:3, 3:4]).T
            if (pc[:, 2] <= 0.05).any():
                continue
            u = pc[:, 0] / pc[:, 2] * sc.fx + sc.cx; w_ = pc[:, 1] / pc[:, 2] * sc.fy + sc.cy
            x0, x1 = int(max(0, math.floor(u.min()))), int(min(W - 1, math.ceil(u.max())))
            y0, y1 = int(max(0, math.floor(w_.min()))), int(min(H - 1, math.ceil(w_.max())))
            if x1 - x0 < 4 or y1 - y0 < 4:
                continue
            bw, bh = x1 - x0, y1 - y0
            if crop is not None:
                ch, cw = min(crop[0], bh), min(crop[1], bw)
                x0 += (bw - cw) // 2; y0 += (bh - ch) // 2; bw, bh = cw, ch
            boxes.append((v, x0, y0, bh, bw))          # FrameId, x, y, h, w  (common.h:18-23)
        ob["boxes"] = np.array(boxes, np.uint32)
    return sc


def colmajor(T):
    return np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(16))


def write_sequence(sc, out_dir):
    """Writes config.yaml / img.txt / groundtruth.txt / rgb|depth|instance PNGs / obj_offline/k.txt."""
    from PIL import Image
    for sub in ("rgb", "depth", "instance", "obj_offline"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    with open(os.path.join(out_dir, "config.yaml"), "w") as f:
        # decoys a substring / first-match reader would trip over (cv::FileStorage matches keys exactly): a comment naming a key, longer keys with the same
        # prefix
        f.write("%%YAML:1.0\n# Camera.fx: 1.0 (an old calibration, commented out)\nCamera.Height_mm: 9999\nCamera.Width_mm: 9999\nCamera.fx: %.6f\nCamera.fy: %.6f\nCamera.cx: %.6f\nCamera.cy: %.6f\nCamera.H: %d\nCamera.W: %d\nDepthMapFactor: %.8f\n"
                % (sc.fx, sc.fy, sc.cx, sc.cy, sc.H, sc.W, 1.0 / 5000.0))
    with open(os.path.join(out_dir, "img.txt"), "w") as fi, open(os.path.join(out_dir, "groundtruth.txt"), "w") as fg:
        fi.write("# stamp name\n"); fg.write("# stamp tx ty tz qx qy qz qw\n")
        for v in range(sc.n_views):
            stamp = "%.6f" % (v * 0.1); name = "%04d.png" % v
            fi.write("%s %s\n" % (stamp, name))
            q = _quat_from_R(sc.Twc[v][:3, :3]); t = sc.Twc[v][:3, 3]
            fg.write("%s %.8f %.8f %.8f %.8f %.8f %.8f %.8f\n" % ((stamp,) + tuple(t) + q))
            Image.fromarray(sc.rgb[v]).save(os.path.join(out_dir, "rgb", name))
            Image.fromarray(np.clip(np.rint(sc.depth[v] * 5000.0), 0, 65535).astype(np.uint16)).save(os.path.join(out_dir, "depth", name))
            Image.fromarray(sc.instance[v]).save(os.path.join(out_dir, "instance", name))
    for k, ob in enumerate(sc.objects):
        with open(os.path.join(out_dir, "obj_offline", "%d.txt" % k), "w") as f:
            f.write("# class tx ty tz qx qy qz qw a1 a2 a3 / stamp x y h w\n")
            q = _quat_from_R(ob["Two"][:3, :3]); t = ob["Two"][:3, 3]
            f.write("%d %.8f %.8f %.8f %.8f %.8f %.8f %.8f %.6f %.6f %.6f\n" % ((ob["cls"],) + tuple(t) + q + tuple(ob["half"])))
            for (v, x, y, h, w) in ob["boxes"]:
                f.write("%.6f %d %d %d %d\n" % (v * 0.1, x, y, h, w))
