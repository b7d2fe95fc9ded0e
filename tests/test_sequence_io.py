"""Sequence-format support of the drop-in boundary: PNG codec (CPU) and the NerfManagerOffline flow (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def test_png_codec_against_pillow(pkg, tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    rgb = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8); gray = rs.randint(0, 256, (37, 53)).astype(np.uint8)
    d16 = rs.randint(0, 65536, (37, 53)).astype(np.uint16); rgba = rs.randint(0, 256, (20, 31, 4)).astype(np.uint8)
    # Pillow -> ours (all PNG filter types appear in Pillow's adaptive filtering)
    Image.fromarray(rgb).save(tmp_path / "a.png"); Image.fromarray(gray).save(tmp_path / "b.png"); Image.fromarray(d16).save(tmp_path / "c.png")
    Image.fromarray(rgba).save(tmp_path / "d.png")
    smooth = (np.add.outer(np.arange(64), np.arange(80))[..., None] * np.array([1, 2, 3])).astype(np.uint8); Image.fromarray(smooth).save(tmp_path / "e.png")
    assert np.array_equal(pkg.png_read(str(tmp_path / "a.png")), rgb)
    assert np.array_equal(pkg.png_read(str(tmp_path / "b.png"))[..., 0], gray)
    assert np.array_equal(pkg.png_read(str(tmp_path / "c.png"))[..., 0], d16)
    assert np.array_equal(pkg.png_read(str(tmp_path / "d.png")), rgba)
    assert np.array_equal(pkg.png_read(str(tmp_path / "e.png")), smooth)
    # ours -> Pillow
    pkg.png_write(str(tmp_path / "w1.png"), rgb); pkg.png_write(str(tmp_path / "w2.png"), gray); pkg.png_write(str(tmp_path / "w3.png"), d16)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "w1.png")), rgb)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "w2.png")), gray)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "w3.png")).astype(np.uint16), d16)
    with pytest.raises(pkg.MonError):
        pkg.png_read(str(tmp_path / "missing.png"))
    (tmp_path / "bad.png").write_bytes(b"not a png at all, definitely" * 4)
    with pytest.raises(pkg.MonError):
        pkg.png_read(str(tmp_path / "bad.png"))
    # damaged files fail cleanly: truncations at every 97th byte, flipped bytes in the header and in the compressed stream
    good = (tmp_path / "a.png").read_bytes()
    for cut in range(8, len(good) - 1, 97):
        (tmp_path / "t.png").write_bytes(good[:cut])
        with pytest.raises(pkg.MonError):
            pkg.png_read(str(tmp_path / "t.png"))
    rs2 = np.random.RandomState(1)
    for pos in list(range(16, 30)) + [int(v) for v in rs2.randint(40, len(good) - 16, 40)]:
        b = bytearray(good); b[pos] ^= 0xFF; (tmp_path / "f.png").write_bytes(bytes(b))
        try:
            out = pkg.png_read(str(tmp_path / "f.png"))          # a flipped pixel byte may still decode; it must not crash
            assert out.ndim == 3
        except pkg.MonError:
            pass


def test_png_reader_takes_what_cv_imread_takes(pkg, tmp_path):
    """Palette (instance masks are often written that way), 1/2/4-bit gray, palette + tRNS, gray + alpha, 16-bit RGB and Adam7-interlaced
    files, each against Pillow's decode; a short IHDR chunk is rejected instead of over-read."""
    import struct
    import zlib
    from PIL import Image
    rs = np.random.RandomState(3); H, W = 23, 41
    idx = rs.randint(0, 7, (H, W)).astype(np.uint8)
    pal = Image.fromarray(idx, mode="P"); palette = rs.randint(0, 256, 7 * 3).astype(np.uint8); pal.putpalette(palette.tolist())
    pal.save(tmp_path / "p8.png"); pal.save(tmp_path / "p4.png", bits=4)
    got = pkg.png_read(str(tmp_path / "p8.png")); want = palette.reshape(-1, 3)[idx]
    assert got.shape == (H, W, 3) and np.array_equal(got, want) and np.array_equal(pkg.png_read(str(tmp_path / "p4.png")), want)
    pal.save(tmp_path / "pt.png", transparency=2)                             # tRNS: palette entry 2 fully transparent -> RGBA like libpng's expand
    got = pkg.png_read(str(tmp_path / "pt.png"))
    assert got.shape == (H, W, 4) and np.array_equal(got, np.asarray(Image.open(tmp_path / "pt.png").convert("RGBA")))
    bw = (rs.rand(H, W) > 0.5); Image.fromarray(bw).save(tmp_path / "g1.png")           # 1-bit gray
    assert np.array_equal(pkg.png_read(str(tmp_path / "g1.png"))[..., 0], bw.astype(np.uint8) * 255)
    # 2- and 4-bit gray, written by hand (Pillow only writes them for palettes): packed samples, filter 0
    for bits in (2, 4):
        v = rs.randint(0, 1 << bits, (H, W)).astype(np.uint8); rows = b""
        for y in range(H):
            acc = 0; nb = 0; line = bytearray()
            for x in range(W):
                acc = (acc << bits) | int(v[y, x]); nb += bits
                if nb == 8: line.append(acc); acc = 0; nb = 0
            if nb: line.append(acc << (8 - nb))
            rows += b"\x00" + bytes(line)
        def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, bits, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b"")
        (tmp_path / ("g%d.png" % bits)).write_bytes(png)
        assert np.array_equal(pkg.png_read(str(tmp_path / ("g%d.png" % bits)))[..., 0], v * (255 // ((1 << bits) - 1)))
        assert np.array_equal(np.asarray(Image.open(tmp_path / ("g%d.png" % bits)).convert("L")), v * (255 // ((1 << bits) - 1)))
    la = rs.randint(0, 256, (H, W, 2)).astype(np.uint8); Image.fromarray(la, mode="LA").save(tmp_path / "la.png")
    assert np.array_equal(pkg.png_read(str(tmp_path / "la.png")), la)
    # Adam7: re-encode a Pillow file's pixels interlaced by hand (Pillow cannot write interlaced PNGs): 7 passes, filter 0
    rgb = rs.randint(0, 256, (H, W, 3)).astype(np.uint8); raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = rgb[y0::dy, x0::dx]
        if sub.size:
            raw += b"".join(b"\x00" + sub[r].tobytes() for r in range(sub.shape[0]))
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    (tmp_path / "i.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 1)) + chunk(b"IDAT",
            zlib.compress(raw)) + chunk(b"IEND", b""))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "i.png")), rgb) and np.array_equal(pkg.png_read(str(tmp_path / "i.png")), rgb)
    # a 12-byte IHDR (crafted) must be refused, not read past
    (tmp_path / "short.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBB", W, H, 8, 2, 0, 0)) + chunk(b"IDAT",
            zlib.compress(b"\x00" * 64)) + chunk(b"IEND", b""))
    with pytest.raises(pkg.MonError):
        pkg.png_read(str(tmp_path / "short.png"))


def test_offline_manager_errors_without_dataset(pkg, tmp_path):
    m = pkg.OfflineManager(str(tmp_path), os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"))
    if pkg.device_count() == 0:
        with pytest.raises(pkg.MonError) as e:
            m.init()
        assert e.value.code == 2                       # no device: fails loudly, no CPU fallback
    else:
        m.init()
        with pytest.raises(pkg.MonError) as e:
            m.read_dataset()
        assert e.value.code == 4
    with pytest.raises(pkg.MonError):
        m.create_nerf(str(tmp_path / "nope.txt"))
    m.close()


@pytest.mark.gpu
def test_offline_nerf_flow_on_disk_sequence(pkg, ss, tmp_path):
    """OfflineNeRF's call sequence (MON/main.cpp:322-340) on a synthetic sequence written in the reference's on-disk layout:
    3 objects, thread per object, test images written as PNG and compared with the ground-truth frames."""
    from PIL import Image
    assert pkg.device_count() >= 1
    sc = ss.make_scene(n_views=10, H=120, W=160, f=130.0, n_objects=3, seed=5)
    seq = str(tmp_path / "seq"); ss.write_sequence(sc, seq)
    pkg.set_offline_schedule(2, 150)
    m = pkg.OfflineManager(seq, os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"), use_dense_depth=True)
    m.init(); m.read_dataset()
    pkg.set_offline_schedule(10, 500)      # (read by init)
    out = str(tmp_path / "out"); os.makedirs(out); m.set_output_dir(out)
    for k in range(3):
        m.create_nerf(os.path.join(seq, "obj_offline", "%d.txt" % k))
    m.wait_threads_end()
    assert m.n_objects() == 3
    # what the OfflineNeRF viewer reads back (MON/main.cpp:55,149-151,334-336)
    fx, fy, cx, cy, H, W = m.intrinsics(); assert (H, W) == (sc.H, sc.W) and abs(fx - sc.fx) < 1e-4 and abs(cy - sc.cy) < 1e-4
    T = m.poses(); assert T.shape == (sc.n_views, 16) and np.allclose(T[3].reshape(4, 4).T, sc.Twc[3], atol=1e-5)
    meta = m.object_meta(1); ob1 = sc.objects[1]
    assert meta["cls"] == ob1["cls"] and np.array_equal(meta["boxes"], ob1["boxes"]) and np.allclose(meta["aabb_max"], ob1["half"], atol=1e-5)
    assert np.allclose(meta["Tow"].reshape(4, 4).T, ob1["Tow"], atol=1e-5)
    for k in range(3):
        loss, dev = m.object_loss(k); assert loss < 0.08 and dev == k % pkg.device_count()
        m.render_test(k, out, 2)
        ob = sc.objects[k]; v, x, y, h, w = (int(q) for q in ob["boxes"][0]); stamp = "%.6f" % (v * 0.1)
        img = np.asarray(Image.open(os.path.join(out, str(k), "test_img", stamp + ".png"))).astype(np.float64) / 255.0
        msk = np.asarray(Image.open(os.path.join(out, str(k), "test_mask", stamp + ".png"))) > 127
        dep = np.asarray(Image.open(os.path.join(out, str(k), "test_depth", stamp + ".png"))).astype(np.float64) / 20000.0
        gm = sc.instance[v, y:y + h, x:x + w] == ob["cls"]
        gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
        assert img.shape == (h, w, 3)
        iou = (msk & gm).sum() / max(1, (msk | gm).sum())
        psnr = -10 * np.log10(np.mean((img - gt) ** 2))
        assert iou > 0.85 and psnr > 18.0, (k, iou, psnr)
        both = msk & gm
        assert np.abs(dep[both] - sc.depth[v, y:y + h, x:x + w][both]).mean() < 0.08
        # mesh: generated every 2nd outer step on the training thread, saved as <out>/<id>.ply at the end and as obj.ply by render_test
        mesh = m.object(k).get_mesh(try_lock=True); nr = mesh["n_verts_real"]
        assert nr > 100 and mesh["indices"].size % 3 == 0 and mesh["indices"].max() < nr
        p_obj = mesh["verts"][:nr] / ob["radii"]
        assert 0.75 < np.median(np.linalg.norm(p_obj, axis=1)) < 1.25
        for ply in (os.path.join(out, "%d.ply" % k), os.path.join(out, str(k), "obj.ply")):
            assert open(ply).readline().strip() == "ply"
    m.close()
    # the headless executable, same sequence, 1 object
    exe = os.path.join(ROOT, "ro-map_amd", "offline_nerf")
    r = subprocess.run([exe, os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"), seq, "0", "1", str(tmp_path / "out2")], capture_output=True,
            text=True, timeout=300)
    assert r.returncode == 0 and "Training completed" in r.stdout, r.stdout + r.stderr
    assert os.path.exists(os.path.join(str(tmp_path / "out2"), "0", "test_img"))
    # ... and with its test images through the in-process RCCL gather (libmon_core_rccl.so, loaded on demand): the same PNG bytes
    r = subprocess.run([exe, os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"), seq, "0", "1", str(tmp_path / "out3"), "gather"],
            capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Training completed" in r.stdout and "gathered to device 0" in r.stdout, r.stdout + r.stderr
    for sub in ("test_img", "test_depth", "test_mask"):
        a_dir, b_dir = os.path.join(str(tmp_path / "out2"), "0", sub), os.path.join(str(tmp_path / "out3"), "0", sub)
        names = sorted(os.listdir(a_dir)); assert names and names == sorted(os.listdir(b_dir))
        for nm in names:
            assert open(os.path.join(a_dir, nm), "rb").read() == open(os.path.join(b_dir, nm), "rb").read(), (sub, nm)


def test_generate_toc_orbit_pose(pkg):
    """GenerateToc (nerf_model.cu:2186-2205): camera at radius r, 30 degrees up, z axis through the object centre, x axis horizontal."""
    for theta in (6.0, 90.0, 201.0, 360.0):
        T = pkg.generate_toc(theta, 30.0, 0.8).reshape(4, 4).T
        R, t = T[:3, :3], T[:3, 3]
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-5
        assert abs(np.linalg.norm(t) - 0.8) < 1e-6 and abs(t[2] - 0.8 * np.sin(np.radians(30))) < 1e-6
        assert np.allclose(R[:, 2], -t / np.linalg.norm(t), atol=1e-6) and abs(R[2, 0]) < 1e-7
        assert np.allclose(np.arctan2(t[1], t[0]) % (2 * np.pi), np.radians(theta) % (2 * np.pi), atol=1e-5) or theta == 360.0


def test_online_manager_errors_without_device(pkg):
    m = pkg.OnlineManager(os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"), False, 50)
    if pkg.device_count() == 0:
        with pytest.raises(pkg.MonError) as e:
            m.init()
        assert e.value.code == 2
    else:
        m.init()
        with pytest.raises(pkg.MonError) as e:          # CreateNeRF before DatasetInit
            m.create_nerf(1, np.eye(4, dtype=np.float32).reshape(16), [-1, -1, -1], [1, 1, 1])
        assert e.value.code == 4
    assert m.get_frame_idx("0.100000") == -1
    with pytest.raises(pkg.MonError):
        m.update_nerf_bbox(3, np.zeros((1, 5), np.uint32), 1)
    m.close()


@pytest.mark.gpu
def test_online_manager_incremental_flow(pkg, ss, tmp_path):
    """The SLAM-side call sequence (REF/src/LocalMapping.cc:1122-1270): frames land one by one, objects are created when first
    seen, boxes arrive per keyframe with train_step=1, training only starts past 10 boxes, WaitThreadsEnd trains once more."""
    sc = ss.make_scene(n_views=24, H=120, W=160, f=130.0, n_objects=2, seed=9)
    m = pkg.OnlineManager(os.path.join(ROOT, "ro-map_amd", "configs", "c1_small.json"), True, 60)
    m.init(); m.dataset_init(sc.fx, sc.fy, sc.cx, sc.cy, sc.H, sc.W, sc.n_views)
    ids = {}
    import time
    for v in range(sc.n_views):
        stamp = "%.6f" % (v * 0.1)
        m.new_frame(v, stamp, sc.rgb[v][..., ::-1], sc.instance[v], ss.colmajor(sc.Twc[v]), sc.depth[v])
        assert m.get_frame_idx(stamp) == v
        for k, ob in enumerate(sc.objects):
            if k not in ids:
                ids[k] = m.create_nerf(ob["cls"], ss.colmajor(ob["Tow"]), -ob["half"] / 1.1, ob["half"] / 1.1)   # the manager inflates by 1.1
            b = ob["boxes"][ob["boxes"][:, 0] == v]
            m.update_nerf_bbox(ids[k], b, 1)
        if v == 8:
            time.sleep(0.3)
            assert all(m.object_info(i)["train_calls"] == 0 for i in ids.values())      # <= 10 boxes: no training yet (nerf.cu:223)
        # an id that is already in use is overwritten: the one case that excludes the training threads
        if v == 20:
            m.new_frame(3, "%.6f" % 0.3, sc.rgb[3][..., ::-1], sc.instance[3], ss.colmajor(sc.Twc[3]), sc.depth[3])
        if v in (14, 18, 22):                                        # a viewer reads while the training threads run (let in between two slices)
            for i in ids.values():
                bx = sc.objects[0]["boxes"][1]; r, d, k_ = m.render(i, bx, ss.colmajor(sc.Twc[int(bx[0])]))
                assert np.isfinite(r).all() and np.isfinite(d).all() and m.object_info(i)["n_boxes"] >= 0
        time.sleep(0.02)
    m.wait_threads_end()
    for k, i in ids.items():
        info = m.object_info(i); ob = sc.objects[k]
        assert info["n_boxes"] == len(ob["boxes"]) and info["train_calls"] >= 3 and info["loss"] < 0.08, info
        v, x, y, h, w = (int(q) for q in ob["boxes"][3])
        rgb, depth, mask = m.render(i, ob["boxes"][3], ss.colmajor(sc.Twc[v]))
        gm = sc.instance[v, y:y + h, x:x + w] == ob["cls"]
        iou = ((mask > 0.5) & gm).sum() / max(1, ((mask > 0.5) | gm).sum())
        assert iou > 0.8, (k, iou)
        mesh = m.object(i).get_mesh(try_lock=True)                    # DrawMesh(idx)'s data
        # the borrowed handle is a full object
        bo = m.object(i); assert bo.cfg.rays_per_batch == 1024 and int(bo.buffer("state")[0]) == bo.info().train_step > 0 and bo.mesh_generation() >= 1
        assert mesh["n_verts_real"] > 50 and mesh["indices"].max() < mesh["n_verts_real"]
    # RenderNeRFsTest (System.cc:610): test images, test.txt / train.txt, 360-degree video, obj.ply
    from PIL import Image
    out = str(tmp_path / "render"); ob = sc.objects[0]; sel = [2, 7]
    stamps = ["%.6f" % (int(ob["boxes"][j][0]) * 0.1) for j in sel]
    m.render_nerfs_test(out, ids[0], stamps, ob["boxes"][sel], np.stack([ss.colmajor(sc.Twc[int(ob["boxes"][j][0])]) for j in sel]), 0.8)
    root = os.path.join(out, "0")
    test_lines = open(os.path.join(root, "test.txt")).read().strip().split("\n"); train_lines = open(os.path.join(root, "train.txt")).read().strip().split("\n")
    assert len(test_lines) == 3 and len(train_lines) == 3 + len(ob["boxes"]) and test_lines[1].split()[0] == stamps[0]
    # object-centric pose of the first test view: Toc = Tow * Twc (nerf.cu:325-329)
    Toc = ob["Tow"] @ sc.Twc[int(ob["boxes"][sel[0]][0])]; vals = [float(q) for q in test_lines[1].split()[5:]]
    assert np.allclose(vals[:3], Toc[:3, 3], atol=1e-5)
    qx, qy, qz, qw = vals[3:]; R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                                             [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                                             [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    assert np.allclose(R, Toc[:3, :3], atol=1e-4)
    for j in sel:
        st = "%.6f" % (int(ob["boxes"][j][0]) * 0.1)
        assert np.asarray(Image.open(os.path.join(root, "test_img", st + ".png"))).shape == (int(ob["boxes"][j][3]), int(ob["boxes"][j][4]), 3)
    frames = [np.asarray(Image.open(os.path.join(root, "video_img", "%d.png" % i))) for i in (0, 29, 59)]
    assert all(fr.shape == (sc.H // 2, sc.W // 2, 3) for fr in frames)
    cover = [(fr.min(axis=2) < 250).mean() for fr in frames]                     # the object is in view from every orbit pose
    assert min(cover) > 0.02, cover
    assert np.asarray(Image.open(os.path.join(root, "video_depth", "0.png"))).max() > 255          # 16-bit depth x 20000
    assert open(os.path.join(root, "obj.ply")).readline().strip() == "ply"
    m.close()
