#!/usr/bin/env python
"""bench.py -- ray-samples/s through hash-encode -> MLP -> composite (train step = fwd + bwd + optimizer) for
per-object NeRFs, one process per GPU.  Contract: see the round brief; prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1]): OfflineNeRF-style training of one object NeRF per GPU with the reference's
base.json defaults (hash L=16 F=2 T=2^16, MLP 64x1, R=4096 rays x S=32 samples = 131072 ray-samples per step) on a
synthetic 'room'-like sequence (40 views, 640x480) that is resident in HBM before the timed region starts.
A "step" is one iteration of NeRF_Model::Train_Step's loop (GenerateBatch -> forward -> composite -> loss
gradient -> backward -> Adam/EMA), nerf_model.cu:1637-1646.  N > 1: objects shard one per rank, no data-path
collective while training; the final render is gathered over RCCL (torch.distributed backend "nccl").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per ray-sample of one training step, SURVEY.md 8(d): 52 + 96*L  (L hash levels, F=2, fp16 table)
def train_bytes_per_sample(L):
    return 52 + 96 * L


def train_bytes_per_sample_fused(L):          # k_fused_train's share: everything except the gradient scatter RMW (64*L)
    return 52 + 32 * L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--backend", type=int, default=-1, help="-1 library default, 0 unfused, 1 fused MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--log2-hashmap-size", type=int, default=0, help="override base.json's T (BASELINE configs[4] stress: 22); 0 = base.json")
    ap.add_argument("--objects-per-gpu", type=int, default=4, help="extra (not the headline): aggregate rate of K objects trained concurrently on one GPU, the manager's thread-per-object mode; 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch
    import importlib
    import __graft_entry__ as ge
    pkg = ge.load_package(); ss = ge.load_tools()
    sharding = importlib.import_module("ro_map_amd.sharding")
    ndev = pkg.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    device = local_rank % ndev
    dist = None; coll_dev = "cpu"
    if world > 1 or os.environ.get("MON_BENCH_FORCE_DIST"):      # FORCE_DIST: exercise the RCCL path with world_size 1 on a 1-GPU box
        import torch.distributed as dist
        # "nccl" IS RCCL on ROCm (xGMI); MON_BENCH_DIST_BACKEND=gloo lets the N>1 path be exercised on a 1-GPU box
        backend = os.environ.get("MON_BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(device); coll_dev = torch.device("cuda", device)
            dist.init_process_group(backend="nccl", device_id=coll_dev)
        else:
            dist.init_process_group(backend=backend)

    # ---- workload: resident in HBM before timing
    sc = ss.make_scene(n_views=args.views, H=480, W=640, f=525.0, seed=0)
    cfg_kw = dict(sample_seed=2024 + rank)            # every rank trains its own object NeRF (independent units)
    if args.log2_hashmap_size:
        cfg_kw["log2_hashmap_size"] = args.log2_hashmap_size
    # host -> HBM hand-over of the frames (mon_dataset_add_frame packs rgb + instance into 4 B/pixel and copies; not in the timed region)
    ds = pkg.Dataset(device, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, sc.n_views, use_depth=False)      # (first HIP call of the process: context creation)
    tu0 = time.perf_counter()
    for v in range(sc.n_views):
        ds.add_frame(v, sc.rgb[v], sc.instance[v], ss.colmajor(sc.Twc[v]))
    pkg.lib().mon_device_synchronize(device); upload_s = time.perf_counter() - tu0
    _, obj = ge.make_problem(pkg, sc, cfg_kw, device=device, dataset=ds)
    if args.backend >= 0:
        obj.set_backend(args.backend)
    cfg = obj.cfg; L = cfg.n_levels; B = cfg.rays_per_batch * cfg.n_samples

    def sync():
        pkg.lib().mon_device_synchronize(device)
        if torch.cuda.device_count() > 0:      # (is_available() can answer False once another HIP user -- the library -- has initialised the device)
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    obj.train(args.warmup)
    barrier(); sync()
    t0 = time.perf_counter()
    obj.train(args.steps)              # K iterations enqueued on the object's HIP stream, one sync at the end
    sync(); barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = sharding.max_over_ranks(dist, torch, dt, coll_dev)
    value = world * args.steps * B / dt

    # ---- roofline of the dominant kernel, HIP events on the kernel's own stream (the object's train stream).
    # a training step moves 52 + 96*L algorithmic bytes per ray-sample.  The fused backend splits them over two
    # kernels: k_fused_train = forward gathers + outputs + dL/dO (52 + 32*L), k_grid_scatter = the gradient
    # scatter read-modify-write (64*L).  The unfused backend is one kernel group timed as a whole.
    obj.set_profiling(True); obj.profile(reset=True)
    sc0 = int(obj.buffer("state")[25])
    obj.train(args.steps); prof = obj.profile(reset=True); obj.set_profiling(False)
    scattered = ((int(obj.buffer("state")[25]) - sc0) % (1 << 32)) / float(args.steps)      # samples with a non-zero gradient per step in this window
    avg = lambda k: prof["ms"][k] / max(1, prof["launches"][k])
    fused = obj_backend(pkg, obj) == 1
    fb_ms, sc_ms, rd_ms = avg(1), avg(4), avg(5)
    dom_bytes = (train_bytes_per_sample_fused(L) if fused else train_bytes_per_sample(L)) * B
    achieved = dom_bytes / (fb_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    base_cfg = not args.log2_hashmap_size            # the committed PMC numbers were collected on the base.json workload only
    if os.path.exists(pmc) and base_cfg:
        try:
            traffic = json.load(open(pmc)).get("k_fused_train_hbm_bytes_per_launch" if fused else "unfused_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "kernel": "k_fused_train" if fused else "unfused fwd+bwd kernel group",
                "avg_launch_ms": round(fb_ms, 4), "algorithmic_bytes_per_launch": dom_bytes,
                "measured_over": "HIP events around every launch on the object's train stream during the %d steps that follow the timed region "
                                 "(events between the launches add ~37 us per step, so the timed region itself runs without them)" % args.steps}
    if fused:
        grp_ms = fb_ms + sc_ms
        sc_bytes = 64 * L * scattered                # the scatter's read-modify-write bytes of the samples that carry a gradient
        roofline["scatter_kernel"] = {"kernel": "k_grid_scatter", "avg_launch_ms": round(sc_ms, 4), "gradient_carrying_samples_per_launch": round(scattered, 1),
                                      "algorithmic_bytes_per_launch": int(sc_bytes), "achieved": round(sc_bytes / (sc_ms * 1e-3) / 1e9, 2),
                                      "frac": round(sc_bytes / (sc_ms * 1e-3) / 1e9 / 8000.0, 4)}
        pair_bytes = train_bytes_per_sample_fused(L) * B + sc_bytes
        roofline["fwd_bwd_pair"] = {"avg_ms": round(grp_ms, 4), "algorithmic_bytes": int(pair_bytes),
                                    "achieved": round(pair_bytes / (grp_ms * 1e-3) / 1e9, 2), "frac": round(pair_bytes / (grp_ms * 1e-3) / 1e9 / 8000.0, 4)}
    roofline["other_kernels_ms"] = {"candidates+frags (folded into k_optimizer in steady state)": round(avg(0), 4),
                                    "reduce_partials (folded into k_grid_scatter)": round(rd_ms, 4), "optimizer": round(avg(2), 4)}
    # MFMA side of the same kernel (the tiny-GEMM of the MLP is the only matrix work): algorithmic flops = 2 * MACs of forward, input
    # gradients and weight gradients (3 GEMMs per layer, no padding counted), against the dense fp16 peak
    W_, NH_, F_in = cfg.n_neurons, cfg.n_hidden_layers, 2 * L
    macs = F_in * W_ + (NH_ - 1) * W_ * W_ + W_ * 4
    mlp_flops = 3 * 2 * macs * B
    roofline["mfma"] = {"algorithmic_flops_per_launch": mlp_flops, "achieved": round(mlp_flops / (fb_ms * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(mlp_flops / (fb_ms * 1e-3) / 1e12 / 2500.0, 4), "note": "the path is gather-bound; MFMA is used only for the MLP's tiny GEMMs"}
    if fused and os.path.exists(pmc) and base_cfg:
        # the bound that actually holds k_fused_train: its 4-byte hash-grid gathers are one L2 request per distinct 64-byte line per
        # instruction, and the chip serves ~270 G of those per second (profiles/r01_microbench.md); requests from the PMC pass
        try:
            pj = json.load(open(pmc)); req = pj.get("k_fused_train_l2_read_requests_per_launch"); rate = pj.get("l2_line_request_rate_measured_per_s")
            if req and rate:
                roofline["l2_request_bound"] = {"requests_per_launch": req, "measured_peak_requests_per_s": rate, "min_ms": round(1e3 * req / rate, 4),
                                                "frac_of_kernel_time": round(1e3 * req / rate / fb_ms, 4)}
        except Exception:
            pass

    # ---- extra, not the headline: the same measurement late in training (the scatter handles only the samples that still carry a
    #      gradient; an OfflineNeRF job runs 5000 iterations)
    late = None
    if fused:
        done = args.warmup + 2 * args.steps; extra = max(0, 800 - done)
        obj.train(extra) if extra else None
        barrier(); sync(); tl0 = time.perf_counter(); obj.train(args.steps); sync(); barrier(); tl = time.perf_counter() - tl0
        if dist is not None:
            tl = sharding.max_over_ranks(dist, torch, tl, coll_dev)
        late = {"after_steps": done + extra, "ms_per_step": round(1e3 * tl / args.steps, 4), "value": round(world * args.steps * B / tl, 1), "unit": "ray-samples/s"}

    # ---- quality: PSNR of a rendered crop vs the synthetic ground truth after (2W + 2K) steps; gathered over RCCL when N > 1
    box = sc.objects[0]["boxes"][0]; v, x, y, h, w = (int(q) for q in box)
    rgb, depth, mask = obj.render(box, ss.colmajor(sc.Twc[v]))
    gm = sc.instance[v, y:y + h, x:x + w] > 0
    gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
    my_psnr = float(-10 * np.log10(max(1e-12, ((rgb - gt) ** 2).mean())))
    # render throughput (NeRF_Model::Render, 2S = 64 samples per pixel ray, nominal count like the reference which evaluates every pixel)
    sync(); tr0 = time.perf_counter(); n_rep = 20
    for _ in range(n_rep):
        obj.render(box, ss.colmajor(sc.Twc[v]))
    sync(); tr = (time.perf_counter() - tr0) / n_rep
    render_info = {"crop": [h, w], "ms_per_crop_incl_d2h": round(1e3 * tr, 3), "nominal_ray_samples_per_s": round(h * w * 2 * cfg.n_samples / tr, 1)}
    psnrs = [my_psnr]
    if dist is not None:
        # RCCL over xGMI: the only collective on the path -- every rank's rendered crop gathered for the final image set
        crops = sharding.gather_crops(dist, torch, [sharding.pack_crop(rgb, depth, mask)], coll_dev)
        psnrs = [float(-10 * np.log10(max(1e-12, ((items[0][0] - gt) ** 2).mean()))) for items in crops if items]

    # ---- CPU baseline: the oracle (port of the same algorithm), bounded sample, rank 0 at N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            orc = ge.load_oracle()
            orc.lib().orc_set_parallel_scatter(1)       # parallel scatter too (fp32 atomics)
            orc.lib().orc_set_threads(int(os.environ.get("MON_CPU_BASELINE_THREADS", min(64, os.cpu_count() or 1))))
            ref = ge.make_oracle(orc, sc, {})
            ref.train(1)
            t1 = time.perf_counter(); n = 0
            while time.perf_counter() - t1 < args.cpu_seconds:
                ref.train(1); n += 1
            cdt = time.perf_counter() - t1
            cpu = {"value": round(n * B / cdt, 1), "unit": "ray-samples/s", "cores": orc.lib().orc_max_threads(), "kind": "port",
                   "sample": "%d full training steps (R=4096 x S=32, base.json network) of oracle/mon_oracle.c with OpenMP in %.1f s" % (n, cdt)}
            ref.close()
        except Exception as e:                       # a reported side figure must not cost the headline line
            cpu = {"value": None, "unit": "ray-samples/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}

    # ---- extra, not the headline: K object NeRFs trained concurrently on this GPU (thread + HIP stream per object, as the managers do,
    #      CORE/src/nerf_manager.cu:89,259); the kernels of different objects overlap, so the aggregate rate says how much of the chip
    #      one object's launch chain leaves idle
    multi = None
    if rank == 0 and world == 1 and args.objects_per_gpu > 1:
        try:
            import threading
            K = args.objects_per_gpu; others = []
            for k in range(1, K):
                _, o2 = ge.make_problem(pkg, sc, dict(sample_seed=3000 + k), device=device, dataset=ds)
                if args.backend >= 0:
                    o2.set_backend(args.backend)
                o2.train(args.warmup + args.steps); others.append(o2)             # same training stage as the first object
            objs = [obj] + others
            sync(); tm0 = time.perf_counter()
            th = [threading.Thread(target=o.train, args=(args.steps,)) for o in objs]
            [t.start() for t in th]; [t.join() for t in th]
            sync(); tm = time.perf_counter() - tm0
            multi = {"objects": K, "value": round(K * args.steps * B / tm, 1), "unit": "ray-samples/s (all objects)", "ms_per_step_per_object": round(1e3 * tm / args.steps, 4),
                     "note": "K independent objects, one host thread and one HIP stream each, same GPU"}
            for o2 in others:
                o2.close()
        except Exception as e:
            multi = {"objects": args.objects_per_gpu, "value": None, "note": "failed: %s" % e}

    if rank == 0:
        out = {"metric": "ray-samples/sec (train: hash-encode->MLP->composite fwd+bwd+optimizer) per object-NeRF", "value": round(value, 1),
               "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16 params/activations, fp32 accumulate + fp32 master",
               "data": "synthetic",
               "config": {"workload": "OfflineNeRF-style training, 1 synthetic 'room'-like object per GPU, base.json defaults (hash L=16 F=2 T=2^16, MLP 64x1), "
                                      "R=4096 rays x S=32 samples/step, %d views 640x480 resident in HBM%s" % (args.views, (", T overridden to 2^%d" % args.log2_hashmap_size) if args.log2_hashmap_size else ""),
                          "objects": world, "rays_per_step": cfg.rays_per_batch, "samples_per_ray": cfg.n_samples, "backend": obj_backend(pkg, obj),
                          "parallelism": "object-per-GPU (no training collective; RCCL all_gather of the final render)"},
               "roofline": roofline, "cpu_baseline": cpu,
               "late_training": late, "multi_object": multi,
               "pcie_inclusive": {"dataset_upload_ms": round(1e3 * upload_s, 2), "dataset_bytes": int(sc.n_views * sc.H * sc.W * 4),
                                  "value_for_a_5000_step_job": round(world * 5000 * B / (upload_s + 5000 * dt / args.steps), 1), "unit": "ray-samples/s",
                                  "note": "host frames -> HBM once per sequence (pack + hipMemcpy), then 5000 steps at the measured step time; never the headline value"},
               "render": render_info, "psnr_db": [round(p, 2) for p in psnrs], "train_steps_before_render": (late["after_steps"] + args.steps) if late else args.warmup + 2 * args.steps,
               "final_loss": round(obj.info().last_loss, 5)}
        print(json.dumps(out))
    obj.close(); ds.close()
    if dist is not None:
        dist.destroy_process_group()


def obj_backend(pkg, obj):
    return int(obj.info().backend)


if __name__ == "__main__":
    main()
