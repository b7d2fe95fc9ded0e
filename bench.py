#!/usr/bin/env python
"""bench.py -- ray-samples/s through hash-encode -> MLP -> composite (train step = fwd + bwd + optimizer) for
per-object NeRFs, one process per GPU.  Contract: see the round brief; prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1]): OfflineNeRF-style training of one object NeRF per GPU with the reference's
base.json defaults (hash L=16 F=2 T=2^16, MLP 64x1, R=4096 rays x S=32 samples = 131072 ray-samples per step) on a
synthetic 'room'-like sequence (40 views, 640x480) that is resident in HBM before the timed region starts.
A "step" is one iteration of NeRF_Model::Train_Step's loop (GenerateBatch -> forward -> composite -> loss
gradient -> backward -> Adam/EMA), nerf_model.cu:1637-1646.

N > 1: objects shard one per rank (CORE/src/nerf.cu:27-33: object k -> device k mod N), no data-path collective while
training; the final render is gathered to rank 0 over RCCL (torch.distributed backend "nccl") from device-resident crops.
Launched by torch.distributed.run the ranks come from the environment; launched plainly as `python bench.py --gpus N`
the script spawns its N ranks itself (rank r -> device r mod visible devices; ranks that share a device use gloo, RCCL
refuses two ranks on one GPU).

Timed region: W warm-up steps, then exactly K steps between barrier + device sync on both sides, max over ranks.  The
region is a few milliseconds, so it is repeated on `--repeats` fresh objects (same seeds => the same steps W..W+K from
init each time) and the MEDIAN repeat is the headline; every repeat is listed in `ms_per_step_repeats`."""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# algorithmic bytes per ray-sample of one training step, SURVEY.md 8(d): 52 + 96*L  (L hash levels, F=2, fp16 table)
def train_bytes_per_sample(L):
    return 52 + 96 * L


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="independent repeats of the timed region (fresh object each); the median is reported")
    ap.add_argument("--backend", type=int, default=-1, help="-1 library default, 0 unfused, 1 fused MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--log2-hashmap-size", type=int, default=0, help="override base.json's T (BASELINE configs[4] stress: 22); 0 = base.json")
    ap.add_argument("--no-stress", action="store_true", help="skip the T = 2^22 side figure")
    ap.add_argument("--no-sustained", action="store_true", help="skip the 45 000 further steps of the offline_job leg (~3 s)")
    ap.add_argument("--objects-per-gpu", type=int, default=4,
            help="extra (not the headline): aggregate rate of K objects trained concurrently on one GPU; 0 = skip")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, 127.0.0.1 rendezvous)."""
    import __graft_entry__ as ge
    ndev = ge.load_package().device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MON_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ndev < args.gpus:
            env.setdefault("MON_BENCH_DIST_BACKEND", "gloo")      # several ranks per device: RCCL needs one GPU per rank
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


def median(v):
    s = sorted(v); return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
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
    dist = None; coll_dev = "cpu"; coll_backend = None
    if world > 1 or os.environ.get("MON_BENCH_FORCE_DIST"):      # FORCE_DIST: exercise the RCCL path with world_size 1 on a 1-GPU box
        import torch.distributed as dist
        # "nccl" IS RCCL on ROCm (xGMI); gloo when ranks share a device (1-GPU box) or when asked for
        coll_backend = os.environ.get("MON_BENCH_DIST_BACKEND", "nccl")
        if coll_backend == "nccl" and world > ndev and "MON_BENCH_DIST_BACKEND" not in os.environ:
            coll_backend = "gloo"                      # more ranks than devices (a launcher on a small box): RCCL refuses two ranks on one device
        if coll_backend == "nccl":
            torch.cuda.set_device(device); coll_dev = torch.device("cuda", device)
            dist.init_process_group(backend="nccl", device_id=coll_dev)
        else:
            dist.init_process_group(backend=coll_backend)

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

    def new_object(seed_kw=None):
        _, o = ge.make_problem(pkg, sc, dict(cfg_kw, **(seed_kw or {})), device=device, dataset=ds)
        if args.backend >= 0:
            o.set_backend(args.backend)
        return o

    def sync():
        pkg.lib().mon_device_synchronize(device)
        if torch.cuda.device_count() > 0:      # (is_available() can answer False once another HIP user -- the library -- has initialised the device)
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- timed region, repeated on fresh objects: W warm-up steps, K timed steps, barrier + sync on both sides, max over ranks
    reps = []; obj = None
    for r in range(max(1, args.repeats)):
        if obj is not None:
            obj.close()
        obj = new_object()
        obj.train(args.warmup)
        barrier(); sync()
        t0 = time.perf_counter()
        obj.train(args.steps)              # K iterations enqueued on the object's HIP stream, one sync at the end
        sync(); barrier()
        dt_r = own_last = time.perf_counter() - t0
        if dist is not None:
            dt_r = sharding.max_over_ranks(dist, torch, dt_r, coll_dev)
        reps.append(dt_r)
    dt = median(reps)
    cfg = obj.cfg; L = cfg.n_levels; B = cfg.rays_per_batch * cfg.n_samples
    value = world * args.steps * B / dt
    per_rank = None; rank_devices = [device]
    if dist is not None:                   # every rank's own last-repeat time (the headline uses the max over ranks) and the device it trained on
        t = torch.tensor([own_last, float(device)], dtype=torch.float64, device=coll_dev); outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t); per_rank = [round(args.steps * B / float(o[0].item()), 1) for o in outs]; rank_devices = [int(o[1].item()) for o in outs]

    # ---- roofline of the dominant kernel: HIP events on the kernel's own stream (the object's train stream) around every launch of
    # the SAME window (a fresh object, steps W..W+K from init), un-timed because the events cost ~37 us per step between the launches.
    # a training step moves 52 + 96*L algorithmic bytes per ray-sample.  The fused backend splits them over two
    # kernels: k_fused_train = forward gathers + outputs + dL/dO (52 + 32*L), k_grid_scatter = the gradient
    # scatter read-modify-write (64*L).  The unfused backend is one kernel group timed as a whole.
    pobj = new_object(); pobj.train(args.warmup)
    pobj.set_profiling(True); pobj.profile(reset=True)
    sc0 = int(pobj.buffer("state")[25])
    pobj.train(args.steps); prof = pobj.profile(reset=True); pobj.set_profiling(False)
    # samples with a non-zero gradient per step in this window (DESIGN.md 3.1 (HISTORY 3.2b))
    scattered = ((int(pobj.buffer("state")[25]) - sc0) % (1 << 32)) / float(args.steps)
    pobj.close()
    avg = lambda k: prof["ms"][k] / max(1, prof["launches"][k])
    fused = obj_backend(pkg, obj) == 1
    n_params = int(obj.info().n_params)
    regime = "dense" if scattered > 0.5 * B else "sparse"          # which committed PMC pass matches this window
    pmc = {}
    base_cfg = not args.log2_hashmap_size            # the committed PMC numbers were collected on the base.json workload only
    # the committed profile belongs to the kernel sources it was measured on: their fingerprint is stored with it (tools/fingerprint.py) and recomputed here.
    # A kernel change without a re-profile makes every profile-derived number STALE: it stays in the line (labelled), but `frac` falls back to the live one.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from fingerprint import kernel_sources_sha16
        sources_now = kernel_sources_sha16()
    except Exception:
        sources_now = None
    try:
        if base_cfg:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pmc = dict(pj.get(regime, {}), source=pj.get(regime, {}).get("source"))
    except Exception:
        pmc = {}
    profile_sources = pmc.get("kernel_sources_sha16")
    traffic_stale = bool(pmc) and not (sources_now and profile_sources == sources_now)
    # Algorithmic bytes per launch, SURVEY 8(d) split over the kernels of a step: per ray-sample position 12 + table gathers 32*L (encode),
    # distance 4 + network output w+r 16 + dL/dO w+r 16 + distance re-read 4 (forward/backward), gradient scatter read-modify-write 64*L of the samples that
    # carry a gradient (scatter); per step 40 B per parameter (optimizer: fp16 gradient, fp32 Adam moments + master r/w, step counter r/w, fp16 copy, EMA r/w).
    step_ms = 1e3 * dt / args.steps
    opt_bpp = 38      # 16-bit saturating step counters (exact for base.json's betas; 32-bit ones are a variant build) read + write 2 B instead of 4 B each
    if fused:
        enc_ms = avg(6) + avg(7)
        kern = [("k_encode_tiles", enc_ms, (12 + 32 * L) * B,
                "VALU issue + LDS reads (level tiles in LDS; HBM traffic is the tile copies)")] if enc_ms > 0 else []
        kern += [("k_fused_train", avg(1), ((40 if enc_ms > 0 else 52 + 32 * L) * B),
                "latency of a ray's MLP / composite / backward chain" if enc_ms > 0 else "L1->L2 line requests of the hash-grid gathers"),
                 ("k_grid_scatter", avg(4) + avg(5), 64 * L * scattered, "VALU issue + LDS integer atomics"),
                 ("k_optimizer", avg(2), opt_bpp * n_params, "HBM / Infinity Cache streaming")]
    else:
        kern = [("unfused fwd+bwd kernel group", avg(0) + avg(1), train_bytes_per_sample(L) * B, "global atomics"), ("k_optimizer", avg(2), opt_bpp * n_params,
                "HBM streaming")]
    table = []
    for name, ms, nbytes, limiter in kern:
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = pmc.get(name + "_hbm_bytes_per_launch")
        # the same kernel's rocprofv3 duration in the committed window of this regime (profiles/pmc_traffic.json): an event pair inflates the launch it brackets
        # by ~2 us, so the event times of a step's kernels sum to MORE than the timed step; the rocprofv3 durations fit it
        rp_us = pmc.get(name + "_avg_us")
        rp_gbs = nbytes / (rp_us * 1e-6) / 1e9 if rp_us else None
        table.append({"kernel": name, "avg_launch_ms": round(ms, 4), "avg_launch_ms_is": "HIP events, live (inflated by ~2 us per launch)",
                "algorithmic_bytes_per_launch": int(nbytes), "achieved": round(gbs, 2), "peak": 8000.0, "unit": "GB/s",
                      "frac": round(gbs / 8000.0, 4), "rocprof_launch_ms": round(rp_us * 1e-3, 5) if rp_us else None,
                      "frac_rocprof": round(rp_gbs / 8000.0, 4) if rp_gbs else None, "traffic": tr, "limited_by": limiter})
    dom = max(table, key=lambda r: r["avg_launch_ms"])
    rp_sum = sum(r["rocprof_launch_ms"] for r in table) if all(r["rocprof_launch_ms"] for r in table) else None
    # "bound" names the roof the contract prices the path against (SURVEY 8(d) accounts it in bytes); what the counters name as the kernel's limiter is
    # `limited_by`
    # achieved / frac: from the kernel's rocprofv3 duration in the committed window when that profile was measured on THESE kernel sources (the event pair
    # inflates the launch it brackets; the rocprofv3 durations are the ones that fit the timed step) -- otherwise from the live HIP-event time, flagged
    use_rp = bool(dom["rocprof_launch_ms"]) and not traffic_stale
    rp_ach = round(dom["algorithmic_bytes_per_launch"] / (dom["rocprof_launch_ms"] * 1e-3) / 1e9, 2) if dom["rocprof_launch_ms"] else None
    roofline = {"bound": "hbm", "bound_note": "priced against HBM bytes as SURVEY 8(d) prescribes; the dominant kernel's measured limiter is in limited_by "
                                                "(it is not HBM-bound at base.json size: tables live in L2 / LDS)",
                "achieved": rp_ach if use_rp else dom["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": dom["frac_rocprof"] if use_rp else dom["frac"],
                "frac_is": "algorithmic bytes / the kernel's rocprofv3 duration in the committed window of these kernel sources" if use_rp
                           else "algorithmic bytes / the live HIP-event time (inflated by ~2 us per launch)" + (
                               ": the committed profile was measured on OTHER kernel sources" if traffic_stale else ""),
                "frac_hip_events": dom["frac"], "achieved_hip_events": dom["achieved"],
                "traffic": dom["traffic"], "traffic_stale": traffic_stale,
                "kernel_sources_sha16": sources_now, "profile_kernel_sources_sha16": profile_sources,
                "traffic_regime": regime if dom["traffic"] else None, "traffic_source": pmc.get("source"),
                "kernel": dom["kernel"], "avg_launch_ms": dom["avg_launch_ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                        "limited_by": dom["limited_by"],
                "rocprof_launch_ms": dom["rocprof_launch_ms"], "frac_rocprof": dom["frac_rocprof"],
                "rocprof_launch_ms_sum_of_the_step": round(rp_sum, 5) if rp_sum else None,
                "rocprof_source": pmc.get("source"),
                "gradient_carrying_samples_per_launch": round(scattered, 1),
                "measured_over": "HIP events around every launch on the object's train stream over steps %d..%d from init of a fresh object -- the same "
                                 "window as "
                                 "the timed region, measured separately because the events add ~35 us per step between the launches; an event pair also "
                                 "inflates the launch it brackets by ~2 us -- rocprof_launch_ms / frac_rocprof carry the kernels' rocprofv3 durations "
                                 "of the same window from the committed profile (their sum fits the timed step, the event times do not)"
                                 % (args.warmup, args.warmup + args.steps),
                "kernels": table}
    # the whole step against the contract's bytes: (52 + 96 L) B per nominal ray-sample + 40 B per parameter, over the TIMED step (no
    # events)
    contract_bytes = train_bytes_per_sample(L) * B + 40 * n_params
    roofline["contract"] = {"bytes_per_step": int(contract_bytes), "bytes_per_ray_sample": train_bytes_per_sample(L), "optimizer_bytes_per_step": 40 * n_params,
                            "ms_per_step": round(step_ms, 4), "achieved": round(contract_bytes / (step_ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                            "frac_of_hbm": round(contract_bytes / (step_ms * 1e-3) / 1e9 / 8000.0, 4),
                            "note": "nominal bytes: every sample counts its scatter bytes whether or not its gradient underflowed to zero (DESIGN.md 3.1 (HISTORY 3.2b))"}
    # MFMA side (the tiny GEMMs of the MLP are the only matrix work): algorithmic flops = 2 * MACs of forward, input gradients and weight gradients
    W_, NH_, F_in = cfg.n_neurons, cfg.n_hidden_layers, 2 * L
    macs = F_in * W_ + (NH_ - 1) * W_ * W_ + W_ * 4
    mlp_flops = 3 * 2 * macs * B
    fb_ms = avg(1)
    roofline["mfma"] = {"kernel": "k_fused_train", "algorithmic_flops_per_launch": mlp_flops, "achieved": round(mlp_flops / (fb_ms * 1e-3) / 1e12,
            2) if fb_ms else None, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(mlp_flops / (fb_ms * 1e-3) / 1e12 / 2500.0, 4) if fb_ms else None,
                                "busy_frac_pmc": pmc.get("k_fused_train_mfma_busy_frac"),
                        "note": "MFMA is used only for the MLP's tiny GEMMs (busy_frac_pmc: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of the committed pass); "
                                "the path is not priced against this roof"}

    # ---- extra, not the headline: the same measurement late in training (the scatter handles only the samples that still carry a
    #      gradient, DESIGN.md 3.1 (HISTORY 3.2b); an OfflineNeRF job runs 5000 iterations)
    late = None
    if fused:
        done = args.warmup + args.steps; extra = max(0, 800 - done)
        obj.train(extra) if extra else None
        tls = []
        for _ in range(3):                              # three consecutive windows of K steps, the median is reported
            barrier(); sync(); tl0 = time.perf_counter(); obj.train(args.steps); sync(); barrier(); tl_r = time.perf_counter() - tl0
            if dist is not None:
                tl_r = sharding.max_over_ranks(dist, torch, tl_r, coll_dev)
            tls.append(tl_r)
        tl = median(tls)
        late = {"after_steps": done + extra, "ms_per_step": round(1e3 * tl / args.steps, 4), "value": round(world * args.steps * B / tl, 1),
                "unit": "ray-samples/s",
                "ms_per_step_windows": [round(1e3 * t / args.steps, 4) for t in tls]}

    # ---- extra, not the headline: the late-training window with occupancy-grid skipping switched on (mon_config::occupancy_skip -- named by
    #      BASELINE.json's north star, absent from the reference, hence opt-in: samples in cells a 64^3 density grid marks empty are not evaluated)
    occ = None
    if fused and rank == 0 and world == 1:
        try:
            oo = new_object(dict(occupancy_skip=1)); oo.train(800); sync()
            tos = []
            for _ in range(3):                          # three consecutive windows of K steps, the median is reported (like late_training)
                to0 = time.perf_counter(); oo.train(args.steps); sync(); tos.append(time.perf_counter() - to0)
            to = median(tos)
            occ = {"after_steps": 800, "ms_per_step": round(1e3 * to / args.steps, 4), "value": round(args.steps * B / to, 1),
                    "unit": "ray-samples/s (nominal: skipped samples count)",
                   "ms_per_step_windows": [round(1e3 * t / args.steps, 4) for t in tos],
                   "note": "opt-in approximation (default off; parity and the headline run without it)"}
            oo.close()
        except Exception as e:
            occ = {"value": None, "note": "failed: %s" % e}

    # ---- quality: PSNR of a rendered crop vs the synthetic ground truth; N > 1: every rank's crop is rendered into a tensor on the
    #      collective's device (HBM for RCCL) and gathered to rank 0 (sizes, then one point-to-point message per peer) -- the only collective on the path
    box = sc.objects[0]["boxes"][0]; v, x, y, h, w = (int(q) for q in box)
    gm = sc.instance[v, y:y + h, x:x + w] > 0
    gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
    psnr_of = lambda img: float(-10 * np.log10(max(1e-12, ((img - gt) ** 2).mean())))
    gather_note = None
    if dist is not None:
        # The throughput above is complete at this point; the gather must not be able to take the JSON line with it (a point-to-point transport problem on a
        # node this script has never seen would otherwise leave the driver without a number): it runs in a worker thread with a deadline, and a rank whose
        # gather did not finish reports so, scores its own crop and leaves the process group alone at exit.
        box_res = {}
        # (the render itself is this rank's own work: only the collective sits behind the deadline)
        packed = sharding.render_packed(obj, box, ss.colmajor(sc.Twc[v]), torch, coll_dev)
        def gather_job():
            try:
                if coll_backend == "nccl":
                    torch.cuda.set_device(coll_dev)
                # gather-to-root: only rank 0 holds (and scores) the crops
                box_res["crops"] = sharding.gather_crops(dist, torch, [packed], coll_dev, root=0)
                box_res["ok"] = True
            except Exception as e:                      # noqa: BLE001 -- reported in the JSON line
                box_res["error"] = "%s: %s" % (type(e).__name__, e)
        gt_thread = threading.Thread(target=gather_job, daemon=True); gt_thread.start()
        gt_thread.join(float(os.environ.get("MON_BENCH_GATHER_TIMEOUT", "180")))
        if box_res.get("ok"):
            crops = box_res["crops"]
            psnrs = [psnr_of(items[0][0]) for items in crops if items] if crops is not None else []
        else:
            gather_note = box_res.get("error", "timed out after %s s" % os.environ.get("MON_BENCH_GATHER_TIMEOUT", "180"))
            rgb, depth, mask = obj.render(box, ss.colmajor(sc.Twc[v])); psnrs = [psnr_of(rgb)] if rank == 0 else []
    else:
        rgb, depth, mask = obj.render(box, ss.colmajor(sc.Twc[v]))
        psnrs = [psnr_of(rgb)]
    # render throughput (NeRF_Model::Render, 2S = 64 samples per pixel ray, nominal count like the reference which evaluates every pixel)
    sync(); tr0 = time.perf_counter(); n_rep = 20
    for _ in range(n_rep):
        obj.render(box, ss.colmajor(sc.Twc[v]))
    sync(); tr = (time.perf_counter() - tr0) / n_rep
    render_info = {"crop": [h, w], "ms_per_crop_incl_d2h": round(1e3 * tr, 3), "nominal_ray_samples_per_s": round(h * w * 2 * cfg.n_samples / tr, 1)}
    # render roofline (the other half of BASELINE's metric): HIP events around the crop's kernels on the object's stream (one pair per crop on the tile
    # path: ray kernel, then per chunk of jobs positions -> k_encode_feat -> k_tile_render), algorithmic bytes 16 + 32 L per sample (position 12,
    # table reads 32 L, distance 4) over the NOMINAL samples (every pixel of the crop, 2S each: what the reference evaluates) and over the EVALUATED ones
    # (rays that hit the object's box)
    roofline_render = None
    try:
        obj.set_profiling(True); obj.profile(reset=True)
        for _ in range(n_rep):
            obj.render(box, ss.colmajor(sc.Twc[v]))
        rprof = obj.profile(reset=True); obj.set_profiling(False)
        k_ms = rprof["ms"][3] / n_rep
        S2 = 2 * cfg.n_samples; nominal = h * w * S2
        try:
            evaluated = obj.render_jobs(0) * S2
        except Exception:
            evaluated = None
        bps = 16 + 32 * L
        roofline_render = {"bound": "hbm",
                           "limited_by": "LDS read-instruction rate + VALU issue of k_encode_feat (level tiles in LDS); priced against HBM bytes as "
                                         "SURVEY 8(d) prescribes",
                           "kernel_ms_per_crop": round(k_ms, 4),
                           "bytes_per_ray_sample": bps, "nominal_samples": nominal, "evaluated_samples": evaluated,
                           "achieved_nominal": round(bps * nominal / (k_ms * 1e-3) / 1e9, 2),
                                   "achieved_evaluated": round(bps * evaluated / (k_ms * 1e-3) / 1e9, 2) if evaluated else None,
                           "peak": 8000.0, "unit": "GB/s", "frac_nominal": round(bps * nominal / (k_ms * 1e-3) / 1e9 / 8000.0, 4),
                           "frac_evaluated": round(bps * evaluated / (k_ms * 1e-3) / 1e9 / 8000.0, 4) if evaluated else None,
                           "nominal_ray_samples_per_s_kernels_only": round(nominal / (k_ms * 1e-3), 1),
                           "traffic": pmc.get("render_hbm_bytes_per_crop"),
                                   "measured_over": "HIP events around the kernels of %d renders of the %dx%d crop after %s training steps" % (n_rep, h, w,
                                   "the bench's"),
                           "path": "level tiles (k_encode_feat + k_tile_render)" if pkg.get_option("tile_render") and h * w >= 4096
                                   else "gathers (k_fused_render)"}
        render_info["kernel_ms_per_crop"] = round(k_ms, 4)
    except Exception as e:
        roofline_render = {"value": None, "note": "failed: %s" % e}
    roofline["render"] = roofline_render

    # ---- one SUSTAINED leg, measured wall to wall: an OfflineNeRF job's training = 10 x Train_Step(500 iterations) of one fresh object
    #      (nerf_manager.cu:89 x nerf_model.cu:1635), ~0.33 s of continuous GPU work, one host sync per Train_Step like the reference's loss read-back
    offline_job = None
    if fused and rank == 0 and world == 1:
        try:
            jo = new_object(); sync(); tj0 = time.perf_counter(); losses = []
            for _ in range(10):
                losses.append(jo.train(500))
            sync(); tj = time.perf_counter() - tj0
            offline_job = {"steps": 5000, "wall_s": round(tj, 4), "ms_per_step": round(1e3 * tj / 5000, 4), "value": round(5000 * B / tj, 1),
                           "unit": "ray-samples/s", "final_loss": round(float(losses[-1]), 5) if losses[-1] is not None else None,
                           "note": "10 x mon_object_train(500) from init, wall clock around all of it (dataset already resident)"}
            # ... and the same object trained on to 50 000 steps (the learning-rate decay of base.json:10 acts from step 20 000): ~3 s of continuous GPU work, long
            # enough for a once-a-second utilisation sampler to see the device busy
            if not args.no_sustained:
                sync(); tk0 = time.perf_counter()
                for _ in range(9):
                    last = jo.train(5000)
                sync(); tk = time.perf_counter() - tk0
                offline_job["continued_to_50000_steps"] = {"steps": 45000, "wall_s": round(tk, 4), "ms_per_step": round(1e3 * tk / 45000, 4),
                                                           "value": round(45000 * B / tk, 1), "final_loss": round(float(last), 5)}
            jo.close()
        except Exception as e:
            offline_job = {"value": None, "note": "failed: %s" % e}
    # ---- BASELINE configs[0] (the reference's CPU-runnable case: R = 1024, hash L = 4, MLP 2 x 32) on the GPU, beside cpu_baseline_c1
    gpu_c1 = None
    if fused and rank == 0 and world == 1:
        try:
            c1_kw = dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2)
            co = new_object(c1_kw); co.train(args.warmup); sync(); tcs = []
            for _ in range(3):
                tc0 = time.perf_counter(); co.train(max(200, args.steps)); sync(); tcs.append((time.perf_counter() - tc0) / max(200, args.steps))
            tc = median(tcs); Bc1 = co.cfg.rays_per_batch * co.cfg.n_samples
            gpu_c1 = {"value": round(Bc1 / tc, 1), "unit": "ray-samples/s", "ms_per_step": round(1e3 * tc, 4), "backend": obj_backend(pkg, co),
                      "workload": "BASELINE configs[0]: R=1024 x S=32, hash L=4, MLP 2x32; steps %d.. from init, median of 3 windows" % args.warmup}
            co.close()
        except Exception as e:
            gpu_c1 = {"value": None, "note": "failed: %s" % e}

    # ---- CPU baseline: the oracle (port of the same algorithm), bounded samples, rank 0 at N=1 only.  `cpu_baseline` is the headline
    #      workload (BASELINE configs[1]); `cpu_baseline_c1` is BASELINE configs[0], the reference's own CPU-runnable case
    cpu = None; cpu_c1 = None; cpu_render = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def time_oracle(kw, seconds, what):
            try:
                orc = ge.load_oracle()
                orc.lib().orc_set_parallel_scatter(1)       # parallel scatter too (fp32 atomics)
                orc.lib().orc_set_threads(int(os.environ.get("MON_CPU_BASELINE_THREADS", min(64, os.cpu_count() or 1))))
                ref = ge.make_oracle(orc, sc, kw)
                ref.train(1)
                Bc = ref.cfg.rays_per_batch * ref.cfg.n_samples
                t1 = time.perf_counter(); n = 0
                while time.perf_counter() - t1 < seconds:
                    ref.train(1); n += 1
                cdt = time.perf_counter() - t1
                out = {"value": round(n * Bc / cdt, 1), "unit": "ray-samples/s", "cores": orc.lib().orc_max_threads(), "kind": "port",
                       "sample": "%d full training steps (%s) of oracle/mon_oracle.c with OpenMP in %.1f s" % (n, what, cdt)}
                ref.close(); return out
            except Exception as e:                       # a reported side figure must not cost the headline line
                return {"value": None, "unit": "ray-samples/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        # CPU-only inference of the same tiny MLP + ray march (north_star): the oracle's orc_render of the same crop with the GPU object's weights
        def time_oracle_render(seconds):
            try:
                orc = ge.load_oracle()
                orc.lib().orc_set_threads(int(os.environ.get("MON_CPU_BASELINE_THREADS", min(64, os.cpu_count() or 1))))
                ref = ge.make_oracle(orc, sc, cfg_kw)
                ref.set_params(obj.get_params(0)); ref.set_ema(obj.get_params(2))
                pose = ss.colmajor(sc.Twc[v])
                rr, rd, rm = ref.render(box, pose)                                   # (first call: page faults, thread start)
                t1 = time.perf_counter(); n = 0
                while n < 1 or time.perf_counter() - t1 < seconds:
                    rr, rd, rm = ref.render(box, pose); n += 1
                cdt = (time.perf_counter() - t1) / n
                grgb, gd, gm = obj.render(box, pose)
                same = gm == rm
                out = {"value": round(h * w * 2 * cfg.n_samples / cdt, 1), "unit": "nominal ray-samples/s", "cores": orc.lib().orc_max_threads(),
                        "kind": "port",
                       "ms_per_crop": round(1e3 * cdt, 2),
                       "sample": "%d renders of the %dx%d crop (2S = %d samples per pixel ray) by oracle/mon_oracle.c orc_render with OpenMP, "
                                 "the GPU object's EMA weights" % (n, h, w, 2 * cfg.n_samples),
                       "gpu_vs_oracle": {"mask_agreement": round(float(same.mean()), 5),
                               "max_abs_rgb_diff_where_masks_agree": round(float(np.abs(grgb - rr)[same].max()), 5) if same.any() else None}}
                ref.close(); return out
            except Exception as e:
                return {"value": None, "unit": "nominal ray-samples/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        cpu_render = time_oracle_render(max(2.0, args.cpu_seconds / 4))
        cpu = time_oracle({}, args.cpu_seconds, "R=4096 x S=32, base.json network")
        cpu_c1 = time_oracle(dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2), max(2.0, args.cpu_seconds / 3),
                "BASELINE configs[0]: R=1024 x S=32, hash L=4, MLP 2x32")

    # ---- extra, not the headline: K object NeRFs trained concurrently on this GPU (a host thread per object, as the managers do,
    #      CORE/src/nerf_manager.cu:89,259; the library threads their work through two shared streams); the kernels of different objects
    # overlap, so the aggregate rate says how much of the chip one object's launch chain leaves idle.  Every object is at the timed window's training stage
    # (steps W..W+K from init).
    multi = None
    if rank == 0 and world == 1 and args.objects_per_gpu > 1:
        try:
            K = args.objects_per_gpu
            # (a short window of four threads is mostly thread start-up; 320 steps = the window README / DESIGN quote)
            msteps = max(320, 5 * args.steps)
            tms = []
            for rep in range(3):                         # three fresh sets of K objects, the median is reported
                objs = [new_object(dict(sample_seed=3000 + 10 * rep + k)) for k in range(K)]
                for o in objs:
                    o.train(args.warmup)
                sync(); tm0 = time.perf_counter()
                th = [threading.Thread(target=o.train, args=(msteps,)) for o in objs]
                [t.start() for t in th]; [t.join() for t in th]
                sync(); tms.append(time.perf_counter() - tm0)
                for o in objs:
                    o.close()
            tm = median(tms)
            multi = {"objects": K, "value": round(K * msteps * B / tm, 1), "unit": "ray-samples/s (all objects)",
                    "ms_per_step_per_object": round(1e3 * tm / msteps / K, 4),
                     "values_of_the_repeats": [round(K * msteps * B / t, 1) for t in tms],
                     "note": "K independent objects, one host thread each, their training work on the device's two training lanes (HISTORY 7.2), same GPU, "
                             "each over steps %d..%d from init; median of 3 fresh sets" % (args.warmup, args.warmup + msteps)}
        except Exception as e:
            multi = {"objects": args.objects_per_gpu, "value": None, "note": "failed: %s" % e}

    # ---- extra, not the headline: the stress configuration's table (BASELINE configs[4]: hash T = 2^22, 105 M parameters per object) on this GPU: the one
    #      case whose kernels are HBM-bound (profiles/r05_window_T22.md).  One object, steps 20..60 from init and 800..840.
    stress = None
    if rank == 0 and world == 1 and not args.log2_hashmap_size and not args.no_stress:
        try:
            so = new_object(dict(log2_hashmap_size=22)); so.train(20); sync()
            ts0 = time.perf_counter(); so.train(40); sync(); ts_init = (time.perf_counter() - ts0) / 40
            so.train(740); sync()
            ts0 = time.perf_counter(); so.train(40); sync(); ts_late = (time.perf_counter() - ts0) / 40
            sp = {}
            try:
                sp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("stress_T22", {})
            except Exception:
                sp = {}
            def hbm_frac(key, ms):
                b = sp.get(key); return round(b / (ms * 1e-3) / 8e12, 4) if b else None
            stress = {"log2_hashmap_size": 22, "n_params": int(so.info().n_params),
                      "from_init": {"steps": "20..60", "ms_per_step": round(1e3 * ts_init, 4), "value": round(B / ts_init, 1),
                                    "frac_of_hbm_from_counters": hbm_frac("step_bytes_beyond_l2_steps_20_40", 1e3 * ts_init)},
                      "late": {"steps": "800..840", "ms_per_step": round(1e3 * ts_late, 4), "value": round(B / ts_late, 1),
                               "frac_of_hbm_from_counters": hbm_frac("step_bytes_beyond_l2_steps_800_820", 1e3 * ts_late)},
                      "unit": "ray-samples/s", "traffic_source": sp.get("source"),
                      "traffic_stale": bool(sp) and not (sources_now and sp.get("kernel_sources_sha16") == sources_now),
                      "note": "frac_of_hbm = committed counter traffic of the step's kernels ((2 FETCH_SIZE + WRITE_SIZE) KB summed over the kernels of a "
                              "step, profiles/r05_window_T22.md) / measured step time / 8 TB/s"}
            so.close()
        except Exception as e:
            stress = {"value": None, "note": "failed: %s" % e}

    if rank == 0:
        out = {"metric": "ray-samples/sec (train: hash-encode->MLP->composite fwd+bwd+optimizer) per object-NeRF", "value": round(value, 1),
               "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16 params/activations, fp32 accumulate + fp32 master",
               "data": "synthetic",
               "config": {"workload": "OfflineNeRF-style training, 1 synthetic 'room'-like object per GPU, base.json defaults (hash L=16 F=2 T=2^16, "
                                      "MLP 64x1), "
                                      "R=4096 rays x S=32 samples/step, %d views 640x480 resident in HBM%s" % (args.views,
                                              (", T overridden to 2^%d" % args.log2_hashmap_size) if args.log2_hashmap_size else ""),
                          "objects": world, "rays_per_step": cfg.rays_per_batch, "samples_per_ray": cfg.n_samples, "backend": obj_backend(pkg, obj),
                          "parallelism": "object-per-GPU (no training collective; %s gather-to-root of the final render%s)" % (
                              {"nccl": "RCCL", None: "RCCL"}.get(coll_backend, coll_backend), ", device-resident crops" if coll_backend == "nccl" else ""),
                          "launcher": "self-spawned ranks" if os.environ.get("MON_BENCH_SPAWNED")
                                       else ("torch.distributed.run" if world > 1 else "single process"),
                          # what the collective itself says: ranks in the process group, its backend, devices visible to this rank
                          "ranks_in_collective": (dist.get_world_size() if dist is not None else 1),
                          "collective_backend": (dist.get_backend() if dist is not None else None), "visible_devices": ndev,
                          "rank_devices": rank_devices},
               "timed_region": "median of %d independent repeats; each repeat: fresh object, %d warm-up steps, %d timed steps from init, barrier + device sync "
                               "on both sides, max over ranks" % (len(reps), args.warmup, args.steps),
               "ms_per_step_repeats": [round(1e3 * r / args.steps, 4) for r in reps], "per_rank_ray_samples_per_s": per_rank,
               "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_c1": cpu_c1, "gpu_c1": gpu_c1, "cpu_baseline_render": cpu_render,
               "offline_job": offline_job,
               "late_training": late, "late_training_with_occupancy_skipping": occ, "multi_object": multi, "stress_T22": stress,
               "pcie_inclusive": {"dataset_upload_ms": round(1e3 * upload_s, 2), "dataset_bytes": int(sc.n_views * sc.H * sc.W * 4),
                                  "value_for_a_5000_step_job": round(world * 5000 * B / (upload_s + 5000 * dt / args.steps), 1), "unit": "ray-samples/s",
                                  "value_for_a_5000_step_job_is": "computed: upload + 5000 x the timed window's step time",
                                  "measured_value_for_a_5000_step_job": round(5000 * B / (upload_s + offline_job["wall_s"]), 1)
                                          if offline_job and offline_job.get("wall_s") else None,
                                  "measured_is": "dataset upload + the offline_job leg's wall clock (5000 steps from init, wall to wall)",
                                  "note": "host frames -> HBM once per sequence (pinned staging + packing kernel), then 5000 steps at the measured step time; "
                                          "never the headline value"},
               "render": render_info, "psnr_db": [round(p, 2) for p in psnrs],
               "render_gather": ("ok" if gather_note is None else "FAILED (%s): psnr_db is rank 0's own crop" % gather_note) if dist is not None else None,
               "train_steps_before_render": (late["after_steps"] + 3 * args.steps) if late else args.warmup + args.steps,
               "final_loss": round(obj.info().last_loss, 5)}
        print(json.dumps(out), flush=True)
    obj.close(); ds.close()
    if dist is not None:
        if gather_note is not None:
            sys.stdout.flush(); os._exit(0)             # (a gather that hangs would also hang the teardown; the line is out)
        dist.destroy_process_group()


def obj_backend(pkg, obj):
    return int(obj.info().backend)


if __name__ == "__main__":
    main()
