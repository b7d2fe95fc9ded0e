#!/usr/bin/env python
"""Commits the evidence of tools/gpu_profile_window.sh runs under profiles/: the per-window kernel durations and PMC tables
(profiles/<round>_window_<regime>.md) and the regime-keyed numbers bench.py attaches to its roofline (profiles/pmc_traffic.json).

   python tools/update_profiles_window.py r02 dense=gpurun_out/<tag> sparse=gpurun_out/<tag>

regime "dense"  = steps 5..25 from init (what `bench.py --gpus 1 --steps 20 --warmup 5` times: every sample carries a gradient),
regime "sparse" = steps 805..825 (late training: a few per cent of the samples carry a gradient)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_encode_tiles", "k_fused_train", "k_grid_scatter", "k_optimizer")
WHAT = {"sparse_occ": "steps 805..825 from init with occupancy-grid skipping switched on (mon_config::occupancy_skip; opt-in, DESIGN.md 3.4)",
        "dense": "steps 5..25 from init -- the window `bench.py --gpus 1 --steps 20 --warmup 5` times; every one of the 131 072 samples carries a gradient",
        "sparse": "steps 805..825 from init -- late training; a few per cent of the samples still carry a gradient (DESIGN.md 3.1 (HISTORY 3.2b))"}


def parse(md):
    val = {}
    for line in md.splitlines():
        m = re.match(r"\| (\S+).*?\| (\w+) \| (\d+) \| ([\d.]+) \|", line)
        if m:
            k = next((k for k in KERNELS if k in m.group(1)), None)
            if k:
                val[(k, m.group(2))] = float(m.group(4))
    return val


def durations(md):
    d = {}
    for line in md.splitlines():
        m = re.match(r"\| (\S+).*?\| \d+ \(\d+\) \| ([\d.]+) \|", line)
        if m:
            k = next((k for k in KERNELS if k in m.group(1)), None)
            if k:
                d[k] = float(m.group(2))
    return d


def main():
    rnd = sys.argv[1]
    pj_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    old = json.load(open(pj_path)) if os.path.exists(pj_path) else {}
    out = dict(old)                      # (keys written by other tools -- fused_floor, gatherbench -- stay)
    out.update({"l2_line_request_rate_measured_per_s": old.get("l2_line_request_rate_measured_per_s", 270e9),
           "note": "rocprofv3 --kernel-trace --pmc, one counter set per pass (tools/gpu_profile_window.sh), means per dispatch over the 20 dispatches of the named window of "
                   "`python tools/profile_window.py` (base.json object, bench scene); hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KB per MI355X_MICROARCH.md (the FETCH_SIZE correction is "
                   "calibrated for wide streams only; 4-byte gathers are uncalibrated). l2_line_request_rate: distinct-line gather rate measured by tools/run_gatherbench.py "
                   "(266-272 G lines/s chip-wide with every lane on its own line = 128 L2 channels x ~2.1 GHz; profiles/r02_gatherbench.md). SQ_* cycle counters are in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES."})
    for k in ("dense", "sparse", "sparse_occ"):
        if k in old and isinstance(old[k], dict):
            out[k] = old[k]
    for arg in sys.argv[2:]:
        regime, src = arg.split("=", 1); tag = os.path.basename(src.rstrip("/"))
        kw = open(os.path.join(src, "kernel_window.md")).read(); pw = open(os.path.join(src, "pmc_window.md")).read()
        # the sources these numbers were measured on: bench.py recomputes the fingerprint and flags the numbers as stale when it differs (tools/fingerprint.py)
        sys.path.insert(0, os.path.join(ROOT, "tools")); from fingerprint import kernel_sources_sha16
        v = parse(pw); du = durations(kw); d = {"source": "profiles/%s_window_%s.md (gpurun %s)" % (rnd, regime, tag), "kernel_sources_sha16": kernel_sources_sha16()}
        for k in KERNELS:
            if (k, "FETCH_SIZE") not in v:
                continue
            f, w, h, mi = v[(k, "FETCH_SIZE")], v[(k, "WRITE_SIZE")], v[(k, "TCC_HIT_sum")], v[(k, "TCC_MISS_sum")]
            d[k + "_avg_us"] = du.get(k)
            d[k + "_FETCH_SIZE_KB"] = f; d[k + "_WRITE_SIZE_KB"] = w; d[k + "_hbm_bytes_per_launch"] = int((2 * f + w) * 1024)
            d[k + "_l2_hit_rate"] = round(h / (h + mi), 4)
            d[k + "_l2_requests_per_launch"] = int(v[(k, "TCC_REQ_sum")]); d[k + "_l2_read_requests_per_launch"] = int(v[(k, "TCP_TCC_READ_REQ_sum")])
            if (k, "SQ_BUSY_CYCLES") in v and v[(k, "SQ_BUSY_CYCLES")] > 0:
                busy = v[(k, "SQ_BUSY_CYCLES")] / 32.0            # summed over the 32 shader engines -> cycles the kernel kept the SQs busy
                d[k + "_sq_busy_cycles"] = int(busy)
                d[k + "_mfma_busy_frac"] = round(v.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0) / (busy * 1024.0), 5)          # of 1024 SIMD matrix pipes
                # SIMD-time with a VALU instruction in flight (quad-cycles -> cycles, 1024 SIMDs)
                d[k + "_valu_busy_frac"] = round(v.get((k, "SQ_ACTIVE_INST_VALU"), 0.0) * 4.0 / (busy * 1024.0), 4)
                if v.get((k, "SQ_INSTS_VALU")):
                    d[k + "_cycles_per_valu_instruction"] = round(v.get((k, "SQ_ACTIVE_INST_VALU"), 0.0) * 4.0 / v[(k, "SQ_INSTS_VALU")], 2)
                d[k + "_lds_active_frac"] = round(v.get((k, "SQ_LDS_IDX_ACTIVE"), 0.0) / (busy * 256.0), 4)                  # of 256 LDS arrays
                d[k + "_lds_bank_conflict_share"] = round(v.get((k, "SQ_LDS_BANK_CONFLICT"), 0.0) / max(1.0, v.get((k, "SQ_LDS_IDX_ACTIVE"), 0.0)), 4)
                wc = v.get((k, "SQ_WAVE_CYCLES"), 0.0)
                if wc:
                    d[k + "_wave_time_split"] = {"active": round(v.get((k, "SQ_ACTIVE_INST_ANY"), 0) / wc, 3),
                            "issue_stall": round(v.get((k, "SQ_WAIT_INST_ANY"), 0) / wc, 3),
                                                 "waitcnt_or_barrier": round(v.get((k, "SQ_WAIT_ANY"), 0) / wc, 3)}
        out[regime] = d
        with open(os.path.join(ROOT, "profiles", "%s_window_%s.md" % (rnd, regime)), "w") as fh:
            fh.write("# %s, regime '%s': %s\n\nCommand profiled: `python tools/profile_window.py --warmup 5 --steps 20%s` under `rocprofv3 --kernel-trace --stats` (durations) and, in separate runs, "
                     "`rocprofv3 --kernel-trace --pmc <one counter set>` (tools/gpu_profile_window.sh, gpurun %s, MI355X). Per kernel, dispatches are put in order and the window's 20 are averaged "
                     "(tools/rocpd_window.py). The trace's VGPR column counts register pairs (x2 = the compiler's .vgpr_count).\n\n## Durations\n\n%s\n## Counters\n\nFETCH_SIZE / WRITE_SIZE in KB; TCC_* / TCP_* in requests; "
                     "SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_BUSY_CYCLES summed over the 32 shader engines; SQ_LDS_* in LDS-array cycles summed over CUs; GRBM_GUI_ACTIVE summed over the 8 XCDs.\n\n%s\n"
                     "## Derived (also in profiles/pmc_traffic.json)\n\n```\n%s\n```\n" % (rnd, regime, WHAT.get(regime, regime),
                             (" --extra 800" if regime.startswith("sparse") else "") + (" --occupancy" if regime.endswith("_occ") else ""), tag,
                                                                                           kw.split("\n\n", 1)[-1], pw.split("\n\n", 1)[-1], json.dumps(d,
                                                                                                   indent=1)))
    json.dump(out, open(pj_path, "w"), indent=1)
    print("profiles updated:", [a.split("=")[0] for a in sys.argv[2:]])


if __name__ == "__main__":
    main()
