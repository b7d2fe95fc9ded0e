#!/usr/bin/env python
"""CPU-only study on the checker (oracle/), BASELINE configs[0] (C1), that pins two numbers the GPU parity tests use:

  (1) the CHAOS FLOOR of the training trajectory: two oracle runs of the SAME schedule whose initial weights differ by one fp16
      unit in the last place (the precision of the working copy) on 1 % of the parameters.  Whatever their mutual PSNR after N steps is, no HIP-vs-oracle
      comparison of trained models can be asked to do better (the HIP path differs from the oracle by expf / summation-order ulps);
  (2) the distance between this repo's numeric contract (fp32 accumulation) and the MODEL of tiny-cuda-nn's
      own fp16 accumulation (oracle flag ORC_NUM_TCNN_HALF): parameters after one step, and PSNR after N steps.

   python tools/oracle_numerics_study.py [steps=300] [out.json]      (writes tests/golden/numerics_study.json by default)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

C1 = dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2)


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)); return 99.0 if mse == 0 else -10.0 * np.log10(mse)


def renders(ref, sc, ss):
    out = []
    for box in sc.objects[0]["boxes"][::4]:
        v, x, y, h, w = (int(q) for q in box); rgb, depth, mask = ref.render(box, ss.colmajor(sc.Twc[v]))
        gm = sc.instance[v, y:y + h, x:x + w] > 0; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
        out.append((rgb, gt))
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "numerics_study.json")
    orc = ge.load_oracle(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)          # tests/conftest.py small_scene
    rows = []
    for seed in (11, 12, 13, 14, 15, 16):
        kw = dict(C1, sample_seed=seed)
        a = ge.make_oracle(orc, sc, kw); b = ge.make_oracle(orc, sc, kw); t = ge.make_oracle(orc, sc, dict(kw, tcnn_half_accum=1))
        p0 = a.buffer("master")
        rs = np.random.RandomState(seed); pert = p0.copy(); sel = rs.rand(p0.size) < 0.01
        # the fp16 working copy moves by ONE fp16 ulp on 1 % of the parameters
        h = pert[sel].astype(np.float16); pert[sel] = np.nextafter(h, np.float16(np.inf)).astype(np.float32); b.set_params(pert)
        # one step: contract vs tcnn model
        a1 = ge.make_oracle(orc, sc, kw); t1 = ge.make_oracle(orc, sc, dict(kw, tcnn_half_accum=1)); a1.train(1); t1.train(1)
        pa, pt = a1.buffer("master"), t1.buffer("master"); nm = a1.n_mlp
        one = dict(frac_mlp_gt_1e4=float((np.abs(pa[:nm] - pt[:nm]) > 1e-4).mean()), frac_grid_gt_1e4=float((np.abs(pa[nm:] - pt[nm:]) > 1e-4).mean()),
                   max_abs=float(np.abs(pa - pt).max()))
        a1.close(); t1.close()
        a.train(steps); b.train(steps); t.train(steps)
        ra, rb, rt = renders(a, sc, ss), renders(b, sc, ss), renders(t, sc, ss)
        row = dict(seed=seed, steps=steps,
                   abs_contract=float(np.mean([psnr(r, g) for r, g in ra])), abs_perturbed=float(np.mean([psnr(r, g) for r, g in rb])),
                           abs_tcnn_half=float(np.mean([psnr(r, g) for r, g in rt])),
                   mutual_contract_vs_perturbed_min=float(min(psnr(x[0], y[0]) for x, y in zip(ra, rb))),
                           mutual_contract_vs_perturbed_mean=float(np.mean([psnr(x[0], y[0]) for x, y in zip(ra, rb)])),
                   mutual_contract_vs_tcnn_half_min=float(min(psnr(x[0], y[0]) for x, y in zip(ra, rt))),
                           mutual_contract_vs_tcnn_half_mean=float(np.mean([psnr(x[0], y[0]) for x, y in zip(ra, rt)])),
                   one_step_contract_vs_tcnn_half=one)
        rows.append(row); print(json.dumps(row), flush=True)
        a.close(); b.close(); t.close()
    f = lambda k: [r[k] for r in rows]
    summary = dict(
        config="BASELINE configs[0] (R=1024, S=32, hash L=4, MLP 2x32), tests/conftest.py small_scene, %d steps, crops = every 4th training box" % steps,
        chaos_floor=dict(what="oracle vs the same oracle run started one fp16 ulp away on 1 % of the parameters",
                mutual_psnr_min_db=min(f("mutual_contract_vs_perturbed_min")),
                         mutual_psnr_mean_db=float(np.mean(f("mutual_contract_vs_perturbed_mean"))),
                                 abs_psnr_diff_max_db=float(max(abs(r["abs_contract"] - r["abs_perturbed"]) for r in rows)),
                         abs_psnr_diff_mean3_max_db=float(max(abs(np.mean(f("abs_contract")[i:i + 3]) - np.mean(f("abs_perturbed")[i:i + 3])) for i in range(0,
                                 len(rows) - 2)))),
        tcnn_half_model=dict(what="contract numerics (fp32 accumulation) vs the model of tiny-cuda-nn's fp16 accumulation, same seeds",
                             mutual_psnr_min_db=min(f("mutual_contract_vs_tcnn_half_min")),
                                     mutual_psnr_mean_db=float(np.mean(f("mutual_contract_vs_tcnn_half_mean"))),
                             abs_psnr_contract_mean_db=float(np.mean(f("abs_contract"))), abs_psnr_tcnn_half_mean_db=float(np.mean(f("abs_tcnn_half"))),
                             abs_psnr_diff_max_db=float(max(abs(r["abs_contract"] - r["abs_tcnn_half"]) for r in rows)),
                             one_step_frac_mlp_gt_1e4_max=max(r["one_step_contract_vs_tcnn_half"]["frac_mlp_gt_1e4"] for r in rows),
                             one_step_frac_grid_gt_1e4_max=max(r["one_step_contract_vs_tcnn_half"]["frac_grid_gt_1e4"] for r in rows)),
        abs_psnr_std_over_seeds_db=float(np.std(f("abs_contract"))))
    json.dump(dict(generated_by="tools/oracle_numerics_study.py %d" % steps, summary=summary, rows=rows), open(out_path, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
