"""PSNR statistics over sampling seeds: HIP backend 0 / backend 1 / CPU oracle, C1 configuration, 400 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle(); ss = ge.load_tools()
sc = ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)
C1 = dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2)
def psnr(a, b): return -10 * np.log10(np.mean((a - b) ** 2))
def score(render):
    out = []
    for box in sc.objects[0]["boxes"][::3]:
        v, x, y, h, w = (int(q) for q in box); rgb, depth, mask = render(box, ss.colmajor(sc.Twc[v]))
        gm = sc.instance[v, y:y + h, x:x + w] > 0; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
        out.append(psnr(rgb, gt))
    return float(np.mean(out))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
res = {"hip0": [], "hip1": [], "oracle": []}
for seed in range(6):
    kw = dict(C1, sample_seed=1000 + seed)
    for be in (0, 1):
        ds, obj = ge.make_problem(pkg, sc, kw); obj.set_backend(be); obj.train(steps)
        res["hip%d" % be].append(score(obj.render)); obj.close(); ds.close()
    if seed < 4:
        ref = ge.make_oracle(orc, sc, kw); ref.train(steps); res["oracle"].append(score(ref.render)); ref.close()
    print(seed, {k: round(v[-1], 2) for k, v in res.items() if v}, flush=True)
for k, v in res.items():
    print("%-7s mean %.2f  std %.2f  n=%d  values %s" % (k, np.mean(v), np.std(v), len(v), [round(x, 2) for x in v]))
