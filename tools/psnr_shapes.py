"""PSNR of a rendered training crop after 1500 steps for the network shapes beyond base.json (1 x 128 on the fused kernels; 16 neurons, 3 x 64, 2 x 128 on the
layer-at-a-time kernels with the LDS grid scatter) next to base.json's: a sanity check that every accepted shape trains (MI355X box)."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
box = sc.objects[0]["boxes"][0]; v, x, y, h, w = (int(q) for q in box)
gm = sc.instance[v, y:y + h, x:x + w] > 0; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
ds = None
for kw in (dict(), dict(n_neurons=16), dict(n_neurons=128), dict(n_neurons=64, n_hidden_layers=3), dict(n_neurons=128, n_hidden_layers=2)):
    ds, obj = ge.make_problem(pkg, sc, dict(sample_seed=2024, **kw), dataset=ds)
    loss = obj.train(1500); rgb, d, m = obj.render(box, ss.colmajor(sc.Twc[v]))
    print(kw, "backend", int(obj.info().backend), "loss %.5f" % loss, "PSNR %.2f dB" % (-10 * np.log10(((rgb - gt) ** 2).mean())), flush=True)
    obj.close()
