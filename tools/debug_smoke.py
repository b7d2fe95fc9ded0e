import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
P = lambda *a: print(*a, flush=True)
variant = sys.argv[1] if len(sys.argv) > 1 else "a"
pkg = ge.load_package(); orc = ge.load_oracle(); ss = ge.load_tools()
P("devices", pkg.device_count(), "variant", variant, "omp threads", orc.lib().orc_max_threads())
if variant == "b":
    orc.lib().orc_set_threads(8)
sc = ss.make_scene(n_views=8, H=96, W=128, f=100.0, seed=3) if variant != "c" else ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)
kw = dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2)
ds, obj = ge.make_problem(pkg, sc, kw); P("problem made")
if variant == "d":
    obj.set_backend(0)
ref = ge.make_oracle(orc, sc, kw); P("oracle made h=%x" % ref.h)
loss = obj.train(1); P("hip train", loss)
r = ref.train(1); P("oracle train returned", r)
P("n_valid", obj.info().last_n_valid, ref.n_valid)
P("h=%x" % ref.h)
P("oracle loss", ref.loss)
a, b = obj.get_params(0), ref.buffer("master"); P("params frac", np.mean(np.abs(a - b) > 1e-4))
box = sc.objects[0]["boxes"][0]
rgb, depth, mask = obj.render(box, ss.colmajor(sc.Twc[box[0]])); P("hip render")
rrgb, rdepth, rmask = ref.render(box, ss.colmajor(sc.Twc[box[0]])); P("oracle render", np.mean(mask != rmask), np.abs(rgb - rrgb).max())
obj.close(); ds.close(); ref.close(); P("closed")
