#!/usr/bin/env python
"""50 000 training steps from init in ten mon_object_train(5000) calls with a render after each (bench scene, one base.json object): wall time, loss, PSNR of a
training crop, learning rate (ExponentialDecay on the device: x0.33 at 20 000, 30 000, ...), skipped batches, finiteness of the parameters."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package(); ss = ge.load_tools()
    sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, seed=0)
    ds, obj = ge.make_problem(pkg, sc, {"sample_seed": 2024})
    ob = sc.objects[0]; bx = ob["boxes"][3]; v, x, y, h, w = (int(q) for q in bx)
    gm = sc.instance[v, y:y + h, x:x + w] == ob["cls"]; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
    t0 = time.perf_counter()
    for k in range(10):
        loss = obj.train(5000)
        rgb, depth, mask = obj.render(bx, ss.colmajor(sc.Twc[v]))
        info = obj.info()
        print("step %6d  wall %.2f s  loss %.5f  PSNR %.2f dB  lr %.3e  skipped %d" % (info.train_step, time.perf_counter() - t0, loss,
                -10 * np.log10(np.mean((rgb - gt) ** 2)), info.learning_rate, info.skipped_batches), flush=True)
    p = obj.get_params(0)
    print("parameters finite:", bool(np.isfinite(p).all()), " %.1f us per step over the whole run" % (1e6 * (time.perf_counter() - t0) / 50000))
    obj.close(); ds.close()


if __name__ == "__main__":
    main()
