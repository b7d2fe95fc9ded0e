"""Generates tests/golden/c2_trained.npz: BASELINE configs[1] (base.json defaults at the full batch, R = 4096 x S = 32) trained for 200 steps by the CPU
oracle with its SERIAL grid scatter (the reference order of accumulation: one fp32 sum per entry in sample order, no atomics), three sampling seeds,
on the 12-view test scene -- the rendered training crops, their PSNR against the synthetic ground truth and the final losses.  The serial scatter costs
~1.3 s per full-size step on 8 cores (13 minutes for the three seeds), which is why this runs in the build container and the GPU box only compares
(tests/test_gpu_parity.py::test_training_parity_c2_three_seeds_against_the_serial_oracle_fixture).
    python tests/golden/make_c2_trained.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge  # noqa: E402

SCENE = dict(n_views=12, H=120, W=160, f=130.0, seed=0)
SEEDS = (21, 22, 23); STEPS = 200; EVERY = 3


def main():
    orc = ge.load_oracle(); ss = ge.load_tools()
    orc.lib().orc_set_parallel_scatter(0)
    sc = ss.make_scene(**SCENE)
    out = dict(seeds=np.array(SEEDS, np.int32), steps=np.int32(STEPS), every=np.int32(EVERY))
    for seed in SEEDS:
        m = ge.make_oracle(orc, sc, dict(sample_seed=seed))
        loss = m.train(STEPS)
        psnrs = []
        for i, box in enumerate(sc.objects[0]["boxes"][::EVERY]):
            v, x, y, h, w = (int(q) for q in box)
            rgb, depth, mask = m.render(box, ss.colmajor(sc.Twc[v]))
            gm = sc.instance[v, y:y + h, x:x + w] > 0; gt = np.where(gm[..., None], sc.rgb[v, y:y + h, x:x + w] / 255.0, 1.0)
            psnrs.append(-10 * np.log10(np.mean((rgb - gt) ** 2)))
            out["rgb_s%d_c%d" % (seed, i)] = rgb.astype(np.float32); out["mask_s%d_c%d" % (seed, i)] = mask.astype(np.uint8)
        out["loss_s%d" % seed] = np.float32(loss); out["psnr_s%d" % seed] = np.array(psnrs, np.float32)
        print("seed %d: loss %.5f, PSNR vs ground truth %s" % (seed, loss, np.round(psnrs, 2)), flush=True)
        m.close()
    np.savez_compressed(os.path.join(HERE, "c2_trained.npz"), **out)


if __name__ == "__main__":
    main()
