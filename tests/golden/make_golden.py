"""Generates the golden fixtures under tests/golden/ (run once, in the build container).

Inputs: the reference's bundled toy pair images/left.pgm and images/right.pgm (5x5,
maxval 2), copied as data.  Outputs come from the CPU oracle (oracle/diffusion_oracle.c),
NOT from the reference binary: the reference is a Rust crate and no Rust toolchain
exists in the build image, and its own tests pin no whole-run result.  Parameters are
the reference main's: max_disparity 40, smoothing_term 1.2, epsilon 1/(10*W*H),
batch 200 (diffusion_stereo/src/main.rs:44-61).

  bundled_seq_100.npz : 100 x diffusion_act in the reference's raster order (f64)
  bundled_cb_100.npz  : 100 iterations in red-black order (f64)
  bundled_main.npz    : main's pipeline (diffusion_while_not_consistent + crossing out +
                        simple_best_labeling) under both schedules
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from oracle.pyoracle import OracleGraph  # noqa: E402

REF_IMAGES = "/root/reference/images"
D, LAM = 40, 1.2


def main():
    for name in ("left.pgm", "right.pgm"):
        shutil.copyfile(os.path.join(REF_IMAGES, name), os.path.join(HERE, name))
    left, W, H = pyoracle.pgm_reader(os.path.join(HERE, "left.pgm"))
    right, _, _ = pyoracle.pgm_reader(os.path.join(HERE, "right.pgm"))

    for tag, step in (("seq", "diffusion_act"), ("cb", "diffusion_act_checkerboard")):
        g = OracleGraph(left, right, D, LAM)
        energies = [g.energy()]
        for _ in range(100):
            getattr(g, step)()
            energies.append(g.energy())
        np.savez_compressed(os.path.join(HERE, "bundled_%s_100.npz" % tag),
                            phi_f64=g.potentials.copy(), energy_f64=np.float64(g.energy()),
                            energies=np.array(energies), depth_map=g.depth_map(),
                            belief_labels=g.belief_labels())
        print(tag, "energy after 100:", g.energy(), "belief labels:\n", g.belief_labels())

    eps = 1.0 / (10 * W * H)
    out = {}
    for tag, schedule in (("seq", 0), ("cb", 1)):
        g = OracleGraph(left, right, D, LAM)
        g.co_alloc(True)
        iters = g.diffusion_while_not_consistent(eps, 200, max_batches=50, schedule=schedule)
        lab = g.simple_best_labeling() if iters > 0 else np.full((H, W), -1, np.int32)
        out["iters_" + tag] = iters
        out["labels_" + tag] = lab
        out["energy_" + tag] = g.energy()
        out["consistent_" + tag] = pyoracle.check_disparity_map(lab) if iters > 0 else False
        print(tag, "main pipeline: iterations", iters, "energy", g.energy(), "labels:\n", lab)
    np.savez_compressed(os.path.join(HERE, "bundled_main.npz"), epsilon=eps, **out)


if __name__ == "__main__":
    main()
