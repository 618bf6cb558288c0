"""Developer tool: time sweep-kernel variants (DS_SWEEP_VARIANT / DS_SWEEP_ROUNDS) back to
back in one process on the benchmark workload.  Prints ms per half-sweep launch."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,1,2,3")
ap.add_argument("--rounds", default="16")
ap.add_argument("--pads", default="0")
ap.add_argument("--layouts", default="0")
ap.add_argument("--H", type=int, default=2160)
ap.add_argument("--W", type=int, default=3840)
ap.add_argument("--D", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--repeat", type=int, default=2)
a = ap.parse_args()
left, right, _ = synth.synthetic_pair(a.H, a.W, a.D)
ref = None
for rep in range(a.repeat):
    for r, pad in [(r, pad) for r in a.rounds.split(",") for pad in a.layouts.split(",")]:
        for v in a.variants.split(","):
            os.environ["DS_SWEEP_VARIANT"] = v
            os.environ["DS_SWEEP_ROUNDS"] = r
            g = diffstereo.DiffusionGraph(left, right, a.D, 1.2, dtype=a.dtype)
            g.iterate(3)
            g.profile_begin(2 * a.iters)
            g.iterate(a.iters)
            p = g.profile_end()
            cs = g.checksum()
            if ref is None:
                ref = cs
            print("rep %d rounds %s layout %s variant %s: %.4f ms/launch (min %.4f max %.4f) checksum %s" % (
                rep, r, pad, v, p["total_ms"] / p["launches"], p["min_ms"], p["max_ms"],
                "same" if cs == ref else "DIFFERENT"), flush=True)
            g.close()
