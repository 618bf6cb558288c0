"""Time ds_crossing_out on a synthetic pair after some diffusion iterations.
env: H, W, D, DTYPE, ITERS (comma list, cumulative), EPS."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256)))
dtype = os.environ.get("DTYPE", "f32")
eps = float(os.environ.get("EPS", 1.0 / (10 * W * H)))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
done = 0
for target in (int(x) for x in os.environ.get("ITERS", "10,50,200").split(",")):
    g.iterate(target - done)
    done = target
    g.synchronize()
    t0 = time.perf_counter()
    r = g.crossing_out(eps)
    dt = time.perf_counter() - t0
    print("iters %d eps %.3g: passes %d not_empty %s labelled %.4f of pixels, %.1f ms (%.2f ms/pass) energy %.6g"
          % (done, eps, r["passes"], r["not_empty"], float((r["labels"] != 0xFFFFFFFF).mean()), dt * 1e3, dt * 1e3 / r["passes"],
             g.energy()), flush=True)
