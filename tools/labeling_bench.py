"""Developer tool: time of ds_find_best_labeling after ds_crossing_out on a converged run.
env H, W, D, DTYPE, ITERS, EPS."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 400), ("W", 400), ("D", 40)))
dtype = os.environ.get("DTYPE", "f64")
iters, eps = int(os.environ.get("ITERS", "400")), float(os.environ.get("EPS", "0.5"))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
g.iterate(iters)
t0 = time.perf_counter()
co = g.crossing_out(eps, with_mask=True)
t1 = time.perf_counter()
several = int((co["alive"].sum(axis=2) > 1).sum())
print("%dx%dx%d %s, %d iterations, eps %g: crossing-out %.3f s (%d passes), not empty %s, %d of %d pixels with several labels"
      % (W, H, D, dtype, iters, eps, t1 - t0, co["passes"], co["not_empty"], several, H * W), flush=True)
if co["not_empty"]:
    t0 = time.perf_counter()
    out = g.find_best_labeling()
    t1 = time.perf_counter()
    same = int((out["labels"] == co["labels"]).sum())
    print("find_best_labeling: %.3f s, %d choices, not empty %s, %d pixels keep simple_best_labeling's label"
          % (t1 - t0, out["choices"], out["not_empty"], same), flush=True)
g.close()
