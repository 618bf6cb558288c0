"""Developer tool: one long crossing-out (an unconverged 4K x 256 f64 image at an epsilon near its
smallest consistent one relaxes for about 2 000 passes), for a kernel trace of the late passes.
env: EPS, ITERS, H, W, D, EXACT."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256)))
eps = float(os.environ.get("EPS", "28.4779"))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
g.iterate(int(os.environ.get("ITERS", "100")))
g.synchronize()
t0 = time.perf_counter()
exact = os.environ.get("EXACT", "0") == "1"   # the reference's comparisons (ds_crossing_out_exact)
r = g.crossing_out(eps, exact=exact)
dt = time.perf_counter() - t0
print(("the reference's comparisons, " if exact else "") + "epsilon %g: %d passes, not empty %s, %.2f s (%.2f ms per pass)" % (eps, r["passes"], r["not_empty"], dt, dt * 1e3 / r["passes"]))
