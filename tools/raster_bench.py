"""Time ds_iterate_raster (anti-diagonal launches) against ds_iterate.  env: H, W, D, DTYPE, N (iterations per call)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256)))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=os.environ.get("DTYPE", "f32"))
N = int(os.environ.get("N", 3))
for name, fn in (("red-black", g.iterate), ("raster", g.iterate_raster)):
    fn(1)
    for n in sorted({1, N}):
        t0 = time.perf_counter()
        fn(n)
        print("%s, %d iteration(s) per call: %.2f ms per iteration" % (name, n, (time.perf_counter() - t0) / n * 1e3),
              flush=True)
