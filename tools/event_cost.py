"""Developer tool: what the per-launch HIP events of ds_profile_* cost.  ds_iterate(N) at the
benchmark's size with and without them.  env: H, W, D, N."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D, N = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256), ("N", 40)))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
g.iterate(6)
for rep in range(2):
    t0 = time.perf_counter()
    g.iterate(N)
    plain = time.perf_counter() - t0
    g.profile_begin(20 * N)
    t0 = time.perf_counter()
    g.iterate(N)
    timed = time.perf_counter() - t0
    g.profile_end()
    n2, ms2 = g.profile_kind(2)
    print("%d iterations per call: %.4f ms per iteration plain, %.4f with an event pair around every launch (two-iteration kernel %.3f ms each)"
          % (N, plain / N * 1e3, timed / N * 1e3, ms2 / max(n2, 1)))
g.close()
