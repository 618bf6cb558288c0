"""Time ds_iterate at the benchmark's size with two iterations per launch: wall time per
iteration and the launches by kind.  env: H, W, D, N (iterations per call), DS_FUSED2_ROWS."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D, N = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256), ("N", 41)))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
print("placement", g.placement_info(), "fused2", g.fused2_info(), flush=True)
g.iterate(5)
g.profile_begin(20 * N)
t0 = time.perf_counter()
g.iterate(N)
wall = time.perf_counter() - t0
g.profile_end()
print("%d iterations per call: %.3f ms per iteration" % (N, wall / N * 1e3))
for kind, name in ((0, "half-sweeps"), (1, "one-iteration launches"), (2, "two-iteration launches")):
    n, ms = g.profile_kind(kind)
    print("  %-24s %4d launches, %8.3f ms in all, %7.3f ms each" % (name, n, ms, ms / max(n, 1)))
