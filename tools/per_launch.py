"""Developer tool: per-launch durations of the sweep kernel (one HIP-event pair each)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = int(os.environ.get("H", "2160")), int(os.environ.get("W", "3840")), int(os.environ.get("D", "256"))
N = int(os.environ.get("ITERS", "8"))
left, right, _ = synth.synthetic_pair(H, W, D)
for rep in range(int(os.environ.get("REPS", "4"))):
    ROWS = int(os.environ.get("ROWS", str(H)))
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=os.environ.get("DTYPE", "f32"), row_begin=0, row_end=ROWS)
    g.sweep(0); g.sweep(1); g.synchronize()
    times = []
    for it in range(N):
        for colour in (0, 1):
            g.profile_begin(1)
            g.sweep(colour)
            times.append(g.profile_end()["total_ms"])
    print("alloc %d:" % rep, g.placement_info(), " ".join("%.2f" % t for t in times), flush=True)
    g.close()
