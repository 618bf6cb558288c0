"""Developer tool: several handles alive at once (different physical memory each); per-colour sweep time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = 2160, 3840, 256
left, right, _ = synth.synthetic_pair(H, W, D)
K = int(os.environ.get("K", "6"))
gs = []
for k in range(K):
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    gs.append(g)
    for h in gs[-1:]:
        h.iterate(1)
        t = []
        for colour in (0, 1):
            h.profile_begin(3)
            for _ in range(3):
                h.sweep(colour)
            t.append(h.profile_end()["total_ms"] / 3)
        print("handle %d (with %d alive): red %.2f black %.2f" % (k, len(gs), t[0], t[1]), flush=True)
print("re-probe all:")
for k, h in enumerate(gs):
    t = []
    for colour in (0, 1):
        h.profile_begin(3)
        for _ in range(3):
            h.sweep(colour)
        t.append(h.profile_end()["total_ms"] / 3)
    print("handle %d: red %.2f black %.2f" % (k, t[0], t[1]), flush=True)
