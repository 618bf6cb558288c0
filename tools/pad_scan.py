"""Developer tool: time the sweep for many paddings inside ONE allocation
(ds_set_padding), so that only the relative placement of the streams changes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DS_POOL_SLACK_MB", "3072")
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = 2160, 3840, 256
unit = int(os.environ.get("UNIT", str(128 * 1024)))
nplane = int(os.environ.get("NPLANE", "9"))
ncol = int(os.environ.get("NCOL", "9"))
rowpad = int(os.environ.get("ROWPAD", "0"))
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
print("unit %d bytes; rows = plane pad / unit, cols = colour pad / unit; cell = ms red/black" % unit)
for a in range(nplane):
    cells = []
    for b in range(ncol):
        g.set_padding(rowpad, a * unit, b * unit)
        g.iterate(1)
        t = []
        for colour in (0, 1):
            g.profile_begin(2)
            g.sweep(colour)
            g.sweep(colour)
            t.append(g.profile_end()["total_ms"] / 2)
        cells.append("%.2f/%.2f" % (t[0], t[1]))
    print("%2d: %s" % (a, "  ".join(cells)), flush=True)
g.close()
