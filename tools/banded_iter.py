"""Experiment: one iteration as row-banded launches, the second colour trailing the first by
one row, so that what a band writes is read again while it may still sit in the last-level
cache.  env: H, W, D, DTYPE, BANDS (comma list of rows per band; 0 = two whole-image launches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256)))
dtype = os.environ.get("DTYPE", "f32")
left, right, _ = synth.synthetic_pair(H, W, D)
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
ref = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
print("placement", g.placement_info(), flush=True)


def banded(rows):
    if rows == 0:
        g.sweep(0)
        g.sweep(1)
        return
    a = 0
    while a < H:
        b = min(a + rows, H)
        g.sweep(0, a, b)
        lo, hi = max(a - 1, 0), (b - 1 if b < H else H)
        if hi > lo:
            g.sweep(1, lo, hi)
        a = b


for rows in (int(x) for x in os.environ.get("BANDS", "0,4,8,16,32,64,128").split(",")):
    for _ in range(2):
        banded(rows)
    g.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        banded(rows)
    g.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("rows/band %4d: %.3f ms per iteration" % (rows, dt * 1e3), flush=True)
# the banded order must not change the result
g2 = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
ref.iterate(2)
g = g2
for _ in range(2):
    banded(8)
g.synchronize()
print("same bits as whole-image launches:", g.checksum() == ref.checksum(), flush=True)
