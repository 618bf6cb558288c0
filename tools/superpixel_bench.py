"""Developer tool: the superpixel variant at the crate's own working point (400 x 400, 5 x 5 windows,
40 disparities): ms per iteration under both schedules, energy, one crossing-out."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import superpixel, synth  # noqa: E402

H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 400), ("W", 400), ("D", 40)))
left, right, _ = synth.synthetic_pair(H, W, D)
for dtype in ("f64", "f32"):
    g = superpixel.SuperpixelGraph(left, right, D, 1.2, 5, 5, perimeters=True, dtype=dtype)
    nodes = g.NI * g.NJ * 2
    for schedule, name, n in ((superpixel.FOUR_COLOURS, "four colours", 200), (superpixel.RASTER, "raster order", 20)):
        g.iterate(2, schedule)
        t0 = time.perf_counter()
        g.iterate(n, schedule)
        ms = (time.perf_counter() - t0) / n * 1e3
        print("%dx%dx%d %s, %d nodes, %s: %.3f ms per iteration (%.2f G node*slot*label*iter/s), energy %.6f"
              % (W, H, D, dtype, nodes, name, ms, nodes * 9 * D / ms / 1e6, g.energy()), flush=True)
    t0 = time.perf_counter()
    out = g.crossing_out(0.5)
    print("  crossing out, eps 0.5: %d passes, not_empty %s, %.1f ms" % (out["passes"], out["not_empty"], (time.perf_counter() - t0) * 1e3))
    g.close()
