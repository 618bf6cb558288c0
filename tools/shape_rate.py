"""Developer tool: rate of ds_iterate for a list of shapes.  env SHAPES = "H,W,D,dtype;..."."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

shapes = os.environ.get("SHAPES", "1080,1920,40,f32;1080,1920,64,f32;1080,1920,100,f32;1080,1920,128,f32;"
                        "1080,1920,200,f32;1080,1920,256,f32")
for spec in shapes.split(";"):
    H, W, D, dtype = spec.split(",")
    H, W, D = int(H), int(W), int(D)
    left, right, _ = synth.synthetic_pair(H, W, D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    g.iterate(3)
    g.synchronize()
    n = int(os.environ.get("ITERS", "40"))
    t0 = time.perf_counter()
    g.iterate(n)
    g.synchronize()
    ms = (time.perf_counter() - t0) / (n + 0.5) * 1e3   # + the half-sweep that ends ds_iterate
    pl = H * (W * D - D * (D - 1) // 2)
    b = 4 * (4 if dtype == "f32" else 8) + 1
    print("%dx%dx%d %s: %.3f ms per iteration, %.0f G pixel*label*iter/s, %.2f of 8 TB/s at %d B"
          % (W, H, D, dtype, ms, pl / ms / 1e6, pl * b / (ms * 1e-3) / 8e12, b), flush=True)
    g.close()
