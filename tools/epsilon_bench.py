"""Developer tool: time of ds_epsilon_gaps (gaps of every vertex and edge, device sort, exact
duplicates removed). env SIZES = "H,W,D;...". That the values equal the CPU restatement's is a test
(tests/test_gpu_epsilon.py), not this tool's business."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

sizes = os.environ.get("SIZES", "48,64,40;100,100,40;200,200,40;400,400,40")
for spec in sizes.split(";"):
    H, W, D = (int(x) for x in spec.split(","))
    left, right, _ = synth.synthetic_pair(H, W, D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    g.iterate(50)
    g.synchronize()
    g.epsilon_gaps()  # warm-up (module load, allocations)
    t0 = time.perf_counter()
    gaps = g.epsilon_gaps()
    t_dev = time.perf_counter() - t0
    n_all = H * W * D * (1 + 4 * D)
    line = "%dx%dx%d: %.3f s on the GPU for %.3g slots -> %d distinct gaps" % (W, H, D, t_dev, n_all, len(gaps))
    print(line, flush=True)
    g.close()
