import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from __graft_entry__ import load_package
load_package()
from stereo_vision_amd import diffstereo, synth
for dtype in ("f64", "f32"):
    H, W, D = 2160, 3840, 256
    left, right, _ = synth.synthetic_pair(H, W, D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    g.iterate(10)
    r = g.crossing_out(600.0)
    print(dtype, "crossing-out", r["passes"], "passes")
    for k in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            g.co_pass()
        print(dtype, "idle pass: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
    os.environ["DS_CO_DIRTY"] = "off"
    t0 = time.perf_counter()
    for _ in range(5):
        g.co_pass()
    print(dtype, "full pass (flags off): %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
    del os.environ["DS_CO_DIRTY"]
    g.close()
