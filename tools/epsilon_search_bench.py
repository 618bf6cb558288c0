"""Developer tool: ds_epsilon_search (the reference's epsilon search without its array of gaps) on
images whose array cannot exist.  env SIZES = "H,W,D;...", ITERS = diffusion iterations before."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402



def heartbeat():
    t0 = time.perf_counter()
    while True:
        time.sleep(30)
        print("... %.0f s" % (time.perf_counter() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
sizes = os.environ.get("SIZES", "400,400,40;1080,1920,128")
iters = int(os.environ.get("ITERS", "200"))
for spec in sizes.split(";"):
    H, W, D = (int(x) for x in spec.split(","))
    left, right, _ = synth.synthetic_pair(H, W, D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    g.iterate(iters)
    g.synchronize()
    t0 = time.perf_counter()
    nxt, largest, count = g.epsilon_next(-1.0)
    t_pass = time.perf_counter() - t0
    t0 = time.perf_counter()
    nxt, largest, count = g.epsilon_next(largest)
    t_pass2 = time.perf_counter() - t0
    print("%dx%dx%d: one pass over the gaps %.3f s" % (W, H, D, t_pass2), flush=True)
    t0 = time.perf_counter()
    if os.environ.get("EXACT", "1") == "1":
        eps_x, stats_x = g.epsilon_search(exact=True, with_stats=True)
    else:  # EXACT=0: only the search with the O(D) comparisons
        eps_x, stats_x = float("nan"), {"crossing_outs": 0, "relaxation_passes": 0}
    t_exact = time.perf_counter() - t0
    print("%dx%dx%d: search with the reference's comparisons %.2f s" % (W, H, D, t_exact), flush=True)
    t0 = time.perf_counter()
    eps, stats = g.epsilon_search(exact=False, with_stats=True)
    t_search = time.perf_counter() - t0
    print("%dx%dx%d after %d iterations: %.4g gaps (an array of %.3g GB), largest %.6g; one pass over them %.3f s "
          "(first, with allocation: %.3f s); search: epsilon = %.17g in %.2f s (%d crossing-outs with %d relaxation "
          "passes, %d gap passes); with the reference's comparisons: epsilon = %.17g in %.2f s (%d crossing-outs, %d relaxation "
          "passes)" % (W, H, D, iters, count, count * 8 / 1e9, largest, t_pass2, t_pass, eps, t_search,
                       stats["crossing_outs"], stats["relaxation_passes"], stats["gap_passes"], eps_x, t_exact,
                       stats_x["crossing_outs"], stats_x["relaxation_passes"]), flush=True)
    g.close()
