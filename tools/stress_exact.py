"""Developer tool: 60 random small graphs (1-8 x 1-13 pixels, 1-70 labels, random or iterated potentials,
both precisions): ds_epsilon_search with the reference's comparisons against the oracle's search, and
ds_crossing_out_exact at a dozen of each graph's gaps against the oracle's masks.  Prints the mismatches."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
load_package()
from stereo_vision_amd import diffstereo
from oracle import pyoracle
from tests.util import random_pair, random_phi

rng = np.random.default_rng(123)
bad = 0
for case in range(60):
    H, W = int(rng.integers(1, 9)), int(rng.integers(1, 14))
    D = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 70]))
    iters = int(rng.choice([0, 0, 1, 3, 10, 40]))
    dtype = "f64" if rng.random() < 0.75 else "f32"
    left, right = random_pair(H, W, seed=1000 + case)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    if iters == 0:
        g.set_phi(random_phi(H, W, D, dtype, seed=case))
    else:
        g.iterate(iters)
    phi = np.asarray(g.get_phi(), dtype=np.float64)
    o = pyoracle.OracleGraph(left.astype(np.uint32), right.astype(np.uint32), D, 1.2)
    o.set_potentials(phi)
    o.co_alloc(True)
    want = o.epsilon_search(o.create_array_of_epsilons(0.0))
    got = g.epsilon_search(exact=True)
    ok = got == want
    gaps = g.epsilon_gaps()
    picks = gaps[rng.integers(0, len(gaps), size=min(12, len(gaps)))]
    for eps in picks:
        o.initialize_with_epsilon(float(eps)); o.crossing_out()
        r = g.crossing_out(float(eps), with_mask=True, exact=True)
        if not np.array_equal(r["alive"], o.vertices) or r["not_empty"] != o.is_not_empty():
            ok = False
    if not ok:
        bad += 1
        print("MISMATCH case", case, H, W, D, iters, dtype, got, want, flush=True)
    g.close()
print("stress done, mismatches:", bad)
