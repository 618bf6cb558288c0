"""Random graphs through ds_crossing_out + ds_find_best_labeling (a script, not collected by pytest:
`python3 tests/fuzz_find_best_labeling.py [seconds] [seed]` on a GPU box).

Small graphs: the reference restatement's masks, verdict and disparity map at an epsilon between
one and three times the smallest consistent one (where a choice leaves the graph empty the
reference panics and the library must report not_empty = 0).  Larger graphs: the same call twice and once
without the skipping of idle blocks (DS_CO_DIRTY=off) -- the same masks and the same map."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from oracle import pyoracle  # noqa: E402
from stereo_vision_amd import diffstereo  # noqa: E402
from tests.util import random_pair, random_phi  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, case, bad, refused, panics = time.time(), 0, [], 0, 0
    while time.time() - t0 < budget:
        case += 1
        rng = np.random.default_rng([seed, case])   # a case can be drawn again from its number alone
        small = rng.random() < 0.5
        if small:
            H, W, D = int(rng.integers(1, 9)), int(rng.integers(1, 13)), int(rng.choice([rng.integers(1, 24), 40, 100]))
        else:
            H, W, D = int(rng.integers(1, 40)), int(rng.integers(1, 150)), int(rng.integers(1, 40))
        dtype = "f64" if rng.random() < 0.6 else "f32"
        iters = int(rng.choice([0, 1, 3, 6, 12, 30]))
        exact = bool(rng.random() < 0.4)
        left, right = random_pair(H, W, seed=case)
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        if iters == 0:
            g.set_phi(random_phi(H, W, D, dtype, seed=case + 1))
        else:
            g.iterate(iters)
        eps = g.epsilon_search(exact=True) * float(rng.uniform(1.0001, 3.0)) + float(rng.choice([0.0, 0.0, 1e-3, 1.0]))
        try:
            first = g.crossing_out(eps, with_mask=True, exact=exact)
        except diffstereo.DiffStereoError:
            refused += 1   # an epsilon below what f32 potentials resolve
            g.close()
            continue
        got = g.find_best_labeling()
        what = []
        if small:
            o = pyoracle.OracleGraph(left.astype(np.uint32), right.astype(np.uint32), D, 1.2)
            o.set_potentials(np.asarray(g.get_phi(), dtype=np.float64))
            o.co_alloc(True)
            o.initialize_with_epsilon(eps)
            o.crossing_out()
            if not np.array_equal(first["alive"], o.vertices) or first["not_empty"] != o.is_not_empty():
                what.append("masks against the restatement")
            if o.is_not_empty():
                try:
                    want = o.find_best_labeling()
                except AssertionError:   # the reference panics: a choice left the graph empty
                    want = None
                    panics += 1
                if (want is None) == bool(got["not_empty"]) or \
                        (want is not None and not np.array_equal(got["labels"].astype(np.int32), want)):
                    what.append("map against the restatement")
        again_mask = g.crossing_out(eps, with_mask=True, exact=exact)
        again = g.find_best_labeling()
        os.environ["DS_CO_DIRTY"] = "off"
        plain_mask = g.crossing_out(eps, with_mask=True, exact=exact)
        plain = g.find_best_labeling()
        del os.environ["DS_CO_DIRTY"]
        for name, m, r in (("second run", again_mask, again), ("without skipping", plain_mask, plain)):
            if not np.array_equal(m["alive"], first["alive"]) or m["not_empty"] != first["not_empty"]:
                what.append("masks, " + name)
            if not np.array_equal(r["labels"], got["labels"]) or r["not_empty"] != got["not_empty"]:
                what.append("map, " + name)
        g.close()
        if what:
            bad.append((case, H, W, D, dtype, iters, eps, exact, what))
        if case % 10 == 0 or what:
            print("case %d: %dx%dx%d %s, %d iterations, epsilon %.6g%s%s, %d choices -> %s (%.0f s)"
                  % (case, H, W, D, dtype, iters, eps, ", the reference's comparisons" if exact else "",
                     ", against the restatement" if small else "", got["choices"], "; ".join(what) or "ok",
                     time.time() - t0), flush=True)
    print("%d cases (%d refused: epsilon below f32 resolution; %d where the reference panics and the library reports an "
          "empty graph), %d mismatches" % (case, refused, panics, len(bad)), flush=True)
    for b in bad:
        print("MISMATCH", b, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
