"""Random graphs through the crossing-out entry points (a script, not collected by pytest:
`python3 tests/fuzz_crossing_out.py [seconds] [seed]` on a GPU box).

Every case draws a shape, a precision and a number of iterations (or random potentials), finds the
smallest epsilon that leaves a graph and relaxes at it, just below it, above it and at a random
gap -- with the O(D) comparisons and with the reference's, each with and without the skipping of
blocks that have nothing to do (DS_CO_DIRTY=off): the same masks, labels and verdict.  Small
cases are also compared with the reference restatement's masks (the exact predicate, byte for
byte), and the smallest epsilon with the restatement's search over its whole array."""
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
    t0, case, bad, longest = time.time(), 0, [], 0
    while time.time() - t0 < budget:
        case += 1
        rng = np.random.default_rng([seed, case])   # a case can be drawn again from its number alone
        small = rng.random() < 0.35
        if small:
            H, W, D = int(rng.integers(1, 8)), int(rng.integers(1, 12)), int(rng.integers(1, 30))
        else:
            H, W = int(rng.integers(1, 130)), int(rng.choice([rng.integers(1, 100), rng.integers(100, 1300)]))
            D = int(rng.choice([rng.integers(1, 17), rng.integers(1, 65)]))
        dtype = "f64" if rng.random() < 0.6 else "f32"
        iters = int(rng.choice([0, 0, 1, 2, 3, 6, 12, 30]))
        left, right = random_pair(H, W, seed=case)
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        if iters == 0:
            g.set_phi(random_phi(H, W, D, dtype, seed=case + 1))
        else:
            g.iterate(iters)
        star = g.epsilon_search(exact=True)
        ok = g.crossing_out(star, exact=True)["not_empty"]
        if not ok:
            print("case %d: the graph at the smallest epsilon %r is empty" % (case, star), flush=True)
        again = g.epsilon_search(exact=True)
        if again != star:
            print("case %d: the search found %r, then %r" % (case, star, again), flush=True)
            ok = False
        eps_list = [star, star * 0.999, star * 1.3, float(rng.uniform(0.0, 2.0 * star + 1e-3))]
        o = None
        if small:
            o = pyoracle.OracleGraph(left.astype(np.uint32), right.astype(np.uint32), D, 1.2)
            o.set_potentials(np.asarray(g.get_phi(), dtype=np.float64))
            o.co_alloc(True)
            ok = ok and star == o.epsilon_search(o.create_array_of_epsilons(0.0))
            gaps = g.epsilon_gaps()
            eps_list += [float(x) for x in gaps[rng.integers(0, len(gaps), size=min(4, len(gaps)))]]
        for eps in eps_list:
            for exact in (False, True):
                try:
                    got = g.crossing_out(eps, with_mask=True, exact=exact)
                except diffstereo.DiffStereoError:
                    # the O(D) comparisons refuse an epsilon below what f32 potentials resolve
                    if exact or dtype != "f32":
                        print("case %d: epsilon %r exact %s: refused" % (case, eps, exact), flush=True)
                        ok = False
                    continue
                os.environ["DS_CO_DIRTY"] = "off"
                want = g.crossing_out(eps, with_mask=True, exact=exact)
                del os.environ["DS_CO_DIRTY"]
                same = np.array_equal(got["alive"], want["alive"]) and np.array_equal(got["labels"], want["labels"]) \
                    and got["not_empty"] == want["not_empty"]
                if not same:
                    print("case %d: epsilon %r exact %s: with / without skipping: masks differ at %d entries, labels at %d, "
                          "verdict %s / %s, passes %d / %d" % (case, eps, exact, int((got["alive"] != want["alive"]).sum()),
                                                               int((got["labels"] != want["labels"]).sum()), got["not_empty"],
                                                               want["not_empty"], got["passes"], want["passes"]), flush=True)
                ok = ok and same
                longest = max(longest, got["passes"])
                if exact and o is not None:
                    o.initialize_with_epsilon(eps)
                    o.crossing_out()
                    ok = ok and np.array_equal(got["alive"], o.vertices) and got["not_empty"] == o.is_not_empty()
        g.close()
        if not ok:
            bad.append((case, H, W, D, dtype, iters))
        if case % 10 == 0 or not ok:
            print("case %d: %dx%dx%d %s, %d iterations, smallest epsilon %.6g%s -> %s (%.0f s)"
                  % (case, H, W, D, dtype, iters, star, ", against the restatement" if small else "",
                     "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("%d cases, %d mismatches, longest relaxation %d passes" % (case, len(bad), longest), flush=True)
    for b in bad:
        print("MISMATCH", b, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
