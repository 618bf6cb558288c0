"""Random shapes through ds_iterate against the O(D) oracle, bit for bit (a script, not collected
by pytest: `python3 tests/fuzz_iterate.py [seconds] [seed] [large]` on a GPU box; `large`: images of 70-700 rows
and 300-4000 columns instead of 1-69 and 1-999).

Every case draws H, W, D, the precision, the launch shape (two iterations per launch / one /
half-sweeps, by the library's own environment switches) and a sequence of call lengths, starts
from random potentials and compares phi after every call, then labels and energy."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from oracle.pyoracle import FastOracle  # noqa: E402
from stereo_vision_amd import diffstereo  # noqa: E402
from tests.util import random_pair, random_phi  # noqa: E402

SWITCHES = [{"DS_CACHE_RESIDENT": "fused"}, {"DS_CACHE_RESIDENT": "fused"}, {"DS_CACHE_RESIDENT": "fused"},
            {}, {"DS_CACHE_RESIDENT": "fused", "DS_FUSED2": "off"}, {"DS_CACHE_RESIDENT": "fused", "DS_FUSED": "off"},
            {"DS_CACHE_RESIDENT": "fused", "DS_STORE_FIRST": "off"}, {"DS_CACHE_RESIDENT": "fused", "DS_FUSED2_PARTIAL": "off"}]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    t0, case, bad = time.time(), 0, []
    while time.time() - t0 < budget:
        case += 1
        rng = np.random.default_rng([seed, case])   # a case can be drawn again from its number alone
        H = int(rng.integers(1, 70))
        W = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1000)]))
        D = int(rng.choice([rng.integers(1, 257), rng.choice([1, 2, 3, 4, 5, 40, 63, 64, 65, 127, 128, 129, 255, 256])]))
        dtype = "f64" if rng.random() < 0.4 else "f32"
        env = SWITCHES[int(rng.integers(0, len(SWITCHES)))]
        calls = [int(x) for x in rng.integers(1, 8, size=int(rng.integers(1, 5)))]
        if large:   # many strips and row segments of unequal length; at most 1.5e8 (pixel, label) pairs
            H, W = int(rng.integers(70, 700)), int(rng.integers(300, 4000))
            while H * W * D > 150_000_000:
                H, W = max(70, H * 2 // 3), max(300, W * 3 // 4)
                if H == 70 and W == 300:
                    break
            calls = [int(x) for x in rng.integers(1, 6, size=int(rng.integers(1, 4)))]
        for k in ("DS_CACHE_RESIDENT", "DS_FUSED2", "DS_FUSED", "DS_STORE_FIRST", "DS_FUSED2_PARTIAL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        left, right = random_pair(H, W, seed=case)
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        f = FastOracle(left, right, D, 1.2, dtype=dtype)
        phi0 = random_phi(H, W, D, dtype, seed=case + 1)
        g.set_phi(phi0)
        f.potentials[...] = phi0
        ok = True
        for n in calls:
            g.iterate(n)
            f.iterate(n)
            ok = ok and np.array_equal(g.get_phi(), f.potentials)
        ok = ok and np.array_equal(g.labels(0), f.labels(0))
        e, want = g.energy(), f.energy()
        ok = ok and abs(e - want) <= (1e-9 if dtype == "f64" else 1e-5) * max(1.0, abs(want))
        info = g.fused2_info()
        g.close()
        if not ok:
            bad.append((case, H, W, D, dtype, env, calls))
        if case % 10 == 0 or not ok or large:
            print("case %d: %dx%dx%d %s %s calls %s two-iteration steps %d -> %s (%.0f s)"
                  % (case, H, W, D, dtype, env, calls, info["steps"], "ok" if ok else "MISMATCH", time.time() - t0),
                  flush=True)
    print("%d cases, %d mismatches" % (case, len(bad)), flush=True)
    for b in bad:
        print("MISMATCH", b, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
