"""60 random small graphs (1-8 x 1-13 pixels, 1-70 labels, random or iterated potentials, both
precisions): ds_epsilon_search with the reference's comparisons against the oracle's search over its
whole array, and ds_crossing_out_exact at a dozen of each graph's gaps against the oracle's masks.
Needs a GPU."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import pyoracle
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo  # noqa: E402


def test_random_graphs_against_the_oracle():
    rng = np.random.default_rng(123)
    bad = []
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
        ok = g.epsilon_search(exact=True) == o.epsilon_search(o.create_array_of_epsilons(0.0))
        gaps = g.epsilon_gaps()
        for eps in gaps[rng.integers(0, len(gaps), size=min(12, len(gaps)))]:
            o.initialize_with_epsilon(float(eps))
            o.crossing_out()
            r = g.crossing_out(float(eps), with_mask=True, exact=True)
            ok = ok and np.array_equal(r["alive"], o.vertices) and r["not_empty"] == o.is_not_empty()
        if not ok:
            bad.append((case, H, W, D, iters, dtype))
        g.close()
    assert not bad, bad
