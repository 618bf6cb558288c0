"""ds_epsilon_gaps (create_array_of_epsilons of diffusion_stereo/src/epsilon_search.rs:39-74 on
the GPU) against the oracle's restatement on the same potentials: the same set of values,
bit for bit, and the same epsilon out of the binary search."""
import os

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import pyoracle
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu


def _graphs(H, W, D, dtype, seed, iters=0):
    load_package()
    from stereo_vision_amd import diffstereo
    left, right = random_pair(H, W, seed=seed)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    # the oracle always in f64: the library widens f32 potentials and computes the gaps in f64
    o = pyoracle.OracleGraph(left.astype(np.uint32), right.astype(np.uint32), D, 1.2)
    if iters:
        g.iterate(iters)
        phi = g.get_phi()
    else:
        phi = random_phi(H, W, D, dtype, seed=seed + 1)
        g.set_phi(phi)
    o.set_potentials(np.asarray(phi, dtype=np.float64))
    return g, o


@pytest.mark.parametrize("H,W,D,dtype", [(5, 5, 40, "f64"), (6, 9, 5, "f64"), (4, 11, 8, "f32"),
                                         (3, 2, 3, "f64"), (1, 7, 4, "f64"), (7, 1, 4, "f32"),
                                         (5, 13, 17, "f64"), (2, 5, 256, "f32"), (3, 4, 130, "f64")])
def test_gaps_match_oracle(H, W, D, dtype):
    g, o = _graphs(H, W, D, dtype, seed=H * 100 + W)
    got = g.epsilon_gaps()
    want = np.unique(o.create_array_of_epsilons(0.0))
    assert got.dtype == np.float64 and np.all(np.diff(got) > 0) and np.all(np.isfinite(got))
    assert np.array_equal(got, want)
    # with the reference's tolerance dedup: the reference's array element for element
    for tol in (0.0, 1e-9, 0.5):
        assert np.array_equal(g.create_array_of_epsilons(tol), o.create_array_of_epsilons(tol))
    g.close()


def test_gaps_after_diffusion_and_search():
    """After real iterations: same array, and the binary search over it with the GPU
    crossing-out as predicate finds the oracle's epsilon."""
    g, o = _graphs(5, 9, 6, "f64", seed=3, iters=30)
    got = g.epsilon_gaps()
    assert np.array_equal(got, np.unique(o.create_array_of_epsilons(0.0)))
    array = pyoracle.dedup_f64(got, 1e-9)
    o.co_alloc(True)
    want = o.epsilon_search(o.create_array_of_epsilons(1e-9))
    first, last = 0, len(array) - 1
    while first < last:                       # epsilon_search.rs:84-112
        median = (first + last) // 2
        if g.crossing_out(array[median])["not_empty"]:
            last = median
        else:
            first = median + 1
    assert array[(first + last) // 2] == want
    g.close()


def test_gaps_need_the_whole_image():
    load_package()
    from stereo_vision_amd import diffstereo
    left, right = random_pair(8, 8, seed=1)
    g = diffstereo.DiffusionGraph(left, right, 4, 1.2, dtype="f64", row_begin=2, row_end=6)
    with pytest.raises(diffstereo.DiffStereoError):
        g.epsilon_gaps()
    g.close()


def _array_search(g, array):
    """epsilon_search.rs:84-112 over `array` with the GPU crossing-out as predicate."""
    first, last = 0, len(array) - 1
    while first < last:
        median = (first + last) // 2
        if g.crossing_out(array[median])["not_empty"]:
            last = median
        else:
            first = median + 1
    return array[(first + last) // 2]


@pytest.mark.parametrize("H,W,D,dtype", [(5, 5, 40, "f64"), (6, 9, 5, "f64"), (4, 11, 8, "f32"), (3, 2, 3, "f64"),
                                         (1, 7, 4, "f64"), (7, 1, 4, "f32"), (5, 13, 17, "f64"), (2, 5, 256, "f32"),
                                         (3, 4, 130, "f64"), (1, 1, 1, "f64"), (2, 3, 1, "f64")])
def test_one_pass_questions_about_the_gaps(H, W, D, dtype):
    """ds_epsilon_next against the oracle's array with every duplicate kept (a negative tolerance
    makes dedup_f64 keep each element): smallest gap above a bound, largest gap, gaps not above."""
    g, o = _graphs(H, W, D, dtype, seed=H * 100 + W)
    raw = np.asarray(o.create_array_of_epsilons(-1.0))
    assert np.all(np.diff(raw) >= 0)
    distinct = np.unique(raw)
    bounds = [-1.0, 0.0, raw[-1], raw[-1] + 1.0, float(np.nextafter(raw[-1], -np.inf))]
    for k in (1, len(distinct) // 3, len(distinct) // 2, len(distinct) - 2):
        if 0 <= k < len(distinct):
            bounds += [distinct[k], float(np.nextafter(distinct[k], np.inf)), float(np.nextafter(distinct[k], -np.inf))]
    for lo in bounds:
        nxt, largest, count = g.epsilon_next(lo)
        above = raw[raw > lo]
        assert nxt == (above[0] if len(above) else np.inf), lo
        assert largest == raw[-1]
        assert count == int((raw <= lo).sum()), lo
    g.close()


@pytest.mark.parametrize("H,W,D,iters,seed", [(5, 9, 6, 30, 3), (5, 5, 40, 100, 1), (8, 12, 10, 25, 7), (6, 20, 16, 60, 2),
                                              (12, 9, 7, 5, 4), (9, 14, 33, 40, 5), (3, 3, 2, 0, 6), (7, 11, 12, 0, 8),
                                              (4, 6, 5, 0, 9), (5, 4, 9, 2, 10), (2, 9, 20, 0, 12)])
def test_search_without_the_array(H, W, D, iters, seed):
    """ds_epsilon_search against the reference's binary search over the whole array (tolerance 0)
    with the same predicate, and -- with the reference's own comparisons as predicate -- against the
    oracle's search."""
    g, o = _graphs(H, W, D, "f64", seed=seed, iters=iters)
    array = g.create_array_of_epsilons(0.0)
    want = _array_search(g, array)
    got, stats = g.epsilon_search(exact=False, with_stats=True)
    assert got == want, (got, want, stats)
    assert stats["gap_passes"] >= 1 and stats["crossing_outs"] >= 1
    o.co_alloc(True)
    want_exact = o.epsilon_search(o.create_array_of_epsilons(0.0))
    got_exact, stats = g.epsilon_search(exact=True, with_stats=True)
    assert got_exact == want_exact, (got_exact, want_exact, stats)
    # the masks the handle is left with are those of the epsilon found
    assert g.crossing_out(got_exact, exact=True)["not_empty"]
    g.close()


def test_search_on_an_f32_handle():
    """exact = 1 widens the potentials: the oracle's search on the same (widened) potentials."""
    g, o = _graphs(6, 9, 8, "f32", seed=21, iters=12)
    o.co_alloc(True)
    assert g.epsilon_search(exact=True) == o.epsilon_search(o.create_array_of_epsilons(0.0))
    g.close()


def test_search_without_the_array_at_a_size_the_array_still_fits():
    """48 x 64 x 40 after 60 iterations: 19.8 million gaps; the streamed search and the search over
    the sorted array agree."""
    g, _ = _graphs(48, 64, 40, "f64", seed=11, iters=60)
    array = g.create_array_of_epsilons(0.0)
    got, stats = g.epsilon_search(exact=False, with_stats=True)
    assert got == _array_search(g, array), stats
    exact, stats = g.epsilon_search(exact=True, with_stats=True)
    assert abs(exact - got) <= 1e-9 * max(1.0, got), (exact, got, stats)
    g.close()


def test_search_refuses_what_it_cannot_do():
    load_package()
    from stereo_vision_amd import diffstereo
    left, right = random_pair(8, 8, seed=1)
    band = diffstereo.DiffusionGraph(left, right, 4, 1.2, dtype="f64", row_begin=2, row_end=6)
    with pytest.raises(diffstereo.DiffStereoError):
        band.epsilon_search()
    with pytest.raises(diffstereo.DiffStereoError):
        band.epsilon_next(0.0)
    band.close()
    single = diffstereo.DiffusionGraph(left, right, 4, 1.2, dtype="f32")
    with pytest.raises(diffstereo.DiffStereoError):
        single.epsilon_search(exact=False)
    single.close()
