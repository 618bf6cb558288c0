"""GPU crossing-out (ds_crossing_out) against the faithful restatement of
CrossingOutGraph (oracle/diffusion_oracle.c, crossing_out_graph.rs:67-327).  Needs a GPU.

The oracle keeps the reference's O(D^2) edge masks and deletes in the reference's
order; the GPU keeps only the vertex mask and re-derives edge tests as min-plus
products.  Both must end in the same greatest consistent sub-graph, so the vertex
masks are compared byte for byte.  epsilon is kept well above rounding so that the
different (but equivalent) floating-point evaluation orders cannot flip a test.
"""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import pyoracle
from oracle.pyoracle import OracleGraph
from tests.util import NumpyCrossingOut, random_pair, random_phi

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo  # noqa: E402

CASES = [(1, 1, 1), (1, 6, 3), (6, 1, 3), (3, 4, 2), (5, 5, 40), (6, 9, 5), (4, 11, 16), (3, 21, 20),
         (2, 12, 65), (2, 9, 130)]


def oracle_crossing_out(left, right, D, lam, phi, eps):
    o = OracleGraph(left, right, D, lam, dtype="f64")
    o.set_potentials(phi)
    o.co_alloc(True)
    o.initialize_with_epsilon(eps)
    o.crossing_out()
    alive = o.vertices.copy()
    not_empty = o.is_not_empty()
    labels = np.where(alive.any(axis=2), alive.argmax(axis=2), -1)
    return alive, not_empty, labels


@pytest.mark.parametrize("iters,eps", [(0, 0.5), (2, 20.0), (2, 60.0), (30, 0.01), (30, 5.0), (30, 20.0)])
@pytest.mark.parametrize("H,W,D", CASES)
def test_vertex_mask_matches_reference_restatement(H, W, D, iters, eps):
    left, right = random_pair(H, W, seed=3 * H + 5 * W + 7 * D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    if iters == 0:
        g.set_phi(random_phi(H, W, D, "f64", seed=11))
    else:
        g.iterate(iters)
    phi = g.get_phi()
    got = g.crossing_out(eps, with_mask=True)
    alive, not_empty, labels = oracle_crossing_out(left, right, D, 1.2, phi, eps)
    assert np.array_equal(got["alive"], alive)
    assert got["not_empty"] == not_empty
    assert np.array_equal(got["labels"].astype(np.int64), np.where(labels < 0, 0xFFFFFFFF, labels))
    assert got["passes"] >= 1
    g.close()


def test_bundled_pair_reference_pipeline(golden_dir):
    """main.rs:44-71 on the bundled pair: batches of 200 iterations until the
    epsilon-consistent sub-graph is non-empty, then the smallest alive label."""
    left, _, _ = pyoracle.pgm_reader(golden_dir + "/left.pgm")
    right, _, _ = pyoracle.pgm_reader(golden_dir + "/right.pgm")
    gold = np.load(golden_dir + "/bundled_main.npz")
    H, W = left.shape
    eps = 1.0 / (10 * W * H)
    g = diffstereo.DiffusionGraph(left, right, 40, 1.2, dtype="f64")
    for batch in range(1, 20):
        g.iterate(200)
        r = g.crossing_out(eps)
        if r["not_empty"]:
            break
    assert r["not_empty"]
    assert batch * 200 == int(gold["iters_cb"])
    assert np.array_equal(r["labels"], gold["labels_cb"].astype(np.uint32))
    g.close()


def test_empty_graph_and_band_refusal():
    # two pixels whose only cheap labels cannot be joined: eps = 0 after a random phi
    left, right = random_pair(3, 8, seed=2)
    g = diffstereo.DiffusionGraph(left, right, 6, 1.2, dtype="f64")
    g.set_phi(random_phi(3, 8, 6, "f64", seed=1))
    r = g.crossing_out(1e-12, with_mask=True)
    alive, not_empty, _ = oracle_crossing_out(left, right, 6, 1.2, g.get_phi(), 1e-12)
    assert np.array_equal(r["alive"], alive) and r["not_empty"] == not_empty and not not_empty
    assert (r["labels"] == 0xFFFFFFFF).any()
    with pytest.raises(diffstereo.DiffStereoError):
        g.crossing_out(-1.0)
    g.close()
    band = diffstereo.DiffusionGraph(left, right, 6, 1.2, row_begin=1, row_end=3)
    with pytest.raises(diffstereo.DiffStereoError):
        band.crossing_out(0.1)  # a band goes through co_begin / co_pass / co_halo / co_finish
    band.close()


def test_f32_and_large_masks_are_consistent():
    """A size the O(D^2) reference-order oracle cannot reach, against a vectorised
    restatement of the same closure; plus idempotence and label range."""
    from stereo_vision_amd import synth
    H, W, D = 24, 96, 64
    left, right, _ = synth.synthetic_pair(H, W, D)
    for dtype in ("f64", "f32"):
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        g.iterate(40)
        eps = 0.05
        r = g.crossing_out(eps, with_mask=True)
        alive = r["alive"].astype(bool)
        phi = g.get_phi().astype(np.float64)
        cost = g.get_cost().astype(np.float64).transpose(1, 2, 0)  # [H][W][D]
        want = NumpyCrossingOut(phi, cost, 1.2, eps).run()
        if dtype == "f64":
            assert np.array_equal(alive, want)
        else:  # f32 thresholds are evaluated in f32: allow a handful of borderline labels
            assert (alive != want).mean() < 1e-3
        r2 = g.crossing_out(eps, with_mask=True)
        assert np.array_equal(r2["alive"], r["alive"]) and r2["passes"] == r["passes"]
        if r["not_empty"]:
            lab = r["labels"].astype(np.int64)
            assert (lab <= np.minimum(np.arange(W)[None, :], D - 1)).all()
        g.close()


@pytest.mark.parametrize("iters,eps", [(30, 5.0), (2, 60.0), (2, 3.0)])
def test_bands_on_one_gpu_match_whole_image(iters, eps):
    """Three band handles in one process, phi and mask halo rows copied device to device
    through ds_halo / ds_co_halo: every band ends with its rows of the whole image's result."""
    import ctypes as C
    hip = diffstereo.hip_runtime()
    H, W, D = 11, 18, 12
    left, right = random_pair(H, W, seed=5)
    whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    cuts = [(0, 4), (4, 5), (5, 11)]
    bands = [diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64", row_begin=a, row_end=b) for a, b in cuts]

    def exchange(halo):
        for up, down in zip(bands[:-1], bands[1:]):
            for src, dst in (((up, 1, 0), (down, 0, 1)), ((down, 0, 0), (up, 1, 1))):
                for colour in (0, 1):
                    sp, n = getattr(src[0], halo)(colour, src[1], src[2])
                    dp, m = getattr(dst[0], halo)(colour, dst[1], dst[2])
                    assert n == m
                    src[0].synchronize()
                    assert hip.hipMemcpy(C.c_void_p(dp), C.c_void_p(sp), C.c_size_t(n), 3) == 0

    for _ in range(iters):
        for colour in (0, 1):
            whole.sweep(colour)
            for b in bands:
                b.sweep(colour)
            exchange("halo")
    want = whole.crossing_out(eps, with_mask=True)
    for b in bands:
        b.co_begin(eps)
    exchange("co_halo")
    while True:
        flags = [b.co_pass() for b in bands]
        changed, lost = any(f[0] for f in flags), any(f[1] for f in flags)
        if lost or not changed:
            break
        exchange("co_halo")
    got = [b.co_finish(lost, with_mask=True) for b in bands]
    assert np.array_equal(np.concatenate([g["alive"] for g in got]), want["alive"])
    assert np.array_equal(np.concatenate([g["labels"] for g in got]), want["labels"])
    assert all(g["not_empty"] for g in got) == want["not_empty"]
    with pytest.raises(diffstereo.DiffStereoError):
        diffstereo.DiffusionGraph(left, right, D, 1.2).co_pass()  # before co_begin
    for b in bands + [whole]:
        b.close()


@pytest.mark.parametrize("H,W,D,dtype,iters", [(3, 4, 3, "f64", 0), (2, 5, 6, "f64", 0), (4, 4, 5, "f64", 6), (1, 6, 8, "f64", 0),
                                               (5, 1, 4, "f64", 3), (3, 7, 12, "f64", 10), (4, 5, 40, "f64", 25),
                                               (3, 5, 7, "f32", 0), (4, 6, 9, "f32", 8), (2, 3, 70, "f64", 4)])
def test_exact_comparisons_at_every_gap_of_the_graph(H, W, D, dtype, iters):
    """ds_crossing_out_exact with epsilon equal to a gap of the graph -- where one comparison of the
    reference is an equality up to its own rounding -- for (a sample of) every gap: the
    reference restatement's vertex masks byte for byte.  f32 handles: the reference's f64
    arithmetic on the widened potentials."""
    left, right = random_pair(H, W, seed=H + 3 * W + 5 * D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    if iters == 0:
        g.set_phi(random_phi(H, W, D, dtype, seed=23))
    else:
        g.iterate(iters)
    phi = np.asarray(g.get_phi(), dtype=np.float64)
    gaps = g.epsilon_gaps()
    picks = gaps if len(gaps) <= 150 else gaps[np.unique(np.linspace(0, len(gaps) - 1, 150).astype(int))]
    o = OracleGraph(left, right, D, 1.2, dtype="f64")
    o.set_potentials(phi)
    o.co_alloc(True)
    differing_fast = 0
    for eps in picks:
        o.initialize_with_epsilon(float(eps))
        o.crossing_out()
        got = g.crossing_out(float(eps), with_mask=True, exact=True)
        assert np.array_equal(got["alive"], o.vertices), eps
        assert got["not_empty"] == o.is_not_empty(), eps
        if dtype == "f64":
            differing_fast += int(not np.array_equal(g.crossing_out(float(eps), with_mask=True)["alive"], o.vertices))
    print("candidates tried: %d, of which the O(D) comparisons decide differently: %d" % (len(picks), differing_fast))
    g.close()


def test_exact_masks_of_an_empty_result_are_the_references():
    """An empty result with masks asked for: the passes go on to the reference's fixed point, where
    vertices held by an edge copy whose twin was never set may still stand (this graph has some at
    three of its gaps; found by tests/fuzz_crossing_out.py).  Without masks the call stops at the
    first pixel that lost every label: same verdict, every label 0xffffffff."""
    H, W, D = 1, 10, 22
    left, right = random_pair(H, W, seed=546)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    g.iterate(3)
    o = OracleGraph(left, right, D, 1.2, dtype="f64")
    o.set_potentials(np.asarray(g.get_phi(), dtype=np.float64))
    o.co_alloc(True)
    standing = 0
    for eps in g.epsilon_gaps():
        o.initialize_with_epsilon(float(eps))
        o.crossing_out()
        got = g.crossing_out(float(eps), with_mask=True, exact=True)
        assert np.array_equal(got["alive"], o.vertices), eps
        assert got["not_empty"] == o.is_not_empty(), eps
        quick = g.crossing_out(float(eps), exact=True)
        assert quick["not_empty"] == got["not_empty"]
        if not got["not_empty"]:
            standing += int(got["alive"].any())
            assert (quick["labels"] == 0xffffffff).all()
    assert standing >= 3
    g.close()


def test_exact_crossing_out_leaves_the_state_other_entry_points_expect():
    left, right = random_pair(6, 9, seed=4)
    g = diffstereo.DiffusionGraph(left, right, 5, 1.2, dtype="f64")
    g.iterate(20)
    r = g.crossing_out(2.0, with_mask=True, exact=True)
    fast = g.crossing_out(2.0, with_mask=True)
    assert np.array_equal(r["alive"], fast["alive"]) and r["not_empty"] == fast["not_empty"]
    g.crossing_out(2.0, exact=True)
    best = g.find_best_labeling()
    g.crossing_out(2.0)
    assert np.array_equal(best["labels"], g.find_best_labeling()["labels"])
    band = diffstereo.DiffusionGraph(left, right, 5, 1.2, dtype="f64", row_begin=1, row_end=4)
    with pytest.raises(diffstereo.DiffStereoError):
        band.crossing_out(2.0, exact=True)
    band.close()
    g.close()


@pytest.mark.parametrize("H,W,D,dtype,iters", [(128, 1024, 4, "f64", 3), (96, 1200, 8, "f32", 2), (70, 900, 12, "f64", 0),
                                               (128, 1024, 4, "f64", 12)])
def test_skipping_blocks_with_nothing_to_do_changes_no_mask(monkeypatch, H, W, D, dtype, iters):
    """Relaxation passes skip blocks whose neighbourhood deleted nothing in the last pass, and from
    the third pass on a block takes 32 row segments.  With DS_CO_DIRTY=off every block runs in every
    pass: the same masks, the same labels, both with the O(D) comparisons and the reference's -- at
    the smallest epsilon that leaves a graph, just below it and well above it, where relaxations are long."""
    left, right = random_pair(H, W, seed=H + W + D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    if iters == 0:
        g.set_phi(random_phi(H, W, D, dtype, seed=5))
    else:
        g.iterate(iters)
    star = g.epsilon_search(exact=True)
    longest = 0
    for eps in (star, star * 0.999, star * 1.3, star * 2.0):
        for exact in (False, True):
            got = g.crossing_out(eps, with_mask=True, exact=exact)
            monkeypatch.setenv("DS_CO_DIRTY", "off")
            want = g.crossing_out(eps, with_mask=True, exact=exact)
            monkeypatch.delenv("DS_CO_DIRTY")
            assert np.array_equal(got["alive"], want["alive"]), (eps, exact)
            assert np.array_equal(got["labels"], want["labels"]) and got["not_empty"] == want["not_empty"]
            longest = max(longest, got["passes"])
    assert g.crossing_out(star, exact=True)["not_empty"] and not g.crossing_out(star * 0.999, exact=True)["not_empty"]
    print("smallest epsilon with a graph %g, longest relaxation %d passes" % (star, longest))
    g.close()
