"""ds_iterate_raster: the reference's sequential sweep order on the GPU (anti-diagonal
launches) against the faithful raster-order restatement (oracle/diffusion_oracle.c,
diffusion_graph.rs:264-341).  Needs a GPU.

Unlike the red-black schedule, this one visits pixels in the reference's own order, so phi
itself (not only the optimum it converges to) must agree with the sequential oracle: 1e-9
(f64) / 1e-5 (f32) norm-wise relative, labels identical."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import pyoracle
from oracle.pyoracle import OracleGraph
from tests.util import random_pair, random_phi, rel_err

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo  # noqa: E402

TOL = {"f64": 1e-9, "f32": 1e-5}
CASES = [(1, 1, 1), (1, 7, 3), (7, 1, 3), (2, 2, 2), (5, 5, 40), (4, 7, 3), (6, 9, 5), (7, 12, 16),
         (9, 8, 13), (3, 20, 64), (2, 12, 65), (3, 9, 130), (12, 5, 4)]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("H,W,D", CASES)
def test_raster_order_matches_sequential_oracle(H, W, D, dtype):
    left, right = random_pair(H, W, seed=H + 17 * W + 3 * D)
    phi0 = random_phi(H, W, D, dtype, seed=9)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    g.set_phi(phi0)
    o = OracleGraph(left, right, D, 1.2, dtype=dtype)
    o.set_potentials(phi0)
    for it in range(3):
        g.iterate_raster(1)
        o.diffusion_act()
        assert rel_err(g.get_phi(), o.potentials) <= TOL[dtype], "iteration %d" % it
    assert abs(g.energy() - o.energy()) <= (1e-9 if dtype == "f64" else 1e-4) * max(1.0, abs(o.energy()))
    # (f_t is 0 up to rounding after a whole iteration, so only the belief argmin is meaningful)
    assert np.array_equal(g.labels(0).astype(np.int64), o.belief_labels().astype(np.int64))
    g.close()


def test_raster_differs_from_red_black_but_both_are_valid():
    left, right = random_pair(6, 9, seed=4)
    a = diffstereo.DiffusionGraph(left, right, 5, 1.2, dtype="f64")
    b = diffstereo.DiffusionGraph(left, right, 5, 1.2, dtype="f64")
    a.iterate(1)
    b.iterate_raster(1)
    assert not np.array_equal(a.get_phi(), b.get_phi())
    for g in (a, b):
        _, resid = g.energy(with_residual=True)
        assert resid <= 1e-9  # f_t = 0 at every pixel after a whole iteration, in either order
        g.close()


def test_bundled_pair_100_sequential_iterations(golden_dir):
    """images/left.pgm, right.pgm with main.rs:44-47's constants: phi, energy and labels of the
    committed sequential golden run (tests/golden/make_golden.py)."""
    left, _, _ = pyoracle.pgm_reader(golden_dir + "/left.pgm")
    right, _, _ = pyoracle.pgm_reader(golden_dir + "/right.pgm")
    gold = np.load(golden_dir + "/bundled_seq_100.npz")
    g = diffstereo.DiffusionGraph(left, right, 40, 1.2, dtype="f64")
    g.iterate_raster(100)
    assert rel_err(g.get_phi(), gold["phi_f64"]) <= 1e-9
    assert abs(g.energy() - float(gold["energy_f64"])) <= 1e-9 * max(1.0, abs(float(gold["energy_f64"])))
    assert np.array_equal(g.labels(0).astype(np.int32), gold["belief_labels"])
    g.close()


@pytest.mark.parametrize("H,W,D,dtype,n", [(6, 9, 5, "f64", 7), (13, 4, 20, "f32", 5), (3, 40, 130, "f32", 4),
                                           (1, 9, 3, "f64", 6), (9, 1, 3, "f64", 6), (24, 31, 64, "f64", 40)])
def test_iterations_in_flight_equal_iterations_one_by_one(H, W, D, dtype, n):
    """One call of n iterations keeps up to n of them in flight (diagonal k of iteration t beside
    diagonal k + 2 of iteration t - 1, ...); every pixel still sees the sequential sweep's
    operands, so the result equals n calls of one iteration bit for bit -- and the oracle."""
    left, right = random_pair(H, W, seed=H + 5 * W + D)
    phi0 = random_phi(H, W, D, dtype, seed=4)
    a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    a.set_phi(phi0)
    b.set_phi(phi0)
    a.iterate_raster(n)
    for _ in range(n):
        b.iterate_raster(1)
    assert np.array_equal(a.get_phi(), b.get_phi())
    if H * W * D * D * n <= 3_000_000:
        o = OracleGraph(left, right, D, 1.2, dtype=dtype)
        o.set_potentials(phi0)
        for _ in range(n):
            o.diffusion_act()
        assert rel_err(a.get_phi(), o.potentials) <= TOL[dtype]
    a.close()
    b.close()


def test_band_refusal():
    left, right = random_pair(4, 4)
    band = diffstereo.DiffusionGraph(left, right, 4, 1.2, row_begin=1, row_end=3)
    with pytest.raises(diffstereo.DiffStereoError):
        band.iterate_raster(1)
    band.close()


@pytest.mark.parametrize("dtype,H,W,D", [("f64", 9, 14, 7), ("f32", 12, 40, 64), ("f64", 5, 5, 40)])
def test_deferred_iterations_equal_one_batch(dtype, H, W, D):
    """ds_iterate_raster_async only notes the iterations; whatever needs the state next queues them
    as one batch.  One at a time + a reader, mixed with the red-black entry point and with
    ds_synchronize: the same bits as ds_iterate_raster."""
    left, right = random_pair(H, W, seed=H * W)
    phi0 = random_phi(H, W, D, dtype, seed=13)
    a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    for g in (a, b):
        g.set_phi(phi0)
    for _ in range(5):
        a.iterate_raster_async(1)
    a.iterate_raster_async(0)
    b.iterate_raster(5)
    assert a.energy() == b.energy()                      # a reader: the five run now
    assert np.array_equal(a.get_phi(), b.get_phi())
    a.iterate_raster_async(2)
    a.iterate(1)                                         # a writer: the two run first
    b.iterate_raster(2)
    b.iterate(1)
    assert np.array_equal(a.get_phi(), b.get_phi())
    a.iterate_raster_async(3)
    a.synchronize()
    a.iterate_raster_async(1)
    a.iterate_raster(1)                                  # the synchronous entry point takes the noted one along
    b.iterate_raster(5)
    assert a.checksum() == b.checksum()
    assert np.array_equal(a.get_phi(), b.get_phi())
    with pytest.raises(diffstereo.DiffStereoError):
        a.iterate_raster_async(-1)
    a.close()
    b.close()
