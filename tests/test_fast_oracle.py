"""The O(D) checkerboard oracle against the faithful O(D^2) restatement (CPU only).

oracle/fast_oracle.c is the bit-exact checker for the HIP kernels; here it is
validated against oracle/diffusion_oracle.c, which the reference's known
answers pin.  Tolerances: the two differ only by rounding (prefix/suffix-min
form against brute force).
"""
import numpy as np
import pytest

from oracle.pyoracle import FastOracle, OracleGraph
from tests.util import random_pair, random_phi, rel_err

SHAPES = [  # (H, W, D)
    (1, 1, 1), (1, 2, 2), (2, 1, 2), (2, 2, 2), (3, 3, 1), (5, 5, 40),
    (4, 7, 3), (6, 9, 5), (7, 12, 16), (5, 33, 40), (9, 8, 13), (3, 70, 64),
]


@pytest.mark.parametrize("H,W,D", SHAPES)
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-12), ("f32", 2e-6)])
def test_sweep_matches_bruteforce(H, W, D, dtype, tol):
    left, right = random_pair(H, W, seed=H * 1000 + W * 10 + D)
    phi0 = random_phi(H, W, D, dtype, seed=7)
    o = OracleGraph(left, right, D, 1.2, dtype=dtype)
    o.set_potentials(phi0)
    f = FastOracle(left, right, D, 1.2, dtype=dtype)
    f.potentials[...] = phi0
    for it in range(4):
        o.diffusion_act_checkerboard()
        f.iterate(1)
        assert rel_err(f.potentials, o.potentials) <= tol, "iteration %d" % it
    # invalid entries (d > j) are never touched
    for j in range(min(W, D)):
        assert np.array_equal(f.potentials[:, j, :, j + 1:], phi0[:, j, :, j + 1:])


@pytest.mark.parametrize("H,W,D", [(5, 5, 40), (6, 9, 5), (7, 12, 16), (1, 1, 1), (1, 4, 3)])
def test_energy_and_labels_match_bruteforce(H, W, D):
    left, right = random_pair(H, W, seed=3)
    o = OracleGraph(left, right, D, 1.2)
    f = FastOracle(left, right, D, 1.2)
    phi0 = random_phi(H, W, D, "f64", seed=11)
    o.set_potentials(phi0)
    f.potentials[...] = phi0
    e_o, e_f = o.energy(), f.energy()
    assert abs(e_o - e_f) <= 1e-10 * max(1.0, abs(e_o))
    assert np.array_equal(f.labels(1).astype(np.int32), o.depth_map())
    assert np.array_equal(f.labels(0).astype(np.int32), o.belief_labels())
    o.diffusion_act_checkerboard()
    f.iterate(1)
    assert abs(o.energy() - f.energy()) <= 1e-10 * max(1.0, abs(o.energy()))
    assert np.array_equal(f.labels(0).astype(np.int32), o.belief_labels())
    if H * W > 1:
        assert f.max_residual() < 1e-10


def test_half_sweep_rows_compose():
    """Sweeping row ranges separately equals sweeping the whole colour at once
    (what the row-band multi-GPU partition relies on)."""
    H, W, D = 9, 11, 6
    left, right = random_pair(H, W, seed=5)
    a = FastOracle(left, right, D, 1.2)
    b = FastOracle(left, right, D, 1.2)
    phi0 = random_phi(H, W, D, "f64", seed=1)
    a.potentials[...] = phi0
    b.potentials[...] = phi0
    for colour in (0, 1, 0, 1):
        a.half_sweep(colour)
        b.half_sweep(colour, 5, H)
        b.half_sweep(colour, 0, 2)
        b.half_sweep(colour, 2, 5)
    assert np.array_equal(a.potentials, b.potentials)


def test_energy_is_monotone_under_diffusion():
    """The sweep is coordinate ascent on the dual energy (diffusion_graph.rs:502-518)."""
    left, right = random_pair(8, 14, seed=9)
    f = FastOracle(left, right, 10, 1.2)
    prev = f.energy()
    for _ in range(15):
        f.iterate(1)
        e = f.energy()
        assert e >= prev - 1e-9 * max(1.0, abs(prev))
        prev = e
