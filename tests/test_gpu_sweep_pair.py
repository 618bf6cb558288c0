"""ds_sweep_pair: two row ranges of one colour in one launch = the two ds_sweep / ds_sweep_next
launches it replaces on a band's boundary rows, bit for bit; and its argument checks."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu


def _pair_of_graphs(H, W, D, dtype):
    load_package()
    from stereo_vision_amd import diffstereo
    left, right = random_pair(H, W, seed=21)
    phi = random_phi(H, W, D, dtype, seed=22)
    graphs = []
    for _ in range(2):
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        g.set_phi(phi)
        graphs.append(g)
    return graphs


@pytest.mark.parametrize("dtype,D", [("f32", 64), ("f64", 12), ("f32", 130)])
def test_pair_equals_two_launches(dtype, D):
    H, W = 14, 150
    a, b = _pair_of_graphs(H, W, D, dtype)
    for colour in (0, 1):
        a.sweep(colour, 2, 4)
        a.sweep(colour, 9, 12)
        b.sweep_pair(colour, False, (2, 4), (9, 12))
        assert np.array_equal(a.get_phi(), b.get_phi())
    # one range empty
    a.sweep(0, 5, 6)
    b.sweep_pair(0, False, (5, 6), (8, 8))
    assert np.array_equal(a.get_phi(), b.get_phi())
    phi = a.get_phi()
    a.close()
    b.close()
    # into the planes the next whole-iteration kernel produces, then that kernel over the rest:
    # a band with a neighbour on either side leaves its first and last row to the half-sweeps
    from stereo_vision_amd import diffstereo
    left, right = random_pair(H, W, seed=21)
    a, b = [diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=1, row_end=H - 1) for _ in range(2)]
    for g in (a, b):
        g.set_phi(phi, 0, H)  # own rows and both halo rows
    fb, fe = a.fused_rows()
    assert (fb, fe) == (2, H - 2)
    a.sweep(0, 1, 3)
    a.sweep(0, H - 3, H - 1)
    a.sweep_next(1, 2)
    a.sweep_next(H - 2, H - 1)
    b.sweep_pair(0, False, (1, 3), (H - 3, H - 1))
    b.sweep_pair(1, True, (1, 2), (H - 2, H - 1))
    for g in (a, b):
        with pytest.raises(diffstereo.DiffStereoError):
            g.iterate_rows(fb + 1, fe)  # a part of the range would leave stale rows behind
        g.iterate_rows(fb, fe)
        with pytest.raises(diffstereo.DiffStereoError):
            g.checksum()                # the first colour's planes are out of date
        g.finish_iterations()
        g.finish_iterations()           # nothing pending: a no-op
    assert np.array_equal(a.get_phi(), b.get_phi())
    assert a.checksum() == b.checksum()
    a.close()
    b.close()


def test_pair_rejects_bad_ranges():
    load_package()
    from stereo_vision_amd import diffstereo
    a, b = _pair_of_graphs(10, 20, 8, "f32")
    for args in [(0, False, (2, 5), (4, 6)),      # overlap
                 (0, False, (-1, 2), (4, 6)),     # outside the image
                 (0, False, (3, 2), (4, 6)),      # reversed
                 (2, False, (0, 1), (4, 6)),      # colour
                 (0, True, (0, 1), (4, 6))]:      # only the second colour has upcoming planes
        with pytest.raises(diffstereo.DiffStereoError):
            a.sweep_pair(*args)
    assert np.array_equal(a.get_phi(), b.get_phi())  # nothing ran
    a.close()
    b.close()
