"""ds_find_best_labeling (crossing_out_graph.rs:349-364 on the GPU) against the oracle's
restatement on the same potentials and epsilon: the same disparity map, pixel for pixel."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import pyoracle
from tests.util import random_pair

pytestmark = pytest.mark.gpu


def _run(H, W, D, dtype, iters, eps, seed):
    load_package()
    from stereo_vision_amd import diffstereo
    left, right = random_pair(H, W, seed=seed)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    g.iterate(iters)
    o = pyoracle.OracleGraph(left.astype(np.uint32), right.astype(np.uint32), D, 1.2)
    o.set_potentials(np.asarray(g.get_phi(), dtype=np.float64))
    o.co_alloc(True)
    o.initialize_with_epsilon(eps)
    o.crossing_out()
    first = g.crossing_out(eps)
    assert first["not_empty"] == o.is_not_empty()
    return g, o, first


@pytest.mark.parametrize("H,W,D,dtype,iters,eps", [(5, 7, 6, "f64", 30, 3.0), (6, 9, 12, "f64", 10, 8.0),
                                                  (4, 12, 5, "f32", 20, 2.0), (9, 9, 40, "f64", 5, 20.0),
                                                  (3, 5, 4, "f64", 0, 1e9), (1, 9, 6, "f64", 15, 4.0),
                                                  (8, 1, 3, "f64", 15, 4.0), (3, 6, 256, "f32", 4, 30.0),
                                                  (4, 5, 100, "f64", 6, 25.0)])
def test_same_map_as_the_oracle(H, W, D, dtype, iters, eps):
    g, o, first = _run(H, W, D, dtype, iters, eps, seed=H * 31 + W)
    assert first["not_empty"]
    want = o.find_best_labeling()
    got = g.find_best_labeling()
    assert got["not_empty"]
    assert np.array_equal(got["labels"].astype(np.int32), want)
    assert 0 <= got["choices"] <= H * W
    # what is left on the device is exactly that labeling: one alive label per pixel
    after = g.co_finish(False, with_mask=True)
    assert np.array_equal(after["alive"].sum(axis=2), np.ones((H, W)))
    assert np.array_equal(after["labels"], got["labels"])
    g.close()


def test_choices_are_made_where_labels_tie():
    """A wide epsilon leaves several labels alive in many pixels: the walk must stop at each."""
    g, o, first = _run(6, 10, 8, "f64", 3, 50.0, seed=5)
    want = o.find_best_labeling()
    got = g.find_best_labeling()
    assert got["not_empty"] and got["choices"] > 0
    assert np.array_equal(got["labels"].astype(np.int32), want)
    g.close()


def test_needs_masks_and_the_whole_image():
    load_package()
    from stereo_vision_amd import diffstereo
    left, right = random_pair(8, 8, seed=1)
    g = diffstereo.DiffusionGraph(left, right, 4, 1.2, dtype="f64")
    with pytest.raises(diffstereo.DiffStereoError):
        g.find_best_labeling()          # no crossing-out yet
    g.close()
    b = diffstereo.DiffusionGraph(left, right, 4, 1.2, dtype="f64", row_begin=2, row_end=6)
    with pytest.raises(diffstereo.DiffStereoError):
        b.find_best_labeling()
    b.close()
