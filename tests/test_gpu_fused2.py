"""Two iterations per launch (ds_iterate: D = 64 / 128 / 256 fill the lanes of a pixel slot, every
other D takes the kernel's instantiation for partly filled slots; f32 in 12-wavefront blocks, f64
in 8-wavefront blocks) against one iteration per launch (DS_FUSED2=off) and the O(D) oracle.
Needs a GPU.

Bar: bit-identical phi (integer-exact comparison of the float bit patterns via array_equal),
identical labels, equal energy."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle.pyoracle import FastOracle
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

# (H, W, D): interiors of a few strips and a few rows; odd widths; rows whose count is odd
SHAPES = [(20, 400, 256), (21, 333, 256), (30, 700, 128), (17, 421, 128), (24, 600, 64), (17, 390, 64),
          (45, 330, 256),
          # D below the lanes' capacity: the reference's own 40, and 100 / 200 / 8 / 132 / 60
          (24, 600, 40), (19, 333, 100), (22, 500, 200), (17, 400, 8), (20, 421, 132), (31, 390, 60),
          # D that is not a multiple of 4: the last label lane also holds padding labels
          (20, 400, 41), (22, 500, 202), (19, 333, 99), (17, 300, 3), (18, 280, 1), (21, 345, 130), (16, 410, 255)]


@pytest.mark.parametrize("H,W,D", SHAPES)
def test_pairs_of_iterations_equal_single_iterations(monkeypatch, H, W, D):
    left, right = random_pair(H, W, seed=H + W + D)
    phi0 = random_phi(H, W, D, "f32", seed=7)
    g2 = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    info = g2.fused2_info()
    assert info["available"] and info["cols"][0] < info["cols"][1] and info["rows"] == (4, H - 4)
    monkeypatch.setenv("DS_FUSED2", "off")
    g1 = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    monkeypatch.delenv("DS_FUSED2")
    assert not g1.fused2_info()["available"]
    f = FastOracle(left, right, D, 1.2, dtype="f32")
    for g in (g1, g2):
        g.set_phi(phi0)
    f.potentials[...] = phi0
    steps = 0
    for n in (3, 4, 7, 2, 1, 5):   # 1, 2, 3, 1, 0, 2 two-iteration steps (the call's last launch also stores the first colour)
        g1.iterate(n)
        g2.iterate(n)
        f.iterate(n)
        steps += n // 2
        phi = g2.get_phi()
        assert np.array_equal(phi, g1.get_phi()), "after a call of %d iterations" % n
        assert np.array_equal(phi, f.potentials), "oracle, after a call of %d iterations" % n
    assert g2.fused2_info()["steps"] == steps and g1.fused2_info()["steps"] == 0
    assert g2.checksum() == g1.checksum()
    assert g2.energy() == g1.energy()
    assert np.array_equal(g2.labels(0), f.labels(0))
    g1.close()
    g2.close()


def test_handles_without_an_interior_fall_back():
    for H, W, D, dtype in ((20, 52, 40, "f32"), (11, 400, 256, "f32"), (20, 60, 41, "f64"), (9, 300, 100, "f64")):
        left, right = random_pair(H, W, seed=3)
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        assert not g.fused2_info()["available"], (H, W, D, dtype)
        g.iterate(5)
        f = FastOracle(left, right, D, 1.2, dtype=dtype)
        f.iterate(5)
        assert np.array_equal(g.get_phi(), f.potentials)
        g.close()
    # an image with no column where every label exists: the two-iteration kernel takes it all the
    # same (labels d > j are masked in its strips left of column D)
    for H, W, D in ((20, 290, 256), (24, 200, 128), (20, 300, 128), (20, 160, 100)):
        left, right = random_pair(H, W, seed=5)
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
        info = g.fused2_info()
        assert info["available"] and info["cols"][0] < D, (H, W, D, info)
        g.iterate(5)
        assert g.fused2_info()["steps"] == 2
        f = FastOracle(left, right, D, 1.2, dtype="f32")
        f.iterate(5)
        assert np.array_equal(g.get_phi(), f.potentials)
        g.close()
    # a band has an interior too (three rows in from a neighbouring band, four from the image's edge)
    band = diffstereo.DiffusionGraph(*random_pair(40, 400, seed=1), 256, 1.2, dtype="f32", row_begin=0, row_end=20)
    assert band.fused2_info()["available"] and band.fused2_info()["rows"] == (4, 17)
    band.close()


def test_full_size_4k_256(monkeypatch):
    """The benchmark's own size: checksum, energy and labels equal to one iteration per launch
    after 9 iterations (4 two-iteration steps + 1), top rows bit-exact against the oracle."""
    H, W, D = 2160, 3840, 256
    monkeypatch.setenv("DS_PLACEMENT_TRIES", "2")
    left, right, _ = synth.synthetic_pair(H, W, D)
    g2 = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    g2.iterate(9)
    assert g2.fused2_info()["steps"] == 4
    c2, e2, l2 = g2.checksum(), g2.energy(), g2.labels(0)
    top = g2.get_phi(0, 20)
    g2.close()
    monkeypatch.setenv("DS_FUSED2", "off")
    g1 = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    g1.iterate(9)
    assert g1.checksum() == c2 and g1.energy() == e2 and np.array_equal(g1.labels(0), l2)
    g1.close()
    f = FastOracle(left[:40], right[:40], D, 1.2, dtype="f32")
    f.iterate(9)
    assert np.array_equal(top, f.potentials[:20])


@pytest.mark.parametrize("dtype,H,W,D", [("f32", 21, 401, 256), ("f32", 19, 333, 128), ("f32", 17, 390, 64),
                                          ("f32", 15, 77, 40), ("f64", 18, 140, 100), ("f64", 13, 300, 256),
                                          ("f32", 9, 35, 24), ("f32", 18, 390, 40), ("f32", 16, 340, 200)])
def test_last_launch_stores_the_first_colour(monkeypatch, dtype, H, W, D):
    """ds_iterate's last launch writes the first colour's planes itself; with DS_STORE_FIRST=off a
    half-sweep does it afterwards from the planes of the iteration before.  Every plane of both
    colours must come out the same, call after call, and equal to the oracle's."""
    left, right = random_pair(H, W, seed=H * W + D)
    phi0 = random_phi(H, W, D, dtype, seed=11)
    a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    f = FastOracle(left, right, D, 1.2, dtype=dtype)
    a.set_phi(phi0)
    b.set_phi(phi0)
    f.potentials[...] = phi0
    for n in (1, 2, 3, 4, 1):
        a.iterate(n)
        monkeypatch.setenv("DS_STORE_FIRST", "off")
        b.iterate(n)
        monkeypatch.delenv("DS_STORE_FIRST")
        f.iterate(n)
        got = a.get_phi()
        assert np.array_equal(got, b.get_phi()), "after a call of %d iterations" % n
        assert np.array_equal(got, f.potentials), "oracle, after a call of %d iterations" % n
        assert a.checksum() == b.checksum() and a.energy() == b.energy()
    a.sweep(0)   # the handle is in the state half-sweeps expect
    b.sweep(0)
    assert np.array_equal(a.get_phi(), b.get_phi())
    a.close()
    b.close()


@pytest.mark.parametrize("left_cols", ["full", "90", "200"])
def test_any_split_between_the_two_kernels_gives_the_same_bits(monkeypatch, left_cols):
    """DS_FUSED2_LEFT moves the column where the one-iteration kernel's frame ends and the
    two-iteration kernel begins (its last frame strip is cut off there): same potentials."""
    H, W, D = 22, 520, 128
    left, right = random_pair(H, W, seed=31)
    phi0 = random_phi(H, W, D, "f32", seed=5)
    ref = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    monkeypatch.setenv("DS_FUSED2_LEFT", left_cols)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    monkeypatch.delenv("DS_FUSED2_LEFT")
    assert g.fused2_info()["available"] and g.fused2_info()["cols"][0] > ref.fused2_info()["cols"][0]
    for x in (ref, g):
        x.set_phi(phi0)
        x.iterate(4)
        x.iterate(3)
    assert g.fused2_info()["steps"] == 3
    assert np.array_equal(g.get_phi(), ref.get_phi())
    ref.close()
    g.close()


def test_partly_filled_slots_can_be_switched_off(monkeypatch):
    """DS_FUSED2_PARTIAL=off keeps D below the lanes' capacity on one iteration per launch: the
    same potentials either way."""
    H, W, D = 20, 500, 40
    left, right = random_pair(H, W, seed=2)
    phi0 = random_phi(H, W, D, "f32", seed=3)
    on = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    monkeypatch.setenv("DS_FUSED2_PARTIAL", "off")
    off = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    monkeypatch.delenv("DS_FUSED2_PARTIAL")
    assert on.fused2_info()["available"] and not off.fused2_info()["available"]
    for g in (on, off):
        g.set_phi(phi0)
        g.iterate(6)
    assert on.fused2_info()["steps"] == 3 and off.fused2_info()["steps"] == 0
    assert np.array_equal(on.get_phi(), off.get_phi())
    on.close()
    off.close()


@pytest.mark.parametrize("H,W,D", [(20, 200, 256), (26, 300, 128), (21, 333, 64), (33, 521, 256), (40, 197, 64),
                                   # partly filled slots: the reference's own D = 40 in its own precision, and others
                                   (24, 400, 40), (19, 233, 100), (22, 300, 200), (17, 200, 8), (20, 321, 132),
                                   (31, 290, 60), (18, 150, 40),
                                   # D that is not a multiple of 4
                                   (20, 400, 41), (20, 400, 102), (18, 300, 130), (17, 200, 5), (19, 333, 255)])
def test_f64_pairs_give_the_same_bits(monkeypatch, H, W, D):
    """The two-iteration kernel in f64 (8-wavefront blocks) against one iteration per launch
    (DS_FUSED2_F64=off) and the oracle."""
    left, right = random_pair(H, W, seed=H + D)
    phi0 = random_phi(H, W, D, "f64", seed=17)
    monkeypatch.setenv("DS_FUSED2_F64", "off")
    ref = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    monkeypatch.delenv("DS_FUSED2_F64")
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    assert g.fused2_info()["available"] and not ref.fused2_info()["available"]
    f = FastOracle(left, right, D, 1.2, dtype="f64")
    for x in (ref, g):
        x.set_phi(phi0)
    f.potentials[...] = phi0
    for n in (4, 3, 1, 2):
        ref.iterate(n)
        g.iterate(n)
        f.iterate(n)
        got = g.get_phi()
        assert np.array_equal(got, ref.get_phi()), "after a call of %d iterations" % n
        assert np.array_equal(got, f.potentials)
    assert g.fused2_info()["steps"] == 2 + 1 + 0 + 1
    assert g.energy() == ref.energy()
    ref.close()
    g.close()
