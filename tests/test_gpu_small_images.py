"""Images whose potentials fit the GPU's last-level cache (8 planes <= 256 MiB: the reference's own
400 x 400 x 40 among them) are iterated by half-sweeps unless DS_CACHE_RESIDENT=fused, which the
other tests run under.  Same bits either way, and as the oracle.  Needs a GPU."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle.pyoracle import FastOracle
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo  # noqa: E402


@pytest.mark.parametrize("H,W,D,dtype", [(60, 400, 40, "f64"), (48, 330, 256, "f32"), (40, 420, 7, "f32"), (80, 300, 64, "f64")])
def test_default_for_cache_resident_images_is_half_sweeps(monkeypatch, H, W, D, dtype):
    left, right = random_pair(H, W, seed=H + W)
    phi0 = random_phi(H, W, D, dtype, seed=9)
    fused = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)      # the test session's setting
    monkeypatch.delenv("DS_CACHE_RESIDENT")
    swept = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)      # the library's default
    assert fused.fused2_info()["available"] and not swept.fused2_info()["available"]
    assert swept.fused_rows() == (0, 0) or swept.fused_rows()[0] >= swept.fused_rows()[1]
    f = FastOracle(left, right, D, 1.2, dtype=dtype)
    for g in (fused, swept):
        g.set_phi(phi0)
    f.potentials[...] = phi0
    for n in (3, 4, 1):
        fused.iterate(n)
        swept.iterate(n)
        f.iterate(n)
        assert np.array_equal(swept.get_phi(), fused.get_phi())
        assert np.array_equal(swept.get_phi(), f.potentials)
    assert swept.fused2_info()["steps"] == 0 and fused.fused2_info()["steps"] == 3
    assert swept.energy() == fused.energy() and swept.checksum() == fused.checksum()
    assert np.array_equal(swept.labels(0), fused.labels(0))
    fused.close()
    swept.close()


def test_images_beyond_the_cache_keep_the_fused_kernels(monkeypatch):
    monkeypatch.delenv("DS_CACHE_RESIDENT")
    left, right = random_pair(300, 600, seed=4)
    g = diffstereo.DiffusionGraph(left, right, 256, 1.2, dtype="f32")   # 8 planes: 740 MB
    assert g.fused2_info()["available"]
    g.iterate(4)
    assert g.fused2_info()["steps"] == 2
    g.close()
