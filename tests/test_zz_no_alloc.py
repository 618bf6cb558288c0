"""ds_energy_static: the energy reduction through the library's static device scratch (for
callers that cannot allocate) against ds_energy, through the C ABI.  Needs a GPU.

Bar: the two entry points share their kernels and their order of summation, so the energy and
the residual must be equal bit for bit -- on two calls in a row, the second on a smaller input
(the scratch keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from __graft_entry__ import load_package
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo  # noqa: E402

D2H = 2
hip = None


class DeviceDoubles:
    """Two doubles of device memory owned by the test (the caller's output buffer)."""

    def __init__(self):
        global hip
        hip = diffstereo.hip_runtime()  # the runtime the library is bound to, not a second copy
        self.ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.ptr), 16) == 0
        assert hip.hipMemset(self.ptr, 0xFF, 16) == 0

    def read(self):
        out = (C.c_double * 2)()
        assert hip.hipMemcpy(out, self.ptr, 16, D2H) == 0
        return out[0], out[1]

    def free(self):
        hip.hipFree(self.ptr)


def _graph(H, W, D, dtype, seed, iters):
    left, right = random_pair(H, W, seed=seed)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    g.set_phi(random_phi(H, W, D, dtype, seed=seed + 1))
    g.iterate(iters)
    return g


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_energy_static_equals_energy_twice_second_smaller(dtype):
    out = DeviceDoubles()
    big = _graph(61, 530, 200, dtype, seed=11, iters=2)
    small = _graph(7, 45, 24, dtype, seed=12, iters=0)  # random phi: a large residual
    for g in (big, small, big):  # big again: the smaller call's leftovers must not matter either
        want = g.energy(with_residual=True)
        g.energy_static(out.ptr.value)
        g.synchronize()
        got = out.read()
        assert got == want, (got, want)
    big.close()
    small.close()
    out.free()


def test_energy_static_rejects_what_it_cannot_take(monkeypatch):
    out = DeviceDoubles()
    g = _graph(5, 9, 4, "f32", seed=3, iters=1)
    host = (C.c_double * 2)()
    with pytest.raises(diffstereo.DiffStereoError) as err:  # pageable host memory: a kernel writes the result
        g.energy_static(C.addressof(host))
    assert err.value.code == -1
    with pytest.raises(diffstereo.DiffStereoError):
        g.energy_static(0)
    g.energy_static(out.ptr.value)  # still usable afterwards
    g.synchronize()
    assert out.read() == g.energy(with_residual=True)
    g.close()
    # a handle beyond the scratch's 2 097 152 partial sums: 2 x 256 blocks per row x 4200 rows
    monkeypatch.setenv("DS_PLACEMENT", "off")
    rng = np.random.default_rng(0)
    H, W = 4200, 16384
    left = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    wide = diffstereo.DiffusionGraph(left, left, 1, 1.2, dtype="f32")
    with pytest.raises(diffstereo.DiffStereoError) as err:
        wide.energy_static(out.ptr.value)
    assert err.value.code == -1 and "2097152" in str(err.value)
    assert wide.energy() == 0.0  # the allocating entry point takes it (identical images, D = 1)
    wide.close()
    out.free()
