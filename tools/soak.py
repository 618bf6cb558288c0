"""Developer tool: longer runs than the tests make, looking for rare ordering faults between the
library's streams.  (1) 4K x 256, many iterations in calls of odd and even length, two iterations
per launch against one: checksum, energy and labels must be equal.  (2) two ranks on one GPU over
the gloo transport on a 500 x 1930 x 256 image against the unpartitioned run (the band protocol
tests' worker at a size where every launch is many blocks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
import numpy as np  # noqa: E402
from stereo_vision_amd import diffstereo, synth  # noqa: E402


def whole_image():
    H, W, D = 2160, 3840, 256
    left, right, _ = synth.synthetic_pair(H, W, D)
    a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    os.environ["DS_FUSED2"] = "off"
    b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    del os.environ["DS_FUSED2"]
    total = 0
    for n in (7, 40, 1, 2, 33, 64, 5, 100):
        a.iterate(n)
        b.iterate(n)
        total += n
        same = a.checksum() == b.checksum() and a.energy() == b.energy()
        print("after %3d iterations: checksum and energy %s" % (total, "equal" if same else "DIFFER"), flush=True)
        assert same
    assert np.array_equal(a.labels(0), b.labels(0))
    print("labels equal; two-iteration steps:", a.fused2_info()["steps"], flush=True)
    a.close()
    b.close()


def two_ranks():
    import torch.multiprocessing as mp
    from tests.test_gpu_band_protocol import _free_port, _gloo_worker
    for world, H, W, D in ((2, 500, 1930, 256), (3, 333, 2500, 128)):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_gloo_worker, args=(world, _free_port(), "f32", D, H, W, out), nprocs=world, join=True)
        print("%d ranks, %d x %d x %d:" % (world, H, W, D), dict(out), flush=True)
        assert all(out[r][:3] == (True, True, True) for r in range(world))


if __name__ == "__main__":
    whole_image()
    two_ranks()
    print("soak ok")
