"""Row-band partition (stereo-vision_amd/bands.py) and the band protocol as tests/band_runner.py
states it, on CPU ranks over gloo with the CPU oracle as the compute engine: every rank's band
must end bit-identical to the unpartitioned run, and the all-reduced energy must equal the
whole image's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from __graft_entry__ import load_package
from oracle.pyoracle import FastOracle
from tests.band_oracle_engine import OracleBandEngine
from tests.band_runner import BandRunner
from tests.util import NumpyCrossingOut, random_pair, random_phi

load_package()
from stereo_vision_amd import bands  # noqa: E402


def test_partition_rows():
    assert bands.partition_rows(10, 1) == [(0, 10)]
    assert bands.partition_rows(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert bands.partition_rows(2160, 8)[3] == (810, 1080)
    p = bands.partition_rows(7, 7)
    assert p[0] == (0, 1) and p[-1] == (6, 7)
    with pytest.raises(ValueError):
        bands.partition_rows(3, 4)
    for H, n in ((2160, 8), (4320, 8), (11, 4)):
        p = bands.partition_rows(H, n)
        assert p[0][0] == 0 and p[-1][1] == H
        assert all(a[1] == b[0] for a, b in zip(p, p[1:]))
        assert max(e - b for b, e in p) - min(e - b for b, e in p) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, H, W, D, iters, out):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)  # a stuck rank ends itself
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        left, right = random_pair(H, W, seed=21)
        phi0 = random_phi(H, W, D, "f64", seed=4)
        rb, re = bands.partition_rows(H, world)[rank]
        eng = OracleBandEngine(left, right, D, 1.2, rb, re)
        eng.set_rows(phi0, rb, re)         # own rows only: halos must come from the exchange
        runner = BandRunner(eng)
        runner.exchange_all()
        runner.iterate(iters)
        energy = runner.energy()
        ref = FastOracle(left, right, D, 1.2)
        ref.potentials[...] = phi0
        ref.iterate(iters)
        ok = np.array_equal(eng.f.potentials[rb:re], ref.potentials[rb:re])
        e_ref = ref.energy()
        e_ok = abs(energy - e_ref) <= 1e-9 * max(1.0, abs(e_ref))
        out[rank] = (bool(ok), bool(e_ok), not bool(np.isnan(eng.f.potentials[rb:re]).any()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 9), (3, 7), (4, 4)])
def test_bands_over_gloo_match_single_process(world, H):
    W, D, iters = 10, 6, 3
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, H, W, D, iters, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        assert out[rank] == (True, True, True), "rank %d: %s" % (rank, out[rank])


def _co_worker(rank, world, port, H, W, D, iters, eps, out):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)  # a stuck rank ends itself
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        left, right = random_pair(H, W, seed=21)
        rb, re = bands.partition_rows(H, world)[rank]
        eng = OracleBandEngine(left, right, D, 1.2, rb, re)
        eng.set_rows(np.zeros((H, W, 4, D)), rb, re)
        runner = BandRunner(eng)
        runner.exchange_all()
        runner.iterate(iters)
        got = runner.crossing_out(eps, with_mask=True)
        ref = FastOracle(left, right, D, 1.2)
        ref.iterate(iters)
        cost = pyoracle.cost_volume(left, right, D).astype(np.float64).transpose(1, 2, 0)
        whole = NumpyCrossingOut(ref.potentials, cost, 1.2, eps)
        want = whole.run()
        out[rank] = (bool(np.array_equal(got["alive"].astype(bool), want[rb:re])),
                     got["not_empty"] == bool(want.any(axis=2).all()),
                     bool(np.array_equal(got["labels"], whole.labels()[rb:re])), got["passes"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,iters,eps", [(2, 9, 30, 5.0), (3, 7, 30, 20.0), (4, 4, 2, 60.0), (2, 6, 2, 3.0)])
def test_crossing_out_over_bands_matches_whole_image(world, H, iters, eps):
    """BandRunner.crossing_out: mask halo exchange + OR-ed flags give every band exactly its
    rows of the whole image's greatest consistent sub-graph (non-empty and empty outcomes)."""
    W, D = 10, 6
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_co_worker, args=(world, _free_port(), H, W, D, iters, eps, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        assert out[rank][:3] == (True, True, True), "rank %d: %s" % (rank, out[rank])
    assert len({out[r][3] for r in range(world)}) == 1  # every rank ran the same number of passes
