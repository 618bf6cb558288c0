"""Rehearsal of the multi-rank path on ONE GPU: two or three ranks (processes) share
cuda:0, each owns a row band through the C ABI, and the halo rows travel rank to rank
over gloo (staged through the host, since RCCL refuses two ranks on one device).  It
exercises exactly the code the N-GPU benchmark runs (bands.BandRunner + HipBandEngine,
ds_sweep on row ranges, ds_halo device views, torch streams) except the RCCL transport."""
import os
import socket

import numpy as np
import pytest

from __graft_entry__ import load_package
from tests.util import random_pair, random_phi

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, dtype, D, H, W, out):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)  # a stuck rank ends itself
    import torch
    import torch.distributed as dist
    load_package()
    from stereo_vision_amd import bands, diffstereo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        iters = 3
        left, right = random_pair(H, W, seed=8)
        phi0 = random_phi(H, W, D, dtype, seed=2)
        rb, re = bands.partition_rows(H, world)[rank]
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re, device=0)
        g.set_phi(phi0[rb:re])
        runner = bands.BandRunner(bands.HipBandEngine(g, stage_through_host=True))
        runner.exchange_all()
        runner.iterate(iters)
        energy = runner.energy()
        whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        whole.set_phi(phi0)
        whole.iterate(iters)
        same = np.array_equal(g.get_phi(), whole.get_phi()[rb:re])
        e_ref = whole.energy()
        e_ok = abs(energy - e_ref) <= 1e-9 * max(1.0, abs(e_ref))
        if not (same and e_ok):
            print("rank", rank, "phi same:", same, "energy", energy, "ref", e_ref, flush=True)
        # the reference's stopping rule over the bands: crossing-out with exchanged mask halos
        co_ok = True
        for eps in (4.0, 40.0):
            got = runner.crossing_out(eps, with_mask=True)
            want = whole.crossing_out(eps, with_mask=True)
            co_ok = co_ok and np.array_equal(got["alive"], want["alive"][rb:re]) \
                and np.array_equal(got["labels"], want["labels"][rb:re]) and got["not_empty"] == want["not_empty"]
        if not co_ok:
            print("rank", rank, "crossing-out differs", flush=True)
        out[rank] = (bool(same), bool(e_ok and co_ok))
        g.close()
        whole.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype,D,H,W", [(2, "f32", 24, 13, 45), (3, "f64", 24, 13, 45), (2, "f32", 256, 29, 75),
                                               (3, "f64", 140, 29, 75), (2, "f64", 64, 17, 130), (3, "f32", 100, 40, 33),
                                               (3, "f32", 64, 23, 70), (4, "f64", 40, 33, 52)])
def test_two_ranks_one_gpu(world, dtype, D, H, W):
    """Bands of 4-7 rows (H = 13) fall back to half-sweeps per colour; bands of 9-15 rows
    (H = 29) iterate with the fused kernel in their interior (bands.BandRunner._fused_iteration).
    H = 23 over three ranks gives bands of 8, 8 and 7 rows: two ranks on the fused protocol next to
    one on half-sweeps (the messages that travel are the same); H = 33 over four: 9, 8, 8, 8."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), dtype, D, H, W, out), nprocs=world, join=True)
    for rank in range(world):
        assert out[rank] == (True, True), "rank %d: %s" % (rank, out[rank])
