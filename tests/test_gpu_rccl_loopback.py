"""The RCCL transport of the halo exchange on ONE GPU: a band whose two neighbours are the
rank itself (world size 1, send/recv to self).  The image that results is not meaningful --
the band's first row comes back as its own upper halo -- but it is deterministic, so the
same loop driven by plain device copies must give the same bits.  What this pins is what
the multi-GPU benchmark relies on and gloo rehearsals cannot show: library-owned device
memory handed to torch.distributed as tensors, and the ordering between the library's
stream and RCCL's (boundary rows swept before the send, halos landed before the next
colour's sweep)."""
import ctypes as C
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


def _worker(rank, port, dtype, D, out):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)  # a stuck rank ends itself
    import torch
    import torch.distributed as dist
    load_package()
    from stereo_vision_amd import bands, diffstereo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        H, W, iters = 40, 96, 3
        rb, re = 13, 29
        left, right = random_pair(H, W, seed=12)
        phi0 = random_phi(H, W, D, dtype, seed=6)
        a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re)
        b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re)
        for g in (a, b):
            g.set_phi(phi0[rb - 1:re + 1], rb - 1, re + 1)
        runner = bands.BandRunner(bands.HipBandEngine(a), rank=0, world=1)
        runner.up = runner.down = 0   # both neighbours are this rank
        runner.world = 2              # take the multi-rank branch
        runner.iterate(iters)
        torch.cuda.synchronize()

        hip = C.CDLL("libamdhip64.so")
        for _ in range(iters):
            for colour in (0, 1):
                b.sweep(colour)
                b.synchronize()
                for side in (0, 1):
                    sp, n = b.halo(colour, side, 0)
                    dp, m = b.halo(colour, side, 1)
                    assert n == m
                    assert hip.hipMemcpy(C.c_void_p(dp), C.c_void_p(sp), C.c_size_t(n), 3) == 0
        # own rows in full; of the halo rows only the slot that travels (the others are never
        # read, and differ between the library's two sets of colour-1 planes)
        pa, pb = a.get_phi(rb - 1, re + 1), b.get_phi(rb - 1, re + 1)
        same = (np.array_equal(pa[1:-1], pb[1:-1]) and np.array_equal(pa[0, :, 3], pb[0, :, 3])
                and np.array_equal(pa[-1, :, 1], pb[-1, :, 1]))
        moved = not np.array_equal(a.get_phi(rb - 1, rb), phi0[rb - 1:rb].astype(a.get_phi(rb, rb + 1).dtype))
        fused = a.fused_rows()[1] > a.fused_rows()[0]
        out[0] = (bool(same), bool(moved), a.checksum() == b.checksum(), bool(fused))
        a.close()
        b.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,D", [("f32", 64), ("f64", 64), ("f32", 200), ("f64", 130)])
def test_rccl_self_exchange_matches_device_copies(dtype, D):
    """The band's interior in the fused iteration kernel (4, 1 and 1 pixels per wavefront for
    D = 64, 200, 130), boundary rows by half-sweeps, both exchanges over RCCL -- against plain
    half-sweeps with device copies."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), dtype, D, out), nprocs=1, join=True)
    assert out[0] == (True, True, True, True), out[0]
