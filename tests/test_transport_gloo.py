"""The host-memory transport a band handle can be given instead of RCCL (bands.GlooTransport),
called through the C function-pointer table the library would call it through: two and three
ranks over gloo on the CPU, with host buffers standing in for device memory.  Also the
neighbour arithmetic of the band partition."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from __graft_entry__ import load_package


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import torch.distributed as dist
    load_package()
    from stereo_vision_amd import bands, diffstereo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = bands.GlooTransport(device=False)
        up, down = bands.neighbours(rank, world)
        peers = [p for p in (up, down) if p >= 0]
        # two buffers per neighbour (as the mask exchange sends two colours to one peer)
        n = 2 * len(peers)
        size = 1000 + 17 * rank
        sizes = [1000 + 17 * min(rank, p) for p in peers for _ in range(2)]
        send = [np.full(sz, 10 * rank + k, dtype=np.uint8) for k, sz in enumerate(sizes)]
        recv = [np.zeros(sz, dtype=np.uint8) for sz in sizes]
        arr_peers = (C.c_int * n)(*[p for p in peers for _ in range(2)])
        arr_send = (C.c_void_p * n)(*[a.ctypes.data for a in send])
        arr_recv = (C.c_void_p * n)(*[a.ctypes.data for a in recv])
        arr_bytes = (C.c_size_t * n)(*sizes)
        rc = t.struct.exchange(None, n, arr_peers, arr_send, arr_recv, arr_bytes, None)
        ok = rc == 0 and t.error is None
        k = 0
        for p in peers:
            # what the peer sent me: its buffers towards me, in its own order
            p_peers = [q for q in bands.neighbours(p, world) if q >= 0]
            base = 2 * p_peers.index(rank)
            for c in range(2):
                ok = ok and bool((recv[k] == 10 * p + base + c).all())
                k += 1
        vals = np.array([1.5 * (rank + 1), -2.0], dtype=np.float64)
        ok = ok and t.struct.all_reduce(None, vals.ctypes.data, 2, diffstereo.DS_REDUCE_SUM_F64, None) == 0
        ok = ok and vals[0] == 1.5 * world * (world + 1) / 2 and vals[1] == -2.0 * world
        flags = np.array([rank, 7 - rank], dtype=np.int32)
        ok = ok and t.struct.all_reduce(None, flags.ctypes.data, 2, diffstereo.DS_REDUCE_MAX_I32, None) == 0
        ok = ok and flags.tolist() == [world - 1, 7]
        ok = ok and t.struct.all_reduce(None, flags.ctypes.data, 1, diffstereo.DS_REDUCE_MIN_I32, None) == 0
        ok = ok and flags[0] == world - 1
        ok = ok and t.struct.all_reduce(None, vals.ctypes.data, 1, 99, None) == 1 and t.error is not None
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()
    del size


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_transport_through_the_c_table(world):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert [out[r] for r in range(world)] == [True] * world


def test_neighbours():
    load_package()
    from stereo_vision_amd import bands
    assert bands.neighbours(0, 1) == (-1, -1)
    assert [bands.neighbours(r, 3) for r in range(3)] == [(-1, 1), (0, 2), (1, -1)]
