"""The band protocol inside the library (ds_band_attach / ds_band_exchange / ds_band_iterate /
ds_band_energy / ds_band_crossing_out) on ONE GPU, in the two ways one GPU allows:

* several ranks (processes) share cuda:0, each owns a band, the halo rows travel through the
  host-memory transport (bands.GlooTransport) -- everything but RCCL: the library's boundary
  stream and events, fused and colour-by-colour bands side by side, the reductions;
* one rank whose two neighbours are itself over the library's own RCCL transport (ds_comm_*,
  ncclSend / ncclRecv to self inside a group): the image that results is not meaningful, but
  it is deterministic, so the same loop with plain device copies must give the same bits.

Bar: bit-identical to the unpartitioned image (phi, labels, masks); energy 1e-9."""
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


def _gloo_worker(rank, world, port, dtype, D, H, W, out, halo="one-row"):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)  # a stuck rank ends itself
    if halo == "deep":
        os.environ["DS_BAND_HALO"] = "deep"
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
        runner = bands.Bands(g, bands.GlooTransport(), rank, world)
        runner.exchange_all()
        runner.iterate(iters)
        runner.iterate(0)
        runner.iterate(1)
        runner.iterate(6)   # bands with an interior: three two-iteration steps, the last one storing the first colour
        energy = runner.energy()
        steps = g.fused2_info()["steps"]
        whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        whole.set_phi(phi0)
        whole.iterate(iters + 7)
        same = np.array_equal(g.get_phi(), whole.get_phi()[rb:re])
        e_ref = whole.energy()
        e_ok = abs(energy - e_ref) <= 1e-9 * max(1.0, abs(e_ref))
        co_ok = True
        for eps in (4.0, 40.0):
            got = runner.crossing_out(eps, with_mask=True)
            want = whole.crossing_out(eps, with_mask=True)
            co_ok = co_ok and np.array_equal(got["alive"], want["alive"][rb:re]) \
                and np.array_equal(got["labels"], want["labels"][rb:re]) and got["not_empty"] == want["not_empty"]
        if not (same and e_ok and co_ok):
            print("rank", rank, "phi same:", same, "energy", energy, "ref", e_ref, "crossing-out", co_ok, flush=True)
        labels_ok = np.array_equal(g.labels(0), whole.labels(0)[rb:re]) and np.array_equal(g.get_cost(), whole.get_cost()[:, rb:re])
        out[rank] = (bool(same), bool(e_ok), bool(co_ok and labels_ok), steps)
        g.close()
        whole.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype,D,H,W", [(2, "f32", 24, 13, 45), (3, "f64", 24, 13, 45), (2, "f32", 256, 29, 75),
                                               (3, "f64", 140, 29, 75), (2, "f64", 64, 17, 130), (3, "f32", 100, 40, 33),
                                               (3, "f32", 64, 23, 70), (4, "f64", 40, 33, 52),
                                               (3, "f32", 64, 60, 330), (2, "f32", 128, 41, 420), (2, "f32", 256, 36, 330),
                                               (3, "f32", 64, 50, 330),
                                               # D that is not a multiple of 4, D = 40 in f64: bands step by two as well
                                               (2, "f32", 41, 40, 500), (2, "f64", 130, 36, 200), (3, "f64", 40, 60, 300)])
def test_ranks_sharing_one_gpu(world, dtype, D, H, W):
    """H = 13: bands of 4-7 rows sweep colour by colour; H = 29 and more: bands of 8 rows or more
    run the fused kernel in their interior; H = 23 over three ranks (8, 8, 7 rows) and H = 33 over
    four (9, 8, 8, 8) put both kinds next to each other.  The four wide f32 cases have bands with an
    interior for the two-iteration kernel (D = 64 / 128 / 256: four, two and one pixel per wavefront)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gloo_worker, args=(world, _free_port(), dtype, D, H, W, out), nprocs=world, join=True)
    for rank in range(world):
        assert out[rank][:3] == (True, True, True), "rank %d: %s" % (rank, out[rank])
    # f32 bands of 14 rows or more, wide enough for a strip of the two-iteration kernel, step by two
    # (1 + 0 + 0 + 3 steps over the calls of 3 + 0 + 1 + 6 iterations); H = 50 over three ranks (17, 17, 16 rows)
    # has every rank on it, next to bands that are not in the other cases
    steps = [out[rank][3] for rank in range(world)]
    assert all(k in (0, 4) for k in steps)
    if (W >= 330 and dtype == "f32") or (dtype, D, H, W) in (("f32", 41, 40, 500), ("f64", 130, 36, 200), ("f64", 40, 60, 300)):
        assert steps == [4] * world, steps   # bands of 14 rows or more with room for a strip: any D, both precisions
    elif H < 28 * world // 2 and W < 100:
        assert steps == [0] * world, steps   # bands under 14 rows step by one
    # the narrow f32 cases with D = 64 / 256: whichever bands have eight interior rows and room for
    # one strip step by two beside neighbours that do not (29 x 75 x 256: the 15-row band only)


@pytest.mark.parametrize("world,D,H,W,dtype", [(3, 64, 60, 330, "f32"), (2, 128, 41, 420, "f32"), (2, 256, 36, 330, "f32"),
                                               (3, 64, 50, 330, "f32"), (2, 40, 40, 500, "f32"), (4, 256, 70, 330, "f32"),
                                               (2, 41, 40, 500, "f32"), (2, 203, 44, 330, "f32"),
                                               (2, 256, 36, 200, "f64"), (3, 40, 60, 300, "f64"), (2, 130, 40, 200, "f64")])
def test_deep_halo_bands(world, D, H, W, dtype):
    """DS_BAND_HALO=deep: bands of 16 rows or more keep four rows of each neighbour and exchange once
    per two-iteration step.  Same bar: every band's potentials, energy, labels, cost and crossing-out
    masks equal to the unpartitioned image's; every band has stepped by two (1 + 0 + 0 + 3 steps)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gloo_worker, args=(world, _free_port(), dtype, D, H, W, out, "deep"), nprocs=world, join=True)
    for rank in range(world):
        assert out[rank][:3] == (True, True, True), "rank %d: %s" % (rank, out[rank])
        assert out[rank][3] == 4, "rank %d ran %d two-iteration steps" % (rank, out[rank][3])


def _mixed_worker(rank, world, port, out):
    """Rank 0 asks for the deep halo, rank 1 does not: attaching must fail on both, not hang."""
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(120, exit=True, file=sys.stderr)
    if rank == 0:
        os.environ["DS_BAND_HALO"] = "deep"
    import torch
    import torch.distributed as dist
    load_package()
    from stereo_vision_amd import bands, diffstereo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        left, right = random_pair(40, 400, seed=8)
        rb, re = bands.partition_rows(40, world)[rank]
        g = diffstereo.DiffusionGraph(left, right, 64, 1.2, dtype="f32", row_begin=rb, row_end=re, device=0)
        try:
            bands.Bands(g, bands.GlooTransport(), rank, world)
            out[rank] = "attached"
        except diffstereo.DiffStereoError as exc:
            out[rank] = "refused" if exc.code == -3 and "DS_BAND_HALO" in str(exc) else repr(exc)
        g.close()
    finally:
        dist.destroy_process_group()


def test_bands_must_agree_on_the_halo_depth():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_mixed_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == "refused" and out[1] == "refused", dict(out)


def _rccl_worker(rank, dtype, D, rows, W, out):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    load_package()
    from stereo_vision_amd import bands, diffstereo
    H, iters = 40, 5 if W >= 330 else 3   # the wide band steps by two iterations twice, then one
    rb, re = rows
    left, right = random_pair(H, W, seed=12)
    phi0 = random_phi(H, W, D, dtype, seed=6)
    a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re, device=0)
    b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re, device=0)
    for g in (a, b):
        g.set_phi(phi0[rb - 1:re + 1], rb - 1, re + 1)
    comm = bands.RcclComm(0, 1, 0)
    runner = bands.Bands(a, comm, 0, 1, up=0, down=0)  # both neighbours are this rank
    runner.iterate(iters)
    e_a = runner.energy()                               # all-reduce over one rank
    hip = diffstereo.hip_runtime()
    for _ in range(iters):
        for colour in (0, 1):
            b.sweep(colour)
            b.synchronize()
            for side in (0, 1):
                sp, n = b.halo(colour, side, 0)
                dp, m = b.halo(colour, side, 1)
                assert n == m
                assert hip.hipMemcpy(C.c_void_p(dp), C.c_void_p(sp), C.c_size_t(n), 3) == 0
    # own rows in full; of the halo rows only the slot that travels (the others are never read,
    # and differ between the library's two sets of colour-1 planes)
    pa, pb = a.get_phi(rb - 1, re + 1), b.get_phi(rb - 1, re + 1)
    same = (np.array_equal(pa[1:-1], pb[1:-1]) and np.array_equal(pa[0, :, 3], pb[0, :, 3])
            and np.array_equal(pa[-1, :, 1], pb[-1, :, 1]))
    moved = not np.array_equal(pa[0], phi0[rb - 1].astype(pa.dtype))
    # bands of 14 rows or more with an interior for the two-iteration kernel (f32, D a multiple of 4,
    # wide enough for a strip) step by two
    by_two = a.fused2_info()["available"] and re - rb >= 14
    # the two-iteration kernel needs room for a strip of its slots: 64-label slots (D = 64) do not fit
    # 96 columns, the single-pixel slots of D = 130 / 200 do; bands under 14 rows step by one
    assert by_two == ((W >= 330 or D in (130, 200)) and re - rb >= 14), (by_two, dtype, D, rows, W)
    stepped = a.fused2_info()["steps"] == (iters // 2 if by_two else 0)
    out[0] = (bool(same), bool(moved), a.checksum() == b.checksum(), e_a == b.energy() and stepped)
    a.close()
    b.close()
    comm.close()


@pytest.mark.parametrize("dtype,D,rows,W", [("f32", 64, (13, 29), 96), ("f64", 64, (13, 29), 96), ("f32", 200, (13, 29), 96),
                                            ("f64", 130, (13, 29), 96), ("f32", 256, (20, 26), 96),
                                            ("f32", 64, (9, 31), 330)])
def test_rccl_self_exchange_matches_device_copies(dtype, D, rows, W):
    """16-row bands: interior in the fused kernel, boundary rows by half-sweeps on the boundary
    stream, both exchanges per iteration over RCCL; the 6-row band sweeps colour by colour; the
    330-column band steps by two iterations (four exchanges per step)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(dtype, D, rows, W, out), nprocs=1, join=True)
    assert out[0] == (True, True, True, True), out[0]


def _wide_worker(rank, world, port, out, halo="one-row"):
    """One of four bands of 270 rows of a 3840-wide image with 256 disparities, f32: the band
    geometry of the 4K benchmark on eight GPUs (its widths, its strips, its unequal row segments)."""
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(280, exit=True, file=sys.stderr)
    if halo == "deep":
        os.environ["DS_BAND_HALO"] = "deep"
    import torch
    import torch.distributed as dist
    load_package()
    from stereo_vision_amd import bands, diffstereo, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        H, W, D, iters = 270 * world, 3840, 256, 7
        left, right, _ = synth.synthetic_pair(H, W, D)
        rb, re = bands.partition_rows(H, world)[rank]
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32", row_begin=rb, row_end=re, device=0)
        runner = bands.Bands(g, bands.GlooTransport(), rank, world)
        runner.exchange_all()
        runner.iterate(iters)                       # three two-iteration steps, then one iteration
        steps = g.fused2_info()["steps"]
        energy = runner.energy()                    # summed over the ranks
        checksum = g.checksum()
        edge = 6                                    # the band's first and last rows, where the three queues meet
        top, bottom = g.get_phi(rb, rb + edge), g.get_phi(re - edge, re)
        g.close()
        dist.barrier()                              # every band is gone before the whole images come
        whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
        whole.iterate(iters)
        same = np.array_equal(top, whole.get_phi(rb, rb + edge)) and np.array_equal(bottom, whole.get_phi(re - edge, re))
        e_ref = whole.energy()
        out[rank] = (bool(same), abs(energy - e_ref) <= 1e-9 * max(1.0, abs(e_ref)), steps, checksum, whole.checksum())
        whole.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("halo", ["one-row", "deep"])
def test_bands_of_the_4k_benchmark_geometry(halo):
    """3840 x 1080 x 256, f32, four ranks sharing the GPU, seven iterations through ds_band_iterate
    against ds_iterate on the whole image: every band's first and last six rows bit for bit, the
    bands' checksums adding up to the image's (so every other row agrees too), the energy.  Once
    with one halo row and four exchanges per two-iteration step, once with the deep halo and one."""
    import torch.multiprocessing as mp
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_wide_worker, args=(world, _free_port(), out, halo), nprocs=world, join=True)
    for rank in range(world):
        assert out[rank][:2] == (True, True), "rank %d: %s" % (rank, out[rank])
        assert out[rank][2] == 3, "rank %d ran %d two-iteration steps" % (rank, out[rank][2])
    assert sum(out[rank][3] for rank in range(world)) % (1 << 64) == out[0][4]
