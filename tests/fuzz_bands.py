"""Random shapes through the library's band protocol, several ranks sharing one GPU over the
host-memory transport, against the unpartitioned image, bit for bit (a script, not collected by
pytest: `python3 tests/fuzz_bands.py [seconds per world size] [seed] [large]` on a GPU box; `large`:
bands of 16-150 rows and 300-3900 columns).

Every case draws H, W, D, the precision, the halo mode and a sequence of call lengths; all ranks
draw the same case from the same seed.  Compared per band: phi after the calls, labels, energy,
crossing-out masks at one epsilon."""
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from tests.util import random_pair, random_phi  # noqa: E402


def worker(rank, world, port, budget, seed, large, out):
    import faulthandler
    faulthandler.dump_traceback_later(int(budget) + 150, exit=True, file=sys.stderr)  # a stuck rank ends itself
    os.environ["DS_CACHE_RESIDENT"] = "fused"
    import torch
    import torch.distributed as dist
    load_package()
    from stereo_vision_amd import bands, diffstereo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t0, case, bad, refusals = time.time(), 0, 0, 0
    try:
        while True:
            # rank 0's clock decides, so that every rank runs the same number of cases
            go = torch.tensor([1 if time.time() - t0 < budget else 0])
            dist.broadcast(go, 0)
            if not go.item():
                break
            case += 1
            rng = np.random.default_rng([seed, case])   # a case can be drawn again from its number alone
            H = int(rng.integers(world, 30 * world))
            W = int(rng.choice([rng.integers(2, 60), rng.integers(60, 350), rng.integers(350, 800)]))
            D = int(rng.choice([rng.integers(1, 257), rng.choice([1, 3, 4, 40, 64, 65, 128, 255, 256])]))
            dtype = "f64" if rng.random() < 0.4 else "f32"
            halo = "deep" if rng.random() < 0.4 else "one-row"
            if halo == "deep" and rng.random() < 0.8:   # mostly shapes whose bands can have it
                H, W = int(rng.integers(16 * world, 30 * world)), int(rng.integers(150, 800))
            if large:   # many strips per band, row segments of unequal length; at most 6e7 (pixel, label) pairs
                H, W = int(rng.integers(16 * world, 150 * world)), int(rng.integers(300, 3900))
                while H * W * D > 60_000_000 and (H > 16 * world or W > 300):
                    H, W = max(16 * world, H * 2 // 3), max(300, W * 3 // 4)
            calls = [int(x) for x in rng.integers(0, 8, size=int(rng.integers(1, 4)))]
            eps = float(rng.choice([0.5, 4.0, 40.0]))
            os.environ["DS_BAND_HALO"] = halo
            left, right = random_pair(H, W, seed=case)
            phi0 = random_phi(H, W, D, dtype, seed=case + 1)
            rb, re = bands.partition_rows(H, world)[rank]
            g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re, device=0)
            g.set_phi(phi0[rb:re])
            # deep halo: every band must qualify (16 rows or more, an interior for the two-iteration
            # kernel) or attaching fails -- on every rank, which is checked here
            try:
                runner = bands.Bands(g, bands.GlooTransport(), rank, world)
                refused = 0
            except diffstereo.DiffStereoError:
                refused = 1
            lo, hi = torch.tensor([refused]), torch.tensor([refused])
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if hi.item():
                g.close()
                if lo.item() != hi.item() or halo != "deep":
                    bad += 1
                    if rank == 0:
                        print("world %d case %d: %dx%dx%d %s halo %s: attach refused on some ranks only, or without "
                              "the deep halo" % (world, case, H, W, D, dtype, halo), flush=True)
                else:
                    refusals += 1
                continue
            runner.exchange_all()
            whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
            whole.set_phi(phi0)
            ok = True
            for n in calls:
                runner.iterate(n)
                whole.iterate(n)
                ok = ok and np.array_equal(g.get_phi(), whole.get_phi()[rb:re])
            energy, e_ref = runner.energy(), whole.energy()
            ok = ok and abs(energy - e_ref) <= 1e-9 * max(1.0, abs(e_ref))
            ok = ok and np.array_equal(g.labels(0), whole.labels(0)[rb:re])
            got, want = runner.crossing_out(eps, with_mask=True), whole.crossing_out(eps, with_mask=True)
            ok = ok and np.array_equal(got["alive"], want["alive"][rb:re]) and got["not_empty"] == want["not_empty"]
            steps = g.fused2_info()["steps"]
            g.close()
            whole.close()
            flag = torch.tensor([1 if ok else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not flag.item():
                bad += 1
            if rank == 0 and (case % 10 == 0 or large or not flag.item()):
                print("world %d case %d: %dx%dx%d %s halo %s calls %s (rank 0: %d two-iteration steps) -> %s (%.0f s)"
                      % (world, case, H, W, D, dtype, halo, calls, steps, "ok" if flag.item() else "MISMATCH",
                         time.time() - t0), flush=True)
        out[rank] = (case, bad, refusals)
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    total_bad = 0
    for world in (2, 3, 4):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = mp.Manager().dict()
        mp.spawn(worker, args=(world, port, budget, seed + world, large, out), nprocs=world, join=True)
        print("world %d: %d cases, %d mismatches (%d cases asked for a deep halo their bands cannot have and were "
              "refused on every rank)" % (world, out[0][0], out[0][1], out[0][2]), flush=True)
        total_bad += out[0][1]
    return 1 if total_bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
