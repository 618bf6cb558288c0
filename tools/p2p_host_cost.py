"""Host-side cost of the halo exchange calls: torch.distributed batch_isend_irecv on the
nccl (RCCL) backend with this rank as its own peer (world size 1), next to the cost of
the sweep launches of one band.  A one-GPU estimate of the per-iteration host time of a
multi-GPU rank; the transfers themselves are not representative."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import bands, diffstereo, synth  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
H, W, D = 2160, 3840, 256
left, right, _ = synth.synthetic_pair(H, W, D)
rb, re = bands.partition_rows(H, 8)[3]
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32", row_begin=rb, row_end=re)
# DUMMY_STREAMS=n: n more streams that have run work, created before the engine's two, to see
# whether the engine's streams still overlap when streams outnumber the hardware queues
dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("DUMMY_STREAMS", "0")))]
for st in dummies:
    with torch.cuda.stream(st):
        torch.zeros(8, device="cuda").add_(1)
torch.cuda.synchronize()
eng = bands.HipBandEngine(g)
runner = bands.BandRunner(eng, rank=0, world=1)
runner.up = runner.down = 0      # loop the exchange back to this rank
runner.world = 2                 # take the multi-rank branch of half_sweep
try:
    runner.iterate(2)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    runner.iterate(n)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("band of %d rows: host returns after %.3f ms per iteration, GPU done after %.3f ms per iteration"
          % (re - rb, t_host / n * 1e3, t_all / n * 1e3), flush=True)
except Exception as exc:  # noqa: BLE001
    print("loopback exchange not possible here: %r" % (exc,), flush=True)
g.close()
dist.destroy_process_group()
