"""One of eight row bands of the 4K x 256 workload iterated by the library's band protocol
(ds_band_iterate) over its own RCCL transport, with this rank as both of its neighbours
(world size 1, ncclSend / ncclRecv to self).  A one-GPU estimate of what a rank of an 8-GPU job
spends per iteration; the transfers themselves are not representative of xGMI.
env: BANDS (default 8), BAND (default 3), WORKLOAD dims H W D, DTYPE."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import bands, diffstereo, synth  # noqa: E402

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
H, W, D = (int(os.environ.get(k, v)) for k, v in (("H", 2160), ("W", 3840), ("D", 256)))
n_bands, which = int(os.environ.get("BANDS", 8)), int(os.environ.get("BAND", 3))
dtype = os.environ.get("DTYPE", "f32")
left, right, _ = synth.synthetic_pair(H, W, D)
rb, re = bands.partition_rows(H, n_bands)[which]
g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=rb, row_end=re, device=0)
comm = bands.RcclComm(0, 1, 0)
runner = bands.Bands(g, comm, 0, 1, up=0 if rb > 0 else -1, down=0 if re < H else -1)
runner.iterate(3)
n = int(os.environ.get("N", 50))
g.profile_begin(12 * n + 8)
t0 = time.perf_counter()
runner.iterate(n)
wall = time.perf_counter() - t0
g.profile_end()
print("band %d of %d (rows %d-%d), %s: %.3f ms per iteration over RCCL loopback" % (which, n_bands, rb, re, dtype, wall / n * 1e3))
print("  two-iteration steps:", g.fused2_info())
for kind, name in ((0, "boundary half-sweeps"), (1, "one-iteration launches"), (2, "two-iteration launches")):
    k, ms = g.profile_kind(kind)
    print("  %-22s %4d launches, %.3f ms per iteration" % (name, k, ms / n))
g.close()
comm.close()
