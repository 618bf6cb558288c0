"""Row-band data parallelism for the diffusion sweep: one process (rank) per GPU.

The reference is single-process (diffusion_stereo/src/diffusion_graph.rs:264-293 is one
raster loop); under the red-black schedule a pixel update reads only the four opposite
colour neighbours, so the image splits into horizontal bands that exchange ONE row of
messages with each neighbouring band after each colour's half-sweep:

  after colour c:  first band row's slot-1 (upward) messages  -> band above (its bottom halo)
                   last band row's slot-3 (downward) messages -> band below (its top halo)

That point-to-point exchange (torch.distributed isend/irecv: RCCL over xGMI on GPUs,
gloo on CPU) is the only communication; the energy is a sum of per-band energies
(all_reduce).  The boundary rows are swept first so that the exchange overlaps the
sweep of the interior rows.  The result is bit-identical to the single-band result:
each pixel sees exactly the same operands.

The product path is `Bands`: the protocol runs inside the library (ds_band_iterate and its
siblings in include/diffstereo.h) over the library's own RCCL transport (`RcclComm`) or over a
transport supplied from here (`GlooTransport`: through host memory, for ranks that share one
GPU and for machines where RCCL point-to-point is unavailable).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import diffstereo


def partition_rows(H, n_bands):
    """Contiguous near-equal bands [(row_begin, row_end)]; every band gets >= 1 row."""
    if n_bands < 1 or n_bands > H:
        raise ValueError("need 1 <= n_bands <= H (got %d bands for %d rows)" % (n_bands, H))
    base, extra = divmod(H, n_bands)
    out, r = [], 0
    for b in range(n_bands):
        k = base + (1 if b < extra else 0)
        out.append((r, r + k))
        r += k
    return out


def neighbours(rank, world):
    """(up_rank, down_rank) of the rank-th of `world` bands; -1 at the image's edges."""
    return (rank - 1 if rank > 0 else -1), (rank + 1 if rank + 1 < world else -1)


class RcclComm:
    """The library's RCCL communicator (ds_comm_*).  Rank 0 makes the unique id; it reaches the
    other ranks through `group`, any torch.distributed group of the same ranks (gloo will do:
    it only carries these 128 bytes)."""

    def __init__(self, rank, world, device, group=None):
        lib = diffstereo.lib()
        ident = (C.c_uint8 * diffstereo.DS_COMM_ID_BYTES)()
        if rank == 0:
            rc = lib.ds_comm_unique_id(ident)
            if rc != 0:
                raise diffstereo.DiffStereoError(rc, (lib.ds_comm_last_error(None) or b"").decode())
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (C.c_uint8 * diffstereo.DS_COMM_ID_BYTES).from_buffer_copy(box[0])
        self._c = C.c_void_p()
        rc = lib.ds_comm_create(ident, int(rank), int(world), int(device), C.byref(self._c))
        if rc != 0:
            raise diffstereo.DiffStereoError(rc, (lib.ds_comm_last_error(None) or b"").decode())
        self.rank, self.world = rank, world
        self.name = "RCCL send/recv inside the library, 1 row per neighbour per colour"

    def attach(self, graph, up, down):
        graph.band_attach_comm(self._c, up, down)

    def close(self):
        if self._c:
            diffstereo.lib().ds_comm_destroy(self._c)
            self._c = C.c_void_p()


class GlooTransport:
    """A ds_transport whose transfers go through host memory and a torch.distributed group
    (gloo).  It blocks the host, which satisfies the stream ordering the library asks for.
    device=False treats the pointers as host memory (the transport itself is then testable
    without a GPU)."""

    _NP = {diffstereo.DS_REDUCE_SUM_F64: (torch.float64, dist.ReduceOp.SUM),
           diffstereo.DS_REDUCE_MAX_F64: (torch.float64, dist.ReduceOp.MAX),
           diffstereo.DS_REDUCE_MAX_I32: (torch.int32, dist.ReduceOp.MAX),
           diffstereo.DS_REDUCE_MIN_I32: (torch.int32, dist.ReduceOp.MIN)}

    def __init__(self, group=None, device=True):
        self.group, self.device, self.error = group, device, None
        self.name = "gloo through host memory"
        if device:
            self._hip = diffstereo.hip_runtime()
        self._keep = (diffstereo.EXCHANGE_FN(self._exchange), diffstereo.ALL_REDUCE_FN(self._all_reduce))
        self.struct = diffstereo.TransportStruct(None, self._keep[0], self._keep[1], diffstereo.LAST_ERROR_FN())

    def _sync(self, stream):
        if self.device and self._hip.hipStreamSynchronize(stream) != 0:
            raise RuntimeError("hipStreamSynchronize failed")

    def _copy(self, dst, src, n):
        if self.device:
            if self._hip.hipMemcpy(dst, src, n, 4) != 0:  # hipMemcpyDefault
                raise RuntimeError("hipMemcpy failed")
        else:
            C.memmove(dst, src, n)

    def _exchange(self, ctx, n, peers, send, recv, nbytes, stream):
        try:
            self._sync(stream)
            ops, landed = [], []
            for k in range(n):
                out = torch.empty(nbytes[k], dtype=torch.uint8)
                self._copy(out.data_ptr(), send[k], nbytes[k])
                got = torch.empty(nbytes[k], dtype=torch.uint8)
                ops.append(dist.P2POp(dist.isend, out, self._global(peers[k]), self.group))
                ops.append(dist.P2POp(dist.irecv, got, self._global(peers[k]), self.group))
                landed.append((recv[k], got, nbytes[k]))
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for dst, got, size in landed:
                self._copy(dst, got.data_ptr(), size)
            return 0
        except Exception as exc:  # noqa: BLE001  (must not unwind through the C caller)
            self.error = exc
            return 1

    def _all_reduce(self, ctx, dev, count, kind, stream):
        try:
            dtype, op = self._NP[kind]
            t = torch.empty(count, dtype=dtype)
            self._sync(stream)
            self._copy(t.data_ptr(), dev, t.numel() * t.element_size())
            dist.all_reduce(t, op=op, group=self.group)
            self._copy(dev, t.data_ptr(), t.numel() * t.element_size())
            return 0
        except Exception as exc:  # noqa: BLE001
            self.error = exc
            return 1

    def _global(self, rank_in_group):
        return dist.get_global_rank(self.group, rank_in_group) if self.group is not None else rank_in_group

    def attach(self, graph, up, down):
        graph.band_attach(self.struct, up, down)

    def close(self):
        pass


class Bands:
    """This rank's band, iterated by the library (ds_band_*): the caller only says "iterate"."""

    def __init__(self, graph, transport, rank, world, up=None, down=None):
        self.g, self.transport, self.rank, self.world = graph, transport, rank, world
        nb = neighbours(rank, world)
        self.up, self.down = (nb[0] if up is None else up), (nb[1] if down is None else down)
        transport.attach(graph, self.up, self.down)

    def _guard(self, call, *args, **kw):
        try:
            return call(*args, **kw)
        except diffstereo.DiffStereoError as exc:
            cause = getattr(self.transport, "error", None)
            if cause is not None:
                raise exc from cause
            raise

    def exchange_all(self):
        self._guard(self.g.band_exchange)

    def iterate(self, n_iters):
        self._guard(self.g.band_iterate, n_iters)

    def energy(self):
        return self._guard(self.g.band_energy)

    def crossing_out(self, epsilon, max_passes=1 << 30, with_mask=False):
        return self._guard(self.g.band_crossing_out, epsilon, max_passes, with_mask)
