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

`BandRunner` states the same protocol in engine-agnostic Python: the engine supplies the sweep
and the halo buffers.  It is how the protocol itself is checked without a GPU (the tests run
it over a CPU engine with 2-4 gloo ranks); the product does not use it.
"""
import contextlib
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


class BandRunner:
    """Runs checkerboard iterations on this rank's band and keeps the halos current."""

    def __init__(self, engine, rank=None, world=None, group=None):
        self.e = engine
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.up = self.rank - 1 if self.rank > 0 else None
        self.down = self.rank + 1 if self.rank + 1 < self.world else None

    def _post(self, colour):
        """Queue this colour's halo exchange; returns (requests, receive buffers)."""
        ops, recvs = [], []
        for side, peer in ((0, self.up), (1, self.down)):
            if peer is None:
                continue
            ops.append(dist.P2POp(dist.isend, self.e.send_buffer(colour, side), peer, self.group))
            rb = self.e.recv_buffer(colour, side)
            recvs.append((side, rb))
            ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
        return (dist.batch_isend_irecv(ops) if ops else []), recvs

    def _on_main(self):
        return self.e.on_main() if hasattr(self.e, "on_main") else contextlib.nullcontext()

    def half_sweep(self, colour):
        with self._on_main():
            self._half_sweep(colour)

    def _half_sweep(self, colour):
        e = self.e
        rb, re = e.row_begin, e.row_end
        if self.world == 1:
            e.sweep(colour, rb, re)
            return
        # boundary rows first, so the exchange overlaps the interior rows
        top_end = min(rb + 1, re) if self.up is not None else rb
        bot_begin = max(re - 1, top_end) if self.down is not None else re
        e.sweep(colour, rb, top_end)
        e.sweep(colour, bot_begin, re)
        reqs, recvs = self._post(colour)
        e.sweep(colour, top_end, bot_begin)
        for r in reqs:
            r.wait()
        for side, buf in recvs:
            e.commit_recv(colour, side, buf)

    def iterate(self, n_iters):
        with self._on_main():
            self._iterate(n_iters)

    def _iterate(self, n_iters):
        if (self.world == 1 and hasattr(self.e, "iterate") and self.e.row_begin == 0
                and self.e.row_end == self.e.H):
            self.e.iterate(n_iters)  # one band = the whole image: the library's own loop
            return
        if n_iters > 0 and self._fused_band():
            for _ in range(n_iters):
                self._fused_iteration()
            self.e.finish_iterations()
            return
        for _ in range(n_iters):
            self._half_sweep(0)
            self._half_sweep(1)

    def _fused_band(self):
        """True if the engine can update the band's interior in one kernel per iteration and
        the band is tall enough for that to be the bulk of it."""
        if not hasattr(self.e, "fused_rows"):
            return False
        fb, fe = self.e.fused_rows()
        return fe - fb >= 4 and self.e.row_end - self.e.row_begin >= 8

    def _fused_iteration(self):
        """One iteration of this band: the boundary rows with half-sweeps, everything else in
        the kernel that keeps the first colour on chip.  Same result as two half_sweep()s.

        The boundary chain (first colour's two first / two last rows -> their exchange -> second
        colour's first / last row, written into the planes the iteration is producing -> their
        exchange) and the interior kernel both read only the planes of the previous iteration
        and write disjoint rows, so engines that have a side stream run the chain there, beside
        the interior kernel; the main stream waits for it before the next iteration."""
        e = self.e
        rb, re = e.row_begin, e.row_end
        fb, fe = e.fused_rows()
        both = self.up is not None and self.down is not None and hasattr(e, "sweep_pair")
        beside = e.beside() if hasattr(e, "beside") else contextlib.nullcontext()
        with beside:
            if both:
                e.sweep_pair(0, False, (rb, rb + 2), (re - 2, re))  # one launch for the four rows
            else:
                if self.up is not None:
                    e.sweep(0, rb, rb + 2)
                if self.down is not None:
                    e.sweep(0, re - 2, re)
            reqs, recvs = self._post(0)
            for r in reqs:
                r.wait()
            for side, buf in recvs:
                e.commit_recv(0, side, buf)
            if both:
                e.sweep_pair(1, True, (rb, rb + 1), (re - 1, re))
            else:
                if self.up is not None:
                    e.sweep_next(rb, rb + 1)
                if self.down is not None:
                    e.sweep_next(re - 1, re)
            ops, staged = [], []
            for side, peer in ((0, self.up), (1, self.down)):
                if peer is None:
                    continue
                ops.append(dist.P2POp(dist.isend, e.send_buffer_next(side), peer, self.group))
                buf, view = e.recv_buffer_next(side)
                staged.append((buf, view))
                ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            for r in reqs:
                r.wait()
            for buf, view in staged:
                if buf is not view:
                    view.copy_(buf)
        # main stream: does not wait for any of the above
        e.iterate_rows(fb, fe)  # the planes just written become the current ones
        if hasattr(e, "join"):
            e.join()

    def exchange_all(self):
        """Bring both colours' halos up to date (after set_phi on the bands)."""
        if self.world == 1:
            return
        with self._on_main():
            for colour in (0, 1):
                reqs, recvs = self._post(colour)
                for r in reqs:
                    r.wait()
                for side, buf in recvs:
                    self.e.commit_recv(colour, side, buf)

    def _reduce_flags(self, values, op):
        t = torch.tensor([int(v) for v in values], dtype=torch.int32)
        if self.world > 1:
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=op, group=self.group)
        return [bool(v) for v in t.tolist()]

    def _co_exchange(self):
        if self.world == 1:
            return
        e = self.e
        e.synchronize()
        with self._on_main():
            ops, recvs = [], []
            for colour in (0, 1):
                for side, peer in ((0, self.up), (1, self.down)):
                    if peer is None:
                        continue
                    ops.append(dist.P2POp(dist.isend, e.co_send_buffer(colour, side), peer, self.group))
                    rb = e.co_recv_buffer(colour, side)
                    recvs.append((colour, side, rb))
                    ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
            for r in (dist.batch_isend_irecv(ops) if ops else []):
                r.wait()
            for colour, side, buf in recvs:
                e.co_commit_recv(colour, side, buf)

    def crossing_out(self, epsilon, max_passes=1 << 30, with_mask=False):
        """CrossingOutGraph::initialize_with_epsilon + crossing_out + simple_best_labeling
        (crossing_out_graph.rs:67-183,307-327) over all bands.  Every band relaxes its rows,
        the boundary rows' masks go to the neighbouring bands, and the two flags (a vertex
        died / a pixel has no vertex left) are OR-ed over the ranks.  Returns this band's
        labels (and mask), the global not_empty and the number of passes."""
        e = self.e
        e.co_begin(epsilon)
        self._co_exchange()
        passes = 0
        while True:
            changed, lost = self._reduce_flags(e.co_pass(), dist.ReduceOp.MAX)
            passes += 1
            if lost or not changed:
                break
            if passes >= max_passes:
                raise RuntimeError("crossing out did not reach its fixed point within max_passes")
            self._co_exchange()
        out = e.co_finish(lost, with_mask)
        out["not_empty"] = self._reduce_flags([out["not_empty"]], dist.ReduceOp.MIN)[0]
        out["passes"] = passes
        return out

    def energy(self):
        """Energy of the whole image (diffusion_graph.rs:502-518): sum over bands."""
        t = torch.tensor([self.e.energy()], dtype=torch.float64)
        if self.world > 1:
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())
