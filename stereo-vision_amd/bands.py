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

`BandRunner` is engine-agnostic: the engine supplies the sweep and the halo buffers.
`HipBandEngine` is the product engine (C ABI of include/diffstereo.h).
"""
import contextlib

import torch
import torch.distributed as dist


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


class _DeviceBytes:
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                         "data": (ptr, False), "version": 2}


class HipBandEngine:
    """One band on the current GPU, driven through the C ABI."""

    def __init__(self, graph, stage_through_host=False):
        self.g = graph
        self.row_begin, self.row_end = graph.row_begin, graph.row_end
        self.H = graph.H
        self.stage = stage_through_host  # gloo rehearsal: halos travel via pinned host copies
        self._views = {}
        # A main stream of the engine's own rather than whatever stream is current when the
        # engine is made: the two streams below are then ordered by this class's events alone
        # (the legacy default stream would add its implicit synchronisation with blocking streams).
        self.main = torch.cuda.Stream()
        self.main.wait_stream(torch.cuda.current_stream())
        # boundary rows + exchanges beside the interior kernel.  Normal priority on purpose: with a
        # high-priority stream in the chain main -> side -> RCCL's stream -> side -> main the
        # device never drained (seen with RCCL send/recv on one MI355X), and the chain is queued
        # before the interior kernel anyway, so its first launch does not wait behind it.
        self.side = torch.cuda.Stream()
        graph.set_stream(self.main.cuda_stream)

    def beside(self):
        """Context: library launches and torch work go to the side stream, which first waits for
        everything queued on the main stream so far.  join() makes the main stream wait for it."""
        engine = self

        class _Beside:
            def __enter__(self):
                engine.side.wait_stream(engine.main)
                self._ctx = torch.cuda.stream(engine.side)
                self._ctx.__enter__()
                engine.g.switch_stream(engine.side.cuda_stream)

            def __exit__(self, *exc):
                engine.g.switch_stream(engine.main.cuda_stream)
                return self._ctx.__exit__(*exc)

        return _Beside()

    def join(self):
        self.main.wait_stream(self.side)

    def on_main(self):
        """Context for everything BandRunner queues: torch's current stream = the engine's main
        stream, so that the communication library orders itself against the library's launches."""
        return torch.cuda.stream(self.main)

    def _view(self, colour, side, which, upcoming=False):
        # the library alternates between two sets of colour-1 planes: views are kept per pointer
        ptr, nbytes = self.g.halo_next(side, which) if upcoming else self.g.halo(colour, side, which)
        if ptr not in self._views:
            self._views[ptr] = torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")
        return self._views[ptr]

    def sweep(self, colour, row_begin, row_end):
        if row_end > row_begin:
            self.g.sweep(colour, row_begin, row_end)

    def iterate(self, n_iters):
        self.g.iterate(n_iters)

    # whole iterations in one kernel on a band (include/diffstereo.h, ds_iterate_rows)
    def fused_rows(self):
        return self.g.fused_rows()

    def iterate_rows(self, row_begin, row_end):
        self.g.iterate_rows(row_begin, row_end)

    def sweep_next(self, row_begin, row_end):
        if row_end > row_begin:
            self.g.sweep_next(row_begin, row_end)

    def sweep_pair(self, colour, upcoming, rows1, rows2):
        self.g.sweep_pair(colour, upcoming, rows1, rows2)

    def finish_iterations(self):
        self.g.finish_iterations()

    def send_buffer_next(self, side):
        v = self._view(1, side, 0, upcoming=True)
        return v.cpu() if self.stage else v

    def recv_buffer_next(self, side):
        v = self._view(1, side, 1, upcoming=True)
        return (torch.empty(v.shape, dtype=v.dtype), v) if self.stage else (v, v)

    def send_buffer(self, colour, side):
        v = self._view(colour, side, 0)
        return v.cpu() if self.stage else v

    def recv_buffer(self, colour, side):
        v = self._view(colour, side, 1)
        return torch.empty(v.shape, dtype=v.dtype) if self.stage else v

    def commit_recv(self, colour, side, buf):
        if self.stage:
            self._view(colour, side, 1).copy_(buf)

    def energy(self):
        return self.g.energy()

    # crossing-out: the mask halo rows travel like the phi halo rows
    def co_begin(self, epsilon):
        self.g.co_begin(epsilon)
        self._co_views = {}

    def co_pass(self):
        return self.g.co_pass()

    def _co_view(self, colour, side, which):
        key = (colour, side, which)
        if key not in self._co_views:
            ptr, nbytes = self.g.co_halo(colour, side, which)
            self._co_views[key] = torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")
        return self._co_views[key]

    def co_send_buffer(self, colour, side):
        v = self._co_view(colour, side, 0)
        return v.cpu() if self.stage else v

    def co_recv_buffer(self, colour, side):
        v = self._co_view(colour, side, 1)
        return torch.empty(v.shape, dtype=v.dtype) if self.stage else v

    def co_commit_recv(self, colour, side, buf):
        if self.stage:
            self._co_view(colour, side, 1).copy_(buf)

    def co_finish(self, empty, with_mask=False):
        return self.g.co_finish(empty, with_mask)

    def synchronize(self):
        self.main.synchronize()


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
