"""The row-band protocol stated in engine-agnostic Python, for tests that have no GPU.

The product runs the protocol inside the library (ds_band_iterate and its siblings,
stereo-vision_amd/csrc/bands.hpp).  `BandRunner` states which rows are swept when and which
halo rows travel after which colour; the engine supplies the sweep and the halo buffers
(tests/band_oracle_engine.py: the CPU oracle), torch.distributed (gloo) carries the rows.
What it states per iteration -- boundary rows of the first colour, their exchange, boundary
rows of the second colour, their exchange, the interior beside them -- is also what a
two-iteration step of an f32 band consists of, twice over (four exchanges per step, the same
messages in the same order as two single iterations), so bands that step by one and by two
iterations are interchangeable neighbours.
"""
import contextlib

import torch
import torch.distributed as dist


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
