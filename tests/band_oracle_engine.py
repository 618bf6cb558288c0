"""TEST-ONLY band engine: the O(D) CPU oracle standing in for the HIP kernels, so that
the row-band partition and the halo exchange of stereo-vision_amd/bands.py can run under
gloo on CPU ranks.  Every row this rank does not own and that is not a halo row is
poisoned with NaN, so any read outside band + halo shows up in the result."""
import numpy as np
import torch

from oracle import pyoracle
from oracle.pyoracle import FastOracle
from tests.util import NumpyCrossingOut


class OracleBandEngine:
    def __init__(self, left, right, D, lam, row_begin, row_end, dtype="f64"):
        self.f = FastOracle(left, right, D, lam, dtype=dtype)
        self.row_begin, self.row_end = row_begin, row_end
        self.H, self.W, self.D = self.f.H, self.f.W, D
        self.f.potentials[:max(row_begin - 1, 0)] = np.nan
        self.f.potentials[min(row_end + 1, self.H):] = np.nan
        self._recv = {}

    def set_rows(self, phi, lo, hi):
        self.f.potentials[lo:hi] = phi[lo:hi]

    def sweep(self, colour, row_begin, row_end):
        if row_end > row_begin:
            self.f.half_sweep(colour, row_begin, row_end)

    def _cols(self, row, colour):
        return np.arange((row + colour) & 1, self.W, 2)

    def send_buffer(self, colour, side):
        row, slot = (self.row_begin, 1) if side == 0 else (self.row_end - 1, 3)
        return torch.from_numpy(np.ascontiguousarray(self.f.potentials[row, self._cols(row, colour), slot, :]))

    def recv_buffer(self, colour, side):
        row = self.row_begin - 1 if side == 0 else self.row_end
        n = len(self._cols(row, colour))
        return torch.empty((n, self.D), dtype=torch.float64 if self.f.dtype == "f64" else torch.float32)

    def commit_recv(self, colour, side, buf):
        row, slot = (self.row_begin - 1, 3) if side == 0 else (self.row_end, 1)
        self.f.potentials[row, self._cols(row, colour), slot, :] = buf.numpy()

    def energy(self):
        """Band share of the energy: vertex and slot-2/3 edge terms of the band's rows."""
        return sum(_row_energy(self.f, i) for i in range(self.row_begin, self.row_end))

    def synchronize(self):
        pass

    # crossing-out on the band (tests/util.NumpyCrossingOut), masks of the halo rows exchanged
    def co_begin(self, epsilon):
        cost = pyoracle.cost_volume(self.f.L, self.f.R, self.D).astype(np.float64).transpose(1, 2, 0)
        with np.errstate(invalid="ignore"):  # poisoned rows outside band + halo
            self.co = NumpyCrossingOut(self.f.potentials, cost, self.f.lam, epsilon, self.row_begin, self.row_end)
        lo, hi = max(self.row_begin - 1, 0), min(self.row_end + 1, self.H)
        self.co.alive[:lo] = False   # anything read beyond the halo rows would kill support
        self.co.alive[hi:] = False

    def co_pass(self):
        return self.co.relax()

    def co_send_buffer(self, colour, side):
        row = self.row_begin if side == 0 else self.row_end - 1
        return torch.from_numpy(self.co.alive[row, self._cols(row, colour), :].astype(np.uint8))

    def co_recv_buffer(self, colour, side):
        row = self.row_begin - 1 if side == 0 else self.row_end
        return torch.empty((len(self._cols(row, colour)), self.D), dtype=torch.uint8)

    def co_commit_recv(self, colour, side, buf):
        row = self.row_begin - 1 if side == 0 else self.row_end
        self.co.alive[row, self._cols(row, colour), :] = buf.numpy().astype(bool)

    def co_finish(self, empty, with_mask=False):
        band = slice(self.row_begin, self.row_end)
        if empty:
            self.co.alive[band] = False
        out = {"labels": self.co.labels(), "not_empty": bool(self.co.alive[band].any(axis=2).all())}
        if with_mask:
            out["alive"] = self.co.alive[band].astype(np.uint8)
        return out


def _row_energy(f, i):
    """Energy terms owned by image row i (vertex minima + edges to the right and below)."""
    from oracle.pyoracle import OracleGraph
    # Brute-force graph of rows [i-1, i+1] clipped to the image, so that slot existence
    # (and with it f_t) at row i matches the full image.
    hi = min(i + 2, f.H)
    a = i - 1 if i > 0 else i
    g = OracleGraph(f.L[a:hi], f.R[a:hi], f.D, f.lam, dtype=f.dtype)
    pot = np.nan_to_num(f.potentials[a:hi], nan=0.0)
    g.set_potentials(pot)
    r = i - a
    e = 0.0
    for j in range(f.W):
        e += g.min_penalty_vertex(r, j)[1]
        if j + 1 < f.W:
            e += g.min_penalty_edge(r, j, 2)
        if i + 1 < f.H:
            e += g.min_penalty_edge(r, j, 3)
    return e
