"""Helpers shared by the tests."""
import numpy as np

NP_REAL = {"f64": np.float64, "f32": np.float32}


def random_pair(H, W, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, size=(H, W), dtype=np.uint8),
            rng.integers(0, 256, size=(H, W), dtype=np.uint8))


def valid_mask(H, W, D):
    """[H][W][4][D] bool: slot exists and d <= j."""
    m = np.zeros((H, W, 4, D), dtype=bool)
    jj = np.arange(W)[None, :, None]
    dd = np.arange(D)[None, None, :]
    ok = np.broadcast_to(dd <= jj, (H, W, D))
    ii = np.arange(H)[:, None, None]
    m[:, :, 0, :] = ok & (jj > 0)
    m[:, :, 1, :] = ok & (ii > 0)
    m[:, :, 2, :] = ok & (jj + 1 < W)
    m[:, :, 3, :] = ok & (ii + 1 < H)
    return m


def random_phi(H, W, D, dtype, seed=0, scale=10.0):
    """Random potentials on existing slots / valid labels, 0 elsewhere (the
    reference never writes the other entries)."""
    rng = np.random.default_rng(seed)
    phi = (rng.standard_normal((H, W, 4, D)) * scale).astype(NP_REAL[dtype])
    phi[~valid_mask(H, W, D)] = 0
    return phi


def rel_err(a, b):
    """max |a-b| / max(1, max|b|): norm-wise relative error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


def elem_err(a, b):
    """max over elements of |a-b| / max(1, |b|): element-wise relative error (every entry is
    held to the tolerance against its own magnitude, not against the largest one)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())


class NumpyCrossingOut:
    """Vectorised restatement of CrossingOutGraph::initialize_with_epsilon + crossing_out
    (crossing_out_graph.rs:67-183) with explicit [D][D] edge tables, for the rows
    [row_begin, row_end) of an image.  phi [H][W][4][D] and cost [H][W][D] are image-sized;
    only the band's rows and the row above / below it are read.  The masks of those two halo
    rows come from outside (`alive[row]`, all-alive until set)."""

    SHIFT = {0: (0, -1), 1: (-1, 0), 2: (0, 1), 3: (1, 0)}

    def __init__(self, phi, cost, lam, eps, row_begin=0, row_end=None):
        H, W, _, D = phi.shape
        self.H, self.W, self.D = H, W, D
        self.rb, self.re = row_begin, H if row_end is None else row_end
        phi = np.asarray(phi, dtype=np.float64)
        dd = np.arange(D)
        valid = np.broadcast_to(dd[None, None, :] <= np.arange(W)[None, :, None], (H, W, D)).copy()
        has = np.zeros((H, W, 4), bool)
        has[:, 1:, 0] = True
        has[1:, :, 1] = True
        has[:, :-1, 2] = True
        has[:-1, :, 3] = True
        self.valid, self.has = valid, has
        band = slice(self.rb, self.re)
        f = cost - np.where(has[..., None], phi, 0.0).sum(axis=2)
        fmin = np.where(valid, f, np.inf).min(axis=2, keepdims=True)
        self.alive = valid.copy()  # halo rows: all alive until the neighbour's masks arrive
        self.alive[band] = (valid & (f <= fmin + eps))[band]
        self.usable = []
        for n in range(4):
            psi = self._neighbour(phi[:, :, n ^ 2, :], n, np.inf)
            nvalid = self._neighbour(valid, n, False)
            e = lam * np.abs(dd[:, None] - dd[None, :])[None, None] + phi[:, :, n, :, None] + psi[:, :, None, :]
            ok = valid[:, :, :, None] & nvalid[:, :, None, :]
            if n == 2:
                ok = ok & (dd[None, :] <= dd[:, None] + 1)[None, None]
            if n == 0:
                ok = ok & (dd[:, None] <= dd[None, :] + 1)[None, None]
            e = np.where(ok, e, np.inf)
            mn = e.reshape(H, W, -1).min(axis=2)[:, :, None, None]
            self.usable.append(ok & (e <= mn + eps))

    def _neighbour(self, a, n, fill):
        """out[i][j] = a[i + di][j + dj] for slot n's neighbour, `fill` outside the image."""
        di, dj = self.SHIFT[n]
        H, W = self.H, self.W
        out = np.full_like(a, fill)
        src = a[max(di, 0):H - max(-di, 0), max(dj, 0):W - max(-dj, 0)]
        out[max(-di, 0):H - max(di, 0), max(-dj, 0):W - max(dj, 0)] = src
        return out

    def relax(self):
        """One Jacobi pass over the band; returns (changed, lost)."""
        band = slice(self.rb, self.re)
        nxt = self.alive.copy()
        for n in range(4):
            partner = self._neighbour(self.alive, n, False)
            support = (self.usable[n] & partner[:, :, None, :]).any(axis=3)
            nxt &= support | ~self.has[:, :, n, None]
        changed = bool((nxt[band] != self.alive[band]).any())
        self.alive[band] = nxt[band]
        lost = bool((~self.alive[band].any(axis=2)).any())
        return changed, lost

    def run(self):
        while self.relax()[0]:
            pass
        return self.alive[self.rb:self.re]

    def labels(self):
        a = self.alive[self.rb:self.re]
        return np.where(a.any(axis=2), a.argmax(axis=2), 0xFFFFFFFF).astype(np.uint32)
