"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product path (stereo-vision_amd/) never does.

`OracleGraph` wraps oracle/liboracle.so (the faithful O(D^2) restatement of
diffusion_stereo/src/diffusion_graph.rs and crossing_out_graph.rs) and keeps
the reference's method names; `fast` wraps oracle/libfastoracle.so (the O(D)
OpenMP restatement of the checkerboard sweep).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_FAST = None

c_real = {"f64": C.c_double, "f32": C.c_float}
np_real = {"f64": np.float64, "f32": np.float32}
sfx = {"f64": "64", "f32": "32"}


def build():
    """Compile the oracle libraries in place (gcc only; no GPU, no reference sources)."""
    subprocess.check_call(["make", "-s", "-C", HERE])


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


def lib():
    global _LIB
    if _LIB is None:
        L = _load("liboracle.so")
        P = C.c_void_p
        I = C.c_int
        for s, R in (("64", C.c_double), ("32", C.c_float)):
            f = lambda n: getattr(L, "og%s_%s" % (s, n))
            f("initialize").restype = P
            f("initialize").argtypes = [P, P, I, I, I, C.c_double]
            f("destroy").argtypes = [P]
            f("potentials").restype = P
            f("potentials").argtypes = [P]
            f("dummy_potentials").restype = P
            f("dummy_potentials").argtypes = [P]
            for n, nargs in (("sum_of_potentials", 3), ("vertex_penalty_with_potentials", 3),
                             ("edge_penalty_with_potential", 5), ("min_edge_between_neighbors", 4),
                             ("min_penalty_edge", 3)):
                f(n).restype = R
                f(n).argtypes = [P] + [I] * nargs
            f("edge_exists").restype = I
            f("edge_exists").argtypes = [P] + [I] * 5
            f("penalty").restype = R
            f("penalty").argtypes = [P, P]
            for n, nargs in (("update_vertex_potential", 4), ("update_edge_potential", 4),
                             ("diffusion_act_vertexes", 2), ("diffusion_act_edges", 2),
                             ("diffusion_act", 0), ("diffusion_act_checkerboard", 0),
                             ("diffusion_half_sweep", 1)):
                f(n).restype = None
                f(n).argtypes = [P] + [I] * nargs
            f("min_penalty_vertex").restype = R
            f("min_penalty_vertex").argtypes = [P, I, I, C.POINTER(C.c_int)]
            f("energy").restype = R
            f("energy").argtypes = [P]
            f("depth_map").argtypes = [P, P]
            f("belief_labels").argtypes = [P, P, P]
            f("co_alloc").restype = I
            f("co_alloc").argtypes = [P, I]
            f("co_vertices").restype = P
            f("co_vertices").argtypes = [P]
            f("co_edges").restype = P
            f("co_edges").argtypes = [P]
            for n in ("co_initialize_vertices", "co_initialize_edges", "co_initialize_with_epsilon"):
                f(n).restype = None
                f(n).argtypes = [P, C.c_double]
            f("co_crossing_out").restype = None
            f("co_crossing_out").argtypes = [P]
            for n in ("co_vertices_exist", "co_edges_exist", "co_is_not_empty"):
                f(n).restype = I
                f(n).argtypes = [P]
            for n in ("co_min_vertex_between_existing", "co_top_vertex_between_existing"):
                f(n).restype = I
                f(n).argtypes = [P, I, I]
            f("co_simple_best_labeling").restype = I
            f("co_simple_best_labeling").argtypes = [P, P]
            f("co_cross_vertex").restype = I
            f("co_cross_vertex").argtypes = [P, I, I, I]
            f("co_find_best_labeling").restype = I
            f("co_find_best_labeling").argtypes = [P, P]
            f("co_diffusion_while_not_consistent").restype = I
            f("co_diffusion_while_not_consistent").argtypes = [P, C.c_double, I, I, I]
            f("create_array_of_epsilons").restype = C.c_long
            f("create_array_of_epsilons").argtypes = [P, C.c_double, C.POINTER(C.POINTER(C.c_double))]
            f("epsilon_search").restype = C.c_double
            f("epsilon_search").argtypes = [P, P, C.c_long]
        L.og_free.argtypes = [P]
        L.og_free.restype = None
        L.og_neighbor_exists.argtypes = [I] * 5
        L.og_neighbor_index.argtypes = [I, I, I] + [C.POINTER(C.c_int)] * 3
        L.og_number_of_neighbors.argtypes = [I] * 4
        L.og_approx_equal.argtypes = [C.c_double] * 3
        L.og_check_disparity_map.argtypes = [P, I, I]
        L.og_dedup_f64.argtypes = [P, I, C.c_double, P]
        L.og_cost_volume.argtypes = [P, P, I, I, I, P]
        L.og_cost_volume.restype = None
        L.og_box_blur.argtypes = [P, I, I, C.c_uint64, C.c_uint64, C.c_uint64, P]
        L.og_pgm_reader.argtypes = [C.c_char_p, P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.og_pgm_writer.argtypes = [P, I, I, C.c_char_p, I]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- utils.rs -------------------------------------------------------------

def neighbor_exists(i, j, n, height, width):
    r = lib().og_neighbor_exists(i, j, n, height, width)
    if r < 0:
        raise ValueError("Non-existent neighbor index: %d" % n)
    return bool(r)


def neighbor_index(i, j, n):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    if lib().og_neighbor_index(i, j, n, C.byref(a), C.byref(b), C.byref(c)) < 0:
        raise ValueError("Non-existent neighbor index: %d" % n)
    return a.value, b.value, c.value


def number_of_neighbors(i, j, height, width):
    return lib().og_number_of_neighbors(i, j, height, width)


def approx_equal(x, y, eps):
    return bool(lib().og_approx_equal(x, y, eps))


def check_disparity_map(m):
    m = np.ascontiguousarray(m, dtype=np.int32)
    return bool(lib().og_check_disparity_map(_ptr(m), m.shape[0], m.shape[1]))


def dedup_f64(arr, tol):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    out = np.empty_like(a)
    n = lib().og_dedup_f64(_ptr(a), len(a), tol, _ptr(out))
    return out[:n].tolist()


def cost_volume(left, right, D):
    """[D][H][W] u8, entries with d > j are 0."""
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    H, W = left.shape
    out = np.empty((D, H, W), dtype=np.uint8)
    lib().og_cost_volume(_ptr(left), _ptr(right), H, W, D, _ptr(out))
    return out


def box_blur(disparity_map, threshold, filter_radius, normalization_term):
    """diffusion_graph.rs:399-424 with a release build's wrapping usize arithmetic.  u64 [H][W]."""
    m = np.ascontiguousarray(disparity_map, dtype=np.uint32)
    out = np.empty(m.shape, dtype=np.uint64)
    if lib().og_box_blur(_ptr(m), m.shape[0], m.shape[1], threshold, filter_radius, normalization_term, _ptr(out)) != 0:
        raise ZeroDivisionError("normalization_term is 0 (the reference divides by it)")
    return out


def pgm_reader(path):
    """Returns (matrix u32 [H][W], width, height) like pgm_handler.rs:36-74."""
    w, h, mx = C.c_int(), C.c_int(), C.c_int()
    p = path.encode()
    if lib().og_pgm_reader(p, None, C.byref(w), C.byref(h), C.byref(mx)) != 0:
        raise IOError("cannot read PGM header: %s" % path)
    out = np.empty((h.value, w.value), dtype=np.uint32)
    if lib().og_pgm_reader(p, _ptr(out), C.byref(w), C.byref(h), C.byref(mx)) != 0:
        raise IOError("malformed PGM: %s" % path)
    return out, w.value, h.value


def pgm_writer(matrix, path, max_intensity):
    m = np.ascontiguousarray(matrix, dtype=np.int32)
    if lib().og_pgm_writer(_ptr(m), m.shape[0], m.shape[1], path.encode(), max_intensity) != 0:
        raise IOError("cannot write %s" % path)


# ---- diffusion_graph.rs + crossing_out_graph.rs ---------------------------

class OracleGraph:
    """DiffusionGraph (+ CrossingOutGraph masks) of the reference, on the CPU."""

    def __init__(self, left, right, max_disparity, smoothing_term, dtype="f64"):
        left = np.ascontiguousarray(left, dtype=np.uint32)
        right = np.ascontiguousarray(right, dtype=np.uint32)
        assert left.shape == right.shape
        self.H, self.W = left.shape
        self.D = int(max_disparity)
        self.dtype = dtype
        self._s = sfx[dtype]
        self._g = self._f("initialize")(_ptr(left), _ptr(right), self.H, self.W, self.D,
                                        float(smoothing_term))
        n = self.H * self.W * 4 * self.D
        shape = (self.H, self.W, 4, self.D)
        if n:
            ct = c_real[dtype] * n
            self.potentials = np.frombuffer(ct.from_address(self._f("potentials")(self._g)),
                                            dtype=np_real[dtype]).reshape(shape)
            self.dummy_potentials = np.frombuffer(
                ct.from_address(self._f("dummy_potentials")(self._g)),
                dtype=np_real[dtype]).reshape(shape)
        else:
            self.potentials = np.zeros(shape, np_real[dtype])
            self.dummy_potentials = np.zeros(shape, np_real[dtype])
        self.vertices = None
        self.edges = None

    def _f(self, name):
        return getattr(lib(), "og%s_%s" % (self._s, name))

    def __del__(self):
        try:
            self._f("destroy")(self._g)
        except Exception:
            pass

    def set_potentials(self, pot):
        """Production state: potentials == dummy_potentials (both buffers)."""
        self.potentials[...] = pot
        self.dummy_potentials[...] = pot

    def sum_of_potentials(self, i, j, d):
        return self._f("sum_of_potentials")(self._g, i, j, d)

    def vertex_penalty_with_potentials(self, i, j, d):
        return self._f("vertex_penalty_with_potentials")(self._g, i, j, d)

    def edge_penalty_with_potential(self, i, j, n, d, n_d):
        return self._f("edge_penalty_with_potential")(self._g, i, j, n, d, n_d)

    def edge_exists(self, i, j, n, d, n_d):
        return bool(self._f("edge_exists")(self._g, i, j, n, d, n_d))

    def penalty(self, disparity_map):
        m = np.ascontiguousarray(disparity_map, dtype=np.int32)
        return self._f("penalty")(self._g, _ptr(m))

    def min_edge_between_neighbors(self, i, j, n, d):
        return self._f("min_edge_between_neighbors")(self._g, i, j, n, d)

    def update_vertex_potential(self, i, j, n, d):
        self._f("update_vertex_potential")(self._g, i, j, n, d)

    def update_edge_potential(self, i, j, n, d):
        self._f("update_edge_potential")(self._g, i, j, n, d)

    def diffusion_act_vertexes(self, i, j):
        self._f("diffusion_act_vertexes")(self._g, i, j)

    def diffusion_act_edges(self, i, j):
        self._f("diffusion_act_edges")(self._g, i, j)

    def diffusion_act(self):
        self._f("diffusion_act")(self._g)

    def diffusion_act_checkerboard(self):
        self._f("diffusion_act_checkerboard")(self._g)

    def diffusion_half_sweep(self, colour):
        self._f("diffusion_half_sweep")(self._g, colour)

    def min_penalty_vertex(self, i, j):
        d = C.c_int()
        v = self._f("min_penalty_vertex")(self._g, i, j, C.byref(d))
        return d.value, v

    def min_penalty_edge(self, i, j, n):
        return self._f("min_penalty_edge")(self._g, i, j, n)

    def energy(self):
        return self._f("energy")(self._g)

    def depth_map(self):
        out = np.empty((self.H, self.W), dtype=np.int32)
        self._f("depth_map")(self._g, _ptr(out))
        return out

    def belief_labels(self, with_belief=False):
        out = np.empty((self.H, self.W), dtype=np.int32)
        bel = np.empty((self.H, self.W, self.D), dtype=np_real[self.dtype]) if with_belief else None
        self._f("belief_labels")(self._g, _ptr(out), _ptr(bel) if with_belief else None)
        return (out, bel) if with_belief else out

    # -- CrossingOutGraph --
    def co_alloc(self, fill=True):
        if self._f("co_alloc")(self._g, 1 if fill else 0) != 0:
            raise MemoryError("crossing-out masks")
        nv = self.H * self.W * self.D
        ne = nv * 4 * self.D
        self.vertices = np.frombuffer((C.c_uint8 * max(nv, 1)).from_address(
            self._f("co_vertices")(self._g)), dtype=np.uint8)[:nv].reshape(self.H, self.W, self.D)
        self.edges = np.frombuffer((C.c_uint8 * max(ne, 1)).from_address(
            self._f("co_edges")(self._g)), dtype=np.uint8)[:ne].reshape(
                self.H, self.W, self.D, 4, self.D)

    def initialize_vertices(self, eps):
        self._f("co_initialize_vertices")(self._g, eps)

    def initialize_edges(self, eps):
        self._f("co_initialize_edges")(self._g, eps)

    def initialize_with_epsilon(self, eps):
        self._f("co_initialize_with_epsilon")(self._g, eps)

    def crossing_out(self):
        self._f("co_crossing_out")(self._g)

    def vertices_exist(self):
        return bool(self._f("co_vertices_exist")(self._g))

    def edges_exist(self):
        return bool(self._f("co_edges_exist")(self._g))

    def is_not_empty(self):
        return bool(self._f("co_is_not_empty")(self._g))

    def min_vertex_between_existing(self, i, j):
        d = self._f("co_min_vertex_between_existing")(self._g, i, j)
        if d < 0:
            raise AssertionError("Disparity for pixel (%d, %d) wasn't found" % (i, j))
        return d

    def top_vertex_between_existing(self, i, j):
        d = self._f("co_top_vertex_between_existing")(self._g, i, j)
        if d < 0:
            raise AssertionError("Disparity for pixel (%d, %d) wasn't found" % (i, j))
        return d

    def simple_best_labeling(self):
        out = np.empty((self.H, self.W), dtype=np.int32)
        if self._f("co_simple_best_labeling")(self._g, _ptr(out)) != 0:
            raise AssertionError("some pixel has no vertex left")
        return out

    def cross_vertex(self, i, j, disparity):
        if self._f("co_cross_vertex")(self._g, i, j, disparity) != 0:
            raise AssertionError("Choosen vertex [%d][%d][%d] is crossed out" % (i, j, disparity))

    def find_best_labeling(self):
        out = np.empty((self.H, self.W), dtype=np.int32)
        if self._f("co_find_best_labeling")(self._g, _ptr(out)) != 0:
            raise AssertionError("Graph is empty")
        return out

    def diffusion_while_not_consistent(self, eps, batch_size, max_batches=1000, schedule=0):
        return self._f("co_diffusion_while_not_consistent")(self._g, eps, batch_size, max_batches,
                                                            schedule)

    # -- epsilon_search.rs --
    def create_array_of_epsilons(self, tolerance):
        out = C.POINTER(C.c_double)()
        n = self._f("create_array_of_epsilons")(self._g, tolerance, C.byref(out))
        if n < 0:
            raise AssertionError("a penalty lies below its minimum (assert_ge!)")
        arr = np.array([out[k] for k in range(n)], dtype=np.float64)
        lib().og_free(out)
        return arr

    def epsilon_search(self, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        assert len(a) != 0
        return self._f("epsilon_search")(self._g, _ptr(a), len(a))


# ---- fast O(D) oracle -----------------------------------------------------

def fastlib():
    global _FAST
    if _FAST is None:
        L = _load("libfastoracle.so")
        P, I, Dbl = C.c_void_p, C.c_int, C.c_double
        for s in ("64", "32"):
            f = lambda n: getattr(L, "fo%s_%s" % (s, n))
            f("half_sweep").restype = None
            f("half_sweep").argtypes = [P, P, P, I, I, I, Dbl, I, I, I]
            f("iterate").restype = None
            f("iterate").argtypes = [P, P, P, I, I, I, Dbl, I]
            f("energy").restype = Dbl
            f("energy").argtypes = [P, P, P, I, I, I, Dbl]
            f("labels").restype = None
            f("labels").argtypes = [P, P, P, I, I, I, Dbl, I, P]
            f("max_residual").restype = Dbl
            f("max_residual").argtypes = [P, P, P, I, I, I]
        _FAST = L
    return _FAST


class FastOracle:
    """O(D) checkerboard sweep on a numpy array pot[H][W][4][D] (reference layout)."""

    def __init__(self, left, right, max_disparity, smoothing_term, dtype="f64"):
        self.L = np.ascontiguousarray(left, dtype=np.uint8)
        self.R = np.ascontiguousarray(right, dtype=np.uint8)
        self.H, self.W = self.L.shape
        self.D = int(max_disparity)
        assert 1 <= self.D <= 256
        self.lam = float(smoothing_term)
        self.dtype = dtype
        self._s = sfx[dtype]
        self.potentials = np.zeros((self.H, self.W, 4, self.D), dtype=np_real[dtype])

    def _f(self, name):
        return getattr(fastlib(), "fo%s_%s" % (self._s, name))

    def _args(self):
        return (_ptr(self.potentials), _ptr(self.L), _ptr(self.R), self.H, self.W, self.D)

    def half_sweep(self, colour, row_begin=0, row_end=None):
        self._f("half_sweep")(*self._args(), self.lam, colour, row_begin,
                              self.H if row_end is None else row_end)

    def iterate(self, n):
        self._f("iterate")(*self._args(), self.lam, n)

    def energy(self):
        return self._f("energy")(*self._args(), self.lam)

    def labels(self, mode):
        out = np.empty((self.H, self.W), dtype=np.uint32)
        self._f("labels")(*self._args(), self.lam, mode, _ptr(out))
        return out

    def max_residual(self):
        return self._f("max_residual")(*self._args())
