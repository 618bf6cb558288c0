"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libsporacle.so, the CPU restatement of the
reference's superpixel_stereo crate (superpixel_stereo/src/{utils,superpixels,diffusion_graph,
crossing_out_graph}.rs).  Only tests/ import it; the product never does.

`SuperpixelOracle` keeps the reference's method names.  Arrays are views into the C state:
potentials[si][sj][superpixel][n][d] (diffusion_graph.rs:44), edge_perimeters[si][sj][superpixel][n]
(superpixels.rs:47), superpixels[i][j] (superpixels.rs:39), vertices[si][sj][superpixel][d] and
edges[si][sj][superpixel][d][n][n_d] (crossing_out_graph.rs:42-45).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
P, I, F = C.c_void_p, C.c_int, C.c_double


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libsporacle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", HERE])
        L = C.CDLL(path)
        L.sp_create.restype = P
        L.sp_create.argtypes = [P, P, I, I, I, F, I, I]
        L.sp_destroy.argtypes = [P]
        for name in ("sp_potentials", "sp_dummy_potentials", "sp_perimeters", "sp_superpixels", "sp_co_vertices",
                     "sp_co_edges"):
            getattr(L, name).restype = P
            getattr(L, name).argtypes = [P]
        for name, n in (("sp_neighbor_exists", 5), ("sp_neighbor_superpixel", 2), ("sp_number_of_neighbors", 4)):
            getattr(L, name).restype = I
            getattr(L, name).argtypes = [I] * n
        L.sp_neighbor_index.restype = I
        L.sp_neighbor_index.argtypes = [I, I, I, I, P, P, P]
        L.sp_approx_equal.restype = I
        L.sp_approx_equal.argtypes = [F, F, F]
        L.sp_check_disparity_map_consistency.restype = I
        L.sp_check_disparity_map_consistency.argtypes = [P, I, I]
        L.sp_dedup_f64.restype = C.c_long
        L.sp_dedup_f64.argtypes = [P, C.c_long, F, P]
        L.sp_median.restype = C.c_long
        L.sp_median.argtypes = [P, C.c_long]
        L.sp_average.restype = C.c_long
        L.sp_average.argtypes = [P, C.c_long]
        L.sp_split_into_superpixels.argtypes = [P]
        L.sp_left_j_in_superpixel.restype = I
        L.sp_left_j_in_superpixel.argtypes = [P, I, I, I]
        L.sp_fill_superpixel_ids.argtypes = [P, P]
        L.sp_calculate_edge_perimeters.restype = I
        L.sp_calculate_edge_perimeters.argtypes = [P]
        for name, n in (("sp_sum_of_potentials", 4), ("sp_vertex_penalty_with_potentials", 4),
                        ("sp_edge_penalty_with_potential", 6), ("sp_min_edge_between_neighbors", 5),
                        ("sp_min_penalty_edge", 4)):
            getattr(L, name).restype = F
            getattr(L, name).argtypes = [P] + [I] * n
        L.sp_edge_exists.restype = I
        L.sp_edge_exists.argtypes = [P] + [I] * 6
        L.sp_penalty.restype = F
        L.sp_penalty.argtypes = [P, P]
        for name, n in (("sp_update_vertex_potential", 5), ("sp_update_edge_potential", 5),
                        ("sp_diffusion_act_vertexes", 3), ("sp_diffusion_act_edges", 3), ("sp_diffusion_act", 0),
                        ("sp_diffusion_act_coloured", 0)):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [P] + [I] * n
        L.sp_min_penalty_vertex.restype = F
        L.sp_min_penalty_vertex.argtypes = [P, I, I, I, P]
        L.sp_energy.restype = F
        L.sp_energy.argtypes = [P]
        L.sp_build_depth_map.argtypes = [P, P]
        L.sp_co_alloc.restype = I
        L.sp_co_alloc.argtypes = [P, I]
        for name in ("sp_initialize_vertices", "sp_initialize_edges", "sp_initialize_with_epsilon"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [P, F]
        L.sp_crossing_out.argtypes = [P]
        for name in ("sp_vertices_exist", "sp_edges_exist", "sp_is_not_empty"):
            getattr(L, name).restype = I
            getattr(L, name).argtypes = [P]
        for name in ("sp_min_vertex_between_existing", "sp_top_vertex_between_existing"):
            getattr(L, name).restype = I
            getattr(L, name).argtypes = [P, I, I, I]
        L.sp_simple_best_labeling.restype = I
        L.sp_simple_best_labeling.argtypes = [P, P]
        L.sp_diffusion_while_not_consistent.restype = I
        L.sp_diffusion_while_not_consistent.argtypes = [P, F, I, I, I]
        for s in ("f64", "f32"):
            getattr(L, "sp_fast_update_node_" + s).argtypes = [P, P, I, I, I]
            getattr(L, "sp_fast_iterate_" + s).argtypes = [P, P, I, I]
            getattr(L, "sp_fast_energy_" + s).restype = F
            getattr(L, "sp_fast_energy_" + s).argtypes = [P, P]
            getattr(L, "sp_fast_depth_map_" + s).argtypes = [P, P]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(P)


def neighbor_exists(si, sj, n, n_vertical, n_horizontal):
    r = lib().sp_neighbor_exists(si, sj, n, n_vertical, n_horizontal)
    if r < 0:
        raise IndexError("Non-existent neighbor index: %d" % n)
    return bool(r)


def neighbor_index(si, sj, n, superpixel):
    a, b, c = I(), I(), I()
    if lib().sp_neighbor_index(si, sj, n, superpixel, C.byref(a), C.byref(b), C.byref(c)) != 0:
        raise IndexError("Non-existent neighbor index: %d" % n)
    return a.value, b.value, c.value


def neighbor_superpixel(superpixel, n):
    return lib().sp_neighbor_superpixel(superpixel, n)


def number_of_neighbors(si, sj, n_vertical, n_horizontal):
    return lib().sp_number_of_neighbors(si, sj, n_vertical, n_horizontal)


def approx_equal(x, y, eps):
    return bool(lib().sp_approx_equal(x, y, eps))


def check_disparity_map_consistency(m):
    m = np.ascontiguousarray(m, dtype=np.int64)
    return bool(lib().sp_check_disparity_map_consistency(_ptr(m), m.shape[0], m.shape[1]))


def dedup_f64(arr, tol):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    out = np.empty(a.size + 1, dtype=np.float64)
    n = lib().sp_dedup_f64(_ptr(a), a.size, tol, _ptr(out))
    if n < 0:
        raise IndexError("dedup_f64 of an empty array")
    return out[:n].copy()


def median(numbers):
    a = np.array(numbers, dtype=np.uint32)
    r = lib().sp_median(_ptr(a), a.size)
    if r < 0:
        raise ValueError("Zero len of numbers list for finding median")
    return r


def average(numbers):
    a = np.array(numbers, dtype=np.uint32)
    r = lib().sp_average(_ptr(a), a.size)
    if r < 0:
        raise ValueError("Zero len of numbers list for finding average")
    return r


class SuperpixelOracle:
    """SuperpixelRepresentation + DiffusionGraph + CrossingOutGraph of the crate, in one object.
    split=True runs split_into_superpixels (as every caller in the reference does), perimeters=True
    also calculate_edge_perimeters (only the crate's main does)."""

    def __init__(self, left, right, max_disparity, smoothing_term, super_height, super_width, split=True,
                 perimeters=False):
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        assert left.shape == right.shape
        self.H, self.W = left.shape
        self.D = int(max_disparity)
        self.super_height, self.super_width = super_height, super_width
        self._g = lib().sp_create(_ptr(left), _ptr(right), self.H, self.W, self.D, float(smoothing_term), super_height,
                                  super_width)
        if not self._g:
            raise ValueError("bad arguments")
        self.NI, self.NJ = self.H // super_height, self.W // super_width
        self.potentials = self._view("sp_potentials", np.float64, (self.NI, self.NJ, 2, 9, self.D))
        self.dummy_potentials = self._view("sp_dummy_potentials", np.float64, (self.NI, self.NJ, 2, 9, self.D))
        self.edge_perimeters = self._view("sp_perimeters", np.float64, (self.NI, self.NJ, 2, 9))
        self.superpixels = self._view("sp_superpixels", np.int32, (self.H, self.W))
        self.vertices = self.edges = None
        if split:
            self.split_into_superpixels()
        if perimeters:
            self.calculate_edge_perimeters()

    def _view(self, getter, dtype, shape):
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        ctype = {np.float64: C.c_double, np.int32: C.c_int32, np.uint8: C.c_uint8}[dtype]
        buf = (ctype * n).from_address(getattr(lib(), getter)(self._g))
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def __del__(self):
        if getattr(self, "_g", None):
            lib().sp_destroy(self._g)
            self._g = None

    # superpixels.rs
    def split_into_superpixels(self):
        lib().sp_split_into_superpixels(self._g)

    def left_j_in_superpixel(self, si, sj, superpixel):
        return lib().sp_left_j_in_superpixel(self._g, si, sj, superpixel)

    def fill_superpixel_ids(self):
        out = np.empty((self.H, self.W), dtype=np.int64)
        lib().sp_fill_superpixel_ids(self._g, _ptr(out))
        return out

    def calculate_edge_perimeters(self):
        if lib().sp_calculate_edge_perimeters(self._g) != 0:
            raise IndexError("calculate_edge_perimeters indexes beyond the image (the reference panics)")

    # diffusion_graph.rs
    def sum_of_potentials(self, si, sj, d, superpixel):
        return lib().sp_sum_of_potentials(self._g, si, sj, d, superpixel)

    def vertex_penalty_with_potentials(self, si, sj, d, superpixel):
        return lib().sp_vertex_penalty_with_potentials(self._g, si, sj, d, superpixel)

    def edge_penalty_with_potential(self, si, sj, n, d, n_d, superpixel):
        return lib().sp_edge_penalty_with_potential(self._g, si, sj, n, d, n_d, superpixel)

    def edge_exists(self, si, sj, n, d, n_d, superpixel):
        return lib().sp_edge_exists(self._g, si, sj, n, d, n_d, superpixel) == 1

    def penalty(self, disparity_map):
        m = np.ascontiguousarray(disparity_map, dtype=np.int64)
        assert m.shape == (self.NI, self.NJ, 2)
        return lib().sp_penalty(self._g, _ptr(m))

    def min_edge_between_neighbors(self, si, sj, n, d, superpixel):
        return lib().sp_min_edge_between_neighbors(self._g, si, sj, n, d, superpixel)

    def update_vertex_potential(self, si, sj, n, d, superpixel):
        lib().sp_update_vertex_potential(self._g, si, sj, n, d, superpixel)

    def update_edge_potential(self, si, sj, n, d, superpixel):
        lib().sp_update_edge_potential(self._g, si, sj, n, d, superpixel)

    def diffusion_act_vertexes(self, si, sj, superpixel):
        lib().sp_diffusion_act_vertexes(self._g, si, sj, superpixel)

    def diffusion_act_edges(self, si, sj, superpixel):
        lib().sp_diffusion_act_edges(self._g, si, sj, superpixel)

    def diffusion_act(self):
        lib().sp_diffusion_act(self._g)

    def diffusion_act_coloured(self):
        lib().sp_diffusion_act_coloured(self._g)

    def min_penalty_vertex(self, si, sj, superpixel):
        d = I()
        v = lib().sp_min_penalty_vertex(self._g, si, sj, superpixel, C.byref(d))
        return d.value, v

    def min_penalty_edge(self, si, sj, n, superpixel):
        return lib().sp_min_penalty_edge(self._g, si, sj, n, superpixel)

    def energy(self):
        return lib().sp_energy(self._g)

    def build_depth_map(self):
        out = np.empty((self.H, self.W), dtype=np.int64)
        lib().sp_build_depth_map(self._g, _ptr(out))
        return out

    # crossing_out_graph.rs
    def co_alloc(self, fill=True):
        if lib().sp_co_alloc(self._g, 1 if fill else 0) != 0:
            raise MemoryError("crossing-out masks")
        self.vertices = self._view("sp_co_vertices", np.uint8, (self.NI, self.NJ, 2, self.D))
        self.edges = self._view("sp_co_edges", np.uint8, (self.NI, self.NJ, 2, self.D, 9, self.D))

    def initialize_vertices(self, eps):
        lib().sp_initialize_vertices(self._g, eps)

    def initialize_edges(self, eps):
        lib().sp_initialize_edges(self._g, eps)

    def initialize_with_epsilon(self, eps):
        lib().sp_initialize_with_epsilon(self._g, eps)

    def crossing_out(self):
        lib().sp_crossing_out(self._g)

    def vertices_exist(self):
        return bool(lib().sp_vertices_exist(self._g))

    def edges_exist(self):
        return bool(lib().sp_edges_exist(self._g))

    def is_not_empty(self):
        return bool(lib().sp_is_not_empty(self._g))

    def min_vertex_between_existing(self, si, sj, superpixel):
        r = lib().sp_min_vertex_between_existing(self._g, si, sj, superpixel)
        if r < 0:
            raise AssertionError("Disparity for pixel (%d, %d) wasn't found" % (si, sj))
        return r

    def top_vertex_between_existing(self, si, sj, superpixel):
        r = lib().sp_top_vertex_between_existing(self._g, si, sj, superpixel)
        if r < 0:
            raise AssertionError("Disparity for pixel (%d, %d) wasn't found" % (si, sj))
        return r

    def simple_best_labeling(self):
        out = np.empty((self.H, self.W), dtype=np.int64)
        if lib().sp_simple_best_labeling(self._g, _ptr(out)) != 0:
            raise AssertionError("a superpixel has no label left")
        return out

    def diffusion_while_not_consistent(self, eps, batch_size, max_batches=1000, schedule=0):
        if self.vertices is None:
            self.co_alloc(True)
        return lib().sp_diffusion_while_not_consistent(self._g, eps, batch_size, max_batches, schedule)

    # the O(D) form the GPU kernels evaluate
    def fast(self, dtype="f64"):
        return FastSuperpixel(self, dtype)


class FastSuperpixel:
    """O(D) restatement of the node update on its own copy of the potentials, f64 or f32."""

    def __init__(self, oracle, dtype="f64"):
        self.o, self.dtype = oracle, dtype
        self.np = np.float64 if dtype == "f64" else np.float32
        self.potentials = np.ascontiguousarray(oracle.potentials, dtype=self.np).copy()

    def _f(self, name):
        return getattr(lib(), "sp_fast_%s_%s" % (name, self.dtype))

    def update_node(self, si, sj, superpixel):
        self._f("update_node")(self.o._g, _ptr(self.potentials), si, sj, superpixel)

    def iterate(self, n, schedule=0):
        self._f("iterate")(self.o._g, _ptr(self.potentials), int(n), int(schedule))

    def energy(self):
        return self._f("energy")(self.o._g, _ptr(self.potentials))

    def depth_map(self):
        out = np.empty((self.o.H, self.o.W), dtype=np.int64)
        self._f("depth_map")(self.o._g, _ptr(self.potentials), _ptr(out))
        return out
