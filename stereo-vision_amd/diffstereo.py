"""ctypes binding of the C ABI in include/diffstereo.h (tests and benchmark use it).

There is no CPU fallback: if csrc/libdiffstereo.so is missing, or a declared symbol
is not exported, `lib()` raises.  `DiffusionGraph` keeps the reference's names where
one exists (diffusion_stereo/src/diffusion_graph.rs:39-539).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DS_LIBRARY") or os.path.join(HERE, "csrc", "libdiffstereo.so")  # DS_LIBRARY: another build
DS_F32, DS_F64 = 0, 1
_DTYPES = {"f32": (DS_F32, np.float32), "f64": (DS_F64, np.float64)}

_P, _I, _D = C.c_void_p, C.c_int, C.c_double
_PP = C.POINTER(C.c_void_p)
_IP, _DP = C.POINTER(C.c_int), C.POINTER(C.c_double)

# name -> (restype, argtypes); must list every symbol include/diffstereo.h declares
SYMBOLS = {
    "ds_create": (_I, [_P, _P, _I, _I, _I, _D, _I, _PP]),
    "ds_create_band": (_I, [_P, _P, _I, _I, _I, _D, _I, _I, _I, _I, _PP]),
    "ds_destroy": (None, [_P]),
    "ds_last_error": (C.c_char_p, [_P]),
    "ds_set_stream": (_I, [_P, _P]),
    "ds_switch_stream": (_I, [_P, _P]),
    "ds_synchronize": (_I, [_P]),
    "ds_build_cost": (_I, [_P]),
    "ds_get_cost": (_I, [_P, _P]),
    "ds_iterate": (_I, [_P, _I]),
    "ds_iterate_raster": (_I, [_P, _I]),
    "ds_iterate_raster_async": (_I, [_P, _I]),
    "ds_sweep": (_I, [_P, _I, _I, _I]),
    "ds_fused_rows": (_I, [_P, _IP, _IP]),
    "ds_sweep_pair": (_I, [_P, _I, _I, _I, _I, _I, _I]),
    "ds_iterate_rows": (_I, [_P, _I, _I]),
    "ds_sweep_next": (_I, [_P, _I, _I]),
    "ds_halo_next": (_I, [_P, _I, _I, _PP, C.POINTER(C.c_size_t)]),
    "ds_finish_iterations": (_I, [_P]),
    "ds_energy": (_I, [_P, _DP, _DP]),
    "ds_energy_static": (_I, [_P, _P]),
    "ds_labels": (_I, [_P, _I, _P]),
    "ds_crossing_out": (_I, [_P, _D, _I, _P, _P, _IP, _IP]),
    "ds_crossing_out_exact": (_I, [_P, _D, _I, _P, _P, _IP, _IP]),
    "ds_find_best_labeling": (_I, [_P, _I, _P, _IP, C.POINTER(C.c_longlong)]),
    "ds_epsilon_gaps": (_I, [_P, C.POINTER(C.c_size_t), _IP]),
    "ds_epsilon_gaps_fetch": (_I, [_P, _P]),
    "ds_dedup_gaps": (_I, [_P, C.c_size_t, _D, _I, _P, C.POINTER(C.c_size_t)]),
    "ds_epsilon_search": (_I, [_P, _I, _I, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "ds_epsilon_next": (_I, [_P, _D, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
    "ds_co_begin": (_I, [_P, _D]),
    "ds_co_pass": (_I, [_P, _IP, _IP]),
    "ds_co_halo": (_I, [_P, _I, _I, _I, _PP, C.POINTER(C.c_size_t)]),
    "ds_co_finish": (_I, [_P, _I, _P, _P, _IP]),
    "ds_comm_unique_id": (_I, [_P]),
    "ds_comm_create": (_I, [_P, _I, _I, _I, _PP]),
    "ds_comm_destroy": (None, [_P]),
    "ds_comm_last_error": (C.c_char_p, [_P]),
    "ds_band_attach_comm": (_I, [_P, _P, _I, _I]),
    "ds_band_attach": (_I, [_P, _P, _I, _I]),
    "ds_band_exchange": (_I, [_P]),
    "ds_band_iterate": (_I, [_P, _I]),
    "ds_band_energy": (_I, [_P, _DP, _DP]),
    "ds_band_crossing_out": (_I, [_P, _D, _I, _P, _P, _IP, _IP]),
    "ds_get_phi_rows": (_I, [_P, _I, _I, _P]),
    "ds_set_phi_rows": (_I, [_P, _I, _I, _P]),
    "ds_get_phi": (_I, [_P, _P]),
    "ds_set_phi": (_I, [_P, _P]),
    "ds_halo": (_I, [_P, _I, _I, _I, _PP, C.POINTER(C.c_size_t)]),
    "ds_profile_begin": (_I, [_P, _I]),
    "ds_profile_end": (_I, [_P, _IP, _DP, _DP, _DP]),
    "ds_profile_split": (_I, [_P, _IP, _DP, _IP, _DP]),
    "ds_profile_kind": (_I, [_P, _I, _IP, _DP]),
    "ds_fused2_info": (_I, [_P, _IP, _IP, _IP, _IP, _IP, C.POINTER(C.c_longlong)]),
    "ds_info": (_I, [_P, _IP, _IP, _IP, _IP, _IP, _IP]),
    "ds_box_blur": (_I, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint64, _P]),
    "ds_placement_info": (_I, [_P, _IP, _DP, _IP]),
    "ds_checksum": (_I, [_P, C.POINTER(C.c_uint64)]),
    # the superpixel variant (superpixel.py)
    "ds_sp_create": (_I, [_P, _P, _I, _I, _I, _D, _I, _I, _I, _I, _PP]),
    "ds_sp_destroy": (None, [_P]),
    "ds_sp_last_error": (C.c_char_p, [_P]),
    "ds_sp_info": (_I, [_P, _IP, _IP, _IP, _IP]),
    "ds_sp_get_superpixels": (_I, [_P, _P]),
    "ds_sp_get_perimeters": (_I, [_P, _P]),
    "ds_sp_set_perimeters": (_I, [_P, _P]),
    "ds_sp_get_left_j": (_I, [_P, _P]),
    "ds_sp_get_cost": (_I, [_P, _P]),
    "ds_sp_get_phi": (_I, [_P, _P]),
    "ds_sp_set_phi": (_I, [_P, _P]),
    "ds_sp_iterate": (_I, [_P, _I, _I]),
    "ds_sp_energy": (_I, [_P, _DP]),
    "ds_sp_labels": (_I, [_P, _P]),
    "ds_sp_crossing_out": (_I, [_P, _D, _I, _P, _P, _IP, _IP]),
}

# struct ds_transport (include/diffstereo.h): a host-supplied way to move halo rows
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
LAST_ERROR_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p)
DS_REDUCE_SUM_F64, DS_REDUCE_MAX_F64, DS_REDUCE_MAX_I32, DS_REDUCE_MIN_I32 = 0, 1, 2, 3
DS_COMM_ID_BYTES = 128


class TransportStruct(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("exchange", EXCHANGE_FN), ("all_reduce", ALL_REDUCE_FN),
                ("last_error", LAST_ERROR_FN)]


_LIB = None


class DiffStereoError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("diffstereo error %d: %s" % (code, text))
        self.code = code


def lib():
    """dlopen the library and bind every declared symbol (raises if one is missing)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP library not built: %s (run __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def hip_runtime():
    """The HIP runtime the library itself is bound to, for callers that move device memory with
    hipMemcpy and the like.  A process can hold a second copy of libamdhip64 (torch bundles one):
    pointers and streams of one copy mean nothing to the other, so do not dlopen the runtime by
    name.  Symbols are looked up through the library's own dependencies."""
    L = lib()
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.hipFree.argtypes = [C.c_void_p]
    L.hipStreamSynchronize.argtypes = [C.c_void_p]
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DiffusionGraph:
    """Device-resident DiffusionGraph (the whole image, or one row band of it)."""

    def __init__(self, left_image, right_image, max_disparity, smoothing_term, dtype="f32",
                 row_begin=None, row_end=None, device=-1):
        self._h = None
        left = np.ascontiguousarray(left_image, dtype=np.uint8)
        right = np.ascontiguousarray(right_image, dtype=np.uint8)
        if left.ndim != 2 or left.shape != right.shape:
            raise ValueError("left and right images must be 2-D and of equal shape")
        self.H, self.W = left.shape
        self.D = int(max_disparity)
        self.dtype = dtype
        self._code, self._np = _DTYPES[dtype]
        self.row_begin = 0 if row_begin is None else int(row_begin)
        self.row_end = self.H if row_end is None else int(row_end)
        h = C.c_void_p()
        rc = lib().ds_create_band(_ptr(left), _ptr(right), self.H, self.W, self.D,
                                  float(smoothing_term), self._code, self.row_begin, self.row_end,
                                  int(device), C.byref(h))
        if rc != 0:
            raise DiffStereoError(rc, (lib().ds_last_error(None) or b"").decode())
        self._h = h
        self.rows = self.row_end - self.row_begin

    def _chk(self, rc):
        if rc != 0:
            raise DiffStereoError(rc, (lib().ds_last_error(self._h) or b"").decode())

    def close(self):
        if self._h is not None:
            lib().ds_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self._chk(lib().ds_set_stream(self._h, C.c_void_p(hip_stream)))

    def switch_stream(self, hip_stream):
        """set_stream without waiting for the stream being left (the caller orders them)."""
        self._chk(lib().ds_switch_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        self._chk(lib().ds_synchronize(self._h))

    def build_cost(self):
        self._chk(lib().ds_build_cost(self._h))

    def get_cost(self):
        out = np.empty((self.D, self.rows, self.W), dtype=np.uint8)
        self._chk(lib().ds_get_cost(self._h, _ptr(out)))
        return out

    def iterate(self, n_iters):
        """n_iters x diffusion_act (diffusion_graph.rs:264-293) under the red-black schedule."""
        self._chk(lib().ds_iterate(self._h, int(n_iters)))

    def iterate_raster(self, n_iters):
        """n_iters x diffusion_act in the reference's raster order (anti-diagonal launches)."""
        self._chk(lib().ds_iterate_raster(self._h, int(n_iters)))

    def iterate_raster_async(self, n_iters):
        """Notes n_iters more raster-order iterations; they run as one batch when the state is needed."""
        self._chk(lib().ds_iterate_raster_async(self._h, int(n_iters)))

    def sweep(self, colour, row_begin=None, row_end=None):
        self._chk(lib().ds_sweep(self._h, int(colour),
                                 self.row_begin if row_begin is None else int(row_begin),
                                 self.row_end if row_end is None else int(row_end)))

    # whole iterations on a band (see include/diffstereo.h)
    def fused_rows(self):
        a, b = C.c_int(), C.c_int()
        self._chk(lib().ds_fused_rows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def iterate_rows(self, row_begin, row_end):
        self._chk(lib().ds_iterate_rows(self._h, int(row_begin), int(row_end)))

    def sweep_next(self, row_begin, row_end):
        self._chk(lib().ds_sweep_next(self._h, int(row_begin), int(row_end)))

    def sweep_pair(self, colour, upcoming, rows1, rows2):
        self._chk(lib().ds_sweep_pair(self._h, int(colour), int(bool(upcoming)), int(rows1[0]), int(rows1[1]),
                                      int(rows2[0]), int(rows2[1])))

    def halo_next(self, side, which):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(lib().ds_halo_next(self._h, side, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def finish_iterations(self):
        self._chk(lib().ds_finish_iterations(self._h))

    def energy(self, with_residual=False):
        e, r = C.c_double(), C.c_double()
        self._chk(lib().ds_energy(self._h, C.byref(e), C.byref(r)))
        return (e.value, r.value) if with_residual else e.value

    def energy_static(self, result_dev_ptr):
        """ds_energy without any allocation: queues the reduction on the handle's stream; its
        last kernel writes (energy, max residual) to the two doubles at the device pointer."""
        self._chk(lib().ds_energy_static(self._h, C.c_void_p(result_dev_ptr)))

    def labels(self, mode=0):
        out = np.empty((self.rows, self.W), dtype=np.uint32)
        self._chk(lib().ds_labels(self._h, int(mode), _ptr(out)))
        return out

    def crossing_out(self, epsilon, max_passes=100000, with_mask=False, exact=False):
        """initialize_with_epsilon + crossing_out + simple_best_labeling on the GPU
        (crossing_out_graph.rs:67-183,307-327).  Returns dict(labels, not_empty, passes[, alive]).
        exact: every comparison as the reference makes it (O(D^2) per pixel and pass)."""
        labels = np.empty((self.rows, self.W), dtype=np.uint32)
        alive = np.empty((self.rows, self.W, self.D), dtype=np.uint8) if with_mask else None
        ne, n = C.c_int(), C.c_int()
        fn = lib().ds_crossing_out_exact if exact else lib().ds_crossing_out
        self._chk(fn(self._h, float(epsilon), int(max_passes), _ptr(labels),
                     _ptr(alive) if with_mask else None, C.byref(ne), C.byref(n)))
        out = {"labels": labels, "not_empty": bool(ne.value), "passes": n.value}
        if with_mask:
            out["alive"] = alive
        return out

    def find_best_labeling(self, max_passes=100000):
        """crossing_out_graph.rs:349-364 on the masks left by crossing_out(): one label per pixel,
        fixed in raster order with a crossing-out after each choice.  dict(labels, not_empty, choices)."""
        labels = np.empty((self.rows, self.W), dtype=np.uint32)
        ne, n = C.c_int(), C.c_longlong()
        self._chk(lib().ds_find_best_labeling(self._h, int(max_passes), _ptr(labels), C.byref(ne), C.byref(n)))
        return {"labels": labels, "not_empty": bool(ne.value), "choices": n.value}

    def epsilon_gaps(self, with_flag=False):
        """create_array_of_epsilons before its tolerance dedup (epsilon_search.rs:39-74): all
        vertex and edge penalty gaps of the current phi, sorted, exact duplicates removed (f64).
        with_flag: also whether the largest value occurred more than once."""
        n, rep = C.c_size_t(), C.c_int()
        self._chk(lib().ds_epsilon_gaps(self._h, C.byref(n), C.byref(rep)))
        out = np.empty(n.value, dtype=np.float64)
        self._chk(lib().ds_epsilon_gaps_fetch(self._h, _ptr(out) if n.value else None))
        return (out, bool(rep.value)) if with_flag else out

    def create_array_of_epsilons(self, tolerance):
        """epsilon_search.rs:39-74 element for element: the gaps, then utils.rs:137-163."""
        gaps, repeated = self.epsilon_gaps(with_flag=True)
        out = np.empty(len(gaps) + 1, dtype=np.float64)
        n = C.c_size_t()
        self._chk(lib().ds_dedup_gaps(_ptr(gaps) if len(gaps) else None, len(gaps), float(tolerance),
                                     int(repeated), _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def epsilon_search(self, exact=True, max_passes=100000, with_stats=False):
        """epsilon_search(create_array_of_epsilons(graph, 0.0)) (epsilon_search.rs:39-112) without
        the array: the smallest gap for which the crossing-out leaves a non-empty graph.
        exact: the predicate compares as the reference does (crossing_out_exact), else crossing_out."""
        eps = C.c_double()
        stats = (C.c_longlong * 3)()
        self._chk(lib().ds_epsilon_search(self._h, int(bool(exact)), int(max_passes), C.byref(eps), stats))
        if with_stats:
            return eps.value, {"crossing_outs": stats[0], "gap_passes": stats[1], "relaxation_passes": stats[2]}
        return eps.value

    def epsilon_next(self, lo):
        """One pass over all gaps: (smallest gap > lo, largest gap, number of gaps <= lo)."""
        nxt, largest, count = C.c_double(), C.c_double(), C.c_ulonglong()
        self._chk(lib().ds_epsilon_next(self._h, float(lo), C.byref(nxt), C.byref(largest), C.byref(count)))
        return nxt.value, largest.value, count.value

    # crossing-out in steps (row bands; see include/diffstereo.h)
    def co_begin(self, epsilon):
        self._chk(lib().ds_co_begin(self._h, float(epsilon)))

    def co_pass(self):
        ch, lo = C.c_int(), C.c_int()
        self._chk(lib().ds_co_pass(self._h, C.byref(ch), C.byref(lo)))
        return bool(ch.value), bool(lo.value)

    def co_halo(self, colour, side, which):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(lib().ds_co_halo(self._h, colour, side, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def co_finish(self, empty, with_mask=False):
        labels = np.empty((self.rows, self.W), dtype=np.uint32)
        alive = np.empty((self.rows, self.W, self.D), dtype=np.uint8) if with_mask else None
        ne = C.c_int()
        self._chk(lib().ds_co_finish(self._h, int(bool(empty)), _ptr(labels),
                                     _ptr(alive) if with_mask else None, C.byref(ne)))
        out = {"labels": labels, "not_empty": bool(ne.value)}
        if with_mask:
            out["alive"] = alive
        return out

    # row bands over several ranks: the protocol runs inside the library (include/diffstereo.h)
    def band_attach_comm(self, comm_ptr, up_rank, down_rank):
        self._chk(lib().ds_band_attach_comm(self._h, comm_ptr, int(up_rank), int(down_rank)))

    def band_attach(self, transport_struct, up_rank, down_rank):
        self._chk(lib().ds_band_attach(self._h, C.byref(transport_struct), int(up_rank), int(down_rank)))

    def band_exchange(self):
        self._chk(lib().ds_band_exchange(self._h))

    def band_iterate(self, n_iters):
        self._chk(lib().ds_band_iterate(self._h, int(n_iters)))

    def band_energy(self, with_residual=False):
        e, r = C.c_double(), C.c_double()
        self._chk(lib().ds_band_energy(self._h, C.byref(e), C.byref(r)))
        return (e.value, r.value) if with_residual else e.value

    def band_crossing_out(self, epsilon, max_passes=1 << 30, with_mask=False):
        labels = np.empty((self.rows, self.W), dtype=np.uint32)
        alive = np.empty((self.rows, self.W, self.D), dtype=np.uint8) if with_mask else None
        ne, n = C.c_int(), C.c_int()
        self._chk(lib().ds_band_crossing_out(self._h, float(epsilon), int(max_passes), _ptr(labels),
                                             _ptr(alive) if with_mask else None, C.byref(ne), C.byref(n)))
        out = {"labels": labels, "not_empty": bool(ne.value), "passes": n.value}
        if with_mask:
            out["alive"] = alive
        return out

    def get_phi(self, row_begin=None, row_end=None):
        rb = self.row_begin if row_begin is None else int(row_begin)
        re = self.row_end if row_end is None else int(row_end)
        out = np.empty((re - rb, self.W, 4, self.D), dtype=self._np)
        self._chk(lib().ds_get_phi_rows(self._h, rb, re, _ptr(out)))
        return out

    def set_phi(self, phi, row_begin=None, row_end=None):
        rb = self.row_begin if row_begin is None else int(row_begin)
        re = self.row_end if row_end is None else int(row_end)
        phi = np.ascontiguousarray(phi, dtype=self._np)
        if phi.shape != (re - rb, self.W, 4, self.D):
            raise ValueError("phi must have shape [rows][W][4][D]")
        self._chk(lib().ds_set_phi_rows(self._h, rb, re, _ptr(phi)))

    def halo(self, colour, side, which):
        """(device pointer, bytes) of a halo row; side 0=top 1=bottom, which 0=send 1=recv."""
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(lib().ds_halo(self._h, colour, side, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def profile_begin(self, max_launches):
        self._chk(lib().ds_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        n, tot, lo, hi = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        self._chk(lib().ds_profile_end(self._h, C.byref(n), C.byref(tot), C.byref(lo), C.byref(hi)))
        return {"launches": n.value, "total_ms": tot.value, "min_ms": lo.value, "max_ms": hi.value}

    def profile_split(self):
        """After profile_end: launches and milliseconds by kind."""
        n0, t0, n1, t1 = C.c_int(), C.c_double(), C.c_int(), C.c_double()
        self._chk(lib().ds_profile_split(self._h, C.byref(n0), C.byref(t0), C.byref(n1), C.byref(t1)))
        return {"half_sweeps": n0.value, "half_sweep_ms": t0.value, "fused": n1.value, "fused_ms": t1.value}

    def box_blur(self, disparity_map, threshold, filter_radius, normalization_term):
        """DiffusionGraph::box_blur (diffusion_graph.rs:399-424), release-build arithmetic.  u64 [H][W]."""
        m = np.ascontiguousarray(disparity_map, dtype=np.uint32)
        if m.shape != (self.H, self.W):
            raise ValueError("disparity_map must have the image's shape")
        out = np.empty(m.shape, dtype=np.uint64)
        self._chk(lib().ds_box_blur(self._h, _ptr(m), int(threshold), int(filter_radius), int(normalization_term), _ptr(out)))
        return out

    def profile_kind(self, kind):
        """After profile_end: (launches, milliseconds) of one kind: 0 half-sweeps, 1 one-iteration
        fused launches, 2 two-iteration launches."""
        n, ms = C.c_int(), C.c_double()
        self._chk(lib().ds_profile_kind(self._h, int(kind), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def fused2_info(self):
        v = [C.c_int() for _ in range(5)]
        steps = C.c_longlong()
        self._chk(lib().ds_fused2_info(self._h, *[C.byref(x) for x in v], C.byref(steps)))
        return {"available": bool(v[0].value), "cols": (v[1].value, v[2].value), "rows": (v[3].value, v[4].value),
                "steps": steps.value}

    def placement_info(self):
        n, ms, sp = C.c_int(), C.c_double(), C.c_int()
        self._chk(lib().ds_placement_info(self._h, C.byref(n), C.byref(ms), C.byref(sp)))
        return {"tries": n.value, "probe_ms": ms.value, "split": bool(sp.value)}

    def checksum(self):
        v = C.c_uint64()
        self._chk(lib().ds_checksum(self._h, C.byref(v)))
        return v.value


def host_checksum(phi, row_begin, H, W, D):
    """The same order-independent checksum as ds_checksum, from a host array
    phi[rows][W][4][D] holding image rows [row_begin, row_begin + rows)."""
    phi = np.ascontiguousarray(phi)
    rows = phi.shape[0]
    ii = (np.arange(rows) + row_begin)[:, None, None, None]
    jj = np.arange(W)[None, :, None, None]
    dd = np.arange(D)[None, None, None, :]
    has = np.stack([np.broadcast_to(jj > 0, (rows, W, 1, 1)), np.broadcast_to(ii > 0, (rows, W, 1, 1)),
                    np.broadcast_to(jj + 1 < W, (rows, W, 1, 1)),
                    np.broadcast_to(ii + 1 < H, (rows, W, 1, 1))], axis=2).reshape(rows, W, 4, 1)
    mask = has & (dd <= jj)
    vals = np.where(phi == 0, 0, phi).astype(phi.dtype)  # -0 -> +0
    bits = vals.view(np.uint32 if phi.dtype == np.float32 else np.uint64).astype(np.uint64)
    return int(np.sum(np.where(mask, bits, np.uint64(0)), dtype=np.uint64))
