"""ctypes binding of the superpixel variant (ds_sp_* in include/diffstereo.h): the reference's
crate superpixel_stereo on the GPU.  `SuperpixelGraph` keeps the crate's names where one exists
(superpixel_stereo/src/{superpixels,diffusion_graph,crossing_out_graph}.rs).  No CPU fallback."""
import ctypes as C

import numpy as np

from . import diffstereo
from .diffstereo import DiffStereoError, _DTYPES, _ptr, lib

FOUR_COLOURS, RASTER = 0, 1


class SuperpixelGraph:
    """SuperpixelRepresentation::initialize + split_into_superpixels (+ calculate_edge_perimeters
    with perimeters=True) + DiffusionGraph::initialize."""

    def __init__(self, left_image, right_image, max_disparity, smoothing_term, super_height, super_width,
                 perimeters=True, dtype="f64"):
        left = np.ascontiguousarray(left_image, dtype=np.uint8)
        right = np.ascontiguousarray(right_image, dtype=np.uint8)
        if left.shape != right.shape or left.ndim != 2:
            raise ValueError("left and right images must be 2-D and of equal shape")
        self.H, self.W = left.shape
        self.D, self.dtype = int(max_disparity), dtype
        self._code, self._np = _DTYPES[dtype]
        self._h = C.c_void_p()
        rc = lib().ds_sp_create(_ptr(left), _ptr(right), self.H, self.W, self.D, float(smoothing_term), int(super_height),
                                int(super_width), 1 if perimeters else 0, self._code, C.byref(self._h))
        if rc != 0:
            raise DiffStereoError(rc, (lib().ds_sp_last_error(None) or b"").decode())
        ni, nj = C.c_int(), C.c_int()
        self._chk(lib().ds_sp_info(self._h, C.byref(ni), C.byref(nj), None, None))
        self.NI, self.NJ = ni.value, nj.value

    def _chk(self, rc):
        if rc != 0:
            raise DiffStereoError(rc, (lib().ds_sp_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib().ds_sp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def superpixels(self):
        out = np.empty((self.H, self.W), dtype=np.uint8)
        self._chk(lib().ds_sp_get_superpixels(self._h, _ptr(out)))
        return out

    def edge_perimeters(self):
        out = np.empty((self.NI, self.NJ, 2, 9), dtype=np.float64)
        self._chk(lib().ds_sp_get_perimeters(self._h, _ptr(out)))
        return out

    def set_edge_perimeters(self, perimeters):
        p = np.ascontiguousarray(perimeters, dtype=np.float64)
        if p.shape != (self.NI, self.NJ, 2, 9):
            raise ValueError("edge_perimeters must be [windows down][windows across][2][9]")
        self._chk(lib().ds_sp_set_perimeters(self._h, _ptr(p)))

    def left_j(self):
        """min(left_j_in_superpixel, D - 1) per node."""
        out = np.empty((self.NI, self.NJ, 2), dtype=np.int32)
        self._chk(lib().ds_sp_get_left_j(self._h, _ptr(out)))
        return out

    def cost(self):
        out = np.empty((self.NI, self.NJ, 2, self.D), dtype=np.uint32)
        self._chk(lib().ds_sp_get_cost(self._h, _ptr(out)))
        return out

    def get_phi(self):
        out = np.empty((self.NI, self.NJ, 2, 9, self.D), dtype=self._np)
        self._chk(lib().ds_sp_get_phi(self._h, _ptr(out)))
        return out

    def set_phi(self, phi):
        p = np.ascontiguousarray(phi, dtype=self._np)
        if p.shape != (self.NI, self.NJ, 2, 9, self.D):
            raise ValueError("potentials must be [windows down][windows across][2][9][D]")
        self._chk(lib().ds_sp_set_phi(self._h, _ptr(p)))

    def iterate(self, n_iters, schedule=FOUR_COLOURS):
        self._chk(lib().ds_sp_iterate(self._h, int(n_iters), int(schedule)))

    def energy(self):
        e = C.c_double()
        self._chk(lib().ds_sp_energy(self._h, C.byref(e)))
        return e.value

    def build_depth_map(self):
        out = np.empty((self.H, self.W), dtype=np.uint32)
        self._chk(lib().ds_sp_labels(self._h, _ptr(out)))
        return out

    def crossing_out(self, epsilon, max_passes=100000, with_mask=False):
        labels = np.empty((self.H, self.W), dtype=np.uint32)
        alive = np.empty((self.NI, self.NJ, 2, self.D), dtype=np.uint8) if with_mask else None
        ok, passes = C.c_int(), C.c_int()
        self._chk(lib().ds_sp_crossing_out(self._h, float(epsilon), int(max_passes), _ptr(labels),
                                           _ptr(alive) if with_mask else None, C.byref(ok), C.byref(passes)))
        return {"labels": labels, "alive": alive, "not_empty": bool(ok.value), "passes": passes.value}


__all__ = ["SuperpixelGraph", "FOUR_COLOURS", "RASTER", "diffstereo"]
