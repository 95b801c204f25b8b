"""Host mirror of the Frame post-processing between extractor and matcher (code/src/Frame.cc): thin ctypes
binding of so_frame_* in include/swarmorb.h.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib

GRID_COLS, GRID_ROWS = 64, 48  # FRAME_GRID_COLS / ROWS, code/include/Frame.h:37-38


class SoCamera(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FramePostProcessor:
    """K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3]) as in the settings yaml (Tracking.cc:60-84)."""

    def __init__(self, K, dist=(0, 0, 0, 0, 0), device=0):
        self._lib = _lib.load_library()
        vp, f, i32 = C.c_void_p, C.c_float, C.c_int32
        self._lib.so_frame_create.argtypes = [C.c_int, C.POINTER(vp)]
        self._lib.so_frame_destroy.argtypes = [vp]
        self._lib.so_frame_destroy.restype = None
        self._lib.so_frame_prepare.argtypes = [vp, C.POINTER(SoCamera), i32, i32, C.c_int, i32, vp, vp, vp, vp, vp, vp,
                                               C.POINTER(i32)]
        self._lib.so_frame_is_in_frustum.argtypes = [vp, C.POINTER(SoCamera), vp, vp, i32, vp, vp, vp, vp, f, f, i32,
                                                     vp, vp, vp, vp, vp]
        d = list(dist) + [0.0] * (5 - len(dist))
        self.cam = SoCamera(*[float(v) for v in K], *[float(v) for v in d])
        self._h = vp()
        _lib.check(self._lib.so_frame_create(int(device), C.byref(self._h)))
        self.bounds = None  # mnMinX, mnMaxX, mnMinY, mnMaxY once computed (static across frames, Frame.cc:247-263)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_frame_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def prepare(self, xy, width, height, grid=True):
        """UndistortKeyPoints (+ ComputeImageBounds on the first call) + AssignFeaturesToGrid.
        Returns dict(xy_un, bounds, cell_of, cell_start, cell_items)."""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        n = len(xy)
        xy_un = np.zeros_like(xy)
        first = self.bounds is None
        bounds = np.zeros(4, np.float32) if first else self.bounds.copy()
        cell_of = np.zeros(n, np.int32)
        cell_start = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
        cell_items = np.zeros(max(n, 1), np.int32)
        n_in = C.c_int32(0)
        g = (cell_of, cell_start, cell_items) if grid else (None, None, None)
        _lib.check(self._lib.so_frame_prepare(self._h, C.byref(self.cam), int(width), int(height), int(first), n,
                                              _vp(xy), _vp(xy_un), _vp(bounds), _vp(g[0]), _vp(g[1]), _vp(g[2]),
                                              C.byref(n_in) if grid else None))
        self.bounds = bounds
        out = dict(xy_un=xy_un, bounds=bounds.copy())
        if grid:
            out.update(cell_of=cell_of, cell_start=cell_start, cell_items=cell_items[:n_in.value])
        return out

    def is_in_frustum(self, Tcw, Xw, normal, max_dist, min_dist, viewing_cos_limit, log_scale_factor, n_levels,
                      bounds=None, init=None):
        """Frame::isInFrustum over a batch; init = (proj_x, proj_y, view_cos, pred_level) values kept where rejected."""
        b = np.ascontiguousarray(self.bounds if bounds is None else bounds, np.float32)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
        X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
        N = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        mx = np.ascontiguousarray(max_dist, np.float32)
        mn = np.ascontiguousarray(min_dist, np.float32)
        n = len(X)
        in_view = np.zeros(n, np.uint8)
        if init is None:
            px, py, vc, lvl = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        else:
            px, py, vc, lvl = [np.ascontiguousarray(a, t).copy() for a, t in zip(init, (np.float32,) * 3 + (np.int32,))]
        _lib.check(self._lib.so_frame_is_in_frustum(self._h, C.byref(self.cam), _vp(b), _vp(T), n, _vp(X), _vp(N),
                                                    _vp(mx), _vp(mn), float(viewing_cos_limit), float(log_scale_factor),
                                                    int(n_levels), _vp(in_view), _vp(px), _vp(py), _vp(vc), _vp(lvl)))
        return dict(in_view=in_view, proj_x=px, proj_y=py, view_cos=vc, pred_level=lvl)
