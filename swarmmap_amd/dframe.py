"""Host mirror of the device-resident frame, the device-resident map-point table and the tracking searches over them
(so_dframe_* / so_map_* / so_track_search_* in include/swarmorb.h).  Thin ctypes binding; no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from .extractor import KP_DTYPE
from .frame import GRID_COLS, GRID_ROWS, SoCamera


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bind(lib):
    if getattr(lib, "_dframe_bound", False):
        return
    vp, i32, f, ip = C.c_void_p, C.c_int32, C.c_float, C.POINTER(C.c_int32)
    lib.so_dframe_create.argtypes = [vp, C.POINTER(SoCamera), C.POINTER(vp)]
    lib.so_dframe_destroy.argtypes = [vp]
    lib.so_dframe_destroy.restype = None
    lib.so_dframe_submit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.so_dframe_submit_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.so_dframe_collect.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int), vp]
    lib.so_dframe_get_grid.argtypes = [vp, vp, vp, ip]
    lib.so_map_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.so_map_destroy.argtypes = [vp]
    lib.so_map_destroy.restype = None
    lib.so_map_size.argtypes = [vp]
    lib.so_map_write.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.so_map_write_positions.argtypes = [vp, i32, vp, vp]
    lib.so_map_read.argtypes = [vp, i32, i32, vp, vp]
    lib.so_search_by_projection_mappoints_dframe.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, f, f, vp, ip]
    lib.so_search_by_projection_lastframe_dframe.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, f, C.c_int,
                                                             vp, ip]
    lib.so_track_search_last_frame.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, f, C.c_int, vp, ip]
    lib.so_track_search_local_map.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, f, f, f, f, vp, vp, ip]
    lib.so_track_stage_last_frame_submit.argtypes = [vp, vp, vp, vp, vp, vp, f, C.c_int, vp, vp]
    lib.so_track_stage_local_map_submit.argtypes = [vp, vp, vp, C.c_int, vp, vp, i32, vp, i32, vp, f, f, f, f, vp, vp]
    lib.so_track_stage_pose_again_submit.argtypes = [vp, vp]
    lib.so_track_stage_wait.argtypes = [vp, vp, ip, vp, ip, vp, vp, vp, ip, vp]
    lib._dframe_bound = True


class DeviceFrame:
    """ORB_SLAM2::Frame's constructor (Frame.cc:218-275) with the results kept in HBM; bound to one ORBextractor."""

    def __init__(self, extractor, K, dist=(0, 0, 0, 0, 0)):
        self._lib = _lib.load_library()
        _bind(self._lib)
        self._ex = extractor  # keeps the extractor alive
        d = list(dist) + [0.0] * (5 - len(dist))
        self.cam = SoCamera(*[float(v) for v in K], *[float(v) for v in d])
        self._h = C.c_void_p()
        _lib.check(self._lib.so_dframe_create(extractor._h, C.byref(self.cam), C.byref(self._h)))
        cap = extractor._cap
        self._cap = cap
        self._kps = np.zeros(cap, KP_DTYPE)
        self._desc = np.zeros((cap, 32), np.uint8)
        self._xy_un = np.zeros((cap, 2), np.float32)
        self.bounds = np.zeros(4, np.float32)
        self.n = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_dframe_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def submit(self, image):
        if image.dtype != np.uint8 or image.ndim != 2 or image.strides[1] != 1:
            raise ValueError("image must be a row-contiguous CV_8UC1 array")
        self._inflight = image
        _lib.check(self._lib.so_dframe_submit(self._h, _vp(image), image.shape[1], image.shape[0], image.strides[0]))

    def submit_device(self, d_ptr, width, height, stride):
        _lib.check(self._lib.so_dframe_submit_device(self._h, C.c_void_p(d_ptr), width, height, stride))

    def collect(self):
        """Returns (keypoints, xy_un, descriptors): host copies of mvKeys, mvKeysUn[i].pt and mDescriptors."""
        n = C.c_int(0)
        _lib.check(self._lib.so_dframe_collect(self._h, _vp(self._kps), _vp(self._xy_un), _vp(self._desc), self._cap,
                                               C.byref(n), _vp(self.bounds)))
        self._inflight = None
        self.n = n.value
        return self._kps[:self.n], self._xy_un[:self.n], self._desc[:self.n]

    def __call__(self, image):
        self.submit(image)
        return self.collect()

    def grid(self):
        cs = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
        items = np.zeros(max(self.n, 1), np.int32)
        n_in = C.c_int32(0)
        _lib.check(self._lib.so_dframe_get_grid(self._h, _vp(cs), _vp(items), C.byref(n_in)))
        return cs, items[:n_in.value].copy()


class DeviceMap:
    """MapPoint fields read by the per-frame operators, resident in HBM, indexed by slot."""

    def __init__(self, device=0):
        self._lib = _lib.load_library()
        _bind(self._lib)
        self._h = C.c_void_p()
        _lib.check(self._lib.so_map_create(int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_map_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __len__(self):
        return int(self._lib.so_map_size(self._h))

    def write(self, first, Xw=None, normal=None, max_dist=None, min_dist=None, desc=None):
        arrs = [None if a is None else np.ascontiguousarray(a, t) for a, t in
                ((Xw, np.float32), (normal, np.float32), (max_dist, np.float32), (min_dist, np.float32), (desc, np.uint8))]
        n = next(len(a.reshape(-1, w)) for a, w in zip(arrs, (3, 3, 1, 1, 32)) if a is not None)
        _lib.check(self._lib.so_map_write(self._h, int(first), n, *[_vp(a) for a in arrs]))

    def append(self, Xw, normal, max_dist, min_dist, desc):
        first = len(self)
        self.write(first, Xw, normal, max_dist, min_dist, desc)
        return first

    def write_positions(self, slots, Xw):
        s = np.ascontiguousarray(slots, np.int32)
        X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
        _lib.check(self._lib.so_map_write_positions(self._h, len(s), _vp(s), _vp(X)))

    def write_rows(self, slots, Xw=None, normal=None, max_dist=None, min_dist=None):
        s = np.ascontiguousarray(slots, np.int32)
        arrs = [None if a is None else np.ascontiguousarray(a, np.float32) for a in (Xw, normal, max_dist, min_dist)]
        self._lib.so_map_write_rows.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        _lib.check(self._lib.so_map_write_rows(self._h, len(s), _vp(s), *[_vp(a) for a in arrs]))

    def read(self, first, n):
        X, d = np.zeros((n, 3), np.float32), np.zeros((n, 32), np.uint8)
        _lib.check(self._lib.so_map_read(self._h, int(first), int(n), _vp(X), _vp(d)))
        return X, d


def search_last_frame(matcher, cur, last, dmap, Tcw, last_slot, th, excluded=None, has_obs=None):
    """so_track_search_last_frame: (nmatches, kp_to_last)."""
    lib = matcher._lib
    _bind(lib)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    slot = np.ascontiguousarray(last_slot, np.int32)
    assert len(slot) == last.n
    ex = None if excluded is None else np.ascontiguousarray(excluded, np.uint8)
    ho = None if has_obs is None else np.ascontiguousarray(has_obs, np.uint8)
    out = np.full(cur.n, -1, np.int32)
    nm = C.c_int32(0)
    _lib.check(lib.so_track_search_last_frame(matcher._h, cur._h, _vp(ex), last._h, dmap._h, _vp(T), _vp(slot), _vp(ho),
                                              float(th), int(matcher.mbCheckOrientation), _vp(out), C.byref(nm)))
    return nm.value, out


def search_local_map(matcher, cur, dmap, Tcw, n_local, th, cos_limit, log_scale_factor, local_slot=None, skip=None,
                     excluded=None, has_obs=None, first_slot=0):
    """so_track_search_local_map: (nmatches, kp_to_local, in_view)."""
    lib = matcher._lib
    _bind(lib)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    slot = None if local_slot is None else np.ascontiguousarray(local_slot, np.int32)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    ex = None if excluded is None else np.ascontiguousarray(excluded, np.uint8)
    ho = None if has_obs is None else np.ascontiguousarray(has_obs, np.uint8)
    in_view = np.zeros(max(n_local, 1), np.uint8)
    out = np.full(cur.n, -1, np.int32)
    nm = C.c_int32(0)
    _lib.check(lib.so_track_search_local_map(matcher._h, cur._h, _vp(ex), dmap._h, _vp(T), int(n_local), _vp(slot),
                                             int(first_slot), _vp(sk), _vp(ho), float(th), float(matcher.mfNNratio), float(cos_limit),
                                             float(log_scale_factor), _vp(in_view), _vp(out), C.byref(nm)))
    return nm.value, out, in_view[:n_local]


SO_RETRY_ON_HOST = 100


def _stage_wait(matcher, n_kp, n_local=0, again=False):
    """so_track_stage_wait: None when the stage was not finished on the device (SO_RETRY_ON_HOST), else a dict."""
    lib = matcher._lib
    k2q = np.full(n_kp, -1, np.int32)
    view = np.zeros(max(n_local, 1), np.uint8)
    edge_kp, edge_out = np.zeros(n_kp, np.int32), np.zeros(n_kp, np.uint8)
    T = np.zeros(12, np.float32)
    nm, ne, ninl = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    info2 = np.zeros(2, np.int32)
    rc = lib.so_track_stage_wait(matcher._h, None if again else _vp(k2q), None if again else C.byref(nm), _vp(view) if n_local else None,
                                 C.byref(ne), _vp(edge_kp), _vp(edge_out), _vp(T), C.byref(ninl), _vp(info2))
    if rc == SO_RETRY_ON_HOST:
        return None
    _lib.check(rc)
    rounds, active = C.c_int32(0), C.c_int32(0)
    lib.so_track_stage_last_rounds.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.so_track_stage_last_rounds(matcher._h, C.byref(rounds), C.byref(active))
    return dict(rounds=rounds.value, active_queries=active.value, nmatches=nm.value, kp_to_q=k2q, in_view=view[:n_local], n_edges=ne.value, edge_kp=edge_kp[:ne.value].copy(),
                edge_outlier=edge_out[:ne.value].copy(), Tcw=T.reshape(3, 4), n_inliers=ninl.value, iterations=int(info2[0]),
                trials=int(info2[1]))


class TrackGroup:
    """so_track_group: the tracking stages of several agents' matchers as one chain of launches.  Members record their
    stages (track_stage_*(..., wait=False) returns the wait as a callable), launch() issues them."""

    def __init__(self, matchers, device=0):
        self._lib = _lib.load_library()
        self._h = C.c_void_p()
        self._lib.so_track_group_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        self._lib.so_matcher_set_track_group.argtypes = [C.c_void_p, C.c_void_p]
        self._lib.so_track_group_launch.argtypes = [C.c_void_p]
        self._lib.so_track_group_pending.argtypes = [C.c_void_p]
        self._lib.so_track_group_destroy.argtypes = [C.c_void_p]
        self._lib.so_track_group_destroy.restype = None
        self._lib.so_track_group_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.check(self._lib.so_track_group_create(device, C.byref(self._h)))
        self.members = list(matchers)
        for m in self.members:
            _lib.check(self._lib.so_matcher_set_track_group(m._h, self._h))

    def pending(self):
        return self._lib.so_track_group_pending(self._h)

    def launch(self):
        _lib.check(self._lib.so_track_group_launch(self._h))

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _lib.check(self._lib.so_track_group_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            for m in self.members:
                if m._h:
                    self._lib.so_matcher_set_track_group(m._h, None)
            self._lib.so_track_group_destroy(self._h)
            self._h = None


def track_stage_last_frame(matcher, cur, last, dmap, Tcw, last_slot, th, K4, level_inv_sigma2, wait=True):
    """so_track_stage_last_frame_submit + so_track_stage_wait: TrackWithMotionModel's search, resolve and PoseOptimization as one
    chain of launches.  None: the caller takes search_last_frame + PoseOptimization.  wait=False: the wait as a callable."""
    lib = matcher._lib
    _bind(lib)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    slot = np.ascontiguousarray(last_slot, np.int32)
    assert len(slot) == last.n
    k4, ls = np.ascontiguousarray(K4, np.float32), np.ascontiguousarray(level_inv_sigma2, np.float32)
    rc = lib.so_track_stage_last_frame_submit(matcher._h, cur._h, last._h, dmap._h, _vp(T), _vp(slot), float(th),
                                              int(matcher.mbCheckOrientation), _vp(k4), _vp(ls))
    if rc == SO_RETRY_ON_HOST:
        return None
    _lib.check(rc)
    n = cur.n
    return _stage_wait(matcher, n) if wait else (lambda: _stage_wait(matcher, n))


def track_stage_local_map(matcher, cur, kp_slot, dmap, Tcw, n_local, th, cos_limit, log_scale_factor, K4, level_inv_sigma2,
                          local_slot=None, skip=None, first_slot=0, kp_slot_is_last_stage=False, wait=True):
    """so_track_stage_local_map_submit + so_track_stage_wait (TrackLocalMap: SearchLocalPoints + PoseOptimization)."""
    lib = matcher._lib
    _bind(lib)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    ks = np.ascontiguousarray(kp_slot, np.int32)
    assert len(ks) == cur.n
    slot = None if local_slot is None else np.ascontiguousarray(local_slot, np.int32)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    k4, ls = np.ascontiguousarray(K4, np.float32), np.ascontiguousarray(level_inv_sigma2, np.float32)
    rc = lib.so_track_stage_local_map_submit(matcher._h, cur._h, _vp(ks), int(bool(kp_slot_is_last_stage)), dmap._h, _vp(T), int(n_local),
                                             _vp(slot), int(first_slot),
                                             _vp(sk), float(th), float(matcher.mfNNratio), float(cos_limit), float(log_scale_factor),
                                             _vp(k4), _vp(ls))
    if rc == SO_RETRY_ON_HOST:
        return None
    _lib.check(rc)
    n = cur.n
    return _stage_wait(matcher, n, n_local) if wait else (lambda: _stage_wait(matcher, n, n_local))


def track_stage_local_map_after(matcher, first, cur, dmap, n_local, th, cos_limit, log_scale_factor, K4, level_inv_sigma2, local_slot=None,
                                skip_static=None, first_slot=0):
    """so_track_stage_local_map_submit_after: TrackLocalMap's stage enqueued behind `first`'s in-flight last-frame stage (no host
    in between).  Returns the wait as a callable taking the first stage's pose (so_track_stage_set_start_pose), or None."""
    lib = matcher._lib
    _bind(lib)
    vp, i32, f = C.c_void_p, C.c_int32, C.c_float
    lib.so_track_stage_local_map_submit_after.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, f, f, f, f, vp, vp]
    lib.so_track_stage_set_start_pose.argtypes = [vp, vp]
    slot = None if local_slot is None else np.ascontiguousarray(local_slot, np.int32)
    sk = None if skip_static is None else np.ascontiguousarray(skip_static, np.uint8)
    k4, ls = np.ascontiguousarray(K4, np.float32), np.ascontiguousarray(level_inv_sigma2, np.float32)
    rc = lib.so_track_stage_local_map_submit_after(matcher._h, first._h, cur._h, dmap._h, int(n_local), _vp(slot), int(first_slot), _vp(sk), float(th),
                                                   float(matcher.mfNNratio), float(cos_limit), float(log_scale_factor), _vp(k4), _vp(ls))
    if rc == SO_RETRY_ON_HOST:
        return None
    _lib.check(rc)
    n = cur.n
    keep = (slot, sk, k4, ls)  # (the submit read them in place: alive until the wait)

    def wait(first_Tcw, _keep=keep):
        T = np.ascontiguousarray(first_Tcw, np.float32).reshape(12)
        _lib.check(lib.so_track_stage_set_start_pose(matcher._h, _vp(T)))
        return _stage_wait(matcher, n, n_local)
    return wait


def track_stage_pose_again(matcher, cur, Tcw, wait=True):
    """so_track_stage_pose_again_submit + wait: PoseOptimization over the last stage's edges from another start pose."""
    lib = matcher._lib
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    _lib.check(lib.so_track_stage_pose_again_submit(matcher._h, _vp(T)))
    n = cur.n
    return _stage_wait(matcher, n, again=True) if wait else (lambda: _stage_wait(matcher, n, again=True))


def search_mappoints_dframe(matcher, cur, mps, th, excluded=None):
    """so_search_by_projection_mappoints_dframe (host-side queries, device-resident candidates)."""
    lib = matcher._lib
    _bind(lib)
    a = {k: np.ascontiguousarray(mps[k], t) for k, t in
         (("in_view", np.uint8), ("proj_x", np.float32), ("proj_y", np.float32), ("view_cos", np.float32),
          ("pred_level", np.int32), ("desc", np.uint8), ("has_obs", np.uint8))}
    ex = None if excluded is None else np.ascontiguousarray(excluded, np.uint8)
    out = np.full(cur.n, -1, np.int32)
    nm = C.c_int32(0)
    _lib.check(lib.so_search_by_projection_mappoints_dframe(
        matcher._h, cur._h, _vp(ex), len(a["proj_x"]), _vp(a["in_view"]), _vp(a["proj_x"]), _vp(a["proj_y"]),
        _vp(a["view_cos"]), _vp(a["pred_level"]), _vp(a["desc"]), _vp(a["has_obs"]), float(th),
        float(matcher.mfNNratio), _vp(out), C.byref(nm)))
    return nm.value, out


def search_lastframe_dframe(matcher, cur, last, th, excluded=None):
    """so_search_by_projection_lastframe_dframe (host-side queries, device-resident candidates)."""
    lib = matcher._lib
    _bind(lib)
    a = {k: np.ascontiguousarray(last[k], t) for k, t in
         (("valid", np.uint8), ("u", np.float32), ("v", np.float32), ("octave", np.int32), ("angle", np.float32),
          ("desc", np.uint8), ("has_obs", np.uint8))}
    ex = None if excluded is None else np.ascontiguousarray(excluded, np.uint8)
    out = np.full(cur.n, -1, np.int32)
    nm = C.c_int32(0)
    _lib.check(lib.so_search_by_projection_lastframe_dframe(
        matcher._h, cur._h, _vp(ex), len(a["u"]), _vp(a["valid"]), _vp(a["u"]), _vp(a["v"]), _vp(a["octave"]),
        _vp(a["angle"]), _vp(a["desc"]), _vp(a["has_obs"]), float(th), int(matcher.mbCheckOrientation), _vp(out),
        C.byref(nm)))
    return nm.value, out
