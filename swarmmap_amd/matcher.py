"""Python mirror of ORB_SLAM2::ORBmatcher's tracking routines (code/include/ORBmatcher.h:41-83) over the C ABI.

Frames and map points are passed as flat numpy arrays (FrameView / dicts); the object-graph side effects of
the reference (F.mvpMapPoints[idx] = pMP) are represented by the returned assignment arrays.
"""
import ctypes as C

import numpy as np

from . import _lib

FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48  # code/include/Frame.h:37-38


class SoFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("desc", C.c_void_p), ("excluded", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("scale_factors", C.c_void_p),
                ("nlevels", C.c_int32), ("has_grid_origin", C.c_int32), ("grid_min_x", C.c_float),
                ("grid_min_y", C.c_float)]


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FrameView:
    """The parts of ORB_SLAM2::Frame the matcher reads (mvKeysUn, mDescriptors, bounds, grid, scale factors)."""

    def __init__(self, x, y, octave, angle, desc, bounds, scale_factors, excluded=None, grid_origin=None,
                 grid_bounds=None):
        """grid_bounds: a KeyFrame's case - `bounds` are its int-truncated mnMinX .. mnMaxY (what IsInImage /
        GetFeaturesInArea use, code/include/KeyFrame.h:220) while the grid and its cell sizes come from the Frame's
        float bounds `grid_bounds` (code/src/KeyFrame.cc:58-72).  grid_origin: only the origin differs."""
        self.x = np.ascontiguousarray(x, np.float32)
        self.y = np.ascontiguousarray(y, np.float32)
        self.octave = np.ascontiguousarray(octave, np.int32)
        self.angle = None if angle is None else np.ascontiguousarray(angle, np.float32)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.excluded = None if excluded is None else np.ascontiguousarray(excluded, np.uint8)
        self.min_x, self.max_x, self.min_y, self.max_y = [np.float32(b) for b in bounds]
        self.scale_factors = np.ascontiguousarray(scale_factors, np.float32)
        self.n = len(self.x)
        # Frame ctor, code/src/Frame.cc:259-260
        gb = [np.float32(b) for b in (grid_bounds if grid_bounds is not None else bounds)]
        self.grid_inv_w = np.float32(FRAME_GRID_COLS) / np.float32(gb[1] - gb[0])
        self.grid_inv_h = np.float32(FRAME_GRID_ROWS) / np.float32(gb[3] - gb[2])
        if grid_bounds is not None and grid_origin is None:
            grid_origin = (gb[0], gb[2])
        self.has_grid_origin = 0 if grid_origin is None else 1
        self.grid_min_x, self.grid_min_y = (np.float32(0), np.float32(0)) if grid_origin is None else \
            (np.float32(grid_origin[0]), np.float32(grid_origin[1]))

    def as_struct(self, cls=SoFrameView):
        return cls(self.n, _vp(self.x), _vp(self.y), _vp(self.octave), _vp(self.angle), _vp(self.desc),
                   _vp(self.excluded), self.min_x, self.max_x, self.min_y, self.max_y, self.grid_inv_w,
                   self.grid_inv_h, _vp(self.scale_factors), len(self.scale_factors), self.has_grid_origin,
                   self.grid_min_x, self.grid_min_y)


def _bind(lib):
    vp, ip, f = C.c_void_p, C.POINTER(C.c_int32), C.c_float
    fv = C.POINTER(SoFrameView)
    lib.so_matcher_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.so_matcher_destroy.argtypes = [vp]
    lib.so_matcher_destroy.restype = None
    lib.so_search_by_projection_mappoints.argtypes = [vp, fv, C.c_int32, vp, vp, vp, vp, vp, vp, vp, f, f, vp, ip]
    lib.so_search_by_projection_lastframe.argtypes = [vp, fv, C.c_int32, vp, vp, vp, vp, vp, vp, vp, f, C.c_int, vp, ip]
    lib.so_search_for_initialization.argtypes = [vp, fv, fv, vp, C.c_int, f, C.c_int, vp, ip]
    lib.so_matcher_topk.argtypes = [vp, fv, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp]
    lib.so_hamming_top2.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, vp, vp, vp]
    lib.so_hamming_top2_device.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, vp, vp, vp]
    lib.so_matcher_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.so_matcher_last_stats.argtypes = [vp, C.POINTER(C.c_double)]
    lib.so_matcher_reuse_frame.argtypes = [vp]


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # code/src/ORBmatcher.cc:37-39

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._lib = _lib.load_library()
        _bind(self._lib)
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self._h = C.c_void_p()
        _lib.check(self._lib.so_matcher_create(int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_matcher_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    @staticmethod
    def DescriptorDistance(a, b):
        """ORBmatcher::DescriptorDistance (ORBmatcher.cc:1511-1525) for one pair, on the host like the reference."""
        x = np.bitwise_xor(np.frombuffer(np.ascontiguousarray(a, np.uint8), np.uint8),
                           np.frombuffer(np.ascontiguousarray(b, np.uint8), np.uint8))
        return int(np.unpackbits(x).sum())

    def reserve(self, n_queries):
        """so_matcher_reserve: staging of the tracking searches for up to n_queries map points, allocated now."""
        self._lib.so_matcher_reserve.argtypes = [C.c_void_p, C.c_int32]
        _lib.check(self._lib.so_matcher_reserve(self._h, int(n_queries)))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _lib.check(self._lib.so_matcher_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    # SearchByProjection(Frame&, const vector<MapPoint*>&, th) — ORBmatcher.cc:44-121
    # MapPoint::ComputeDistinctiveDescriptors for a batch of map points — MapPoint.cc:323-392
    def ComputeDistinctiveDescriptors(self, offsets, descriptors):
        off = np.ascontiguousarray(offsets, np.int32)
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(off) - 1
        idx, med = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._lib.so_distinctive_descriptors.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.check(self._lib.so_distinctive_descriptors(self._h, n, _vp(off), _vp(d), _vp(idx), _vp(med)))
        return idx, med

    def reuse_frame(self):
        """The next search looks at the same frame as the previous one on this matcher (one-shot)."""
        _lib.check(self._lib.so_matcher_reuse_frame(self._h))

    def last_stats(self):
        st = (C.c_double * 4)()
        _lib.check(self._lib.so_matcher_last_stats(self._h, st))
        return dict(enqueue_ms=st[0], wait_ms=st[1], launches=int(st[2]), staged_bytes=int(st[3]))

    def SearchByProjectionMapPoints(self, F, mps, th=1.0):
        n_mp = len(mps["proj_x"])
        a = {k: np.ascontiguousarray(mps[k], t) for k, t in
             (("in_view", np.uint8), ("proj_x", np.float32), ("proj_y", np.float32), ("view_cos", np.float32),
              ("pred_level", np.int32), ("desc", np.uint8), ("has_obs", np.uint8))}
        out = np.full(F.n, -1, np.int32)
        nm = C.c_int32(0)
        fs = F.as_struct()
        _lib.check(self._lib.so_search_by_projection_mappoints(
            self._h, C.byref(fs), n_mp, _vp(a["in_view"]), _vp(a["proj_x"]), _vp(a["proj_y"]), _vp(a["view_cos"]),
            _vp(a["pred_level"]), _vp(a["desc"]), _vp(a["has_obs"]), th, self.mfNNratio, _vp(out), C.byref(nm)))
        return nm.value, out

    # SearchByProjection(Frame& cur, const Frame& last, th, bMono) — ORBmatcher.cc:1223-1354
    def SearchByProjectionLastFrame(self, cur, last, th):
        n = len(last["u"])
        a = {k: np.ascontiguousarray(last[k], t) for k, t in
             (("valid", np.uint8), ("u", np.float32), ("v", np.float32), ("octave", np.int32),
              ("angle", np.float32), ("desc", np.uint8), ("has_obs", np.uint8))}
        out = np.full(cur.n, -1, np.int32)
        nm = C.c_int32(0)
        fs = cur.as_struct()
        _lib.check(self._lib.so_search_by_projection_lastframe(
            self._h, C.byref(fs), n, _vp(a["valid"]), _vp(a["u"]), _vp(a["v"]), _vp(a["octave"]), _vp(a["angle"]),
            _vp(a["desc"]), _vp(a["has_obs"]), th, int(self.mbCheckOrientation), _vp(out), C.byref(nm)))
        return nm.value, out

    # SearchForInitialization — ORBmatcher.cc:375-479
    def SearchForInitialization(self, F1, F2, prev_matched, windowSize=10):
        pm = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
        out = np.full(F1.n, -1, np.int32)
        nm = C.c_int32(0)
        f1, f2 = F1.as_struct(), F2.as_struct()
        _lib.check(self._lib.so_search_for_initialization(self._h, C.byref(f1), C.byref(f2), _vp(pm), int(windowSize),
                                                           self.mfNNratio, int(self.mbCheckOrientation), _vp(out),
                                                           C.byref(nm)))
        return nm.value, out, pm

    def topk(self, F, u, v, r, min_level, max_level, qdesc, K, active=None, limit=None):
        nq = len(u)
        u, v, r = [np.ascontiguousarray(t, np.float32) for t in (u, v, r)]
        mn, mx = np.ascontiguousarray(min_level, np.int32), np.ascontiguousarray(max_level, np.int32)
        qd = np.ascontiguousarray(qdesc, np.uint8)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        lim = None if limit is None else np.ascontiguousarray(limit, np.int32)
        idx = np.zeros((nq, K), np.int32)
        dist = np.zeros((nq, K), np.int32)
        cnt = np.zeros(nq, np.int32)
        fs = F.as_struct()
        _lib.check(self._lib.so_matcher_topk(self._h, C.byref(fs), _vp(lim), nq, _vp(u), _vp(v), _vp(r), _vp(mn),
                                              _vp(mx), _vp(act), _vp(qd), K, _vp(idx), _vp(dist), _vp(cnt)))
        return idx, dist, cnt

    def hamming_top2(self, A, B):
        A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32)
        B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
        bi, bd, sd = [np.zeros(len(A), np.int32) for _ in range(3)]
        _lib.check(self._lib.so_hamming_top2(self._h, _vp(A), len(A), _vp(B), len(B), _vp(bi), _vp(bd), _vp(sd)))
        return bi, bd, sd

    def hamming_top2_device(self, dA_ptr, na, dB_ptr, nb):
        bi, bd, sd = [np.zeros(na, np.int32) for _ in range(3)]
        _lib.check(self._lib.so_hamming_top2_device(self._h, C.c_void_p(dA_ptr), na, C.c_void_p(dB_ptr), nb,
                                                     _vp(bi), _vp(bd), _vp(sd)))
        return bi, bd, sd


# ------------------------------------------------------------------------------------------------
# M3 / M5 / M6 / M7 (LocalMapping, loop closing, relocalisation routines)
# ------------------------------------------------------------------------------------------------
class SoFeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("off", C.c_void_p), ("idx", C.c_void_p)]


class FeatureVector:
    """DBoW2::FeatureVector flattened (node ids ascending; per node the feature indices in stored order)."""

    def __init__(self, node_of_feature):
        node_of_feature = np.asarray(node_of_feature, np.int64)
        order = np.argsort(node_of_feature, kind="stable")  # DBoW2 appends indices in ascending feature order
        nodes, counts = np.unique(node_of_feature, return_counts=True)
        self.node_id = nodes.astype(np.int32)
        self.off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.idx = order.astype(np.int32)

    def as_struct(self, cls=SoFeatVec):
        return cls(len(self.node_id), _vp(self.node_id), _vp(self.off), _vp(self.idx))


def _bind_ext(lib):
    vp, ip, f = C.c_void_p, C.POINTER(C.c_int32), C.c_float
    fv, fr = C.POINTER(SoFeatVec), C.POINTER(SoFrameView)
    lib.so_search_by_bow.argtypes = [vp, C.c_int, C.c_int32, vp, vp, vp, fv, C.c_int32, vp, vp, vp, fv, f, C.c_int,
                                     vp, vp, ip]
    lib.so_search_for_triangulation.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, fv, C.c_int32, vp, vp, vp, vp, vp,
                                                vp, fv, vp, f, f, vp, vp, C.c_int32, C.c_int, vp, ip]
    lib.so_search_window_best.argtypes = [vp, fr, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    lib.so_search_window_greedy.argtypes = [vp, fr, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int,
                                            vp, ip]


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _SearchByBoW(self, variant, kf1, fv1, kf2, fv2):
    """kf1/kf2: dicts with desc, angle, valid.  Returns (nmatches, match_of_2, match_of_1)."""
    _bind_ext(self._lib)
    d1, a1, v1 = _u8(kf1["desc"]), _f32(kf1["angle"]), _u8(kf1["valid"])
    d2, a2 = _u8(kf2["desc"]), _f32(kf2["angle"])
    v2 = _u8(kf2["valid"]) if "valid" in kf2 else np.ones(len(d2), np.uint8)
    n1, n2 = len(d1), len(d2)
    m2, m1 = np.full(n2, -1, np.int32), np.full(n1, -1, np.int32)
    nm = C.c_int32(0)
    s1, s2 = fv1.as_struct(), fv2.as_struct()
    _lib.check(self._lib.so_search_by_bow(self._h, variant, n1, _vp(d1), _vp(a1), _vp(v1), C.byref(s1), n2, _vp(d2),
                                          _vp(a2), _vp(v2), C.byref(s2), self.mfNNratio, int(self.mbCheckOrientation),
                                          _vp(m2), _vp(m1), C.byref(nm)))
    return nm.value, m2, m1


def _SearchForTriangulation(self, kf1, fv1, kf2, fv2, F12, epipole, scale_factors2, level_sigma2_2):
    _bind_ext(self._lib)
    x1, y1, a1, d1, f1 = _f32(kf1["x"]), _f32(kf1["y"]), _f32(kf1["angle"]), _u8(kf1["desc"]), _u8(kf1["free"])
    x2, y2, o2, a2, d2, f2 = (_f32(kf2["x"]), _f32(kf2["y"]), _i32(kf2["octave"]), _f32(kf2["angle"]),
                              _u8(kf2["desc"]), _u8(kf2["free"]))
    F = _f32(F12).reshape(9)
    sf, ls = _f32(scale_factors2), _f32(level_sigma2_2)
    out = np.full(len(x1), -1, np.int32)
    nm = C.c_int32(0)
    s1, s2 = fv1.as_struct(), fv2.as_struct()
    _lib.check(self._lib.so_search_for_triangulation(
        self._h, len(x1), _vp(x1), _vp(y1), _vp(a1), _vp(d1), _vp(f1), C.byref(s1), len(x2), _vp(x2), _vp(y2), _vp(o2),
        _vp(a2), _vp(d2), _vp(f2), C.byref(s2), _vp(F), float(epipole[0]), float(epipole[1]), _vp(sf), _vp(ls),
        len(sf), int(self.mbCheckOrientation), _vp(out), C.byref(nm)))
    if getattr(self, "_batching", False):  # outputs are complete after batch_end(); the resolve reads the angle arrays
        self._batch_keep += [a1, a2, out, nm]
        return nm, out
    return nm.value, out


def _batch_begin(self):
    """so_matcher_batch_begin: the SearchForTriangulation / Fuse / FuseSim3 calls up to batch_end() are staged and
    launched together; inside a batch they return their output holders (counts as ctypes ints: read .value afterwards)."""
    self._lib.so_matcher_batch_begin.argtypes = [C.c_void_p]
    self._lib.so_matcher_batch_end.argtypes = [C.c_void_p]
    _lib.check(self._lib.so_matcher_batch_begin(self._h))
    self._batching, self._batch_keep = True, []


def _batch_end(self):
    self._batching = False
    try:
        _lib.check(self._lib.so_matcher_batch_end(self._h))
    finally:
        self._batch_keep = []


def _batch_abort(self):
    """so_matcher_batch_abort: leaves a batch without running it (a call inside it failed)."""
    self._lib.so_matcher_batch_abort.argtypes = [C.c_void_p]
    self._batching, self._batch_keep = False, []
    _lib.check(self._lib.so_matcher_batch_abort(self._h))


class _Batch:
    """`with matcher.batch():` - batch_begin / batch_end, batch_abort when a call inside raises."""

    def __init__(self, m):
        self.m = m

    def __enter__(self):
        self.m.batch_begin()
        return self.m

    def __exit__(self, et, ev, tb):
        if et is None:
            self.m.batch_end()
        else:
            self.m.batch_abort()
        return False


def _SearchWindowBest(self, KF, q, chi2_gate=False, inv_sigma2=None):
    """Core of Fuse / SearchBySim3: q has valid, u, v, radius, pred_level, desc."""
    _bind_ext(self._lib)
    nq = len(q["u"])
    a = dict(valid=_u8(q["valid"]), u=_f32(q["u"]), v=_f32(q["v"]), radius=_f32(q["radius"]),
             pred_level=_i32(q["pred_level"]), desc=_u8(q["desc"]))
    inv = None if inv_sigma2 is None else _f32(inv_sigma2)
    bi, bd = np.full(nq, -1, np.int32), np.full(nq, 256, np.int32)
    fs = KF.as_struct()
    _lib.check(self._lib.so_search_window_best(self._h, C.byref(fs), nq, _vp(a["valid"]), _vp(a["u"]), _vp(a["v"]),
                                               _vp(a["radius"]), _vp(a["pred_level"]), _vp(a["desc"]), int(chi2_gate),
                                               _vp(inv), _vp(bi), _vp(bd)))
    return bi, bd


def _SearchWindowGreedy(self, F, q, max_dist):
    """SearchByProjection(KF, Scw, ...) / SearchByProjection(Frame, KF, sAlreadyFound, th, ORBdist) core."""
    _bind_ext(self._lib)
    nq = len(q["u"])
    a = dict(valid=_u8(q["valid"]), u=_f32(q["u"]), v=_f32(q["v"]), radius=_f32(q["radius"]),
             min_level=_i32(q["min_level"]), max_level=_i32(q["max_level"]), desc=_u8(q["desc"]),
             angle=_f32(q["angle"]))
    out = np.full(F.n, -1, np.int32)
    nm = C.c_int32(0)
    fs = F.as_struct()
    _lib.check(self._lib.so_search_window_greedy(self._h, C.byref(fs), nq, _vp(a["valid"]), _vp(a["u"]), _vp(a["v"]),
                                                 _vp(a["radius"]), _vp(a["min_level"]), _vp(a["max_level"]),
                                                 _vp(a["desc"]), _vp(a["angle"]), int(max_dist),
                                                 int(self.mbCheckOrientation), _vp(out), C.byref(nm)))
    return nm.value, out


ORBmatcher.SearchByBoW = _SearchByBoW
ORBmatcher.batch_begin = _batch_begin
ORBmatcher.batch_end = _batch_end
ORBmatcher.batch_abort = _batch_abort
ORBmatcher.batch = lambda self: _Batch(self)
ORBmatcher.SearchForTriangulation = _SearchForTriangulation
ORBmatcher.SearchWindowBest = _SearchWindowBest
ORBmatcher.SearchWindowGreedy = _SearchWindowGreedy


# ---- Fuse / SearchBySim3 / keyframe-side SearchByProjection with the projection on the device (rows M6 / M7) --------
class SoMapPointView(C.Structure):
    _fields_ = [("n", C.c_int32), ("Xw", C.c_void_p), ("normal", C.c_void_p), ("max_dist", C.c_void_p),
                ("min_dist", C.c_void_p), ("desc", C.c_void_p), ("valid", C.c_void_p)]


class SoWindowQueries(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("active", "u", "v", "radius", "level")]


class SoCameraM(C.Structure):  # so_camera
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


def _bind_proj(lib):
    vp, ip, f, i32 = C.c_void_p, C.POINTER(C.c_int32), C.c_float, C.c_int32
    fr, cam = C.POINTER(SoFrameView), C.POINTER(SoCameraM)
    mp, wq = C.POINTER(SoMapPointView), C.POINTER(SoWindowQueries)
    lib.so_fuse.argtypes = [vp, fr, cam, vp, f, vp, mp, f, vp, vp, ip, wq]
    lib.so_fuse_sim3.argtypes = [vp, fr, cam, vp, f, mp, f, vp, vp, ip, wq]
    lib.so_search_by_sim3.argtypes = [vp, fr, fr, cam, vp, vp, f, vp, vp, f, f, mp, mp, f, vp, ip, wq, wq]
    lib.so_search_by_projection_sim3.argtypes = [vp, fr, cam, vp, f, mp, C.c_int, vp, ip, wq]
    lib.so_search_by_projection_keyframe.argtypes = [vp, fr, cam, vp, f, mp, vp, f, i32, C.c_int, vp, ip, wq]


def _mp_struct(mp):
    a = dict(Xw=_f32(mp["Xw"]), normal=_f32(mp["normal"]) if mp.get("normal") is not None else None,
             max_dist=_f32(mp["max_dist"]), min_dist=_f32(mp["min_dist"]), desc=_u8(mp["desc"]),
             valid=_u8(mp["valid"]) if mp.get("valid") is not None else None)
    n = len(a["max_dist"])
    return SoMapPointView(n, _vp(a["Xw"]), _vp(a["normal"]), _vp(a["max_dist"]), _vp(a["min_dist"]), _vp(a["desc"]),
                          _vp(a["valid"])), a


def _wq_struct(n):
    q = dict(active=np.zeros(n, np.uint8), u=np.zeros(n, np.float32), v=np.zeros(n, np.float32),
             radius=np.zeros(n, np.float32), level=np.zeros(n, np.int32))
    return SoWindowQueries(_vp(q["active"]), _vp(q["u"]), _vp(q["v"]), _vp(q["radius"]), _vp(q["level"])), q


def _cam(K):
    return SoCameraM(float(K[0]), float(K[1]), float(K[2]), float(K[3]), 0, 0, 0, 0, 0)


def _Fuse(self, KF, K, Tcw, log_scale_factor, inv_level_sigma2, mp, th=3.0):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) up to the map side effects.
    Returns (nFused, best_idx, best_dist, queries) - queries = the projection half's output per map point."""
    _bind_proj(self._lib)
    ms, _keep = _mp_struct(mp)
    qs, q = _wq_struct(ms.n)
    bi, bd = np.full(ms.n, -1, np.int32), np.full(ms.n, 256, np.int32)
    nf = C.c_int32(0)
    fs, cam, T, inv = KF.as_struct(), _cam(K), _f32(Tcw).reshape(12), _f32(inv_level_sigma2)
    _lib.check(self._lib.so_fuse(self._h, C.byref(fs), C.byref(cam), _vp(T), float(log_scale_factor), _vp(inv),
                                 C.byref(ms), float(th), _vp(bi), _vp(bd), C.byref(nf), C.byref(qs)))
    if getattr(self, "_batching", False):
        self._batch_keep += [bi, bd, q, nf]
        return nf, bi, bd, q
    return nf.value, bi, bd, q


def _FuseSim3(self, KF, K, Scw, log_scale_factor, mp, th=4.0):
    """ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)."""
    _bind_proj(self._lib)
    ms, _keep = _mp_struct(mp)
    qs, q = _wq_struct(ms.n)
    bi, bd = np.full(ms.n, -1, np.int32), np.full(ms.n, 256, np.int32)
    nf = C.c_int32(0)
    fs, cam, S = KF.as_struct(), _cam(K), _f32(Scw).reshape(12)
    _lib.check(self._lib.so_fuse_sim3(self._h, C.byref(fs), C.byref(cam), _vp(S), float(log_scale_factor), C.byref(ms),
                                      float(th), _vp(bi), _vp(bd), C.byref(nf), C.byref(qs)))
    if getattr(self, "_batching", False):
        self._batch_keep += [bi, bd, q, nf]
        return nf, bi, bd, q
    return nf.value, bi, bd, q


def _SearchBySim3(self, KF1, KF2, K, T1w, T2w, s12, R12, t12, log_sf1, log_sf2, mp1, mp2, th=7.5):
    """ORBmatcher::SearchBySim3.  Returns (nFound, match12, queries1_in_2, queries2_in_1)."""
    _bind_proj(self._lib)
    m1, _k1 = _mp_struct(mp1)
    m2, _k2 = _mp_struct(mp2)
    q1s, q1 = _wq_struct(m1.n)
    q2s, q2 = _wq_struct(m2.n)
    out = np.full(m1.n, -1, np.int32)
    nf = C.c_int32(0)
    f1, f2, cam = KF1.as_struct(), KF2.as_struct(), _cam(K)
    a, b, R, t = _f32(T1w).reshape(12), _f32(T2w).reshape(12), _f32(R12).reshape(9), _f32(t12).reshape(3)
    _lib.check(self._lib.so_search_by_sim3(self._h, C.byref(f1), C.byref(f2), C.byref(cam), _vp(a), _vp(b), float(s12),
                                           _vp(R), _vp(t), float(log_sf1), float(log_sf2), C.byref(m1), C.byref(m2),
                                           float(th), _vp(out), C.byref(nf), C.byref(q1s), C.byref(q2s)))
    return nf.value, out, q1, q2


def _SearchByProjectionSim3(self, KF, K, Scw, log_scale_factor, mp, th=10):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th).  Returns (nmatches, kp_to_point, queries)."""
    _bind_proj(self._lib)
    ms, _keep = _mp_struct(mp)
    qs, q = _wq_struct(ms.n)
    out = np.full(KF.n, -1, np.int32)
    nm = C.c_int32(0)
    fs, cam, S = KF.as_struct(), _cam(K), _f32(Scw).reshape(12)
    _lib.check(self._lib.so_search_by_projection_sim3(self._h, C.byref(fs), C.byref(cam), _vp(S), float(log_scale_factor),
                                                      C.byref(ms), int(th), _vp(out), C.byref(nm), C.byref(qs)))
    return nm.value, out, q


def _SearchByProjectionKeyFrame(self, F, K, Tcw, log_scale_factor, mp, mp_angle, th, ORBdist):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist, bGlobal)."""
    _bind_proj(self._lib)
    ms, _keep = _mp_struct(mp)
    qs, q = _wq_struct(ms.n)
    out = np.full(F.n, -1, np.int32)
    nm = C.c_int32(0)
    fs, cam, T, ang = F.as_struct(), _cam(K), _f32(Tcw).reshape(12), _f32(mp_angle)
    _lib.check(self._lib.so_search_by_projection_keyframe(
        self._h, C.byref(fs), C.byref(cam), _vp(T), float(log_scale_factor), C.byref(ms), _vp(ang), float(th),
        int(ORBdist), int(self.mbCheckOrientation), _vp(out), C.byref(nm), C.byref(qs)))
    return nm.value, out, q


ORBmatcher.Fuse = _Fuse
ORBmatcher.FuseSim3 = _FuseSim3
ORBmatcher.SearchBySim3 = _SearchBySim3
ORBmatcher.SearchByProjectionSim3 = _SearchByProjectionSim3
ORBmatcher.SearchByProjectionKeyFrame = _SearchByProjectionKeyFrame


# ---- HBM-resident keyframes (so_kframe_*) ---------------------------------------------------------------------------
class KFrame:
    """A keyframe's matcher-side data uploaded once (so_kframe_create): FrameView + optional FeatureVector."""

    def __init__(self, matcher, KF, fv=None, level_sigma2=None):
        self._lib = matcher._lib
        vp = C.c_void_p
        self._lib.so_kframe_create.argtypes = [vp, C.POINTER(SoFrameView), C.POINTER(SoFeatVec), vp, C.POINTER(vp)]
        self._lib.so_kframe_destroy.argtypes = [vp]
        self._lib.so_kframe_destroy.restype = None
        self._h = vp()
        fs = KF.as_struct()
        fvs = fv.as_struct() if fv is not None else None
        ls = _f32(level_sigma2) if level_sigma2 is not None else None
        _lib.check(self._lib.so_kframe_create(matcher._h, C.byref(fs), C.byref(fvs) if fvs is not None else None, _vp(ls),
                                              C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_kframe_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def _FuseKFrame(self, kframe, K, Tcw, log_scale_factor, inv_level_sigma2, mp, th=3.0):
    """so_fuse_kframe: ORBmatcher::Fuse(pKF, vpMapPoints, th) against an HBM-resident keyframe."""
    _bind_proj(self._lib)
    vp, f = C.c_void_p, C.c_float
    self._lib.so_fuse_kframe.argtypes = [vp, vp, C.POINTER(SoCameraM), vp, f, vp, C.POINTER(SoMapPointView), f, vp, vp,
                                         C.POINTER(C.c_int32), C.POINTER(SoWindowQueries)]
    ms, _keep = _mp_struct(mp)
    qs, q = _wq_struct(ms.n)
    bi, bd = np.full(ms.n, -1, np.int32), np.full(ms.n, 256, np.int32)
    nf = C.c_int32(0)
    cam, T, inv = _cam(K), _f32(Tcw).reshape(12), _f32(inv_level_sigma2)
    _lib.check(self._lib.so_fuse_kframe(self._h, kframe._h, C.byref(cam), _vp(T), float(log_scale_factor), _vp(inv), C.byref(ms),
                                        float(th), _vp(bi), _vp(bd), C.byref(nf), C.byref(qs)))
    if getattr(self, "_batching", False):
        self._batch_keep += [bi, bd, q, nf, kframe]
        return nf, bi, bd, q
    return nf.value, bi, bd, q


def _SearchForTriangulationKFrame(self, kf1, fv1, kframe2, free2, F12, epipole):
    """so_search_for_triangulation_kframe: keyframe 2 HBM-resident (created with its feature vector)."""
    vp, f, i32 = C.c_void_p, C.c_float, C.c_int32
    self._lib.so_search_for_triangulation_kframe.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.POINTER(SoFeatVec), vp, vp, vp, f, f,
                                                             C.c_int, vp, C.POINTER(i32)]
    x1, y1, a1, d1, f1 = _f32(kf1["x"]), _f32(kf1["y"]), _f32(kf1["angle"]), _u8(kf1["desc"]), _u8(kf1["free"])
    f2 = _u8(free2)
    F = _f32(F12).reshape(9)
    out = np.full(len(x1), -1, np.int32)
    nm = C.c_int32(0)
    s1 = fv1.as_struct()
    _lib.check(self._lib.so_search_for_triangulation_kframe(
        self._h, len(x1), _vp(x1), _vp(y1), _vp(a1), _vp(d1), _vp(f1), C.byref(s1), kframe2._h, _vp(f2), _vp(F),
        float(epipole[0]), float(epipole[1]), int(self.mbCheckOrientation), _vp(out), C.byref(nm)))
    if getattr(self, "_batching", False):
        self._batch_keep += [a1, out, nm, kframe2]
        return nm, out
    return nm.value, out


def _FuseKFrameMap(self, kframe, K, Tcw, log_scale_factor, inv_level_sigma2, dmap, slots, valid=None, th=3.0):
    """so_fuse_kframe_map: Fuse against an HBM-resident keyframe with the map points read from a DeviceMap by slot."""
    vp, f, i32 = C.c_void_p, C.c_float, C.c_int32
    self._lib.so_fuse_kframe_map.argtypes = [vp, vp, C.POINTER(SoCameraM), vp, f, vp, vp, i32, vp, vp, f, vp, vp,
                                             C.POINTER(i32), C.POINTER(SoWindowQueries)]
    sl = np.ascontiguousarray(slots, np.int32)
    n = len(sl)
    va = None if valid is None else _u8(valid)
    qs, q = _wq_struct(n)
    bi, bd = np.full(n, -1, np.int32), np.full(n, 256, np.int32)
    nf = i32(0)
    cam, T, inv = _cam(K), _f32(Tcw).reshape(12), _f32(inv_level_sigma2)
    _lib.check(self._lib.so_fuse_kframe_map(self._h, kframe._h, C.byref(cam), _vp(T), float(log_scale_factor), _vp(inv), dmap._h, n,
                                            _vp(sl), _vp(va), float(th), _vp(bi), _vp(bd), C.byref(nf), C.byref(qs)))
    if getattr(self, "_batching", False):
        self._batch_keep += [bi, bd, q, nf, kframe, dmap]
        return nf, bi, bd, q
    return nf.value, bi, bd, q


class SoTriNeighbour(C.Structure):
    _fields_ = [("kf2", C.c_void_p), ("free2", C.c_void_p), ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float),
                ("matches12", C.c_void_p), ("nmatches", C.c_void_p)]


def _SearchForTriangulationKFrames(self, kframe1, free1, neighbours):
    """so_search_for_triangulation_kframes: one resident keyframe against several resident neighbours, the queries built on
    the device.  neighbours: list of (kframe2, free2, F12, (ex, ey)).  Returns [(nmatches, matches12)] (inside a batch:
    ctypes ints, read .value after batch_end())."""
    self._lib.so_search_for_triangulation_kframes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int]
    f1 = _u8(free1)
    arr = (SoTriNeighbour * max(len(neighbours), 1))()
    keep, outs = [f1], []
    for j, (k2, free2, F12, epi) in enumerate(neighbours):
        f2, F = _u8(free2), _f32(F12).reshape(9)
        m12, nm = np.full(len(f1), -1, np.int32), C.c_int32(0)
        arr[j].kf2, arr[j].free2 = k2._h, _vp(f2)
        arr[j].F12[:] = [float(v) for v in F]
        arr[j].ex, arr[j].ey = float(epi[0]), float(epi[1])
        arr[j].matches12, arr[j].nmatches = _vp(m12), C.cast(C.pointer(nm), C.c_void_p)
        keep += [f2, m12, nm, k2]
        outs.append((nm, m12))
    _lib.check(self._lib.so_search_for_triangulation_kframes(self._h, kframe1._h, _vp(f1), len(neighbours), C.cast(arr, C.c_void_p),
                                                             int(self.mbCheckOrientation)))
    if getattr(self, "_batching", False):
        self._batch_keep += keep + [arr, kframe1]
        return outs
    return [(nm.value, m12) for nm, m12 in outs]


ORBmatcher.SearchForTriangulationKFrames = _SearchForTriangulationKFrames
ORBmatcher.FuseKFrame = _FuseKFrame
ORBmatcher.FuseKFrameMap = _FuseKFrameMap
ORBmatcher.SearchForTriangulationKFrame = _SearchForTriangulationKFrame


# ---- the local-mapping thread's per-point loops between the matcher and local BA -----------------------------------
class SoTriKeyframe(C.Structure):
    _fields_ = [("Tcw", C.c_float * 12)] + [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "invfx", "invfy")] + \
               [("scale_factors", C.c_void_p), ("level_sigma2", C.c_void_p), ("nlevels", C.c_int32)]


def _so_tri_kf(kf, keep):
    sf, ls = _f32(kf["scale_factors"]), _f32(kf["level_sigma2"])
    keep += [sf, ls]
    fx, fy, cx, cy = [np.float32(v) for v in kf["K"]]
    s = SoTriKeyframe()
    s.Tcw[:] = [float(v) for v in _f32(kf["Tcw"]).reshape(12)]
    s.fx, s.fy, s.cx, s.cy = float(fx), float(fy), float(cx), float(cy)
    s.invfx, s.invfy = float(np.float32(1.0) / fx), float(np.float32(1.0) / fy)
    s.scale_factors, s.level_sigma2, s.nlevels = _vp(sf), _vp(ls), len(sf)
    return s


def _TriangulateMatches(self, kf1, kf2_list, ratio_factor, kf2_of_match, xy1, octave1, xy2, octave2):
    """so_triangulate_matches: CreateNewMapPoints' per-match body for the matches of kf1 with several neighbours in one
    launch.  kf dicts: Tcw (12), K (fx fy cx cy), scale_factors, level_sigma2.  Returns (ok, x3D)."""
    vp, i32 = C.c_void_p, C.c_int32
    self._lib.so_triangulate_matches.argtypes = [vp, C.POINTER(SoTriKeyframe), i32, vp, C.c_float, i32, vp, vp, vp, vp, vp, vp, vp]
    keep = []
    a = _so_tri_kf(kf1, keep)
    arr = (SoTriKeyframe * max(len(kf2_list), 1))(*[_so_tri_kf(k, keep) for k in kf2_list])
    of = _i32(kf2_of_match)
    x1, o1, x2, o2 = _f32(xy1).reshape(-1, 2), _i32(octave1), _f32(xy2).reshape(-1, 2), _i32(octave2)
    n = len(o1)
    ok, X = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
    _lib.check(self._lib.so_triangulate_matches(self._h, C.byref(a), len(kf2_list), arr, float(ratio_factor), n, _vp(of), _vp(x1),
                                                _vp(o1), _vp(x2), _vp(o2), _vp(ok), _vp(X)))
    return ok, X


def _UpdateNormalAndDepth(self, offsets, obs_Ow, Xw, ref_Ow, ref_level_scale, ref_last_scale, normal, max_dist, min_dist):
    """so_update_normal_and_depth: MapPoint::UpdateNormalAndDepth for a batch.  Returns (normal, max_dist, min_dist)."""
    vp = C.c_void_p
    self._lib.so_update_normal_and_depth.argtypes = [vp, C.c_int32] + [vp] * 9
    off = _i32(offsets)
    a = [_f32(v) for v in (obs_Ow, Xw, ref_Ow, ref_level_scale, ref_last_scale)]
    nrm, mx, mn = [np.array(v, np.float32, copy=True) for v in (normal, max_dist, min_dist)]
    _lib.check(self._lib.so_update_normal_and_depth(self._h, len(off) - 1, _vp(off), _vp(a[0]), _vp(a[1]), _vp(a[2]), _vp(a[3]),
                                                    _vp(a[4]), _vp(nrm), _vp(mx), _vp(mn)))
    return nrm, mx, mn


def _UpdateNormalAndDepthIndexed(self, offsets, obs_kf, kf_Ow, Xw, ref_kf, ref_level_scale, ref_last_scale, normal, max_dist, min_dist):
    """so_update_normal_and_depth_indexed: the observers as indices into kf_Ow (n_kf x 3 camera centres)."""
    vp, i32 = C.c_void_p, C.c_int32
    self._lib.so_update_normal_and_depth_indexed.argtypes = [vp, i32, vp, vp, i32] + [vp] * 8
    off, ok, rk = _i32(offsets), _i32(obs_kf), _i32(ref_kf)
    c, X, ls, ll = [_f32(v) for v in (kf_Ow, Xw, ref_level_scale, ref_last_scale)]
    nrm, mx, mn = [np.array(v, np.float32, copy=True) for v in (normal, max_dist, min_dist)]
    _lib.check(self._lib.so_update_normal_and_depth_indexed(self._h, len(off) - 1, _vp(off), _vp(ok), len(c.reshape(-1, 3)), _vp(c), _vp(X),
                                                            _vp(rk), _vp(ls), _vp(ll), _vp(nrm), _vp(mx), _vp(mn)))
    return nrm, mx, mn


def _TriangulateNewPoints(self, kf1, kf2_list, ratio_factor, kf2_of_match, xy1, octave1, xy2, octave2):
    """so_triangulate_new_points: so_triangulate_matches + the new points' normal / distance range in the same launch.
    Returns (ok, x3D, normal, max_dist, min_dist); the last three are zero where ok is 0."""
    vp, i32 = C.c_void_p, C.c_int32
    self._lib.so_triangulate_new_points.argtypes = [vp, C.POINTER(SoTriKeyframe), i32, vp, C.c_float, i32] + [vp] * 10
    keep = []
    a = _so_tri_kf(kf1, keep)
    arr = (SoTriKeyframe * max(len(kf2_list), 1))(*[_so_tri_kf(k, keep) for k in kf2_list])
    of = _i32(kf2_of_match)
    x1, o1, x2, o2 = _f32(xy1).reshape(-1, 2), _i32(octave1), _f32(xy2).reshape(-1, 2), _i32(octave2)
    n = len(o1)
    ok, X = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
    nrm, mx, mn = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    _lib.check(self._lib.so_triangulate_new_points(self._h, C.byref(a), len(kf2_list), arr, float(ratio_factor), n, _vp(of), _vp(x1),
                                                   _vp(o1), _vp(x2), _vp(o2), _vp(ok), _vp(X), _vp(nrm), _vp(mx), _vp(mn)))
    return ok, X, nrm, mx, mn


ORBmatcher.TriangulateMatches = _TriangulateMatches
ORBmatcher.TriangulateNewPoints = _TriangulateNewPoints
ORBmatcher.UpdateNormalAndDepth = _UpdateNormalAndDepth
ORBmatcher.UpdateNormalAndDepthIndexed = _UpdateNormalAndDepthIndexed
