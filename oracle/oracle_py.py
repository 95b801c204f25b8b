"""ctypes binding of the CPU parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (swarmmap_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_MAX_LEVELS = 16
ORC_FAST_CAP = 10000


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrcConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class OrcTables(C.Structure):
    _fields_ = [("scale", C.c_float * ORC_MAX_LEVELS), ("inv_scale", C.c_float * ORC_MAX_LEVELS),
                ("sigma2", C.c_float * ORC_MAX_LEVELS), ("inv_sigma2", C.c_float * ORC_MAX_LEVELS),
                ("features_per_level", C.c_int32 * ORC_MAX_LEVELS), ("umax", C.c_int32 * 16)]


def build(force=False):
    """Compile oracle/*.c into oracle/liboracle.so when missing or stale (needs gcc + make)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_atan2f.restype = C.c_float
        _LIB.orc_atan2f.argtypes = [C.c_float, C.c_float]
        _LIB.orc_sincosf_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def config(nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
    return OrcConfig(nfeatures, scale_factor, nlevels, ini_th, min_th)


def make_tables(cfg):
    t = OrcTables()
    lib().orc_make_tables(C.byref(cfg), C.byref(t))
    return t


def level_sizes(cfg, w, h):
    t = make_tables(cfg)
    out = []
    for l in range(cfg.nlevels):
        lw, lh = C.c_int(), C.c_int()
        lib().orc_level_size(w, h, C.c_float(t.inv_scale[l]), C.byref(lw), C.byref(lh))
        out.append((lw.value, lh.value))
    return out


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian7(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty_like(src)
    lib().orc_gaussian7(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def border_reflect101(src, border=19):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.empty((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_border_reflect101(_p(src), w, h, src.strides[0], _p(dst), border, dst.strides[0])
    return dst


def fast_score(img, x, y, th):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return int(lib().orc_fast_score(_p(img), img.strides[0], int(x), int(y), int(th)))


def fast_detect(level, th_high=20, th_low=7, cap=ORC_FAST_CAP):
    level = np.ascontiguousarray(level, dtype=np.uint8)
    xs = np.empty(cap, np.int16)
    ys = np.empty(cap, np.int16)
    sc = np.empty(cap, np.uint8)
    n = lib().orc_fast_detect(_p(level), level.shape[1], level.shape[0], level.strides[0], th_high, th_low,
                              _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def distribute_octree(xs, ys, scores, roi_w, roi_h, n_target):
    xs = np.ascontiguousarray(xs, np.int16)
    ys = np.ascontiguousarray(ys, np.int16)
    scores = np.ascontiguousarray(scores, np.uint8)
    out = np.empty(n_target + 64, np.int32)
    n = lib().orc_distribute_octree(_p(xs), _p(ys), _p(scores), len(xs), roi_w, roi_h, n_target, _p(out),
                                    len(out))
    return out[:n].copy()


def ic_angle(level, x, y, umax):
    level = np.ascontiguousarray(level, dtype=np.uint8)
    um = np.ascontiguousarray(umax, np.int32)
    return float(lib().orc_ic_angle(_p(level), level.strides[0], int(x), int(y), _p(um)))


def brief(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, dtype=np.uint8)
    d = np.empty(32, np.uint8)
    lib().orc_brief(_p(blurred), blurred.strides[0], int(x), int(y), C.c_float(angle_deg), _p(d))
    return d


def extract(cfg, img, debug=False, cand_cap=ORC_FAST_CAP):
    """Returns (keypoints[KP_DTYPE], descriptors[n,32]); with debug also per-level candidates + pyramid."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = cfg.nfeatures + 2 * cfg.nlevels + 64
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    if not debug:
        n = lib().orc_extract(C.byref(cfg), _p(img), w, h, img.strides[0], _p(kps), _p(desc), cap)
        return kps[:n].copy(), desc[:n].copy()
    nl = cfg.nlevels
    cx = np.zeros((nl, cand_cap), np.int16)
    cy = np.zeros((nl, cand_cap), np.int16)
    cs = np.zeros((nl, cand_cap), np.uint8)
    cnt = np.zeros(nl, np.int32)
    sizes = level_sizes(cfg, w, h)
    pyr = np.zeros(sum(a * b for a, b in sizes), np.uint8)
    n = lib().orc_extract_debug(C.byref(cfg), _p(img), w, h, img.strides[0], _p(kps), _p(desc), cap, _p(cx),
                                _p(cy), _p(cs), _p(cnt), cand_cap, _p(pyr))
    levels, off = [], 0
    for (lw, lh) in sizes:
        levels.append(pyr[off:off + lw * lh].reshape(lh, lw))
        off += lw * lh
    cands = [(cx[l, :cnt[l]].copy(), cy[l, :cnt[l]].copy(), cs[l, :cnt[l]].copy()) for l in range(nl)]
    return kps[:n].copy(), desc[:n].copy(), cands, levels


# ------------------------------------------------------------------------------------------------
# matcher oracle (matcher_oracle.c); FrameView objects come from swarmmap_amd.matcher (plain data holder)
# ------------------------------------------------------------------------------------------------
class OrcFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("desc", C.c_void_p), ("excluded", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("scale_factors", C.c_void_p),
                ("nlevels", C.c_int32), ("has_grid_origin", C.c_int32), ("grid_min_x", C.c_float),
                ("grid_min_y", C.c_float)]


def _fv(F):
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    return OrcFrameView(F.n, p(F.x), p(F.y), p(F.octave), p(F.angle), p(F.desc), p(F.excluded), F.min_x, F.max_x,
                        F.min_y, F.max_y, F.grid_inv_w, F.grid_inv_h, p(F.scale_factors), len(F.scale_factors),
                        getattr(F, "has_grid_origin", 0), getattr(F, "grid_min_x", 0.0), getattr(F, "grid_min_y", 0.0))


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_descriptor_distance(_p(a), _p(b)))


def features_in_area(F, x, y, r, min_level=-1, max_level=-1):
    out = np.zeros(max(F.n, 1), np.int32)
    fs = _fv(F)
    n = lib().orc_features_in_area(C.byref(fs), C.c_float(x), C.c_float(y), C.c_float(r), int(min_level),
                                   int(max_level), _p(out), len(out))
    return out[:n].copy()


def search_by_projection_mappoints(F, mps, th, nn_ratio):
    a = {k: np.ascontiguousarray(mps[k], t) for k, t in
         (("in_view", np.uint8), ("proj_x", np.float32), ("proj_y", np.float32), ("view_cos", np.float32),
          ("pred_level", np.int32), ("desc", np.uint8), ("has_obs", np.uint8))}
    out = np.full(F.n, -1, np.int32)
    fs = _fv(F)
    nm = lib().orc_search_by_projection_mappoints(C.byref(fs), len(a["proj_x"]), _p(a["in_view"]), _p(a["proj_x"]),
                                                  _p(a["proj_y"]), _p(a["view_cos"]), _p(a["pred_level"]),
                                                  _p(a["desc"]), _p(a["has_obs"]), C.c_float(th), C.c_float(nn_ratio),
                                                  _p(out))
    return nm, out


def search_by_projection_lastframe(cur, last, th, check_ori=True):
    a = {k: np.ascontiguousarray(last[k], t) for k, t in
         (("valid", np.uint8), ("u", np.float32), ("v", np.float32), ("octave", np.int32), ("angle", np.float32),
          ("desc", np.uint8), ("has_obs", np.uint8))}
    out = np.full(cur.n, -1, np.int32)
    fs = _fv(cur)
    nm = lib().orc_search_by_projection_lastframe(C.byref(fs), len(a["u"]), _p(a["valid"]), _p(a["u"]), _p(a["v"]),
                                                  _p(a["octave"]), _p(a["angle"]), _p(a["desc"]), _p(a["has_obs"]),
                                                  C.c_float(th), int(check_ori), _p(out))
    return nm, out


def search_for_initialization(F1, F2, prev_matched, window, nn_ratio, check_ori=True):
    pm = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
    out = np.full(F1.n, -1, np.int32)
    f1, f2 = _fv(F1), _fv(F2)
    nm = lib().orc_search_for_initialization(C.byref(f1), C.byref(f2), _p(pm), int(window), C.c_float(nn_ratio),
                                             int(check_ori), _p(out))
    return nm, out, pm


def hamming_top2(A, B):
    A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32)
    B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
    bi, bd, sd = [np.zeros(len(A), np.int32) for _ in range(3)]
    lib().orc_hamming_top2(_p(A), len(A), _p(B), len(B), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


# ------------------------------------------------------------------------------------------------
# bundle-adjustment oracle (ba_oracle.c)
# ------------------------------------------------------------------------------------------------
class OrcBaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("Tcw", C.c_void_p), ("fixed", C.c_void_p), ("intr", C.c_void_p),
                ("n_points", C.c_int32), ("Xw", C.c_void_p), ("n_edges", C.c_int32), ("edge_pose", C.c_void_p),
                ("edge_point", C.c_void_p), ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p)]


class OrcBaOptions(C.Structure):
    _fields_ = [("its_stage1", C.c_int32), ("its_stage2", C.c_int32), ("robust", C.c_int32),
                ("huber_delta", C.c_float), ("chi2_threshold", C.c_float)]


class OrcBaInfo(C.Structure):
    _fields_ = [("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("iterations_stage1", C.c_int32), ("iterations_stage2", C.c_int32), ("lm_trials", C.c_int32),
                ("aborted", C.c_int32), ("n_outliers", C.c_int32)]


def ba_arrays(prob):
    """Normalise a problem dict (see swarmmap_amd.synth.make_ba_problem) to contiguous arrays of the ABI dtypes."""
    return dict(Tcw=np.ascontiguousarray(prob["Tcw"], np.float32).reshape(-1, 12),
                fixed=np.ascontiguousarray(prob["fixed"], np.uint8),
                intr=np.ascontiguousarray(prob["intr"], np.float32).reshape(-1, 4),
                Xw=np.ascontiguousarray(prob["Xw"], np.float32).reshape(-1, 3),
                edge_pose=np.ascontiguousarray(prob["edge_pose"], np.int32),
                edge_point=np.ascontiguousarray(prob["edge_point"], np.int32),
                obs=np.ascontiguousarray(prob["obs"], np.float32).reshape(-1, 2),
                inv_sigma2=np.ascontiguousarray(prob["inv_sigma2"], np.float32))


def bundle_adjust(prob, its1=5, its2=10, robust=True, huber_delta=None, chi2_threshold=5.991, stop=None):
    """Optimizer::LocalBundleAdjustment (its2 > 0) / BundleAdjustment (its2 == 0) on a flattened problem."""
    a = ba_arrays(prob)
    P = OrcBaProblem(len(a["Tcw"]), _p(a["Tcw"]), _p(a["fixed"]), _p(a["intr"]), len(a["Xw"]), _p(a["Xw"]),
                     len(a["edge_pose"]), _p(a["edge_pose"]), _p(a["edge_point"]), _p(a["obs"]), _p(a["inv_sigma2"]))
    if huber_delta is None:
        huber_delta = np.float32(np.sqrt(5.991))  # const float thHuberMono = sqrt(5.991), Optimizer.cc:547
    O = OrcBaOptions(its1, its2, int(robust), float(huber_delta), float(chi2_threshold))
    Tout = np.zeros_like(a["Tcw"])
    Xout = np.zeros_like(a["Xw"])
    outl = np.zeros(len(a["edge_pose"]), np.uint8)
    chi2 = np.zeros(len(a["edge_pose"]), np.float64)
    info = OrcBaInfo()
    stop_p = None if stop is None else stop.ctypes.data_as(C.c_void_p)
    rc = lib().orc_bundle_adjust(C.byref(P), C.byref(O), stop_p, _p(Tout), _p(Xout), _p(outl), _p(chi2), C.byref(info))
    assert rc == 0
    return dict(Tcw=Tout, Xw=Xout, outlier=outl, chi2=chi2,
                info={k: getattr(info, k) for k, _ in OrcBaInfo._fields_})


class OrcFeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("off", C.c_void_p), ("idx", C.c_void_p)]


def _fvs(fv):
    return OrcFeatVec(len(fv.node_id), _p(fv.node_id), _p(fv.off), _p(fv.idx))


def search_by_bow(variant, kf1, fv1, kf2, fv2, nn_ratio, check_ori=True):
    d1, a1, v1 = [np.ascontiguousarray(kf1[k], t) for k, t in (("desc", np.uint8), ("angle", np.float32), ("valid", np.uint8))]
    d2, a2 = np.ascontiguousarray(kf2["desc"], np.uint8), np.ascontiguousarray(kf2["angle"], np.float32)
    v2 = np.ascontiguousarray(kf2["valid"], np.uint8) if "valid" in kf2 else np.ones(len(d2), np.uint8)
    m2, m1 = np.full(len(d2), -1, np.int32), np.full(len(d1), -1, np.int32)
    s1, s2 = _fvs(fv1), _fvs(fv2)
    nm = lib().orc_search_by_bow(int(variant), len(d1), _p(d1), _p(a1), _p(v1), C.byref(s1), len(d2), _p(d2), _p(a2),
                                 _p(v2), C.byref(s2), C.c_float(nn_ratio), int(check_ori), _p(m2), _p(m1))
    return nm, m2, m1


def search_for_triangulation(kf1, fv1, kf2, fv2, F12, epipole, scale_factors2, level_sigma2_2, check_ori=True):
    g = lambda d, k, t: np.ascontiguousarray(d[k], t)  # noqa: E731
    x1, y1, a1, d1, f1 = g(kf1, "x", np.float32), g(kf1, "y", np.float32), g(kf1, "angle", np.float32), \
        g(kf1, "desc", np.uint8), g(kf1, "free", np.uint8)
    x2, y2, o2, a2, d2, f2 = g(kf2, "x", np.float32), g(kf2, "y", np.float32), g(kf2, "octave", np.int32), \
        g(kf2, "angle", np.float32), g(kf2, "desc", np.uint8), g(kf2, "free", np.uint8)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sf, ls = np.ascontiguousarray(scale_factors2, np.float32), np.ascontiguousarray(level_sigma2_2, np.float32)
    out = np.full(len(x1), -1, np.int32)
    s1, s2 = _fvs(fv1), _fvs(fv2)
    nm = lib().orc_search_for_triangulation(len(x1), _p(x1), _p(y1), _p(a1), _p(d1), _p(f1), C.byref(s1), len(x2),
                                            _p(x2), _p(y2), _p(o2), _p(a2), _p(d2), _p(f2), C.byref(s2), _p(F),
                                            C.c_float(epipole[0]), C.c_float(epipole[1]), _p(sf), _p(ls),
                                            int(check_ori), _p(out))
    return nm, out


def search_window_best(KF, q, chi2_gate=False, inv_sigma2=None):
    nq = len(q["u"])
    a = {k: np.ascontiguousarray(q[k], t) for k, t in
         (("valid", np.uint8), ("u", np.float32), ("v", np.float32), ("radius", np.float32),
          ("pred_level", np.int32), ("desc", np.uint8))}
    inv = np.zeros(8, np.float32) if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    bi, bd = np.full(nq, -1, np.int32), np.full(nq, 256, np.int32)
    fs = _fv(KF)
    lib().orc_search_window_best(C.byref(fs), nq, _p(a["valid"]), _p(a["u"]), _p(a["v"]), _p(a["radius"]),
                                 _p(a["pred_level"]), _p(a["desc"]), int(chi2_gate), _p(inv), _p(bi), _p(bd))
    return bi, bd


def search_window_greedy(F, q, max_dist, check_ori=True):
    nq = len(q["u"])
    a = {k: np.ascontiguousarray(q[k], t) for k, t in
         (("valid", np.uint8), ("u", np.float32), ("v", np.float32), ("radius", np.float32),
          ("min_level", np.int32), ("max_level", np.int32), ("desc", np.uint8), ("angle", np.float32))}
    out = np.full(F.n, -1, np.int32)
    fs = _fv(F)
    nm = lib().orc_search_window_greedy(C.byref(fs), nq, _p(a["valid"]), _p(a["u"]), _p(a["v"]), _p(a["radius"]),
                                        _p(a["min_level"]), _p(a["max_level"]), _p(a["desc"]), _p(a["angle"]),
                                        int(max_dist), int(check_ori), _p(out))
    return nm, out


def pose_optimization(Tcw, intr, Xw, obs, inv_sigma2):
    """Optimizer::PoseOptimization on flattened inputs.  Returns (n_inliers, Tcw_out[12], outlier[n], info)."""
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    K = np.ascontiguousarray(intr, np.float32).reshape(4)
    X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
    O = np.ascontiguousarray(obs, np.float32).reshape(-1, 2)
    W = np.ascontiguousarray(inv_sigma2, np.float32)
    Tout = T.copy()
    outl = np.zeros(len(X), np.uint8)
    info = np.zeros(2, np.int32)
    n = lib().orc_pose_optimization(_p(T), _p(K), len(X), _p(X), _p(O), _p(W), _p(Tout), _p(outl), _p(info))
    return n, Tout, outl, {"iterations": int(info[0]), "lm_trials": int(info[1])}


# ---- Frame post-processing (frame_oracle.h) --------------------------------------------------------
class OrcCamera(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


def camera(K, dist=(0, 0, 0, 0, 0)):
    d = list(dist) + [0.0] * (5 - len(dist))
    return OrcCamera(*[float(v) for v in K], *[float(v) for v in d])


def det_log(x):
    lib().orc_log.restype = C.c_double
    lib().orc_log.argtypes = [C.c_double]
    return float(lib().orc_log(float(x)))


def undistort_keypoints(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    lib().orc_undistort_keypoints(C.byref(cam), C.c_int32(len(xy)), _p(xy), _p(out))
    return out


def image_bounds(cam, width, height):
    b = np.zeros(4, np.float32)
    lib().orc_image_bounds(C.byref(cam), C.c_int32(width), C.c_int32(height), _p(b))
    return b


def assign_features_to_grid(xy_un, bounds):
    xy = np.ascontiguousarray(xy_un, np.float32).reshape(-1, 2)
    n = len(xy)
    b = np.ascontiguousarray(bounds, np.float32)
    cell_of = np.zeros(n, np.int32)
    cell_start = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(max(n, 1), np.int32)
    lib().orc_assign_features_to_grid.restype = C.c_int32
    inside = lib().orc_assign_features_to_grid(C.c_int32(n), _p(xy), _p(b), _p(cell_of), _p(cell_start), _p(items))
    return dict(cell_of=cell_of, cell_start=cell_start, cell_items=items[:inside])


def is_in_frustum(cam, bounds, Tcw, Xw, normal, max_dist, min_dist, viewing_cos_limit, log_scale_factor, n_levels,
                  init=None):
    b = np.ascontiguousarray(bounds, np.float32)
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
    lib().orc_is_in_frustum(C.byref(cam), _p(b), _p(T), C.c_int32(n), _p(X), _p(N), _p(mx), _p(mn),
                            C.c_float(viewing_cos_limit), C.c_float(log_scale_factor), C.c_int32(n_levels), _p(in_view),
                            _p(px), _p(py), _p(vc), _p(lvl))
    return dict(in_view=in_view, proj_x=px, proj_y=py, view_cos=vc, pred_level=lvl)


def project_last_frame(cam, bounds, Tcw, Xw, has_point):
    """Projection step of SearchByProjection(cur, last) (ORBmatcher.cc:1251-1270): (valid, u, v)."""
    b = np.ascontiguousarray(bounds, np.float32)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
    hp = np.ascontiguousarray(has_point, np.uint8)
    n = len(X)
    valid, u, v = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib().orc_project_last_frame(C.byref(cam), _p(b), _p(T), C.c_int32(n), _p(X), _p(hp), _p(valid), _p(u), _p(v))
    return valid, u, v


def distinctive_descriptors(offsets, descriptors):
    """MapPoint::ComputeDistinctiveDescriptors per map point: (best_idx, best_median) arrays, -1 for empty points."""
    off = np.ascontiguousarray(offsets, np.int32)
    d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
    n = len(off) - 1
    idx, med = np.full(n, -1, np.int32), np.full(n, 2 ** 31 - 1, np.int32)
    lib().orc_distinctive_descriptor.restype = C.c_int
    for p in range(n):
        k = int(off[p + 1] - off[p])
        if k > 0:
            m = C.c_int32(0)
            blk = np.ascontiguousarray(d[off[p]:off[p + 1]])
            idx[p] = lib().orc_distinctive_descriptor(_p(blk), C.c_int32(k), C.byref(m))
            med[p] = m.value
    return idx, med


# ---- cross-agent keyframe candidate search (oracle/kfsearch_oracle.c + orc_search_by_bow) ----------------------
class _OneNode:
    """DBoW2::FeatureVector with every feature in ONE node, indices in stored (ascending) order."""

    def __init__(self, n):
        self.node_id = np.zeros(1, np.int32)
        self.off = np.array([0, n], np.int32)
        self.idx = np.arange(n, dtype=np.int32)


def kf_votes(query, kf, th_low=50, nn_ratio=0.75):
    """Detection score of one stored keyframe for one query keyframe; both dicts: desc (n,32) u8, valid (n,) u8."""
    d1, v1 = np.ascontiguousarray(query["desc"], np.uint8), np.ascontiguousarray(query["valid"], np.uint8)
    d2, v2 = np.ascontiguousarray(kf["desc"], np.uint8), np.ascontiguousarray(kf["valid"], np.uint8)
    f = lib().orc_kf_votes
    f.restype = C.c_int
    return int(f(C.c_int32(len(d1)), _p(d1), _p(v1), C.c_int32(len(d2)), _p(d2), _p(v2), C.c_int32(int(th_low)),
                 C.c_float(nn_ratio)))


def kf_search(query, store, th_low=50, nn_ratio=0.75, check_ori=True, min_votes=20, min_matches=20, max_candidates=16):
    """AgentMediator::CheckOverlapCandidates + GetSim3's matching (code/src/AgentMediator.cc:177-191,204-262) over a
    store = list of keyframe dicts (agent, keyframe_id, desc, angle, valid) in slot order (None = empty slot); query: the
    same fields.  Returns (votes per slot (-1: empty / own agent), list of (slot, votes, n_matches, match_of_1) for the
    candidates that reach min_matches, number of keyframes phase 2 looked at)."""
    assert th_low == 50, "orc_search_by_bow has TH_LOW built in (code/src/ORBmatcher.cc:38)"
    votes = np.full(len(store), -1, np.int32)
    for k, kf in enumerate(store):
        if kf is None or kf["agent"] == query["agent"]:
            continue
        votes[k] = kf_votes(query, kf, th_low, nn_ratio)
    order = [k for k in np.argsort(-votes, kind="stable") if votes[k] >= min_votes and votes[k] > 0][:max_candidates]
    out = []
    for k in order:
        kf = store[k]
        nm, _, m1 = search_by_bow(1, query, _OneNode(len(query["desc"])), kf, _OneNode(len(kf["desc"])), nn_ratio, check_ori)
        if nm >= min_matches:
            out.append((int(k), int(votes[k]), int(nm), m1))
    return votes, out, len(order)


# ---- projection + gating half of Fuse / SearchBySim3 / keyframe-side SearchByProjection (project_oracle.h) ----------
class OrcMapPointView(C.Structure):
    _fields_ = [("n", C.c_int32), ("Xw", C.c_void_p), ("normal", C.c_void_p), ("max_dist", C.c_void_p),
                ("min_dist", C.c_void_p), ("desc", C.c_void_p), ("valid", C.c_void_p)]


class OrcWindowQueries(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("active", "u", "v", "radius", "level")]


def _mpv(mp):
    """mp: dict with Xw, normal, max_dist, min_dist, desc, valid.  Returns (struct, keep-alive arrays)."""
    a = dict(Xw=np.ascontiguousarray(mp["Xw"], np.float32), normal=np.ascontiguousarray(mp["normal"], np.float32),
             max_dist=np.ascontiguousarray(mp["max_dist"], np.float32),
             min_dist=np.ascontiguousarray(mp["min_dist"], np.float32), desc=np.ascontiguousarray(mp["desc"], np.uint8),
             valid=np.ascontiguousarray(mp["valid"], np.uint8))
    n = len(a["max_dist"])
    return OrcMapPointView(n, _p(a["Xw"]), _p(a["normal"]), _p(a["max_dist"]), _p(a["min_dist"]), _p(a["desc"]),
                           _p(a["valid"])), a


def _wq(n):
    q = dict(active=np.zeros(n, np.uint8), u=np.zeros(n, np.float32), v=np.zeros(n, np.float32),
             radius=np.zeros(n, np.float32), level=np.zeros(n, np.int32))
    return OrcWindowQueries(_p(q["active"]), _p(q["u"]), _p(q["v"]), _p(q["radius"]), _p(q["level"])), q


def _f12(a, n=12):
    return np.ascontiguousarray(a, np.float32).reshape(n)


def sim3_decompose(Scw):
    S = _f12(Scw)
    R, t, Ow = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_sim3_decompose(_p(S), _p(R), _p(t), _p(Ow))
    return R.reshape(3, 3), t, Ow


def fuse_queries(KF, cam, Tcw, log_sf, mp, th):
    ms, _keep = _mpv(mp)
    qs, q = _wq(ms.n)
    fs, T = _fv(KF), _f12(Tcw)
    lib().orc_fuse_queries(C.byref(fs), C.byref(cam), _p(T), C.c_float(log_sf), C.byref(ms), C.c_float(th), C.byref(qs))
    return q


def sim3_world_queries(KF, cam, Scw, log_sf, mp, th):
    ms, _keep = _mpv(mp)
    qs, q = _wq(ms.n)
    fs, S = _fv(KF), _f12(Scw)
    lib().orc_sim3_world_queries(C.byref(fs), C.byref(cam), _p(S), C.c_float(log_sf), C.byref(ms), C.c_float(th),
                                 C.byref(qs))
    return q


def sim3_pair_queries(KF_target, cam, T_src_w, s12, R12, t12, to_2, log_sf, mp, th):
    """One direction of SearchBySim3: to_2 = True projects keyframe 1's points into keyframe 2 (sR21, t21)."""
    R, t = _f12(R12, 9), _f12(t12, 3)
    sR12, sR21, t21 = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib().orc_sim3_relative(C.c_float(s12), _p(R), _p(t), _p(sR12), _p(sR21), _p(t21))
    ms, _keep = _mpv(mp)
    qs, q = _wq(ms.n)
    fs, T = _fv(KF_target), _f12(T_src_w)
    sR, tt = (sR21, t21) if to_2 else (sR12, t)
    lib().orc_sim3_pair_queries(C.byref(fs), C.byref(cam), _p(T), _p(sR), _p(tt), C.c_float(log_sf), C.byref(ms),
                                C.c_float(th), C.byref(qs))
    return q


def frame_kf_queries(F, cam, Tcw, log_sf, mp, th):
    ms, _keep = _mpv(mp)
    qs, q = _wq(ms.n)
    fs, T = _fv(F), _f12(Tcw)
    lib().orc_frame_kf_queries(C.byref(fs), C.byref(cam), _p(T), C.c_float(log_sf), C.byref(ms), C.c_float(th),
                               C.byref(qs))
    return q


def fuse(KF, cam, Tcw, log_sf, inv_sigma2, mp, th):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) up to the side effects.  Returns (nFused, best_idx, best_dist)."""
    ms, _keep = _mpv(mp)
    bi, bd = np.full(ms.n, -1, np.int32), np.full(ms.n, 256, np.int32)
    fs, T, inv = _fv(KF), _f12(Tcw), np.ascontiguousarray(inv_sigma2, np.float32)
    n = lib().orc_fuse(C.byref(fs), C.byref(cam), _p(T), C.c_float(log_sf), _p(inv), C.byref(ms), C.c_float(th), _p(bi),
                       _p(bd))
    return n, bi, bd


def fuse_sim3(KF, cam, Scw, log_sf, mp, th):
    ms, _keep = _mpv(mp)
    bi, bd = np.full(ms.n, -1, np.int32), np.full(ms.n, 256, np.int32)
    fs, S = _fv(KF), _f12(Scw)
    n = lib().orc_fuse_sim3(C.byref(fs), C.byref(cam), _p(S), C.c_float(log_sf), C.byref(ms), C.c_float(th), _p(bi), _p(bd))
    return n, bi, bd


def search_by_sim3(KF1, KF2, cam, T1w, T2w, s12, R12, t12, log_sf1, log_sf2, mp1, mp2, th):
    m1, _k1 = _mpv(mp1)
    m2, _k2 = _mpv(mp2)
    out = np.full(m1.n, -1, np.int32)
    f1, f2 = _fv(KF1), _fv(KF2)
    a, b, R, t = _f12(T1w), _f12(T2w), _f12(R12, 9), _f12(t12, 3)
    n = lib().orc_search_by_sim3(C.byref(f1), C.byref(f2), C.byref(cam), _p(a), _p(b), C.c_float(s12), _p(R), _p(t),
                                 C.c_float(log_sf1), C.c_float(log_sf2), C.byref(m1), C.byref(m2), C.c_float(th), _p(out))
    return n, out


def search_by_projection_sim3(KF, cam, Scw, log_sf, mp, th):
    ms, _keep = _mpv(mp)
    out = np.full(KF.n, -1, np.int32)
    fs, S = _fv(KF), _f12(Scw)
    n = lib().orc_search_by_projection_sim3(C.byref(fs), C.byref(cam), _p(S), C.c_float(log_sf), C.byref(ms), int(th),
                                            _p(out))
    return n, out


def search_by_projection_frame_kf(F, cam, Tcw, log_sf, mp, mp_angle, th, orb_dist, check_ori=True):
    ms, _keep = _mpv(mp)
    out = np.full(F.n, -1, np.int32)
    fs, T, ang = _fv(F), _f12(Tcw), np.ascontiguousarray(mp_angle, np.float32)
    n = lib().orc_search_by_projection_frame_kf(C.byref(fs), C.byref(cam), _p(T), C.c_float(log_sf), C.byref(ms), _p(ang),
                                                C.c_float(th), int(orb_dist), int(check_ori), _p(out))
    return n, out


# ---- the local-mapping thread's per-point loops (mapping_oracle.h) ------------------------------------------------
class OrcTriKeyframe(C.Structure):
    _fields_ = [("Tcw", C.c_float * 12)] + [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "invfx", "invfy")] + \
               [("scale_factors", C.c_void_p), ("level_sigma2", C.c_void_p)]


def _tri_kf(kf, keep):
    sf, ls = np.ascontiguousarray(kf["scale_factors"], np.float32), np.ascontiguousarray(kf["level_sigma2"], np.float32)
    keep += [sf, ls]
    fx, fy, cx, cy = [np.float32(v) for v in kf["K"]]
    s = OrcTriKeyframe()
    s.Tcw[:] = [float(v) for v in np.asarray(kf["Tcw"], np.float32).reshape(12)]
    s.fx, s.fy, s.cx, s.cy = float(fx), float(fy), float(cx), float(cy)
    s.invfx, s.invfy = float(np.float32(1.0) / fx), float(np.float32(1.0) / fy)  # invfx = 1.0f / fx, Frame.cc:266
    s.scale_factors, s.level_sigma2 = _p(sf), _p(ls)
    return s


def svd4_last_row(A):
    A = np.ascontiguousarray(A, np.float32).reshape(16)
    v = np.zeros(4, np.float32)
    lib().orc_svd4_last_row(_p(A), _p(v))
    return v


def triangulate_matches(kf1, kf2, ratio_factor, xy1, octave1, xy2, octave2):
    """CreateNewMapPoints' per-match body for the matches of kf1 with ONE neighbour kf2.  Returns (ok, x3D)."""
    keep = []
    a, b = _tri_kf(kf1, keep), _tri_kf(kf2, keep)
    x1, o1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2), np.ascontiguousarray(octave1, np.int32)
    x2, o2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2), np.ascontiguousarray(octave2, np.int32)
    n = len(o1)
    ok, X = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
    lib().orc_triangulate_matches(C.byref(a), C.byref(b), C.c_float(ratio_factor), n, _p(x1), _p(o1), _p(x2), _p(o2), _p(ok), _p(X))
    return ok, X


def update_normal_and_depth(offsets, obs_Ow, Xw, ref_Ow, ref_level_scale, ref_last_scale, normal, max_dist, min_dist):
    off = np.ascontiguousarray(offsets, np.int32)
    n = len(off) - 1
    a = [np.ascontiguousarray(v, np.float32) for v in (obs_Ow, Xw, ref_Ow, ref_level_scale, ref_last_scale)]
    nrm, mx, mn = [np.array(v, np.float32, copy=True) for v in (normal, max_dist, min_dist)]
    lib().orc_update_normal_and_depth(n, _p(off), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(nrm), _p(mx), _p(mn))
    return nrm, mx, mn
