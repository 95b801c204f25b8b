"""Python mirror of ORB_SLAM2::Optimizer's bundle adjustment entry points (code/include/Optimizer.h:41-46)
over the C ABI, on flattened problems (see swarmmap_amd.synth.make_ba_problem for the dict layout)."""
import ctypes as C

import numpy as np

from . import _lib


class SoBaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("Tcw", C.c_void_p), ("fixed", C.c_void_p), ("intr", C.c_void_p),
                ("n_points", C.c_int32), ("Xw", C.c_void_p), ("n_edges", C.c_int32), ("edge_pose", C.c_void_p),
                ("edge_point", C.c_void_p), ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p)]


class SoBaOptions(C.Structure):
    _fields_ = [("its_stage1", C.c_int32), ("its_stage2", C.c_int32), ("robust", C.c_int32),
                ("huber_delta", C.c_float), ("chi2_threshold", C.c_float)]


class SoBaInfo(C.Structure):
    _fields_ = [("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("iterations_stage1", C.c_int32), ("iterations_stage2", C.c_int32), ("lm_trials", C.c_int32),
                ("aborted", C.c_int32), ("n_outliers", C.c_int32), ("gpu_ms", C.c_float), ("wall_ms", C.c_float),
                ("solve_ms", C.c_float), ("n_solves", C.c_int32), ("solve_gflop_structural", C.c_double),
                ("solve_gflop_dense", C.c_double), ("nnz_tiles", C.c_double), ("solver_path", C.c_int32), ("n_free_keyframes", C.c_int32),
                ("flow_timeouts", C.c_int32), ("pcg_iterations", C.c_int32)]


class SoPoseProblem(C.Structure):
    _fields_ = [("Tcw12", C.c_void_p), ("intr", C.c_void_p), ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("Tcw_out12", C.c_void_p), ("outlier", C.c_void_p), ("n_inliers", C.c_void_p),
                ("info", C.c_void_p)]


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def problem_struct(prob):
    """(SoBaProblem, arrays it points into) for a problem dictionary (synth.make_ba_problem layout)."""
    a = dict(Tcw=np.ascontiguousarray(prob["Tcw"], np.float32).reshape(-1, 12),
             fixed=np.ascontiguousarray(prob["fixed"], np.uint8),
             intr=np.ascontiguousarray(prob["intr"], np.float32).reshape(-1, 4),
             Xw=np.ascontiguousarray(prob["Xw"], np.float32).reshape(-1, 3),
             edge_pose=np.ascontiguousarray(prob["edge_pose"], np.int32),
             edge_point=np.ascontiguousarray(prob["edge_point"], np.int32),
             obs=np.ascontiguousarray(prob["obs"], np.float32).reshape(-1, 2),
             inv_sigma2=np.ascontiguousarray(prob["inv_sigma2"], np.float32))
    P = SoBaProblem(len(a["Tcw"]), _vp(a["Tcw"]), _vp(a["fixed"]), _vp(a["intr"]), len(a["Xw"]), _vp(a["Xw"]),
                    len(a["edge_pose"]), _vp(a["edge_pose"]), _vp(a["edge_point"]), _vp(a["obs"]), _vp(a["inv_sigma2"]))
    return P, a


class BaGroup:
    """so_ba_group: the local bundle adjustments of several Optimizer contexts (one per agent, each called from its own
    thread) as one chain of launches."""

    def __init__(self, device=0, window_us=0.0):
        self._lib = lib = _lib.load_library()
        lib.so_ba_group_create.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_void_p)]
        lib.so_ba_group_destroy.argtypes = [C.c_void_p]
        lib.so_ba_group_destroy.restype = None
        lib.so_ba_group_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._h = C.c_void_p()
        _lib.check(lib.so_ba_group_create(int(device), float(window_us), C.byref(self._h)))

    def stats(self):
        a = np.zeros(8, np.float64)
        _lib.check(self._lib.so_ba_group_stats(self._h, _vp(a)))
        return dict(zip(("rounds", "members_total", "grouped_launches", "ungrouped_launches", "rows_launched", "members", "window_us"), a.tolist()))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_ba_group_destroy(self._h)
            self._h = C.c_void_p()


class Optimizer:
    """One solver context per LocalMapping thread (device buffers are reused from call to call)."""

    def __init__(self, device=0):
        self._lib = lib = _lib.load_library()
        vp = C.c_void_p
        lib.so_ba_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.so_ba_destroy.argtypes = [vp]
        lib.so_ba_destroy.restype = None
        lib.so_ba_options_local.argtypes = [C.POINTER(SoBaOptions)]
        lib.so_ba_options_local.restype = None
        lib.so_ba_options_global.argtypes = [C.POINTER(SoBaOptions), C.c_int32, C.c_int32]
        lib.so_ba_options_global.restype = None
        lib.so_bundle_adjust.argtypes = [vp, C.POINTER(SoBaProblem), C.POINTER(SoBaOptions), vp, vp, vp, vp, vp,
                                         C.POINTER(SoBaInfo)]
        lib.so_bundle_adjust_set_solve_timing.argtypes = [vp, C.c_int]
        self._h = vp()
        _lib.check(lib.so_ba_create(int(device), C.byref(self._h)))

    def set_solve_timing(self, enabled):
        """HIP events around every reduced-system solve (info["solve_ms"], info["n_solves"]); off by default - the two
        event records idle the stream ~12 us per LM trial."""
        _lib.check(self._lib.so_bundle_adjust_set_solve_timing(self._h, 1 if enabled else 0))

    def set_linear_solver(self, solver="direct", rel_tolerance=0.0, max_iterations=0):
        """so_ba_set_linear_solver: "direct" (block-skyline Cholesky, default) or "pcg" (block-Jacobi preconditioned conjugate
        gradients over the nonzero 6 x 6 blocks of the reduced camera system; maps of 80 free keyframes and more)."""
        self._lib.so_ba_set_linear_solver.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
        _lib.check(self._lib.so_ba_set_linear_solver(self._h, {"direct": 0, "pcg": 1}[solver], float(rel_tolerance), int(max_iterations)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_ba_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def resident_stages(self):
        """so_ba_resident_stages: LM stages of this context that ran as one resident launch (SWARMORB_BA_RESIDENT=1, an experiment)."""
        n = C.c_longlong(0)
        self._lib.so_ba_resident_stages.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        _lib.check(self._lib.so_ba_resident_stages(self._h, C.byref(n)))
        return int(n.value)

    def set_group(self, group):
        """so_ba_set_group: this context's local bundle adjustments go out merged with the other members' (None: leave)."""
        self._lib.so_ba_set_group.argtypes = [C.c_void_p, C.c_void_p]
        _lib.check(self._lib.so_ba_set_group(self._h, group._h if group is not None else None))

    def _solve(self, prob, opt, pbStopFlag):
        a = dict(Tcw=np.ascontiguousarray(prob["Tcw"], np.float32).reshape(-1, 12),
                 fixed=np.ascontiguousarray(prob["fixed"], np.uint8),
                 intr=np.ascontiguousarray(prob["intr"], np.float32).reshape(-1, 4),
                 Xw=np.ascontiguousarray(prob["Xw"], np.float32).reshape(-1, 3),
                 edge_pose=np.ascontiguousarray(prob["edge_pose"], np.int32),
                 edge_point=np.ascontiguousarray(prob["edge_point"], np.int32),
                 obs=np.ascontiguousarray(prob["obs"], np.float32).reshape(-1, 2),
                 inv_sigma2=np.ascontiguousarray(prob["inv_sigma2"], np.float32))
        P = SoBaProblem(len(a["Tcw"]), _vp(a["Tcw"]), _vp(a["fixed"]), _vp(a["intr"]), len(a["Xw"]), _vp(a["Xw"]),
                        len(a["edge_pose"]), _vp(a["edge_pose"]), _vp(a["edge_point"]), _vp(a["obs"]),
                        _vp(a["inv_sigma2"]))
        Tout, Xout = np.zeros_like(a["Tcw"]), np.zeros_like(a["Xw"])
        outl = np.zeros(len(a["edge_pose"]), np.uint8)
        chi2 = np.zeros(len(a["edge_pose"]), np.float64)
        info = SoBaInfo()
        _lib.check(self._lib.so_bundle_adjust(self._h, C.byref(P), C.byref(opt), _vp(pbStopFlag), _vp(Tout), _vp(Xout),
                                              _vp(outl), _vp(chi2), C.byref(info)))
        return dict(Tcw=Tout, Xw=Xout, outlier=outl, chi2=chi2,
                    info={k: getattr(info, k) for k, _ in SoBaInfo._fields_})

    # Optimizer::LocalBundleAdjustment(pKF, pbStopFlag, pMap) on the gathered local window
    def LocalBundleAdjustment(self, window, pbStopFlag=None):
        opt = SoBaOptions()
        self._lib.so_ba_options_local(C.byref(opt))
        return self._solve(window, opt, pbStopFlag)

    # Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust)
    def BundleAdjustment(self, problem, nIterations=5, pbStopFlag=None, bRobust=True):
        opt = SoBaOptions()
        self._lib.so_ba_options_global(C.byref(opt), int(nIterations), int(bool(bRobust)))
        return self._solve(problem, opt, pbStopFlag)

    GlobalBundleAdjustment = BundleAdjustment

    # int Optimizer::PoseOptimization(Frame* pFrame, bGlobal) on the keypoints that have a map point
    def PoseOptimization(self, Tcw, intr, Xw, obs, inv_sigma2):
        lib = self._lib
        vp = C.c_void_p
        lib.so_pose_optimization.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(C.c_int32), vp]
        T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
        K = np.ascontiguousarray(intr, np.float32).reshape(4)
        X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
        O = np.ascontiguousarray(obs, np.float32).reshape(-1, 2)
        W = np.ascontiguousarray(inv_sigma2, np.float32)
        Tout = T.copy()
        outl = np.zeros(len(X), np.uint8)
        info = np.zeros(2, np.int32)
        n_in = C.c_int32(0)
        _lib.check(lib.so_pose_optimization(self._h, _vp(T), _vp(K), len(X), _vp(X), _vp(O), _vp(W), _vp(Tout),
                                            _vp(outl), C.byref(n_in), _vp(info)))
        return n_in.value, Tout, outl, {"iterations": int(info[0]), "lm_trials": int(info[1])}

    # so_pose_optimization_batch: several frames' problems (e.g. the agents a thread drives in lockstep) in ONE launch
    def PoseOptimizationBatch(self, problems):
        """problems: list of (Tcw, intr, Xw, obs, inv_sigma2); returns the list of PoseOptimization results."""
        arr = (SoPoseProblem * len(problems))()
        keep, outs = [], []
        for k, (Tcw, intr, Xw, obs, w) in enumerate(problems):
            T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
            K = np.ascontiguousarray(intr, np.float32).reshape(4)
            X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
            O = np.ascontiguousarray(obs, np.float32).reshape(-1, 2)
            W = np.ascontiguousarray(w, np.float32)
            Tout, outl, info, n_in = T.copy(), np.zeros(len(X), np.uint8), np.zeros(2, np.int32), np.zeros(1, np.int32)
            keep.append((T, K, X, O, W))
            outs.append((n_in, Tout, outl, info))
            arr[k] = SoPoseProblem(T.ctypes.data, K.ctypes.data, len(X), X.ctypes.data, O.ctypes.data, W.ctypes.data,
                                   Tout.ctypes.data, outl.ctypes.data, n_in.ctypes.data, info.ctypes.data)
        self._lib.so_pose_optimization_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        _lib.check(self._lib.so_pose_optimization_batch(self._h, len(problems), C.cast(arr, C.c_void_p)))
        return [(int(n_in[0]), Tout, outl, {"iterations": int(info[0]), "lm_trials": int(info[1])}) for n_in, Tout, outl, info in outs]

    # the same call in two halves (so_pose_optimization_submit / _wait): host work placed between them runs under the kernel
    def PoseOptimizationSubmit(self, Tcw, intr, Xw, obs, inv_sigma2):
        lib = self._lib
        vp = C.c_void_p
        lib.so_pose_optimization_submit.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp]
        T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
        K = np.ascontiguousarray(intr, np.float32).reshape(4)
        X = np.ascontiguousarray(Xw, np.float32).reshape(-1, 3)
        O = np.ascontiguousarray(obs, np.float32).reshape(-1, 2)
        W = np.ascontiguousarray(inv_sigma2, np.float32)
        _lib.check(lib.so_pose_optimization_submit(self._h, _vp(T), _vp(K), len(X), _vp(X), _vp(O), _vp(W)))
        self._pose_pending = (T, len(X))

    def PoseOptimizationWait(self):
        lib = self._lib
        vp = C.c_void_p
        lib.so_pose_optimization_wait.argtypes = [vp, vp, vp, C.POINTER(C.c_int32), vp]
        pending = getattr(self, "_pose_pending", None)
        n = pending[1] if pending else 0
        Tout = pending[0].copy() if pending else np.zeros(12, np.float32)
        outl = np.zeros(n, np.uint8)
        info = np.zeros(2, np.int32)
        n_in = C.c_int32(0)
        self._pose_pending = None
        _lib.check(lib.so_pose_optimization_wait(self._h, _vp(Tout), _vp(outl), C.byref(n_in), _vp(info)))
        return n_in.value, Tout, outl, {"iterations": int(info[0]), "lm_trials": int(info[1])}

    def pose_kernel_ms(self):
        ms = C.c_float(0)
        self._lib.so_pose_optimization_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _lib.check(self._lib.so_pose_optimization_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def solve(self, problem, its1, its2, robust, huber_delta, chi2_threshold=5.991, pbStopFlag=None):
        opt = SoBaOptions(int(its1), int(its2), int(bool(robust)), float(huber_delta), float(chi2_threshold))
        return self._solve(problem, opt, pbStopFlag)
