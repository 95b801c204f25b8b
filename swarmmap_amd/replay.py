"""ctypes view of swarmmap_amd/host/replay.cc (libswarmorb_replay.so): the tracking thread's chained, device-resident
per-frame calls and the local-mapping thread as a C++ host loop over the C ABI — the host side a SwarmMap integration
has.  Used by bench.py (timed region) and by tests/ (frame-by-frame comparison with the CPU oracle chain)."""
import ctypes as C
import os

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))

STAT = ("steps", "extract_ms", "m2_ms", "pose1_ms", "m1_ms", "pose2_ms", "pose3_ms", "map_ms", "lba_ms", "n_kp", "n_m2",
        "n_m1", "n_inliers", "match_kernel_ms", "pose_kernel_ms", "pose_trials", "pose_calls", "pose_points", "n_lba",
        "lba_busy_ms", "lba_gpu_ms", "solve_ms", "n_solves", "n_local", "n_in_view", "n_keyframes", "n_map_points",
        "m2_enqueue_ms", "m2_wait_ms", "m1_enqueue_ms", "m1_wait_ms", "pose_timed_calls", "timed_frames")


LM_STAT = ("jobs", "wall_ms", "node_ms", "tri_calls", "tri_ms", "tri_kernel_ms", "tri_matches", "fuse_calls", "fuse_ms",
           "fuse_kernel_ms", "fused", "fuse_points", "tri_queries", "batch_ms", "batch_end_ms", "batch_kernel_ms",
           "triangulate_ms", "triangulate_kernel_ms", "new_points", "stage_tri_ms", "stage_fuse_ms", "stage_back_ms",
           "batch_enqueue_ms", "batch_wait_ms",
           # the closed loop's job (closedloop.cc): ProcessNewKeyFrame + culling, the Fuse batch's kernels, applying the Fuse
           # results, the window gather, the so_bundle_adjust call, its write-back, local BA as a whole, the whole job
           "cl_process_ms", "cl_fuse_kernel_ms", "cl_apply_ms", "cl_gather_ms", "cl_solver_ms", "cl_writeback_ms", "cl_lba_ms",
           "cl_job_ms", "cl_wb_apply_ms", "cl_und_build_ms", "cl_und_call_ms", "cl_packet_ms", "cl_tri_batch_end_ms")


def make_vocabulary(n=100, seed=20221001):
    """The vocabulary stand-in of the local-mapping matcher job: n seeded random 256-bit centroids; a feature's node is
    its nearest centroid (lowest index on ties) - what one level of a DBoW2 tree does with trained centroids."""
    return np.random.default_rng(seed).integers(0, 256, (n, 32)).astype(np.uint8)


class Replay:
    def __init__(self, dev, w, h, nfeatures, lba_every, K, dist=None, keyframe_every=8, keyframe_ratio=0.7, plane_z=2.0,
                 local_keyframes=0, third_pose=True):
        from .optimizer import SoBaProblem  # noqa: F401  (binds the BA structs)
        _lib.load_library()  # binds HIP through torch's runtime first
        path = os.path.join(_HERE, "libswarmorb_replay.so")
        if not os.path.exists(path):
            raise _lib.SwarmOrbError("libswarmorb_replay.so is missing: run __graft_entry__.build()")
        self.lib = lib = C.CDLL(path)
        vp, i32, f = C.c_void_p, C.c_int, C.c_float
        lib.so_replay_create.argtypes = [i32, i32, i32, i32, i32, vp, vp, i32, f, f, i32, i32, C.POINTER(vp)]
        lib.so_replay_destroy.argtypes = [vp]; lib.so_replay_destroy.restype = None
        lib.so_replay_error.argtypes = [vp]; lib.so_replay_error.restype = C.c_char_p
        lib.so_replay_set_frames.argtypes = [vp, vp, i32, i32]
        lib.so_replay_set_window.argtypes = [vp, vp]
        lib.so_replay_set_profiling.argtypes = [vp, i32]
        lib.so_replay_preallocate.argtypes = [vp]
        lib.so_replay_prime.argtypes = [vp, i32]
        lib.so_replay_run.argtypes = [vp, i32, i32, i32]
        lib.so_replay_drain.argtypes = [vp]
        lib.so_replay_finish.argtypes = [vp]
        lib.so_replay_stats.argtypes = [vp, vp]
        lib.so_replay_log_size.argtypes = [vp]
        lib.so_replay_log.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.so_replay_last_frame.argtypes = [vp, C.POINTER(vp), C.POINTER(i32)]
        lib.so_replay_extractor.argtypes = [vp]; lib.so_replay_extractor.restype = vp
        lib.so_replay_matcher.argtypes = [vp]; lib.so_replay_matcher.restype = vp
        lib.so_replay_last_dframe.argtypes = [vp]; lib.so_replay_last_dframe.restype = vp
        lib.so_fleet_run.argtypes = [vp, i32, i32, i32, i32]
        lib.so_replay_set_vocabulary.argtypes = [vp, vp, i32, i32]
        lib.so_replay_lm_stats.argtypes = [vp, vp]
        lib.so_replay_lm_log.argtypes = [vp, vp, i32]
        self.h = vp()
        K4 = np.ascontiguousarray(K, np.float32)
        d5 = None if dist is None else np.ascontiguousarray(list(dist) + [0.0] * (5 - len(dist)), np.float32)
        self._check(lib.so_replay_create(dev, w, h, nfeatures, lba_every, self._p(K4), None if d5 is None else self._p(d5),
                                         keyframe_every, keyframe_ratio, plane_z, local_keyframes, int(third_pose),
                                         C.byref(self.h)), "create")
        self._keep = []

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("so_replay_%s failed (%d): %s" % (what, rc, (self.lib.so_replay_error(self.h) or b"").decode()))

    @staticmethod
    def _p(a):
        return a.ctypes.data

    def set_frames(self, ptrs, on_device=False):
        a = np.array(ptrs, np.uint64)
        self._check(self.lib.so_replay_set_frames(self.h, self._p(a), len(a), int(on_device)), "set_frames")

    def set_host_frames(self, frames):
        """frames: list of (h, w) uint8 C-contiguous arrays (kept alive by this object)."""
        self._keep.append(frames)
        self.set_frames([f.ctypes.data for f in frames], on_device=False)

    def set_window(self, prob):
        from .optimizer import problem_struct
        st, keep = problem_struct(prob)
        self._keep.append(keep)
        self._check(self.lib.so_replay_set_window(self.h, C.byref(st)), "set_window")

    def set_vocabulary(self, centroids, neighbours=20):
        """Switches the local-mapping thread's matcher job on (SearchForTriangulation + Fuse against the last
        `neighbours` keyframes before every window): centroids = the vocabulary stand-in, n x 32 bytes."""
        c = np.ascontiguousarray(centroids, np.uint8).reshape(-1, 32)
        self._check(self.lib.so_replay_set_vocabulary(self.h, self._p(c), len(c), int(neighbours)), "set_vocabulary")

    def lm_stats(self):
        """Sums over the timed matcher jobs of the local-mapping thread."""
        a = np.zeros(40, np.float64)
        self.lib.so_replay_lm_stats(self.h, self._p(a))
        return dict(zip(LM_STAT, a[:len(LM_STAT)].tolist()))

    def lm_log(self):
        """Every matcher job so far: rows (frame, neighbours, triangulation matches, fused into neighbours, fused back, new
        map points)."""
        n = self.lib.so_replay_lm_log(self.h, None, 0)
        out = np.zeros((max(n, 1), 6), np.int32)
        n = min(n, self.lib.so_replay_lm_log(self.h, self._p(out), len(out)))
        return out[:n]

    def set_closed_loop(self, kf_every=5, delay=None, n_free=25, n_fixed=40, policy=0):
        """so_replay_set_closed_loop: every keyframe's local-mapping results flow back into the tracked map
        (swarmmap_amd/closedloop.py is the same loop in Python).  After set_vocabulary, before the first frame."""
        self.lib.so_replay_set_closed_loop.argtypes = [C.c_void_p] + [C.c_int] * 5
        self._check(self.lib.so_replay_set_closed_loop(self.h, kf_every, kf_every if delay is None else delay, n_free, n_fixed,
                                                       policy), "set_closed_loop")

    def set_fleet_offset(self, offset):
        """so_replay_set_fleet_offset: inside fleet_run this agent tracks frame (tick + offset) - run it alone for `offset` frames first."""
        self.lib.so_replay_set_fleet_offset.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.so_replay_set_fleet_offset(self.h, int(offset)), "set_fleet_offset")

    def fleet_ticks(self):
        """so_replay_fleet_ticks (on the fleet's first agent): (ticks driven, agent places filled) of fleet_run's elastic ticks."""
        t, n = C.c_longlong(0), C.c_longlong(0)
        self.lib.so_replay_fleet_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        self._check(self.lib.so_replay_fleet_ticks(self.h, C.byref(t), C.byref(n)), "fleet_ticks")
        return int(t.value), int(n.value)

    def set_track_chain(self, on):
        """so_replay_set_track_chain: the tracking stages as device chains (default) or as separate calls (same results)."""
        self.lib.so_replay_set_track_chain.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.so_replay_set_track_chain(self.h, int(bool(on))), "set_track_chain")

    def closed_loop_log(self):
        """After drain(): dict(lm_log (rows of closedloop.LM_LOG_COLUMNS), kf_t, kf_poses (final), ref_kf, Tcr,
        final_centres (System::SaveTrajectoryTUM: Tcr x final keyframe pose), kf_centres, counts)."""
        vp, i32 = C.c_void_p, C.c_int
        lib = self.lib
        lib.so_replay_cl_log.argtypes = [vp, vp, i32]
        lib.so_replay_cl_keyframes.argtypes = [vp, vp, vp, i32]
        lib.so_replay_cl_frames.argtypes = [vp, vp, vp, i32]
        lib.so_replay_cl_counts.argtypes = [vp, vp, vp]
        n = lib.so_replay_cl_log(self.h, None, 0)
        lm = np.zeros((max(n, 1), 12), np.int64)
        n = min(n, lib.so_replay_cl_log(self.h, self._p(lm), len(lm)))
        nk = lib.so_replay_cl_keyframes(self.h, None, None, 0)
        kf_t, kf_T = np.zeros(max(nk, 1), np.int32), np.zeros((max(nk, 1), 12), np.float32)
        lib.so_replay_cl_keyframes(self.h, self._p(kf_t), self._p(kf_T), nk)
        nf = lib.so_replay_cl_frames(self.h, None, None, 0)
        ref, Tcr = np.zeros(max(nf, 1), np.int32), np.zeros((max(nf, 1), 4, 4), np.float64)
        lib.so_replay_cl_frames(self.h, self._p(ref), self._p(Tcr), nf)
        counts, wait = np.zeros(8, np.int64), C.c_double(0)
        self._check(lib.so_replay_cl_counts(self.h, self._p(counts), C.byref(wait)), "cl_counts")
        kf_t, kf_T, ref, Tcr = kf_t[:nk], kf_T[:nk], ref[:nf], Tcr[:nf]

        def t44(p):
            T = np.eye(4)
            T[:3, :4] = np.asarray(p, np.float64).reshape(3, 4)
            return T
        fin = []
        for rk, Tc in zip(ref, Tcr):
            Tw = Tc @ t44(kf_T[rk])
            fin.append(-Tw[:3, :3].T @ Tw[:3, 3])
        kfc = [-t44(p)[:3, :3].T @ t44(p)[:3, 3] for p in kf_T]
        lib.so_replay_cl_bindings.argtypes = [vp, i32, vp, i32]
        lib.so_replay_cl_points.argtypes = [vp, vp, vp, i32]
        binds = []
        for k in range(nk):
            m = lib.so_replay_cl_bindings(self.h, k, None, 0)
            a = np.full(max(m, 1), -1, np.int32)
            lib.so_replay_cl_bindings(self.h, k, self._p(a), m)
            binds.append(a[:m])
        npnt = lib.so_replay_cl_points(self.h, None, None, 0)
        bad, repl = np.zeros(max(npnt, 1), np.uint8), np.zeros(max(npnt, 1), np.int32)
        lib.so_replay_cl_points(self.h, self._p(bad), self._p(repl), npnt)
        return dict(lm_log=lm[:n], kf_t=kf_t, kf_poses=kf_T, ref_kf=ref, Tcr=Tcr, final_centres=np.array(fin).reshape(-1, 3),
                    kf_bindings=binds, point_bad=bad[:npnt], point_replaced_by=repl[:npnt],
                    kf_centres=np.array(kfc).reshape(-1, 3),
                    counts=dict(zip(("jobs", "windows", "windows_aborted", "interrupt_ba", "map_slots", "bad_points", "keyframes",
                                     "packet_apply_us"), counts[:8].tolist())), wait_ms=wait.value)

    def preallocate(self):
        self._check(self.lib.so_replay_preallocate(self.h), "preallocate")

    def set_profiling(self, on):
        self.lib.so_replay_set_profiling(self.h, int(on))

    def prime(self, t):
        self._check(self.lib.so_replay_prime(self.h, t), "prime")

    def run(self, first_t, n, timed):
        self._check(self.lib.so_replay_run(self.h, first_t, n, int(timed)), "run")

    def run_live(self, first_t, n):
        """so_replay_run_live: frames handed over one by one, nothing extracted ahead (a live camera).  Returns
        (image-in -> pose-out ms per frame, whole-step ms per frame)."""
        pose, step = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
        self.lib.so_replay_run_live.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._check(self.lib.so_replay_run_live(self.h, first_t, n, self._p(pose), self._p(step)), "run_live")
        return pose[:n], step[:n]

    @staticmethod
    def fleet_run(agents, first_t, n, timed):
        """so_fleet_run: the agents (Replay objects of one GPU, created after private_streams(True)) walk through n
        frames in lockstep on the calling thread; PoseOptimization of all agents goes out as one launch."""
        arr = (C.c_void_p * len(agents))(*[a.h for a in agents])
        rc = agents[0].lib.so_fleet_run(arr, len(agents), first_t, n, int(timed))
        if rc != 0:
            errs = [(a.lib.so_replay_error(a.h) or b"").decode() for a in agents]
            raise RuntimeError("so_fleet_run failed (%d): %s" % (rc, "; ".join(e for e in errs if e)))

    def drain(self):
        self._check(self.lib.so_replay_drain(self.h), "drain")

    def finish(self):
        self._check(self.lib.so_replay_finish(self.h), "finish")

    def stats(self):
        a = np.zeros(48, np.float64)
        self.lib.so_replay_stats(self.h, self._p(a))
        d = dict(zip(STAT, a[:len(STAT)].tolist()))
        from .extractor import STAGES
        d["stages"] = dict(zip(STAGES, a[len(STAT):len(STAT) + len(STAGES)].tolist()))
        d["n_reruns"], d["n_wide_m2"] = a[len(STAT) + len(STAGES)], a[len(STAT) + len(STAGES) + 1]
        d["lba_trials"] = a[len(STAT) + len(STAGES) + 2]  # all windows; solve_ms / n_solves cover the event-timed ones (1 in 4)
        n = self.lib.so_replay_log_size(self.h)
        ms = np.zeros(max(n, 1), np.float32)
        self.lib.so_replay_frame_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        got = self.lib.so_replay_frame_ms(self.h, self._p(ms), n)
        d["frame_ms"] = ms[:max(got, 0)]  # wall time of every tracked frame so far (warm-up frames first)
        return d

    def log(self):
        n = self.lib.so_replay_log_size(self.h)
        poses = np.zeros((n, 12), np.float32)
        cols = [np.zeros(n, np.int32) for _ in range(4)]
        self._check(self.lib.so_replay_log(self.h, self._p(poses), *[self._p(c) for c in cols]), "log")
        T = poses.reshape(n, 3, 4).astype(np.float64)
        centres = -np.einsum("nji,nj->ni", T[:, :, :3], T[:, :, 3])
        return dict(poses=poses, centres=centres, matches_last=cols[0], matches_map=cols[1], inliers=cols[2],
                    n_map_points=cols[3])

    def last_descriptors(self):
        ptr, n = C.c_void_p(), C.c_int(0)
        self.lib.so_replay_last_frame(self.h, C.byref(ptr), C.byref(n))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (max(n.value, 1) * 32,))[:n.value * 32].reshape(-1, 32).copy()

    def last_dframe(self):
        """so_dframe handle of the frame tracked last (device-resident descriptors for the exchange tick)."""
        return C.c_void_p(self.lib.so_replay_last_dframe(self.h))

    def last_bindings(self):
        """(keypoint -> map slot of the frame tracked last, -1 = none; its pose Tcw as 12 floats)."""
        ptr, n = C.c_void_p(), C.c_int(0)
        T = np.zeros(12, np.float32)
        self.lib.so_replay_last_bindings.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
        self._check(self.lib.so_replay_last_bindings(self.h, C.byref(ptr), C.byref(n), self._p(T)), "last_bindings")
        if n.value <= 0:
            return np.zeros(0, np.int32), T
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), (n.value,)).copy(), T

    def candidates_total(self, nlevels=8, cap=10000):
        exh = self.lib.so_replay_extractor(self.h)
        base = _lib.load_library()
        tot = 0
        xs, ys, sc = np.zeros(cap, np.int16), np.zeros(cap, np.int16), np.zeros(cap, np.uint8)
        for l in range(nlevels):
            n = C.c_int(0)
            base.so_extractor_get_candidates(C.c_void_p(exh), l, self._p(xs), self._p(ys), self._p(sc), cap, C.byref(n))
            tot += n.value
        return tot

    def close(self):
        if self.h:
            self.lib.so_replay_destroy(self.h)
            self.h = None


def private_streams(enabled=True):
    """so_runtime_private_streams: extractors / matchers created afterwards get HIP streams of their own (for a thread
    that drives several agents)."""
    lib = _lib.load_library()
    lib.so_runtime_private_streams.argtypes = [C.c_int]
    _lib.check(lib.so_runtime_private_streams(int(bool(enabled))))
