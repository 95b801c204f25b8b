"""A minimal monocular tracking replay over the hot path: extract -> frame post-processing -> SearchByProjection
(last frame) -> PoseOptimization -> isInFrustum -> SearchByProjection (local map) -> PoseOptimization, and, at
every "keyframe", LocalBundleAdjustment over the recent keyframes.

It is NOT the reference's Tracking state machine (out of scope): it is the shortest host loop that
chains every per-frame operator the way Tracking::TrackWithMotionModel (code/src/Tracking.cc:984-1050) and
Tracking::TrackLocalMap (:1052-1100) do, so that a *trajectory* can be produced from images alone and compared
between two implementations of the operators ("ATE").  The scene is the synthetic stream of
`synth.FrameStream`: a textured plane at depth `plane_z` seen by a camera translating parallel to it, so map points
are created by back-projecting keypoints onto the known plane (as an RGB-D front-end would) and ground truth is
known in closed form.

The operators come from a backend object; `HipBackend` binds the product (C ABI).  tests/ bind the CPU oracle to
the same interface and compare trajectories.
"""
import numpy as np


TH_LAST_FRAME = 15.0      # Tracking.cc:1014 (monocular)
MIN_MATCHES_MOTION = 20   # Tracking.cc:1020
COS_LIMIT = 0.5           # Tracking.cc:1133 isInFrustum(pMP, 0.5)


class HipBackend:
    """The product's operators (C ABI through the ctypes mirrors), device-resident: two DeviceFrames alternate on one
    extractor, the map lives in a DeviceMap, the two tracking searches build their queries on the GPU."""

    name = "hip"

    def __init__(self, K, nfeatures=1000, device=0, dist=(0, 0, 0, 0, 0)):
        from .dframe import DeviceFrame, DeviceMap
        from .extractor import ORBextractor
        from .matcher import ORBmatcher
        from .optimizer import Optimizer
        self.ex = ORBextractor(nfeatures, 1.2, 8, 20, 7, device=device)
        self.frames = [DeviceFrame(self.ex, K, dist), DeviceFrame(self.ex, K, dist)]
        self.cur = -1                                          # index of the current frame's handle
        self.map = DeviceMap(device)
        self.m_last = ORBmatcher(0.9, True, device=device)    # Tracking.cc:998
        self.m_map = ORBmatcher(0.8, True, device=device)     # Tracking.cc:1153
        self.opt = Optimizer(device=device)
        self.lba = Optimizer(device=device)                   # the local-mapping thread's solver context

    def tables(self):
        return (self.ex.GetScaleFactors(), self.ex.GetInverseScaleSigmaSquares())

    def new_frame(self, img):
        self.cur = (self.cur + 1) % 2
        f = self.frames[self.cur]
        kps, xy_un, desc = f(img)
        return kps.copy(), xy_un.copy(), desc.copy(), f.bounds.copy()

    def search_last(self, Tcw, last_slot, th):
        from . import dframe as dfm
        return dfm.search_last_frame(self.m_last, self.frames[self.cur], self.frames[1 - self.cur], self.map, Tcw,
                                     last_slot, th)

    def search_local(self, Tcw, first, n_local, skip, excluded, th, log_sf):
        from . import dframe as dfm
        return dfm.search_local_map(self.m_map, self.frames[self.cur], self.map, Tcw, n_local, th, COS_LIMIT, log_sf,
                                    skip=skip, excluded=excluded, first_slot=first)

    def map_append(self, X, normal, max_d, min_d, desc):
        self.map.append(X, normal, max_d, min_d, desc)

    def map_set_positions(self, slots, X):
        self.map.write_positions(slots, X)

    def pose(self, Tcw, intr, Xw, obs, w):
        n, T, outl, _ = self.opt.PoseOptimization(Tcw, intr, Xw, obs, w)
        return n, T, outl

    def local_ba(self, window):
        r = self.lba.LocalBundleAdjustment(window)
        return r["Tcw"], r["Xw"], r["outlier"]

    def close(self):
        for o in self.frames + [self.map, self.ex, self.m_last, self.m_map, self.opt, self.lba]:
            o.close()


def _T44(T12):
    T = np.eye(4)
    T[:3, :4] = np.asarray(T12, np.float64).reshape(3, 4)
    return T


def ground_truth(stream, n_frames, K, plane_z):
    """Camera centres of `synth.FrameStream` frames 0..n-1 in the frame-0 camera's coordinates."""
    fx, fy = float(K[0]), float(K[1])
    o = np.array([stream.offset(t) for t in range(n_frames)], np.float64)
    c = np.zeros((n_frames, 3))
    c[:, 0] = (o[:, 0] - o[0, 0]) * plane_z / fx
    c[:, 1] = (o[:, 1] - o[0, 1]) * plane_z / fy
    return c


def _local_window(kfs, mp_X, intr, n_free=6, n_fixed=8):
    """The window Optimizer::LocalBundleAdjustment gathers (code/src/Optimizer.cc:436-560), flattened: the last
    n_free keyframes are free (keyframes 0 and 1 never: together they pin the monocular scale, as the initial map
    does in the reference), the map points they see are the local points, up to n_fixed older keyframes seeing those
    points are fixed.  Points with a single observation in the window are left out (they
    were back-projected from one view).  Returns (problem, pose -> keyframe index, point -> map point index,
    edge -> (keyframe, observation) index) or None."""
    first_free = max(2, len(kfs) - n_free)
    free = list(range(first_free, len(kfs)))
    if not free:
        return None
    local = np.unique(np.concatenate([kfs[i]["mp"] for i in free]))
    older = [i for i in range(first_free - 1, -1, -1) if np.isin(kfs[i]["mp"], local).any()][:n_fixed]
    if len(older) < 2:
        older = sorted(set(older) | {0, 1})
    poses = sorted(older) + free
    count = np.zeros(len(mp_X), np.int32)
    for i in poses:
        np.add.at(count, kfs[i]["mp"][np.isin(kfs[i]["mp"], local)], 1)
    pts = local[count[local] >= 2]
    if len(pts) < 10:
        return None
    slot = np.full(len(mp_X), -1, np.int64)
    slot[pts] = np.arange(len(pts))
    e_pose, e_pt, obs, w, ref = [], [], [], [], []
    for p, i in enumerate(poses):
        sel = np.nonzero(slot[kfs[i]["mp"]] >= 0)[0]
        e_pose.append(np.full(len(sel), p, np.int32)); e_pt.append(slot[kfs[i]["mp"][sel]].astype(np.int32))
        obs.append(kfs[i]["uv"][sel]); w.append(kfs[i]["w"][sel])
        ref.append(np.stack([np.full(len(sel), i), sel], 1))
    prob = dict(Tcw=np.stack([kfs[i]["T"][:3, :4].reshape(12) for i in poses]).astype(np.float32),
                fixed=np.array([0 if i in free else 1 for i in poses], np.uint8),
                intr=np.tile(np.asarray(intr, np.float32), (len(poses), 1)), Xw=mp_X[pts].astype(np.float32),
                edge_pose=np.concatenate(e_pose), edge_point=np.concatenate(e_pt),
                obs=np.concatenate(obs).astype(np.float32), inv_sigma2=np.concatenate(w).astype(np.float32))
    return prob, poses, pts, np.concatenate(ref)


def track(backend, stream, n_frames, K, plane_z=2.0, keyframe_every=8, keyframe_ratio=0.7, local_ba=False,
          local_keyframes=0, third_pose=False, frames=None, on_frame=None):
    """Returns dict(centres (n,3), poses (n,12), matches_last, matches_map, inliers, n_map_points[, lba_*]).
    local_keyframes > 0: the local map (Tracking::UpdateLocalMap) is the points created at the last that many
    keyframes instead of the whole map.  third_pose: one more PoseOptimization per frame, from the last frame's pose
    over the final matches (what Tracking::TrackReferenceKeyFrame does when the motion model fails; the per-frame
    replay of SURVEY.md 8d counts three calls), result not used.  frames: optional list of images (else
    stream.frame(t)).  on_frame(t): called after each tracked frame; returning False ends the run."""
    intr = np.asarray(K, np.float32)
    fx, fy, cx, cy = [float(v) for v in intr]
    sf, inv_sigma2 = backend.tables()
    sf = np.asarray(sf, np.float32)
    inv_sigma2 = np.asarray(inv_sigma2, np.float32)
    nlevels = len(sf)
    log_sf = float(np.log(np.float32(1.2)))

    # the host's copy of the map: world position, descriptor, reference normal and scale-invariance distances
    # (MapPoint.cc:395-433); the backend keeps its own (device-resident for the HIP path)
    mp_X = np.zeros((0, 3), np.float32)
    kf_first_slot = []    # first map slot created at each keyframe

    def add_points(T, xy_un, kps, desc, sel):
        nonlocal mp_X
        R, t = T[:3, :3], T[:3, 3]
        rays = np.stack([(xy_un[sel, 0] - cx) / fx, (xy_un[sel, 1] - cy) / fy, np.ones(sel.sum())], 1)
        Ow = -R.T @ t
        dirs = rays @ R                                    # R^T r per row
        d = (plane_z - Ow[2]) / dirs[:, 2]
        X = Ow[None, :] + dirs * d[:, None]
        PO = X - Ow[None, :]
        dist = np.linalg.norm(PO, axis=1)
        mx = dist * sf[kps["octave"][sel]]
        first = len(mp_X)
        mp_X = np.concatenate([mp_X, X.astype(np.float32)])
        backend.map_append(X.astype(np.float32), (PO / dist[:, None]).astype(np.float32),
                           (1.2 * mx).astype(np.float32),                    # GetMaxDistanceInvariance
                           (0.8 * mx / sf[nlevels - 1]).astype(np.float32),  # GetMinDistanceInvariance
                           desc[sel])
        kf_first_slot.append(first)
        return first

    poses, centres = [], []
    kfs = []              # keyframes: pose + the observations LocalBundleAdjustment uses
    log = dict(matches_last=[], matches_map=[], inliers=[], n_map_points=[], lba_edges=[], lba_outliers=[])
    T_last = np.eye(4)
    velocity = np.eye(4)
    last = None           # (kps, kp_mp, outlier)
    kf_inliers = 0

    for t in range(n_frames):
        img = frames[t] if frames is not None else stream.frame(t)
        kps, xy_un, desc, bounds = backend.new_frame(img)
        n = len(kps)
        kp_mp = np.full(n, -1, np.int64)
        if t == 0:
            T = np.eye(4)
            first = add_points(T, xy_un, kps, desc, np.ones(n, bool))
            kp_mp[:] = first + np.arange(n)
            kf_inliers = n
            outlier = np.zeros(n, bool)
            kfs.append(dict(T=T.copy(), mp=kp_mp.copy(), uv=xy_un.copy(), w=inv_sigma2[kps["octave"]]))
            log["matches_last"].append(0); log["matches_map"].append(0); log["inliers"].append(n)
        else:
            # ---- TrackWithMotionModel: project the last frame's map points with the predicted pose -------------
            T_pred = (velocity @ T_last)
            Tp = T_pred[:3, :4].astype(np.float32)
            lk, lmp, lout = last
            last_slot = np.where((lmp >= 0) & ~lout, lmp, -1).astype(np.int32)
            nm, k2l = backend.search_last(Tp.reshape(12), last_slot, TH_LAST_FRAME)
            if nm < MIN_MATCHES_MOTION:
                nm, k2l = backend.search_last(Tp.reshape(12), last_slot, 2 * TH_LAST_FRAME)
            bound = k2l >= 0
            kp_mp[bound] = lmp[k2l[bound]]
            log["matches_last"].append(int(nm))
            idx = np.nonzero(kp_mp >= 0)[0]
            _, T12, outl = backend.pose(Tp.reshape(12), intr, mp_X[kp_mp[idx]], xy_un[idx],
                                        inv_sigma2[kps["octave"][idx]])
            kp_mp[idx[outl.astype(bool)]] = -1              # Tracking.cc:1030-1046 (outliers dropped)
            T_a = np.asarray(T12, np.float32).reshape(3, 4)

            # ---- TrackLocalMap: SearchLocalPoints + PoseOptimization --------------------------------------------
            n_map = len(mp_X)
            first = 0
            if local_keyframes > 0 and len(kf_first_slot) > local_keyframes:
                first = kf_first_slot[-local_keyframes]
            n_local = n_map - first
            skip = np.zeros(n_local, np.uint8)
            cur_mp = kp_mp[kp_mp >= 0]
            skip[cur_mp[cur_mp >= first] - first] = 1       # already matched: mbTrackInView = false (:1117-1124)
            excluded = (kp_mp >= 0).astype(np.uint8)
            nm2, k2m, _ = backend.search_local(T_a.reshape(12), first, n_local, skip, excluded, 1.0, log_sf)
            newly = k2m >= 0
            kp_mp[newly] = first + k2m[newly]
            log["matches_map"].append(int(nm2))
            idx = np.nonzero(kp_mp >= 0)[0]
            n_in, T12, outl = backend.pose(T_a.reshape(12), intr, mp_X[kp_mp[idx]], xy_un[idx],
                                           inv_sigma2[kps["octave"][idx]])
            if third_pose:
                backend.pose(T_last[:3, :4].astype(np.float32).reshape(12), intr, mp_X[kp_mp[idx]], xy_un[idx],
                             inv_sigma2[kps["octave"][idx]])
            outlier = np.zeros(n, bool)
            outlier[idx[outl.astype(bool)]] = True
            T = _T44(np.asarray(T12, np.float32))
            log["inliers"].append(int(n_in))

            # ---- "keyframe": create map points for the unmatched keypoints --------------------------------------
            if n_in < keyframe_ratio * kf_inliers or t % keyframe_every == 0:
                fresh = kp_mp < 0
                if fresh.any():
                    first = add_points(T, xy_un, kps, desc, fresh)
                    kp_mp[fresh] = first + np.arange(int(fresh.sum()))
                kf_inliers = max(n_in, 1)
                if local_ba:
                    keep = (kp_mp >= 0) & ~outlier
                    kfs.append(dict(T=T.copy(), mp=kp_mp[keep].copy(), uv=xy_un[keep].copy(),
                                    w=inv_sigma2[kps["octave"][keep]]))
                    win = _local_window(kfs, mp_X, intr)
                    if win is not None:
                        prob, pidx, pts, ref = win
                        T_out, X_out, e_out = backend.local_ba(prob)
                        for p, i in enumerate(pidx):          # SetPose of the free keyframes (Optimizer.cc:713-727)
                            if not prob["fixed"][p]:
                                kfs[i]["T"] = _T44(np.asarray(T_out[p], np.float32))
                        mp_X[pts] = np.asarray(X_out, np.float32)            # SetWorldPos
                        backend.map_set_positions(pts, mp_X[pts])
                        bad = np.asarray(e_out).astype(bool)                 # EraseMapPointMatch / EraseObservation
                        for i in np.unique(ref[bad, 0]):
                            drop = ref[bad & (ref[:, 0] == i), 1]
                            m = np.ones(len(kfs[i]["mp"]), bool); m[drop] = False
                            for key in ("mp", "uv", "w"):
                                kfs[i][key] = kfs[i][key][m]
                        T = kfs[-1]["T"].copy()               # the current frame is the newest keyframe
                        log["lba_edges"].append(len(bad)); log["lba_outliers"].append(int(bad.sum()))
            velocity = T @ np.linalg.inv(T_last)

        poses.append(T[:3, :4].reshape(12).copy())
        centres.append(-T[:3, :3].T @ T[:3, 3])
        log["n_map_points"].append(len(mp_X))
        last = (kps, kp_mp, outlier)
        T_last = T
        if on_frame is not None and on_frame(t) is False:
            break
    out = dict(centres=np.array(centres), poses=np.array(poses))
    out.update({k: np.array(v) for k, v in log.items()})
    return out


def umeyama(src, dst, with_scale=False):
    """Least-squares similarity (or rigid) transform dst ~ s R src + t (Umeyama 1991)."""
    src = np.asarray(src, np.float64); dst = np.asarray(dst, np.float64)
    ms, md = src.mean(0), dst.mean(0)
    a, b = src - ms, dst - md
    U, S, Vt = np.linalg.svd(b.T @ a / len(src))
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    var = (a ** 2).sum() / len(src)
    s = float((S * np.diag(D)).sum() / var) if with_scale and var > 0 else 1.0
    return s, R, md - s * R @ ms


def ate_rmse(est, ref, with_scale=False, align=True):
    """Absolute trajectory error: RMSE of camera-centre distances after the optimal alignment."""
    est = np.asarray(est, np.float64); ref = np.asarray(ref, np.float64)
    if align:
        s, R, t = umeyama(est, ref, with_scale)
        est = s * est @ R.T + t
    return float(np.sqrt(((est - ref) ** 2).sum(1).mean()))
