"""A minimal monocular tracking replay over the hot path: extract -> frame post-processing -> SearchByProjection
(last frame) -> PoseOptimization -> isInFrustum -> SearchByProjection (local map) -> PoseOptimization, and, at
every "keyframe", LocalBundleAdjustment over the recent keyframes.

It is NOT the reference's Tracking state machine (out of scope): it is the shortest host loop that
chains every per-frame operator the way Tracking::TrackWithMotionModel (code/src/Tracking.cc:714-768) and
Tracking::TrackLocalMap (:770-807) do, so that a *trajectory* can be produced from images alone and compared
between two implementations of the operators ("ATE").  The scene is the synthetic stream of
`synth.FrameStream`: a textured plane at depth `plane_z` seen by a camera translating parallel to it, so map points
are created by back-projecting keypoints onto the known plane (as an RGB-D front-end would) and ground truth is
known in closed form.

The operators come from a backend object; `HipBackend` binds the product (C ABI).  tests/ bind the CPU oracle to
the same interface and compare trajectories.
"""
import numpy as np


TH_LAST_FRAME = 15.0      # Tracking.cc:731 (monocular)
MIN_MATCHES_MOTION = 20   # Tracking.cc:734
COS_LIMIT = 0.5           # Tracking.cc:991 isInFrustum(pMP, 0.5)


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

    def search_local(self, Tcw, first, n_local, skip, excluded, th, log_sf, local_slot=None):
        from . import dframe as dfm
        return dfm.search_local_map(self.m_map, self.frames[self.cur], self.map, Tcw, n_local, th, COS_LIMIT, log_sf,
                                    local_slot=local_slot, skip=skip, excluded=excluded, first_slot=first)

    def map_append(self, X, normal, max_d, min_d, desc):
        self.map.append(X, normal, max_d, min_d, desc)

    def map_set_positions(self, slots, X):
        self.map.write_positions(slots, X)

    def map_write_rows(self, slots, X=None, normal=None, max_d=None, min_d=None):
        self.map.write_rows(slots, X, normal, max_d, min_d)

    def pose(self, Tcw, intr, Xw, obs, w):
        n, T, outl, _ = self.opt.PoseOptimization(Tcw, intr, Xw, obs, w)
        return n, T, outl

    def local_ba(self, window):
        r = self.lba.LocalBundleAdjustment(window)
        return r["Tcw"], r["Xw"], r["outlier"]

    # ---- the local-mapping thread's matcher operators (local_mapping_matcher_job below) ----
    def _lm(self):
        if getattr(self, "m_lm", None) is None:
            from .matcher import ORBmatcher
            self.m_lm = ORBmatcher(0.6, True)  # LocalMapping.cc:200 / :451: ORBmatcher matcher(0.6, false) / matcher
        return self.m_lm

    def assign_nodes(self, desc, vocab):
        return self._lm().hamming_top2(desc, vocab)[0]

    def search_for_triangulation(self, kf1, fv1, kf2, fv2, F12, epipole, sf, level_sigma2, check_ori=True):
        if not check_ori:  # CreateNewMapPoints' matcher: ORBmatcher(0.6, false), LocalMapping.cc:197
            if getattr(self, "m_tri", None) is None:
                from .matcher import ORBmatcher
                self.m_tri = ORBmatcher(0.6, False)
            return self.m_tri.SearchForTriangulation(kf1, fv1, kf2, fv2, F12, epipole, sf, level_sigma2)
        return self._lm().SearchForTriangulation(kf1, fv1, kf2, fv2, F12, epipole, sf, level_sigma2)

    def fuse(self, KF, K, Tcw, log_sf, inv_sigma2, mp, th):
        return self._lm().Fuse(KF, K, Tcw, log_sf, inv_sigma2, mp, th)[0]

    def triangulate(self, kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2):
        return self._lm().TriangulateMatches(kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2)

    def fuse_idx(self, KF, K, Tcw, log_sf, inv_sigma2, mp, th):
        return self._lm().Fuse(KF, K, Tcw, log_sf, inv_sigma2, mp, th)[:3]

    def triangulate_new_points(self, kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2):
        return self._lm().TriangulateNewPoints(kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2)

    def update_normal_and_depth(self, *a):
        return self._lm().UpdateNormalAndDepth(*a)

    def close(self):
        for o in self.frames + [self.map, self.ex, self.m_last, self.m_map, self.opt, self.lba]:
            o.close()
        if getattr(self, "m_lm", None) is not None:
            self.m_lm.close()
        if getattr(self, "m_tri", None) is not None:
            self.m_tri.close()


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


def fundamental_and_epipole(K, T1, T2):
    """LocalMapping::ComputeF12 (code/src/LocalMapping.cc:593-609) and the epipole of SearchForTriangulation
    (code/src/ORBmatcher.cc:605-613) from two float poses [R | t] (12 floats each); both keyframes share K."""
    fx, fy, cx, cy = [float(v) for v in K]
    T1 = np.asarray(T1, np.float32).astype(np.float64).reshape(3, 4)
    T2 = np.asarray(T2, np.float32).astype(np.float64).reshape(3, 4)
    R1, t1, R2, t2 = T1[:, :3], T1[:, 3], T2[:, :3], T2[:, 3]
    R12 = R1 @ R2.T
    t12 = -R12 @ t2 + t1
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]])
    F12 = (Ki.T @ tx @ R12 @ Ki).astype(np.float32)
    C2 = R2 @ (-R1.T @ t1) + t2
    return F12, (np.float32(fx * C2[0] / C2[2] + cx), np.float32(fy * C2[1] / C2[2] + cy))


def keyframe_view(kf, sf):
    """FrameView of a keyframe snapshot with a KeyFrame's bounds: the Frame's truncated to int for IsInImage /
    GetFeaturesInArea, the Frame's float ones for the grid it copied (include/swarmorb.h, so_frame_view)."""
    from .matcher import FrameView
    b = [float(v) for v in kf["bounds"]]
    return FrameView(kf["x"], kf["y"], kf["octave"], kf["angle"], kf["desc"], [float(int(v)) for v in b], sf, grid_bounds=b)


def local_mapping_matcher_job(backend, ring, c, K, sf, inv_sigma2, log_sf, vocab):
    """What the local-mapping thread does with a new keyframe before local BA, operators from `backend`:
    CreateNewMapPoints' SearchForTriangulation against every neighbour (code/src/LocalMapping.cc:197-246) and
    SearchInNeighbors' Fuse into every neighbour and back (:451-481).  The same job as swarmmap_amd/host/replay.cc's
    lm_matcher_job.  c / ring entries: dicts x, y, angle, octave, desc, mp (slot or -1), mpX, mpN, mpMax, mpMin, mpDesc,
    T (12), bounds.  Returns (triangulation matches, fused into neighbours, fused back, new map points)."""
    from .matcher import FeatureVector
    c["fv"] = FeatureVector(np.asarray(backend.assign_nodes(c["desc"], vocab), np.int32))
    kfeat = lambda k: dict(x=k["x"], y=k["y"], angle=k["angle"], octave=k["octave"], desc=k["desc"],  # noqa: E731
                           free=(k["mp"] < 0).astype(np.uint8))
    n_tri = n_fused = n_back = 0
    level_sigma2 = (np.asarray(sf, np.float32) * np.asarray(sf, np.float32)).astype(np.float32)
    tri = []  # (neighbour index, matches12)
    for j, k2 in enumerate(ring):
        F12, epi = fundamental_and_epipole(K, c["T"], k2["T"])
        nm, m12 = backend.search_for_triangulation(kfeat(c), c["fv"], kfeat(k2), k2["fv"], F12, epi, sf, level_sigma2)
        n_tri += nm
        tri.append((j, m12))
    mp_c = dict(Xw=c["mpX"], normal=c["mpN"], max_dist=c["mpMax"], min_dist=c["mpMin"], desc=c["mpDesc"])
    for k2 in ring:
        in_kf = np.isin(c["mp"], k2["mp"][k2["mp"] >= 0])
        mp_c["valid"] = ((c["mp"] >= 0) & ~in_kf).astype(np.uint8)
        n_fused += backend.fuse(keyframe_view(k2, sf), K, k2["T"], log_sf, inv_sigma2, mp_c, 3.0)
    if ring:
        seen, rows = set(), []
        for k2 in ring:
            for i in np.nonzero(k2["mp"] >= 0)[0]:
                s = int(k2["mp"][i])
                if s not in seen:
                    seen.add(s)
                    rows.append((k2, i, s))
        own = set(int(s) for s in c["mp"][c["mp"] >= 0])
        cand = dict(Xw=np.array([k["mpX"][i] for k, i, _ in rows], np.float32).reshape(-1, 3),
                    normal=np.array([k["mpN"][i] for k, i, _ in rows], np.float32).reshape(-1, 3),
                    max_dist=np.array([k["mpMax"][i] for k, i, _ in rows], np.float32),
                    min_dist=np.array([k["mpMin"][i] for k, i, _ in rows], np.float32),
                    desc=np.array([k["mpDesc"][i] for k, i, _ in rows], np.uint8).reshape(-1, 32),
                    valid=np.array([0 if s in own else 1 for _, _, s in rows], np.uint8))
        n_back = backend.fuse(keyframe_view(c, sf), K, c["T"], log_sf, inv_sigma2, cand, 3.0)
    # CreateNewMapPoints' per-match body (LocalMapping.cc:263-420) for the matches of all neighbours, then
    # MapPoint::UpdateNormalAndDepth of the new points (two observations each, the new keyframe is the reference)
    n_new = 0
    if tri:
        tk = lambda k: dict(Tcw=k["T"], K=K, scale_factors=sf, level_sigma2=level_sigma2)  # noqa: E731
        of, i1, i2 = [], [], []
        for j, m12 in tri:
            sel = np.nonzero(m12 >= 0)[0]
            of.append(np.full(len(sel), j, np.int32)); i1.append(sel); i2.append(m12[sel])
        of, i1 = np.concatenate(of), np.concatenate(i1)
        if len(of):
            xy1 = np.stack([c["x"][i1], c["y"][i1]], 1)
            o1 = c["octave"][i1]
            xy2 = np.concatenate([np.stack([ring[j]["x"][b], ring[j]["y"][b]], 1) for (j, _), b in zip(tri, i2)])
            o2 = np.concatenate([ring[j]["octave"][b] for (j, _), b in zip(tri, i2)])
            ok, X = backend.triangulate(tk(c), [tk(k) for k in ring], float(np.float32(1.5) * np.float32(1.2)), of, xy1, o1, xy2, o2)
            ok = ok.astype(bool)
            n_new = int(ok.sum())
            if n_new:
                centre = lambda T: (-(np.asarray(T, np.float32).astype(np.float64).reshape(3, 4)[:, :3].T  # noqa: E731
                                      @ np.asarray(T, np.float32).astype(np.float64).reshape(3, 4)[:, 3])).astype(np.float32)
                Owc = centre(c["T"])
                Ow2 = np.stack([centre(k["T"]) for k in ring])
                obs = np.stack([np.tile(Owc, (n_new, 1)), Ow2[of[ok]]], 1).reshape(-1, 3)
                sfa = np.asarray(sf, np.float32)
                backend.update_normal_and_depth(np.arange(0, 2 * n_new + 1, 2, dtype=np.int32), obs, X[ok], np.tile(Owc, (n_new, 1)),
                                                sfa[o1[ok]], np.full(n_new, sfa[-1], np.float32), np.zeros((n_new, 3), np.float32),
                                                np.zeros(n_new, np.float32), np.zeros(n_new, np.float32))
    return int(n_tri), int(n_fused), int(n_back), n_new


def track(backend, stream, n_frames, K, plane_z=2.0, keyframe_every=8, keyframe_ratio=0.7, local_ba=False,
          local_keyframes=0, third_pose=False, frames=None, on_frame=None, lm_every=0, vocab=None, lm_neighbours=20,
          on_keyframe=None):
    """Returns dict(centres (n,3), poses (n,12), matches_last, matches_map, inliers, n_map_points[, lba_*]).
    local_keyframes > 0: the local map (Tracking::UpdateLocalMap) is the points created at the last that many
    keyframes instead of the whole map.  third_pose: one more PoseOptimization per frame, from the last frame's pose
    over the final matches (what Tracking::TrackReferenceKeyFrame does when the motion model fails; the per-frame
    replay of SURVEY.md 8d counts three calls), result not used.  frames: optional list of images (else
    stream.frame(t)).  on_frame(t): called after each tracked frame; returning False ends the run.
    lm_every > 0 with a vocabulary (replay.make_vocabulary): every lm_every-th frame is handed to the local-mapping
    matcher job (local_mapping_matcher_job) as a new keyframe against the last lm_neighbours ones, as the bench's
    local-mapping thread does; its counts land in out["lm_log"]; results are not fed back.  on_keyframe(job): instead
    of running the job inline, hand the closure over (the CPU baseline runs it on its local-mapping thread)."""
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
    mp_F = dict(N=np.zeros((0, 3), np.float32), mx=np.zeros(0, np.float32), mn=np.zeros(0, np.float32),
                D=np.zeros((0, 32), np.uint8))  # normal, mfMaxDistance, mfMinDistance, descriptor (matcher job only)
    kf_first_slot = []    # first map slot created at each keyframe
    lm_ring, lm_log = [], []

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
        fields = ((PO / dist[:, None]).astype(np.float32),
                  (1.2 * mx).astype(np.float32),                    # GetMaxDistanceInvariance
                  (0.8 * mx / sf[nlevels - 1]).astype(np.float32),  # GetMinDistanceInvariance
                  desc[sel])
        backend.map_append(X.astype(np.float32), *fields)
        if lm_every > 0:
            for key, a in zip(("N", "mx", "mn", "D"), fields):
                mp_F[key] = np.concatenate([mp_F[key], a])
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
        map_size_at_begin = len(mp_X)
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
            kp_mp[idx[outl.astype(bool)]] = -1              # Tracking.cc:745-760 (outliers dropped)
            T_a = np.asarray(T12, np.float32).reshape(3, 4)

            # ---- TrackLocalMap: SearchLocalPoints + PoseOptimization --------------------------------------------
            n_map = len(mp_X)
            first = 0
            if local_keyframes > 0 and len(kf_first_slot) > local_keyframes:
                first = kf_first_slot[-local_keyframes]
            n_local = n_map - first
            skip = np.zeros(n_local, np.uint8)
            cur_mp = kp_mp[kp_mp >= 0]
            skip[cur_mp[cur_mp >= first] - first] = 1       # already matched: mbTrackInView = false (:966-978)
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

        if lm_every > 0 and vocab is not None and t % lm_every == 0:
            # the frame as the local-mapping thread's new keyframe: bindings to map points that existed before it
            pre = np.where((kp_mp >= 0) & (kp_mp < map_size_at_begin) & ~outlier, kp_mp, -1).astype(np.int64)
            s0 = np.maximum(pre, 0)
            has = (pre >= 0)
            snap = dict(x=xy_un[:, 0].copy(), y=xy_un[:, 1].copy(), angle=kps["angle"].copy(),
                        octave=kps["octave"].astype(np.int32), desc=desc.copy(), mp=pre,
                        mpX=np.where(has[:, None], mp_X[s0] if len(mp_X) else 0, 0).astype(np.float32),
                        mpN=np.where(has[:, None], mp_F["N"][s0] if len(mp_X) else 0, 0).astype(np.float32),
                        mpMax=np.where(has, mp_F["mx"][s0] if len(mp_X) else 0, 0).astype(np.float32),
                        mpMin=np.where(has, mp_F["mn"][s0] if len(mp_X) else 0, 0).astype(np.float32),
                        mpDesc=np.where(has[:, None], mp_F["D"][s0] if len(mp_X) else 0, 0).astype(np.uint8),
                        T=T[:3, :4].astype(np.float32).reshape(12), bounds=np.asarray(bounds, np.float32), t=t)

            def job(snap=snap):
                res = local_mapping_matcher_job(backend, list(lm_ring), snap, K, sf, inv_sigma2, log_sf, vocab)
                lm_log.append((snap["t"], len(lm_ring)) + res)
                lm_ring.append(snap)
                del lm_ring[:max(0, len(lm_ring) - lm_neighbours)]
            if on_keyframe is not None:
                on_keyframe(job)
            else:
                job()
        poses.append(T[:3, :4].reshape(12).copy())
        centres.append(-T[:3, :3].T @ T[:3, 3])
        log["n_map_points"].append(len(mp_X))
        last = (kps, kp_mp, outlier)
        T_last = T
        if on_frame is not None and on_frame(t) is False:
            break
    out = dict(centres=np.array(centres), poses=np.array(poses))
    out.update({k: np.array(v) for k, v in log.items()})
    out["lm_log"] = np.array(lm_log, np.int32).reshape(-1, 6)
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
