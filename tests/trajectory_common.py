"""CPU-oracle binding of the minitrack backend interface (test infrastructure)."""
import numpy as np

from oracle import oracle_py as orc
from swarmmap_amd import minitrack
from swarmmap_amd.matcher import FrameView


class OracleBackend:
    """Same interface as minitrack.HipBackend, every operator from the CPU oracle, all state in host arrays."""

    name = "oracle"

    def __init__(self, K, nfeatures=1000, dist=(0, 0, 0, 0, 0)):
        self.cfg = orc.config(nfeatures, 1.2, 8, 20, 7)
        self.tab = orc.make_tables(self.cfg)
        self.cam = orc.camera(K, dist)
        self.bounds = None
        self.sf = np.array(self.tab.scale[:self.cfg.nlevels], np.float32)
        self.cur = self.last = None  # (kps, xy_un, desc)
        self.X = np.zeros((0, 3), np.float32)
        self.normal = np.zeros((0, 3), np.float32)
        self.max_d = np.zeros(0, np.float32)
        self.min_d = np.zeros(0, np.float32)
        self.desc = np.zeros((0, 32), np.uint8)

    def tables(self):
        n = self.cfg.nlevels
        return (self.sf.copy(), np.array(self.tab.inv_sigma2[:n], np.float32))

    def new_frame(self, img):
        kps, desc = orc.extract(self.cfg, img)
        if self.bounds is None:
            self.bounds = orc.image_bounds(self.cam, img.shape[1], img.shape[0])
        xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        xy_un = orc.undistort_keypoints(self.cam, xy)
        self.last, self.cur = self.cur, (kps, xy_un, desc)
        return kps, xy_un, desc, self.bounds.copy()

    def _view(self, excluded=None):
        kps, xy_un, desc = self.cur
        return FrameView(xy_un[:, 0], xy_un[:, 1], kps["octave"], kps["angle"], desc, self.bounds, self.sf, excluded)

    def search_last(self, Tcw, last_slot, th):
        lk = self.last[0]
        slot = np.asarray(last_slot)
        has = slot >= 0
        s0 = np.maximum(slot, 0)
        valid, u, v = orc.project_last_frame(self.cam, self.bounds, Tcw, self.X[s0], has)
        lastd = dict(valid=valid, u=u, v=v, octave=lk["octave"], angle=lk["angle"], desc=self.desc[s0],
                     has_obs=np.ones(len(lk), np.uint8))
        return orc.search_by_projection_lastframe(self._view(), lastd, th, True)

    def search_local(self, Tcw, first, n_local, skip, excluded, th, log_sf, local_slot=None):
        sl = slice(first, first + n_local) if local_slot is None else np.asarray(local_slot, np.int64)
        fr = orc.is_in_frustum(self.cam, self.bounds, Tcw, self.X[sl], self.normal[sl], self.max_d[sl], self.min_d[sl],
                               minitrack.COS_LIMIT, log_sf, self.cfg.nlevels)
        in_view = fr["in_view"] & (1 - np.asarray(skip, np.uint8))
        mps = dict(in_view=in_view, proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"],
                   pred_level=fr["pred_level"], desc=self.desc[sl], has_obs=np.ones(n_local, np.uint8))
        nm, k2m = orc.search_by_projection_mappoints(self._view(excluded), mps, th, 0.8)
        return nm, k2m, in_view

    def map_append(self, X, normal, max_d, min_d, desc):
        self.X = np.concatenate([self.X, X]); self.normal = np.concatenate([self.normal, normal])
        self.max_d = np.concatenate([self.max_d, max_d]); self.min_d = np.concatenate([self.min_d, min_d])
        self.desc = np.concatenate([self.desc, desc])

    def map_set_positions(self, slots, X):
        self.X[np.asarray(slots)] = X

    def map_write_rows(self, slots, X=None, normal=None, max_d=None, min_d=None):
        s = np.asarray(slots, np.int64)
        for dst, src in ((self.X, X), (self.normal, normal), (self.max_d, max_d), (self.min_d, min_d)):
            if src is not None:
                dst[s] = src

    def pose(self, Tcw, intr, Xw, obs, w):
        n, T, outl, _ = orc.pose_optimization(Tcw, intr, Xw, obs, w)
        return n, T, outl

    def local_ba(self, window):
        r = orc.bundle_adjust(window)
        return r["Tcw"], r["Xw"], r["outlier"]

    # ---- the local-mapping thread's matcher operators (minitrack.local_mapping_matcher_job) ----
    def assign_nodes(self, desc, vocab):
        return orc.hamming_top2(desc, vocab)[0]

    def search_for_triangulation(self, kf1, fv1, kf2, fv2, F12, epipole, sf, level_sigma2, check_ori=True):
        return orc.search_for_triangulation(kf1, fv1, kf2, fv2, F12, epipole, sf, level_sigma2, bool(check_ori))

    def fuse(self, KF, K, Tcw, log_sf, inv_sigma2, mp, th):
        return orc.fuse(KF, orc.camera(K), Tcw, log_sf, inv_sigma2, mp, th)[0]

    def fuse_idx(self, KF, K, Tcw, log_sf, inv_sigma2, mp, th):
        return orc.fuse(KF, orc.camera(K), Tcw, log_sf, inv_sigma2, mp, th)

    def triangulate_new_points(self, kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2):
        """CreateNewMapPoints' per-match body + the new points' UpdateNormalAndDepth (two observations: kf1, then the
        neighbour; kf1 the reference keyframe), as so_triangulate_new_points."""
        ok, X = self.triangulate(kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2)
        n = len(ok)
        nrm, mx, mn = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        sel = np.nonzero(ok)[0]
        if len(sel):
            centre = lambda T: (-(np.asarray(T, np.float32).astype(np.float64).reshape(3, 4)[:, :3].T  # noqa: E731
                                  @ np.asarray(T, np.float32).astype(np.float64).reshape(3, 4)[:, 3])).astype(np.float32)
            O1 = centre(kf1["Tcw"])
            O2 = np.stack([centre(k["Tcw"]) for k in kf2_list])
            obs = np.stack([np.tile(O1, (len(sel), 1)), O2[np.asarray(kf2_of)[sel]]], 1).reshape(-1, 3)
            sfa = np.asarray(kf1["scale_factors"], np.float32)
            a, b, c = orc.update_normal_and_depth(np.arange(0, 2 * len(sel) + 1, 2, dtype=np.int32), obs, X[sel], np.tile(O1, (len(sel), 1)),
                                                  sfa[np.asarray(o1)[sel]], np.full(len(sel), sfa[-1], np.float32),
                                                  np.zeros((len(sel), 3), np.float32), np.zeros(len(sel), np.float32),
                                                  np.zeros(len(sel), np.float32))
            nrm[sel], mx[sel], mn[sel] = a, b, c
        return ok, X, nrm, mx, mn

    def triangulate(self, kf1, kf2_list, ratio_factor, kf2_of, xy1, o1, xy2, o2):
        ok, X = np.zeros(len(o1), np.uint8), np.zeros((len(o1), 3), np.float32)
        for j, kf2 in enumerate(kf2_list):  # the oracle takes one neighbour at a time
            sel = np.nonzero(np.asarray(kf2_of) == j)[0]
            if len(sel):
                a, b = orc.triangulate_matches(kf1, kf2, ratio_factor, xy1[sel], o1[sel], xy2[sel], o2[sel])
                ok[sel], X[sel] = a, b
        return ok, X

    def update_normal_and_depth(self, *a):
        return orc.update_normal_and_depth(*a)

    def close(self):
        pass
