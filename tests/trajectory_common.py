"""CPU-oracle binding of the minitrack backend interface (test infrastructure)."""
import numpy as np

from oracle import oracle_py as orc
from swarmmap_amd import minitrack


class OracleBackend:
    name = "oracle"

    def __init__(self, K, nfeatures=1000):
        self.cfg = orc.config(nfeatures, 1.2, 8, 20, 7)
        self.tab = orc.make_tables(self.cfg)
        self.cam = orc.camera(K)
        self.bounds = None

    def tables(self):
        n = self.cfg.nlevels
        return (np.array(self.tab.scale[:n], np.float32), np.array(self.tab.inv_sigma2[:n], np.float32))

    def extract(self, img):
        return orc.extract(self.cfg, img)

    def prepare(self, xy, w, h):
        if self.bounds is None:
            self.bounds = orc.image_bounds(self.cam, w, h)
        return orc.undistort_keypoints(self.cam, xy), self.bounds.copy()

    def search_last(self, F, last, th):
        return orc.search_by_projection_lastframe(F, last, th, True)

    def frustum(self, bounds, Tcw, Xw, normal, max_d, min_d, log_sf, nlevels):
        return orc.is_in_frustum(self.cam, bounds, Tcw, Xw, normal, max_d, min_d, minitrack.COS_LIMIT, log_sf, nlevels)

    def search_map(self, F, mps, th):
        return orc.search_by_projection_mappoints(F, mps, th, 0.8)

    def pose(self, Tcw, intr, Xw, obs, w):
        n, T, outl, _ = orc.pose_optimization(Tcw, intr, Xw, obs, w)
        return n, T, outl

    def local_ba(self, window):
        r = orc.bundle_adjust(window)
        return r["Tcw"], r["Xw"], r["outlier"]

    def close(self):
        pass
