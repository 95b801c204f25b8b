"""ATE of the HIP path against the CPU oracle path and against ground truth on the synthetic streams
(SURVEY.md 8d).  Prints one JSON line per stream; run on the GPU box:  python tools/trajectory_ate.py [frames]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import numpy as np  # noqa: E402

from swarmmap_amd import minitrack, synth  # noqa: E402
from trajectory_common import OracleBackend  # noqa: E402

PLANE_Z = 2.0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for name, size, K, nfeat, lba in (("euroc_752x480", synth.EUROC, synth.EUROC_K, 1000, False),
                                      ("kitti_1241x376", synth.KITTI, synth.KITTI_K, 2000, False),
                                      ("euroc_752x480 + LocalBundleAdjustment", synth.EUROC, synth.EUROC_K, 1000, True)):
        st = synth.FrameStream(size=size)
        hip = minitrack.HipBackend(K, nfeat)
        t0 = time.perf_counter()
        a = minitrack.track(hip, st, n, K, plane_z=PLANE_Z, local_ba=lba)
        t1 = time.perf_counter()
        hip.close()
        b = minitrack.track(OracleBackend(K, nfeat), st, n, K, plane_z=PLANE_Z, local_ba=lba)
        t2 = time.perf_counter()
        gt = minitrack.ground_truth(st, n, K, PLANE_Z)
        path = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())
        print(json.dumps(dict(
            stream=name, frames=n, plane_z_m=PLANE_Z, pixel_m=PLANE_Z / float(K[0]), path_length_m=path,
            ate_hip_vs_oracle_m=minitrack.ate_rmse(a["centres"], b["centres"], align=False),
            ate_hip_vs_gt_m=minitrack.ate_rmse(a["centres"], gt, align=False),
            ate_oracle_vs_gt_m=minitrack.ate_rmse(b["centres"], gt, align=False),
            ate_hip_vs_gt_sim3_m=minitrack.ate_rmse(a["centres"], gt, with_scale=True),
            ate_oracle_vs_gt_sim3_m=minitrack.ate_rmse(b["centres"], gt, with_scale=True),
            max_pose_entry_diff=float(np.abs(a["poses"] - b["poses"]).max()),
            frames_with_different_match_counts=int(((a["matches_last"] != b["matches_last"]) |
                                                    (a["matches_map"] != b["matches_map"])).sum()),
            frames_with_different_inlier_counts=int((a["inliers"] != b["inliers"]).sum()),
            map_points=[int(a["n_map_points"][-1]), int(b["n_map_points"][-1])],
            mean_inliers=float(a["inliers"][1:].mean()), lba_windows=int(len(a["lba_edges"])),
            lba_edges_mean=float(a["lba_edges"].mean()) if len(a["lba_edges"]) else 0.0,
            python_loop_s=dict(hip=t1 - t0, oracle=t2 - t1))))


if __name__ == "__main__":
    main()
