"""PoseOptimization kernel time against the number of matched points (developer tool, GPU box)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.optimizer import Optimizer
o = Optimizer()
for n in ([int(a) for a in sys.argv[1:]] or (100, 300, 500, 640, 800, 1000, 1500, 2000, 3000, 4000)):
    c = synth.make_pose_case(7, n=n)
    ks = []
    for _ in range(12):
        o.PoseOptimization(c["Tcw"], c["intr"], c["Xw"], c["obs"], c["inv_sigma2"])
        ks.append(o.pose_kernel_ms())
    r = o.PoseOptimization(c["Tcw"], c["intr"], c["Xw"], c["obs"], c["inv_sigma2"])
    print(n, "kernel us %.1f" % (np.median(ks[2:]) * 1e3), "inliers", r[0], r[3])
