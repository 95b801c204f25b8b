import sys, time; sys.path.insert(0,'.')
import numpy as np, swarmmap_amd
from swarmmap_amd import synth
from oracle import oracle_py
o = swarmmap_amd.Optimizer()
for n in (500, 2000, 3000, 3500):
    c = synth.make_pose_case(7, n)
    for _ in range(3): r = o.PoseOptimization(c["Tcw"], c["intr"], c["Xw"], c["obs"], c["inv_sigma2"])
    t0=time.perf_counter()
    for _ in range(20): r = o.PoseOptimization(c["Tcw"], c["intr"], c["Xw"], c["obs"], c["inv_sigma2"])
    dt=(time.perf_counter()-t0)/20
    ro = oracle_py.pose_optimization(c["Tcw"], c["intr"], c["Xw"], c["obs"], c["inv_sigma2"])
    print(n, "ms/call %.3f" % (dt*1e3), r[0], ro[0], r[3], ro[3], float(np.abs(r[1]-ro[1]).max()), int((r[2]!=ro[2]).sum()))
