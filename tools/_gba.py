import sys, time; sys.path.insert(0,'.')
import numpy as np, swarmmap_amd
from swarmmap_amd import synth
o = swarmmap_amd.Optimizer()
for name in ("GBA-1", "GBA-2"):
    p = synth.make_ba_case(name, 1)
    t0=time.perf_counter(); r = o.BundleAdjustment(p, nIterations=10, bRobust=True); dt=time.perf_counter()-t0
    print(name, "edges", len(p["edge_pose"]), "wall %.1f ms" % (dt*1e3), r["info"])
