import sys; sys.path.insert(0,'.')
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.optimizer import Optimizer
o=Optimizer()
o.set_solve_timing(True)
for nf in (48,64,72,80,88,96,104,112,128):
    w=synth.make_ba_problem(0, nf, (3*nf)//2, 150*nf, max_obs="auto")
    for _ in range(2): r=o.LocalBundleAdjustment(w)
    i=r["info"]
    print(nf, "hessian kf", i["n_free_keyframes"], "tiles", i["nnz_tiles"], "path", i["solver_path"], "solve_ms/solve %.4f"%(i["solve_ms"]/max(i["n_solves"],1)), "n_solves", i["n_solves"], "gpu_ms %.3f wall %.3f"%(i["gpu_ms"], i["wall_ms"]))
