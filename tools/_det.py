import sys; sys.path.insert(0,'.')
import numpy as np, swarmmap_amd
from swarmmap_amd import synth
o = swarmmap_amd.Optimizer()
for name, seed in (("LBA-S",1),("LBA-M",3)):
    p = synth.make_ba_case(name, seed)
    ref = None; bad = 0; infos=set()
    for i in range(40):
        r = o.LocalBundleAdjustment(p)
        key = (r["Tcw"].tobytes(), r["Xw"].tobytes(), r["info"]["iterations_stage2"], r["info"]["lm_trials"])
        infos.add((r["info"]["iterations_stage1"], r["info"]["iterations_stage2"], r["info"]["lm_trials"], round(r["info"]["chi2_final"],3)))
        if ref is None: ref = key
        elif key != ref: bad += 1
    print(name, "runs differing from the first:", bad, sorted(infos))
