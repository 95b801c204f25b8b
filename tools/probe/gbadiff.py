import sys, numpy as np
sys.path.insert(0,'/root/repo')
import swarmmap_amd
from swarmmap_amd import synth
for name, fx in (("GBA-2","gba2_norobust.npz"),("GBA-2r","gba2r_norobust.npz")):
    g=np.load('/root/repo/tests/golden/'+fx)
    p=synth.make_ba_case(name,1)
    o=swarmmap_amd.Optimizer()
    r=o.BundleAdjustment(p,nIterations=10,bRobust=False)
    inf=dict(zip([str(k) for k in g["info_keys"]], g["info_vals"]))
    dT=np.abs(r["Tcw"]-g["Tcw"]); dX=np.abs(r["Xw"][::8]-g["Xw_every8"])
    print(name, "trials", r["info"]["lm_trials"], inf["lm_trials"], "chi2f", r["info"]["chi2_final"], inf["chi2_final"], "lambda", r["info"]["lambda_final"], inf["lambda_final"])
    print(" dT max", dT.max(), "99.9%", np.quantile(dT,0.999), "argmax pose", np.unravel_index(dT.argmax(), dT.shape), " dX max", dX.max(), np.quantile(dX,0.999))
    print(" |T| at argmax", g["Tcw"][np.unravel_index(dT.argmax(), dT.shape)[0]])
    o.close()
