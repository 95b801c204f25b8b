import sys, time, os
sys.path.insert(0,'.')
import numpy as np, swarmmap_amd
g=np.load('tests/golden/closed_loop_window.npz'); p={k:g[k] for k in g.files}
o=swarmmap_amd.Optimizer()
for _ in range(3): r=o.LocalBundleAdjustment(p)
ts=[]
for _ in range(7):
    t0=time.perf_counter(); r=o.LocalBundleAdjustment(p); ts.append(time.perf_counter()-t0)
print('wall ms', np.median(ts)*1e3, {k:r['info'][k] for k in ('gpu_ms','lm_trials','iterations_stage1','iterations_stage2','n_outliers','solver_path','n_free_keyframes','chi2_initial','chi2_final')})
o.set_solve_timing(True); r=o.LocalBundleAdjustment(p); print('solve ms per', r['info']['solve_ms']/max(r['info']['n_solves'],1), r['info']['n_solves'])
