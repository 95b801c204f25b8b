import sys, os
sys.path.insert(0, '/root/repo')
import swarmmap_amd
from swarmmap_amd import synth
o = swarmmap_amd.Optimizer()
p = synth.make_ba_case("LBA-M", seed=100)
for i in range(4):
    r = o.LocalBundleAdjustment(p)
print(r["info"]["wall_ms"], r["info"]["gpu_ms"])
