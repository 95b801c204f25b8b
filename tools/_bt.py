import sys; sys.path.insert(0,'.')
import swarmmap_amd
from swarmmap_amd import synth
ba = swarmmap_amd.Optimizer()
w = synth.make_ba_case("LBA-M", seed=100)
for i in range(6):
    r = ba.LocalBundleAdjustment(w)["info"]
print(r)
