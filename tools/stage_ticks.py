"""Phase times of track_resolve_kernel inside the closed loop (SWARMORB_STAGE_DEBUG=1 makes so_track_stage_wait print the
kernel's own 100 MHz tick counters): loads / rounds / rotation check / by-keypoint tables / outputs, per stage call.
    SWARMORB_STAGE_DEBUG=1 python tools/stage_ticks.py [frames=40] 2>&1 | grep "\[stage" | tail"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swarmmap_amd import synth  # noqa: E402
from swarmmap_amd.replay import Replay, make_vocabulary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
K, dist = synth.EUROC_K, synth.EUROC_DIST
st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=K, dist=dist)
block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
view = block.numpy()
for t in range(n + 2):
    view[t] = st.frame(t)
rp = Replay(0, st.w, st.h, 1000, 5, K, dist, plane_z=2.0, local_keyframes=12, third_pose=True)
rp.set_frames([block.data_ptr() + i * st.w * st.h for i in range(n + 2)], on_device=False)
rp.set_vocabulary(make_vocabulary())
rp.set_closed_loop()
rp.prime(0)
rp.run(0, n, True)
rp.drain()
rp.finish()
rp.close()
