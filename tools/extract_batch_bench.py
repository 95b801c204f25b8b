"""Front-end throughput with A agents' frames per chain of launches (so_extractor_group + so_dframe_group_submit):
wall time from the group submit to the last member's frame complete on the device, nothing else on the GPU.
    python tools/extract_batch_bench.py [euroc|kitti] [A ...]      (default A = 1 2 4 8 16 32)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd as S  # noqa: E402
from swarmmap_amd import synth  # noqa: E402


def run(A, kitti, reps=60):
    size, K, dist, nf = (synth.KITTI, synth.KITTI_K, None, 2000) if kitti else (synth.EUROC, synth.EUROC_K, synth.EUROC_DIST, 1000)
    st = synth.FrameStream(seed=20221001, size=size, K=K, dist=dist)
    nimg = 12
    block = torch.empty((nimg, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(nimg):
        view[t] = st.frame(t)
    exs = [S.ORBextractor(nf, 1.2, 8, 20, 7) for _ in range(A)]
    frs = [S.DeviceFrame(ex, K, dist if dist is not None else (0.0, 0.0, 0.0, 0.0, 0.0)) for ex in exs]
    grp = S.ExtractorGroup(exs)
    lib = frs[0]._lib
    lib.so_dframe_wait.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    ts, nk_total = [], 0
    for r in range(reps + 10):
        imgs = [view[(r + a) % nimg] for a in range(A)]
        t0 = time.perf_counter()
        grp.submit(imgs, frames=frs)
        nk = C.c_int(0)
        for f in frs:
            lib.so_dframe_wait(f._h, C.byref(nk), None)
        t1 = time.perf_counter()
        for f in frs:
            f.collect()
            nk_total += f.n
        if r >= 10:
            ts.append(t1 - t0)
    grp.close()
    for f in frs:
        f.close()
    for ex in exs:
        ex.close()
    ms = float(np.median(ts)) * 1e3
    return {"stream": "kitti" if kitti else "euroc", "agents_per_chain": A, "ms_per_chain": ms, "frames_per_s": A / (ms * 1e-3),
            "keypoints_per_frame": nk_total / ((reps + 10) * A)}


if __name__ == "__main__":
    kitti = "kitti" in sys.argv[1:]
    As = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8, 16, 32]
    for A in As:
        print(json.dumps(run(A, kitti)), flush=True)
