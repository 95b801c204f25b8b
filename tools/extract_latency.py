"""Wall time of ONE frame through the device-resident Frame constructor, nothing else on the GPU: image in pinned host
memory -> so_dframe_submit (ingest + pyramid + FAST + quadtree + describe as one hipGraph, + frame_prepare) ->
so_dframe_wait (the frame is complete on the device) [-> so_dframe_collect: host mirrors].  The extraction part of the
live image-to-pose latency (bench.py: latency_ms_image_to_pose).
    python tools/extract_latency.py [euroc|kitti]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd  # noqa: E402
from swarmmap_amd import synth  # noqa: E402


def main():
    kitti = "kitti" in sys.argv[1:]
    size, K, dist, nf = (synth.KITTI, synth.KITTI_K, None, 2000) if kitti else (synth.EUROC, synth.EUROC_K, synth.EUROC_DIST, 1000)
    st = synth.FrameStream(seed=20221001, size=size, K=K, dist=dist)
    n = 120
    block = torch.empty((n, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n):
        view[t] = st.frame(t)
    ex = swarmmap_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    f = swarmmap_amd.DeviceFrame(ex, K, dist if dist is not None else (0.0, 0.0, 0.0, 0.0, 0.0))
    lib = f._lib
    lib.so_dframe_wait.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    bounds = np.zeros(4, np.float32)
    wait_ms, full_ms = [], []
    for t in range(n):
        t0 = time.perf_counter()
        f.submit(view[t])
        nk = C.c_int(0)
        lib.so_dframe_wait(f._h, C.byref(nk), bounds.ctypes.data)
        t1 = time.perf_counter()
        f.collect()
        t2 = time.perf_counter()
        if t >= 20:
            wait_ms.append((t1 - t0) * 1e3)
            full_ms.append((t2 - t0) * 1e3)
    p = lambda a, q: float(np.percentile(a, q))  # noqa: E731
    print(json.dumps({"stream": "kitti" if kitti else "euroc", "frames": len(wait_ms),
                      "submit_to_device_complete_ms": {"p50": p(wait_ms, 50), "p90": p(wait_ms, 90), "min": float(np.min(wait_ms))},
                      "submit_to_host_mirrors_ms": {"p50": p(full_ms, 50), "p90": p(full_ms, 90)}}))
    f.close(); ex.close()


if __name__ == "__main__":
    main()
