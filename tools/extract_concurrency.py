"""How do independent front-end chains share one GPU?  N threads, each with its own extractor + device-resident frame on
a stream of its own, each running submit -> wait back to back: aggregate frames/s for N = 1, 2, 4, 8.
    python tools/extract_concurrency.py"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd  # noqa: E402
from swarmmap_amd import synth  # noqa: E402
from swarmmap_amd.replay import private_streams  # noqa: E402


def worker(view, n_frames, go, out, k):
    try:
        _worker(view, n_frames, go, out, k)
    except BaseException as e:  # noqa: BLE001 - reported, and the others must not wait for this thread
        print("worker %d failed: %r" % (k, e), flush=True)
        go.abort()


def _worker(view, n_frames, go, out, k):
    ex = swarmmap_amd.ORBextractor(1000, 1.2, 8, 20, 7)
    f = swarmmap_amd.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST)
    lib = f._lib
    lib.so_dframe_wait.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    bounds = np.zeros(4, np.float32)
    nk = C.c_int(0)
    for t in range(10):
        f.submit(view[t]); lib.so_dframe_wait(f._h, C.byref(nk), bounds.ctypes.data); f.collect()
    go.wait(timeout=60)
    t0 = time.perf_counter()
    for t in range(n_frames):
        f.submit(view[t % len(view)])
        lib.so_dframe_wait(f._h, C.byref(nk), bounds.ctypes.data)
        f.collect()
    out[k] = time.perf_counter() - t0
    f.close(); ex.close()


def main():
    st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=synth.EUROC_K, dist=synth.EUROC_DIST)
    n = 32
    block = torch.empty((n, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n):
        view[t] = st.frame(t)
    private_streams(True)
    for N in (1, 2, 4, 8):
        go = threading.Barrier(N + 1)
        out = [0.0] * N
        frames = 300
        ths = [threading.Thread(target=worker, args=(view, frames, go, out, k)) for k in range(N)]
        for th in ths:
            th.start()
        go.wait(timeout=60)
        for th in ths:
            th.join()
        print(json.dumps({"threads": N, "frames_per_s_aggregate": N * frames / max(out), "ms_per_frame_per_thread": 1e3 * max(out) / frames}), flush=True)


if __name__ == "__main__":
    main()
