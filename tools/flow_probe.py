"""Stage timeline of dense_flow_kernel (developer tool, GPU box):  python tools/flow_probe.py [free keyframes, default 64]
Builds a copy of the library with -DSO_FLOW_PROBE (wall-clock marks per workgroup and stage) under tools/probe/, solves
one window with it and prints, per tile, when each stage ended (us since the first workgroup started)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "swarmmap_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probe", "libswarmorb_flowprobe.so")


def build():
    objs = []
    for src in ("orb_kernels.hip", "quadtree_kernel.hip", "extractor.cpp", "quadtree.cpp", "match_kernels.hip", "matcher.cpp",
                "frame_kernels.hip", "frame.cpp", "dframe.cpp", "kfstore_kernels.hip", "kfstore.cpp", "exchange.cpp", "ba_kernels.hip",
                "ba.cpp", "record.cpp", "capi.cpp"):
        objs.append(os.path.join(CSRC, "build", os.path.splitext(src)[0] + ".o"))
    probe_o = os.path.join(ROOT, "tools", "probe", "ba_dense_flowprobe.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-DSO_FLOW_PROBE", "-c", os.path.join(CSRC, "ba_dense.hip"), "-o", probe_o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs + [probe_o, "-ldl"])


if __name__ == "__main__":
    if "--build" in sys.argv or not os.path.exists(OUT):
        build()
        if "--build" in sys.argv:
            sys.exit(0)
    import swarmmap_amd._lib as L
    L.library_path = lambda: OUT
    import numpy as np
    from swarmmap_amd import synth
    from swarmmap_amd.optimizer import Optimizer
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    nf = int(args[0]) if args else 64
    w = synth.make_ba_problem(0, nf, (3 * nf) // 2, 150 * nf, max_obs="auto")
    o = Optimizer()
    for _ in range(3):
        o.LocalBundleAdjustment(w)
    lib = L.load_library()
    marks = np.zeros((256, 16), np.uint64)
    lib.so_debug_flow_marks.argtypes = [C.c_void_p]
    rc = lib.so_debug_flow_marks(marks.ctypes.data)
    assert rc == 0, rc
    T = (6 * nf + 95) // 96
    tiles = [(i, j) for j in range(T) for i in range(j, T)]
    t0 = min(int(marks[b, 0]) for b in range(len(tiles)))
    names = ["start", "accumulated", "Linv seen", "trsm done", "L published", "syrk done", "potrf done", "Linv published",
             "fwd vec / y", "bwd vec / x"]
    print("tile      " + " ".join("%14s" % n for n in names))
    for b, (i, j) in enumerate(tiles):
        row = []
        for k in range(10):
            v = int(marks[b, k])
            row.append("%14.2f" % ((v - t0) / 100.0) if v >= t0 and v - t0 < 10**8 else "%14s" % "-")  # 100 MHz wall clock
        print("(%2d,%2d)   " % (i, j) + " ".join(row))
