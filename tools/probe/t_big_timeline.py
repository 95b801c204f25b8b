"""Pace of dense_flow_big_kernel's three chains on a large map (developer tool, probe build): when each block column's
factor, y and x were published (us since the first factor)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import swarmmap_amd._lib as L
L.library_path = lambda: os.path.join(ROOT, "tools", "probe", "libswarmorb_flowprobe.so")
import numpy as np
import swarmmap_amd
from swarmmap_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "GBA-2"
p = synth.make_ba_case(name, 1)
o = swarmmap_amd.Optimizer()
r = o.BundleAdjustment(p, nIterations=2, bRobust=True)
lib = L.load_library()
d = np.zeros((10, 512), np.uint64)
lib.so_debug_flow_diag.argtypes = [C.c_void_p]
assert lib.so_debug_flow_diag(d.ctypes.data) == 0
T = (6 * int((p["fixed"] == 0).sum()) + 95) // 96
t0 = int(d[0, 0])
f = [(int(d[0, j]) - t0) / 100.0 for j in range(T)]
y = [(int(d[1, j]) - t0) / 100.0 for j in range(T)]
x = [(int(d[2, j]) - t0) / 100.0 for j in range(T)]
print(name, "T", T, "solve_ms", r["info"]["solve_ms"] / max(r["info"]["n_solves"], 1), "path", r["info"]["solver_path"], "tiles", r["info"]["nnz_tiles"], flush=True)
rel = lambda row, j: (int(d[row, j]) - t0) / 100.0 if int(d[row, j]) else float("nan")
print("col   picked up   history done   Linv_{J-1} seen   factor out        y        x | tile (J+1,J): picked up, history done, Linv_J seen, published", flush=True)
for j in (list(range(0, T, max(1, T // 12))) + list(range(40, 48))):
    print("%3d  %10.1f  %13.1f  %16.1f  %11.1f  %7.1f  %7.1f" % (j, rel(3, j), rel(4, j), rel(5, j), rel(0, j), rel(1, j), rel(2, j)) + "   | %9.1f %9.1f %9.1f %9.1f" % (rel(6, j), rel(7, j), rel(8, j), rel(9, j)), flush=True)
