"""Repeated solves through both tile-dataflow kernels, alone and from two threads at once (developer tool, GPU box):
every result must equal the first one bit for bit, and nothing may hang.  python tools/flow_stress.py [rounds]"""
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np
import swarmmap_amd
from swarmmap_amd import synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = {"w64": synth.make_ba_problem(0, 64, 96, 9600, max_obs="auto"), "w128": synth.make_ba_problem(1, 128, 192, 19200, max_obs="auto"),
         "GBA-1r": synth.make_ba_case("GBA-1r", 1), "GBA-2r": synth.make_ba_case("GBA-2r", 1)}
o = swarmmap_amd.Optimizer()
ref = {}
t0 = time.time()
for name, p in cases.items():
    local = name.startswith("w")
    for k in range(rounds if local else max(3, rounds // 4)):
        r = o.LocalBundleAdjustment(p) if local else o.BundleAdjustment(p, nIterations=4, bRobust=True)
        key = (r["Tcw"].tobytes(), r["Xw"].tobytes(), r["info"]["solver_path"])
        if name not in ref:
            ref[name] = key
            print(name, "path", r["info"]["solver_path"], "tiles", r["info"]["nnz_tiles"], flush=True)
        assert key == ref[name], (name, k)
print("sequential ok %.1f s" % (time.time() - t0), flush=True)
others = [swarmmap_amd.Optimizer() for _ in range(2)]
errs = []
def work(i, name):
    p = cases[name]
    for k in range(rounds // 2):
        r = others[i].LocalBundleAdjustment(p) if name.startswith("w") else others[i].BundleAdjustment(p, nIterations=4, bRobust=True)
        if (r["Tcw"].tobytes(), r["Xw"].tobytes()) != ref[name][:2]:
            errs.append((name, k, r["info"]["solver_path"]))
for pair in (("w64", "w128"), ("w128", "GBA-1r"), ("GBA-2r", "w64")):
    ths = [threading.Thread(target=work, args=(i, n)) for i, n in enumerate(pair)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    print("concurrent", pair, "mismatches", len(errs), flush=True)
# the same solves while another stream keeps the CUs busy (workgroups of the dataflow launches then start late and in
# odd orders - the condition under which a race between tiles would show)
import torch
stop = []
def noise():
    x = torch.randn(6144, 6144, device="cuda")
    while not stop:
        y = x @ x
        torch.cuda.synchronize()
nt = threading.Thread(target=noise)
nt.start()
try:
    for name, p in cases.items():
        local = name.startswith("w")
        for k in range(rounds if local else max(3, rounds // 4)):
            r = o.LocalBundleAdjustment(p) if local else o.BundleAdjustment(p, nIterations=4, bRobust=True)
            if (r["Tcw"].tobytes(), r["Xw"].tobytes()) != ref[name][:2]:
                errs.append(("noise", name, k, r["info"]["solver_path"]))
finally:
    stop.append(1)
    nt.join()
print("under load: mismatches", len(errs), flush=True)
print("done %.1f s" % (time.time() - t0), "errors", errs[:5], flush=True)
sys.exit(1 if errs else 0)
