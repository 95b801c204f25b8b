#!/usr/bin/env python3
"""profiles/<tag>_gba_pmc_mfma.json + profiles/<tag>_gba_kernel_stats.csv from what tools/profile_gba.sh leaves in
gpurun_out/: per map (GBA-1, GBA-2, GBA-1r, GBA-2r) and per solver kernel (dense_flow_kernel: a workgroup per tile;
dense_flow_big_kernel: tiles by ticket) the rocprofv3 kernel-trace duration and the MFMA counters of separate --pmc passes.
    SQ_INSTS_VALU_MFMA_MOPS_F64  x 512 = FP64 flop the matrix cores executed (v_mfma_f64_16x16x4_f64 = 2048 flop = 4 MOPS)
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -> share of the kernel's SIMD-cycles the matrix pipes were busy
Usage: make_gba_pmc.py <tag> [gpurun_out] [profiles]"""
import csv
import json
import os
import re
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles"
PEAK = 78.6
KERNELS = ("dense_flow_kernel", "dense_flow_big_kernel")
bench = {}
for line in open(os.path.join(src, "%s_gba.json" % tag)):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        bench[d["case"]] = d
out = {"command": "bash tools/profile_gba.sh %s: per map `rocprofv3 --kernel-trace --stats -- python3 tools/gba_bench.py <case>` and one "
                  "`rocprofv3 --pmc <counter> -- python3 tools/gba_bench.py <case>` pass per counter (2 warm-up + 10 timed LM "
                  "iterations per pass = 12 solves)" % tag,
       "fp64_peak_tflops": PEAK, "cases": {}}
stats_rows = [["case", "kernel", "calls", "total_ns", "average_ns", "min_ns", "max_ns"]]
for case in ("GBA-1", "GBA-2", "GBA-1r", "GBA-2r"):
    ks = os.path.join(src, "%s_gba_%s_kernel_stats.csv" % (tag, case))
    if not os.path.exists(ks):
        continue
    rec = {}
    for r in csv.DictReader(open(ks)):
        m = re.search(r"so::(\w+)", r["Name"])
        name = m.group(1) if m else r["Name"][:40]
        stats_rows.append([case, name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
        if name in KERNELS:
            rec = {"kernel": name, "calls": int(r["Calls"]), "average_us_kernel_trace": float(r["AverageNs"]) / 1e3}
    for c in ("SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        p = os.path.join(src, "%s_gba_%s_pmc_%s.csv" % (tag, case, c))
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if rec and ("so::%s(" % rec["kernel"]) in r["kernel"]:
                rec[c + "_per_launch"] = float(r["mean_per_dispatch"])
    if not rec:
        continue
    b = bench.get(case, {}).get("solve", {})
    us = rec["average_us_kernel_trace"]
    if "SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch" in rec:
        rec["mfma_flop_per_launch_from_counter"] = 512.0 * rec["SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch"]
        rec["mfma_tflops_from_counter"] = rec["mfma_flop_per_launch_from_counter"] / (us * 1e-6) / 1e12
    if b:
        rec["structural_flop_per_solve"] = b["structural_flop"]
        rec["dense_flop_per_solve"] = b["dense_flop"]
        rec["achieved_tflops_min_count"] = min(b["structural_flop"], b["dense_flop"]) / (us * 1e-6) / 1e12
        rec["frac_of_fp64_peak"] = rec["achieved_tflops_min_count"] / PEAK
        rec["ms_per_solve_hip_events_unprofiled"] = b["ms_per_solve"]
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in rec and rec.get("GRBM_GUI_ACTIVE_per_launch", 0) > 0:
        # GRBM_GUI_ACTIVE sums the 8 XCDs' clocks; SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 SIMDs' matrix
        # pipes (= 64 cycles per v_mfma_f64_16x16x4_f64: MOPS / 4 x 64 reproduces it exactly)
        cycles = rec["GRBM_GUI_ACTIVE_per_launch"] / 8.0
        rec["shader_clock_ghz_during_kernel"] = cycles / (us * 1e-6) / 1e9
        rec["mfma_pipe_busy_share"] = rec["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / (cycles * 1024.0)
    out["cases"][case] = rec
json.dump(out, open(os.path.join(dst, "%s_gba_pmc_mfma.json" % tag), "w"), indent=1)
csv.writer(open(os.path.join(dst, "%s_gba_kernel_stats.csv" % tag), "w")).writerows(stats_rows)
print(json.dumps(out["cases"], indent=1))
