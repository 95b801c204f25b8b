#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two per-counter summaries tools/profile_round.sh leaves
(<tag>_pmc_FETCH_SIZE_summary.csv, <tag>_pmc_WRITE_SIZE_summary.csv).  FETCH_SIZE is doubled: on gfx950 the counter
reads half the streamed bytes (calibrated with tools/pmc_calibrate.py: 2^30 bytes read 524300 KB,
profiles/r1_pmc/cal_FETCH_SIZE_summary.csv); both counters are in KB.
Usage: make_pmc_traffic.py <FETCH summary> <WRITE summary> <out.json>"""
import csv
import json
import re
import sys

KEEP = ("fast_score_kernel", "fast_low_kernel", "describe_qt_kernel", "resize_kernel", "topk_window_kernel",
        "stage_in_kernel", "pose_opt_lds_kernel", "pose_opt_reg_kernel", "pose_opt_chain_kernel", "track_resolve_kernel", "ba_solve_la_kernel", "ba_solve_mfma_kernel",
        "ba_schur_gather_kernel", "ba_build_kernel", "ba_update_kernel", "ba_update_errors_kernel", "quadtree_kernel", "frame_prepare_kernel", "ingest_kernel", "ingest16_kernel",
        "project_queries_batch_kernel", "hamming_top2_kernel", "triangulate_kernel", "normal_depth_kernel", "ba_mark_outliers_kernel",
        "ba_errors_kernel", "map_scatter_rows_kernel")


def load(path):
    """by kernel name; template instances also under their own name ("topk_window_kernel<4>": the batched search is not the
    tracking searches), the bare name holding the dispatch-weighted mean over the instances"""
    out, acc = {}, {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"so::(\w+)(<[\w, ]+>)?", r["kernel"])
        if m and m.group(1) in KEEP:
            mean, n = float(r["mean_per_dispatch"]), int(r["dispatches"])
            if m.group(2):
                out[m.group(1) + m.group(2)] = (mean, n)
            t = acc.setdefault(m.group(1), [0.0, 0])
            t[0] += mean * n
            t[1] += n
    for k, (tot, n) in acc.items():
        out[k] = (tot / max(n, 1), n)
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (one counter per pass, no other trace domain) of "
                 "`bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs` on MI355X (tools/profile_round.sh); "
                 "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE x2 per the calibration in "
                 "profiles/r1_pmc/cal_FETCH_SIZE_summary.csv"}
for k in sorted(set(fetch) & set(write)):
    if True:
        f, w = fetch[k][0], write[k][0]
        res[k] = {"fetch_kb_per_launch": round(f, 1), "write_kb_per_launch": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024), "dispatches": fetch[k][1]}
json.dump(res, open(sys.argv[3], "w"), indent=1)
