#!/bin/bash
# What bounds the front-end kernels when they are throughput-bound (16 frames per chain): per-kernel instruction and LDS
# counters, one rocprofv3 --pmc pass per counter (never combined with another trace domain).
#   bash tools/profile_frontend_pmc.sh [frames per chain, default 16]   -> gpurun_out/fe_pmc_<counter>.csv + a table
A=${1:-16}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  SWARMORB_NO_GRAPH=1 rocprofv3 --pmc $c --output-format csv -d gpurun_out/fe_pmc_$c -- python3 tools/extract_batch_bench.py $A > gpurun_out/fe_pmc_$c.log 2>&1
  f=$(find gpurun_out/fe_pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" gpurun_out/fe_pmc_${c}.csv
  rm -rf gpurun_out/fe_pmc_$c
done
python3 - <<'PY'
import csv, glob, re, os
tab = {}
for f in sorted(glob.glob("gpurun_out/fe_pmc_*.csv")):
    c = os.path.basename(f)[len("fe_pmc_"):-4]
    for r in csv.DictReader(open(f)):
        m = re.search(r"so::(\w+)", r["kernel"])
        if m and "batch" in m.group(1):
            tab.setdefault(m.group(1), {})[c] = float(r["mean_per_dispatch"])
cols = sorted({c for v in tab.values() for c in v})
print("%-28s" % "kernel" + " ".join("%14s" % c.replace("SQ_", "")[:14] for c in cols))
for k, v in tab.items():
    print("%-28s" % k + " ".join("%14.3g" % v.get(c, float("nan")) for c in cols))
PY
