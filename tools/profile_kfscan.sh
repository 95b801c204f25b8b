#!/bin/bash
# Profiler evidence for the detection scan of the cross-agent candidate search:  bash tools/profile_kfscan.sh <tag>
set -u
tag=${1:-rX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
export SWARMORB_KFSCAN_ONLY_DEFAULT=1
python3 tools/kfscan_bench.py > $out/${tag}_kfscan.json 2> $out/${tag}_kfscan.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_kfscan -- python3 tools/kfscan_bench.py > $out/${tag}_kfscan_prof.log 2>&1
f=$(find $out/prof_${tag}_kfscan -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kfscan_kernel_stats.csv
rm -rf $out/prof_${tag}_kfscan
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM FETCH_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_kfscan_$c -- python3 tools/kfscan_bench.py > $out/${tag}_kfscan_pmc_$c.log 2>&1
  f=$(find $out/pmc_${tag}_kfscan_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" $out/${tag}_kfscan_pmc_${c}.csv
  rm -rf $out/pmc_${tag}_kfscan_$c
done
grep -h "kf_scan" $out/${tag}_kfscan_kernel_stats.csv $out/${tag}_kfscan_pmc_*.csv | cut -c1-200
