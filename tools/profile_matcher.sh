#!/bin/bash
# Profiler evidence for the ORBmatcher routines and the batched keyframe load:  bash tools/profile_matcher.sh <tag>
# (kernel trace in its own pass, one --pmc counter per pass, nothing else traced)
set -u
tag=${1:-rX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 tools/matcher_bench.py > $out/${tag}_matcher.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_matcher -- python3 tools/matcher_bench.py > $out/${tag}_matcher_prof.log 2>&1
f=$(find $out/prof_${tag}_matcher -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_matcher_kernel_stats.csv
rm -rf $out/prof_${tag}_matcher
: > $out/${tag}_matcher_pmc.txt
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_matcher_$c -- python3 tools/matcher_bench.py > $out/${tag}_matcher_pmc_$c.log 2>&1
  f=$(find $out/pmc_${tag}_matcher_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" $out/${tag}_matcher_pmc_${c}.csv && grep -h "topk_window_kernel<4>\|project_queries\|topk_window_kernel<0>\|distinctive\|hamming_top2" $out/${tag}_matcher_pmc_${c}.csv >> $out/${tag}_matcher_pmc.txt
  rm -rf $out/pmc_${tag}_matcher_$c $out/${tag}_matcher_pmc_${c}.csv $out/${tag}_matcher_pmc_$c.log
done
cat $out/${tag}_matcher_pmc.txt | cut -c1-220
