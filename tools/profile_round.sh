#!/bin/bash
# Refresh the judged summaries on the GPU box:  bash tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>_*)
# kernel-trace/stats and each PMC counter run in their own pass (never combined with other trace domains).
set -u
tag=${1:-rX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cp profiles/last_bench_full.json $out/${tag}_bench_full.json   # (the profiler passes below overwrite profiles/last_bench_full.json with their own, slower, runs)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_args.json 2>> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -- python3 bench.py --steps 200 --no-cpu-baseline --no-configs > $out/${tag}_prof.log 2>&1
f=$(find $out/prof_${tag} -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  SWARMORB_NO_GRAPH=1 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_$c -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $out/${tag}_pmc_$c.log 2>&1
  f=$(find $out/pmc_${tag}_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" $out/${tag}_pmc_${c}_summary.csv
  rm -rf $out/pmc_${tag}_$c
done
rm -rf $out/prof_${tag}
tail -1 $out/${tag}_bench.json | cut -c1-400
