export TMPDIR=/tmp
for n in 64 96 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lba$n -- python3 tools/lba_bench.py --only $n > gpurun_out/r3_lba${n}.log 2>&1
  f=$(find gpurun_out/prof_lba$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r3_lba${n}_kernel_stats.csv
  rm -rf gpurun_out/prof_lba$n
  tail -1 gpurun_out/r3_lba${n}.log
done
python3 tools/lba_bench.py --sweep
