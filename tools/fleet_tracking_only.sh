cd /root/repo
for A in 1 4 8; do
flag="--lockstep"; [ $A = 1 ] && flag=""
env SWARMORB_BENCH_LOOP=open SWARMORB_BENCH_LBA_EVERY=100000 SWARMORB_BENCH_LM_MATCHER=0 timeout 300 python bench.py --agents-per-gpu $A $flag --steps 300 --warmup 20 --no-configs --no-cpu-baseline 2>/tmp/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tracking only, agents', $A, 'fps', round(d['value']), 'tick p50', d['config']['frame_ms_percentiles']['p50'], 'pose kernel', d['config']['pose_kernel_ms_per_call'])" || tail -5 /tmp/err.txt
done
