#!/bin/bash
# closed loop, A agents on one GPU: thread pairs vs lockstep with grouped stages
cd /root/repo
for A in 1 2 4 8; do
  for mode in threads lockstep; do
    if [ $A = 1 ] && [ $mode = lockstep ]; then continue; fi
    flag=""; [ $mode = lockstep ] && flag="--lockstep"
    timeout 300 python bench.py --agents-per-gpu $A $flag --steps 300 --warmup 20 --no-configs --no-cpu-baseline 2>gpurun_out/fleet_err_${A}_$mode.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'agents_per_gpu': $A, 'mode': '$mode', 'frames_per_s_aggregate': round(d['value'],1), 'fps_per_agent': round(d['fps_per_agent'],1), 'lm_ms_per_keyframe': d['config'].get('local_mapping_ms_per_keyframe'), 'ba_ms': d['config'].get('so_bundle_adjust_ms'), 'pose_kernel_ms': d['config'].get('pose_kernel_ms_per_call'), 'frame_p50': d['config']['frame_ms_percentiles'].get('p50')}))" || tail -3 gpurun_out/fleet_err_${A}_$mode.txt
  done
done
