#!/bin/bash
# bash tools/fleet_ab.sh "<env assignments>" <bench flags...>   -> one line
cd /root/repo
envs="$1"; shift
env $envs timeout 300 python bench.py "$@" --steps 300 --warmup 20 --no-configs --no-cpu-baseline 2>/tmp/fleet_ab_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
f=json.load(open('profiles/last_bench_full.json'))
L=f['config']['closed_loop']['local_mapping_ms_per_keyframe']
print(json.dumps({'env': '$envs', 'flags': '$*', 'frames_per_s_aggregate': round(d['value'],1), 'lm_job_ms': round(L['whole_job'],3), 'ba_ms': round(L['so_bundle_adjust'],3), 'tri_ms': round(L['triangulation_and_new_points'],3), 'und_ms': round(L['write_back_split']['so_update_normal_and_depth'],3), 'fuse_wait_ms': round(L['fuse_batch_launch_wait_resolve'],3), 'pose_kernel_ms': d['config'].get('pose_kernel_ms_per_call'), 'frame_p50': d['config']['frame_ms_percentiles'].get('p50'), 'waited_ms': d['config'].get('tracking_thread_waited_ms'), 'host_cpu': d.get('host_cpu'), 'ticks': d.get('fleet_ticks')}))" || tail -5 /tmp/fleet_ab_err.txt
