#!/bin/bash
# Front-end latency check on the GPU box:  bash tools/profile_frontend.sh [tag]
# wall-clock submit -> complete of one frame (no profiler), then the per-kernel timeline of the chain under
# rocprofv3 --kernel-trace (tools/frame_timeline.py).  Outputs under gpurun_out/<tag>_*.
set -u
tag=${1:-fe}
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 tools/extract_latency.py > gpurun_out/${tag}_extract_latency.json
python3 tools/extract_latency.py kitti >> gpurun_out/${tag}_extract_latency.json
cat gpurun_out/${tag}_extract_latency.json
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_tl -- python3 tools/extract_latency.py > gpurun_out/${tag}_tl.log 2>&1
python3 tools/frame_timeline.py gpurun_out/${tag}_tl | tee gpurun_out/${tag}_frame_timeline.txt
rm -rf gpurun_out/${tag}_tl
