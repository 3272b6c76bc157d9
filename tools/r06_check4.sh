#!/bin/bash
# round 6: FAST-9 RAW in one launch (stage + band gather): parity, timing A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_algos.py tests/test_gpu_edges.py tests/test_gpu_video_steps.py tests/test_gpu_video_extruder.py -m gpu -x -q -k "fast or extruder or detect or keypoint" > gpurun_out/gputests4.log 2>&1; echo "gpu tests exit $?"; tail -5 gpurun_out/gputests4.log
for i in 1 2; do
echo "== fast9 timing, fused write on (default)"; timeout 120 python tools/fast_time.py 2>&1 | grep -v amdgpu.ids
echo "== fast9 timing, fused write off"; timeout 120 python tools/fast_time.py fast9.raw_fused=0 2>&1 | grep -v amdgpu.ids
done
