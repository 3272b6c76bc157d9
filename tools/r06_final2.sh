#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
fails=0
for i in $(seq 1 25); do timeout 120 python -m pytest tests/test_gpu_core.py -m gpu -x -q -k "cross_host_threads or several_threads" > /tmp/t_$i.log 2>&1 || { fails=$((fails+1)); tail -5 /tmp/t_$i.log; }; done
echo "thread tests: $fails failures of 25 runs"
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_final.log 2>&1; echo "gpu tests exit $?"; grep -E "passed|failed" gpurun_out/gputests_final.log | tail -2
