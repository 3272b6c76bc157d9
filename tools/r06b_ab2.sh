#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sdof.py -x -q 2>&1 | tail -3
bash tools/flow_lib_ab.sh "$@"
