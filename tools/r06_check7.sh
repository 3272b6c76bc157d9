#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 tests/cpp/_build/device_lambda_test time 2>&1 | tail -25
echo "== window off"; VPP_PW_WINDOW=0 timeout 900 tests/cpp/_build/device_lambda_test time 2>&1 | grep "int box"
timeout 600 python -m pytest tests/test_gpu_sdof.py -m gpu -x -q 2>&1 | tail -2
