#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python tools/flow_replicated_share.py 2 8 > gpurun_out/r06_flow_replicated_share.md 2> gpurun_out/flow_share.err; echo "share exit $?"; cat gpurun_out/r06_flow_replicated_share.md | head -40; tail -3 gpurun_out/flow_share.err
timeout 300 benchmarks/lambda_call_bench 200
timeout 600 tests/cpp/_build/device_lambda_test time 2>&1 | tail -8
