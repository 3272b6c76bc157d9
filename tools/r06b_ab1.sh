#!/bin/bash
# round 6, second session: descent variant A/B (sdof.descent_var 0/1) + the sdof parity tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sdof.py -x -q 2>&1 | tail -3
timeout 600 python tools/flow_knob_ab.py sdof.descent_var 0 1 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/ab_descent_var.log
