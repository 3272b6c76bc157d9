#!/bin/bash
# A/B of one tuning knob over several values (tools/flow_knob_ab.py) after the sdof parity tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sdof.py -x -q 2>&1 | tail -3
timeout 600 python tools/flow_knob_ab.py "$@" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids"
