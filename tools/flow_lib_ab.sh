#!/bin/bash
# A/B of several builds of the library on ONE box: tools/flow_lib_ab.sh tmp_ab/libA.so tmp_ab/libB.so ... (each through tools/time_flow_min.py, twice, interleaved)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for L in "$@"; do
    echo "== $L (pass $rep)"
    VPP_AMD_LIB=$R/$L timeout 200 python $R/tools/time_flow_min.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids"
  done
done
