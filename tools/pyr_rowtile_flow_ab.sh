#!/bin/bash
# the flow's pair with the pyramid's row tiles (VPP_TUNE presets), interleaved on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do for t in "pyr.row_tiles=0" "pyr.row_tiles=4,pyr.xcd=0" "pyr.row_tiles=4,pyr.xcd=1" "pyr.row_tiles=2,pyr.xcd=1"; do
  echo "== $t (pass $rep)"; VPP_TUNE="$t" timeout 200 python tools/time_flow_min.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids"
done; done
