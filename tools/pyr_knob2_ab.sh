#!/bin/bash
# the packed pyramid kernel by tuning presets (VPP_TUNE), interleaved on one box: the kernel alone and the flow's pair
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do for t in "$@"; do
  echo "== $t (pass $rep)"
  VPP_TUNE="$t" timeout 200 python tools/time_pyr_knob.py pyr.xcd 1 2>&1 | grep "x" | head -3 | cut -c1-60
  VPP_TUNE="$t" timeout 200 python tools/time_flow_min.py 2>&1 | grep "2160p"
done; done
