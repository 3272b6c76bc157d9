#!/bin/bash
# several builds of the library on one box: vpp_pyramid_build alone (tools/time_pyr_knob.py) and the flow's pair (tools/time_flow_min.py): tools/pyr_lib_ab.sh tmp_ab/libA.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do for L in "$@"; do
  echo "== $L (pass $rep)"
  VPP_AMD_LIB=$R/$L timeout 200 python tools/time_pyr_knob.py pyr.row_tiles 4 2>&1 | grep "x" | head -3
  VPP_AMD_LIB=$R/$L timeout 200 python tools/time_flow_min.py 2>&1 | grep "2160p"
done; done
