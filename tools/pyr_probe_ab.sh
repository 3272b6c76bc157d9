#!/bin/bash
# same-box timing of several builds of the library on vpp_pyramid_build (tools/time_pyr_knob.py): tools/pyr_probe_ab.sh tmp_ab/libA.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for L in "$@"; do
    echo "== $L (pass $rep)"
    VPP_AMD_LIB=$R/$L timeout 200 python $R/tools/time_pyr_knob.py pyr.wide 1 2>&1 | grep "x"
  done
done
