#!/bin/bash
# tools/flow_tl_lib.sh tmp_ab/libX.so ...: tools/flow_timeline.sh with another build of the library in place of the default one (timing experiments)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp vpp_amd/csrc/libvpp_amd.so /tmp/lib_keep.so
for L in "$@"; do echo "== $L"; cp $L vpp_amd/csrc/libvpp_amd.so; bash tools/flow_timeline.sh 2>&1 | grep "^| "; done
cp /tmp/lib_keep.so vpp_amd/csrc/libvpp_amd.so
