#!/bin/bash
# the 4K tracker benchmark with several builds of the library in place, interleaved (tools/r06b_ve_ab.sh tmp_ab/libA.so ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp vpp_amd/csrc/libvpp_amd.so /tmp/lib_keep.so
for rep in 1 2; do for L in "$@"; do cp $L vpp_amd/csrc/libvpp_amd.so; echo "== $L"; timeout 200 benchmarks/video_extruder_bench 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_update_median_steady'], d['ms_per_frame_frames_3_to_end_incl_detection_frames']['frames_in_hbm'])"; done; done
cp /tmp/lib_keep.so vpp_amd/csrc/libvpp_amd.so
