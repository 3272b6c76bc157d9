#!/bin/bash
# FAST-9 timing of several builds on one box (tools/fast_time.py), interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2 3; do for L in "$@"; do echo "== $L (pass $rep)"; VPP_AMD_LIB=$R/$L timeout 200 python tools/fast_time.py 2>&1 | grep "^mode\|^async"; done; done
