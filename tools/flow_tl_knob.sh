#!/bin/bash
# tools/flow_tl_knob.sh knob v1 v2 ...: the flow's per-dispatch timeline (tools/flow_timeline.sh) under each value of one tuning knob (VPP_TUNE="knob=value" read by tools/rounds_ab.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; k=$1; shift
for v in "$@"; do echo "== $k=$v"; VPP_TUNE="$k=$v" bash tools/flow_timeline.sh 2>&1 | grep "^| "; done
