#!/bin/bash
# the whole GPU suite + the randomized flow sweep (round 6, second session)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_r06b.log 2>&1; echo "gpu tests exit $?"; tail -4 gpurun_out/gputests_r06b.log
timeout 600 python tools/stress_parity.py 2>&1 | tail -5
