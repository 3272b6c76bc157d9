#!/bin/bash
# Round 6: does editing a kernel node of a graph under capture (what rounds 4-5 did for record-time batching) abort under rocprofv3, in a HIP-only program?
# tools/capture_setparams_repro {edit, deps, plain} x RUNS under `rocprofv3 --kernel-trace`, and once each without the profiler; exit statuses are counted.
# Run on the GPU box from the repo root: bash tools/capture_repro.sh [runs]; summary in gpurun_out/capture_repro.txt
set -u
RUNS=${1:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/capture_repro
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXE=$R/tools/capture_setparams_repro
{
for mode in ${MODES:-edit deps plain timed}; do
  timeout -k 5 60 $EXE $mode 40 > $OUT/${mode}_bare.log 2>&1; echo "$mode without profiler: exit $?"
  ok=0; bad=0
  for i in $(seq $RUNS); do
    rm -rf $OUT/kt
    timeout -k 5 120 rocprofv3 --kernel-trace -d $OUT/kt -o t -- $EXE $mode 40 > $OUT/${mode}_$i.log 2>&1
    rc=$?
    if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "  $mode run $i: exit $rc: $(grep -m1 -iE 'free\(\)|corrupt|abort|malloc|segm' $OUT/${mode}_$i.log)"; fi
  done
  echo "$mode under rocprofv3 --kernel-trace: $ok ok, $bad failed of $RUNS"
done
} | tee $R/gpurun_out/capture_repro.txt
rm -rf $OUT/kt
