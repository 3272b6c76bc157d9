#!/bin/bash
# Round 5: bench.py died under rocprofv3 with a glibc heap-corruption abort in a non-main thread (1 run of 3 survived).  Up to N attempts with stage
# markers; on a crash: the stage reached and — when a core file exists — the backtraces of all threads.
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-3}
cd /tmp && export TMPDIR=/tmp
ulimit -c unlimited
for k in $(seq 1 $N); do
  rm -rf /tmp/kt_dbg; rm -f /tmp/core* 2>/dev/null
  VPP_BENCH_TRACE=1 timeout -k 10 150 rocprofv3 --kernel-trace --stats -d /tmp/kt_dbg -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > /tmp/dbg.out 2> /tmp/dbg.err
  rc=$?
  echo "== attempt $k rc=$rc"
  if [ $rc -ne 0 ]; then
    grep "bench stage\|free()\|malloc\|corrupt\|Assertion" /tmp/dbg.err | tail -8
    cat /proc/sys/kernel/core_pattern
    C=$(ls -t /tmp/core* core* 2>/dev/null | head -1)
    echo "core: $C"
    if [ -n "$C" ]; then timeout 120 rocgdb -batch -ex "thread apply all bt 14" $(which python) $C 2>&1 | grep -v "^\[New LWP\|^warning" | head -150; fi
    break
  fi
done
