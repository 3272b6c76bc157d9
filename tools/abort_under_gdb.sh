#!/bin/bash
# Round 6: whose free() trips glibc's heap check when bench.py runs under rocprofv3?  The profiled process runs under rocgdb (batch mode): a SIGABRT stops it before any
# handler runs and the backtraces of the aborting thread and of the main thread are printed.  Up to N passes; stops at the first abort.
#   bash tools/abort_under_gdb.sh [passes]        output: gpurun_out/abort_gdb.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-12}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gdbcmds <<'GDB'
set pagination off
set confirm off
handle SIGABRT stop print nopass
handle SIGSEGV stop print
handle SIG32 nostop noprint pass
handle SIG33 nostop noprint pass
handle SIG34 nostop noprint pass
run
echo \n==== stopped: aborting thread ====\n
bt 40
echo \n==== where the libraries are loaded ====\n
info sharedlibrary rocprofiler
info sharedlibrary hsa-runtime
info sharedlibrary vpp_amd
info sharedlibrary amdhip
echo \n==== main thread ====\n
thread 1
bt 25
echo \n==== threads inside libvpp_amd / libamdhip64 (none expected) ====\n
thread apply all -s -q bt 6
kill
quit
GDB
for i in $(seq 1 $N); do
  rm -rf /tmp/kt_gdb
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_gdb -o bench -- rocgdb -q -batch -x /tmp/gdbcmds --args python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > /tmp/gdb_$i.out 2> /tmp/gdb_$i.err
  rc=$?
  if grep -q "==== stopped" /tmp/gdb_$i.out && grep -qE "SIGABRT|SIGSEGV" /tmp/gdb_$i.out; then
    echo "pass $i: stopped on a signal (exit $rc)"
    { echo "# pass $i of $N: bench.py under rocprofv3 --kernel-trace under rocgdb"; grep -E "free\(\)|corrupt|malloc" /tmp/gdb_$i.err | head -5; sed -n '/Thread .* received signal/,$p' /tmp/gdb_$i.out | grep -v -A3 gomp_barrier_wait_end | head -400; } > $R/gpurun_out/abort_gdb.txt
    head -120 $R/gpurun_out/abort_gdb.txt
    exit 0
  fi
  echo "pass $i: exit $rc, no abort ($(grep -c 'metric' /tmp/gdb_$i.out) bench lines)"
done
echo "no abort in $N passes" | tee $R/gpurun_out/abort_gdb.txt
