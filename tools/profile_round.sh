#!/bin/bash
# Round profile: rocprofv3 kernel trace of bench.py itself (default flags and the driver's --steps 20 --warmup 5) + separate PMC
# passes over the headline kernels (tools/run_kernels.py) and the algorithm kernels (tools/run_algos.py): HBM traffic
# (FETCH_SIZE / WRITE_SIZE), SQ instruction counts, SQ busy / wait cycles, and the L2 (TCC) request mix.
# Run on the GPU box from the repo root: bash tools/profile_round.sh <tag>; outputs under gpurun_out/prof_<tag>.
# Every rocprofv3 run is its own process with --kernel-trace + --pmc only (no other trace domains) and a timeout.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout -k 10 ${PASS_LIMIT:-150}"   # -k: round 5 lost 40 GPU-minutes to a profiled process that ignored TERM after the profiler's own worker thread had aborted
# (the default-flag run — 500 steps, 32 000 recorded per-frame calls — aborted inside rocprofv3's worker thread in round 5 ("corrupted size vs. prev_size"); the driver's command is the one profiled)
# (bench.py under rocprofv3 died twice in round 5 — a glibc heap-check abort in a non-main thread within the first seconds, 2 of 9 runs, never without the profiler
# and never under MALLOC_CHECK_=3: up to three attempts; tools/bench_under_rocprof_debug.sh is the script that chases it)
for attempt in 1 2 3; do
  rm -rf $OUT/kt20
  $T rocprofv3 --kernel-trace --stats -d $OUT/kt20 -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $OUT/bench_s20_under_rocprof.json 2> $OUT/bench_s20_under_rocprof.err
  [ -s $OUT/bench_s20_under_rocprof.json ] && break
  echo "bench under rocprofv3: attempt $attempt failed" >> $OUT/failed.txt
done
$T rocprofv3 --kernel-trace --stats -d $OUT/kt_algos -o algos -- python $R/tools/run_algos.py > $OUT/run_algos.log 2>&1
pmc() {  # pmc <dir> <script> <counters...>
  local d=$1 s=$2; shift 2
  $T rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o p -- python $R/tools/${s%% *} ${s#* } > $OUT/$d.log 2>&1 || echo "pass $d failed / timed out" >> $OUT/failed.txt
}
pmc pmc_fetch "run_kernels.py all 16" FETCH_SIZE
pmc pmc_write "run_kernels.py all 16" WRITE_SIZE
pmc pmc_sq "run_kernels.py all 16" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc pmc_busy "run_kernels.py all 16" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
pmc pmc_wait "run_kernels.py all 16" SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
[ -n "${WAIT2:-}" ] && pmc pmc_wait2 "run_kernels.py all 16" SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pmc pmc_sq_algos "run_algos.py x" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pmc pmc_busy_algos "run_algos.py x" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
pmc pmc_wait_algos "run_algos.py x" SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
[ -n "${WAIT2:-}" ] && pmc pmc_wait2_algos "run_algos.py x" SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
# L2 request mix, one counter per pass (a multi-counter TCC pass hung the profiler in round 1)
for c in ${TCC_PASSES:-}; do   # e.g. TCC_PASSES="TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" (off by default: the box kernel has not changed since r04_tcc.md)
  pmc pmc_tcc_$c "run_kernels.py box 16" $c
done
cd $R
find $OUT -name "*_results.db" | sort
cat $OUT/failed.txt 2>/dev/null
# summarise on the box (the rocpd databases are too large to travel back), then drop the raw outputs
PUB=$R/gpurun_out/pub_$TAG
mkdir -p $PUB
bash tools/publish_profiles.sh $TAG $PUB
rm -rf $OUT/*/ 
# per-dispatch timelines of one 4K flow pair and of the tracker's updates, and the fused sweeps' own event log
{ echo "# Round ${TAG#r} — per-dispatch timeline of one 4K frame pair of vpp_semi_dense_optical_flow (tools/flow_timeline.sh: rocprofv3 --kernel-trace of tools/rounds_ab.py, the 10th call; durations include the dispatch's ramp)"; echo
  bash tools/flow_timeline.sh 2>/dev/null | grep "^|"; echo; echo "## The fused sweeps' event log of the same scene (tools/sweep_log.py, tuning sdof.stats = 1: 100 MHz wall clock, no profiler)"; echo; echo '```'
  timeout 120 python tools/sweep_log.py 2>/dev/null | grep " us "; echo '```'; } > $PUB/${TAG}_flow_timeline.md
{ echo "# Round ${TAG#r} — per-dispatch timelines of the 4K tracker (benchmarks/video_extruder_bench under rocprofv3 --kernel-trace, tools/tracker_timeline.sh)"; echo
  echo "## A steady update (no re-detection)"; echo; bash tools/tracker_timeline.sh 2>/dev/null | grep "^|"; echo
  echo "## From a re-detection frame's ve_finish_kernel to the end of the next update's flow"; echo; DETECT=1 bash tools/tracker_timeline.sh 2>/dev/null | grep "^|"; } > $PUB/${TAG}_tracker_timeline.md
rm -rf $R/gpurun_out/flow_tl $R/gpurun_out/ve_tl
