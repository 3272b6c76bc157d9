#!/bin/bash
# Summarise gpurun_out/prof_<tag> (written by tools/profile_round.sh, same call, on the GPU box) into <dest> (default profiles/);
# the summaries are then copied into the tracked profiles/ directory.
set -u
TAG=${1:-r05}
D=${2:-profiles}
P=gpurun_out/prof_$TAG
{ echo "# Round ${TAG#r} — rocprofv3 --kernel-trace --stats of the driver's \`python bench.py --gpus 1 --steps 20 --warmup 5\` (--no-cpu) and of tools/run_algos.py"; echo
  echo "One row per (kernel symbol, grid): the 64-frame launches of the timed regions are the rows with the largest grids."; echo
  echo "Produced by tools/profile_round.sh on one MI355X; summarised from the rocpd sqlite outputs by tools/prof_summary.py."
  echo "bench.py's own JSON line from the same (profiled) run: ${TAG}_bench_s20_under_rocprof.json — roofline.avg_launch_us"
  echo "there is bench.py's event-timed sample; the trace averages below also contain the warm-up, preheat and graph-upload launches."; echo
  python tools/prof_summary.py $P/kt20/bench_results.db $P/kt_algos/algos_results.db; } > $D/${TAG}_bench_kernel_trace.md
{ echo "# Round ${TAG#r} — PMC passes (separate runs, rocprofv3 --kernel-trace --pmc …) over tools/run_kernels.py / tools/run_algos.py"; echo
  echo "FETCH_SIZE / WRITE_SIZE are KB per dispatch; FETCH_SIZE under-reports wide reads by 2x on gfx950 (MI355X_MICROARCH.md), corrected in ${TAG}_traffic.json."
  echo "SQ_* cycle counters are summed over all SIMDs / CUs of the chip (1024 SIMDs): divide SQ_ACTIVE_INST_VALU by SQ_INSTS_VALU for the issue cycles per wave64 VALU instruction."; echo
  python tools/prof_summary.py $P/pmc_fetch/p_results.db $P/pmc_write/p_results.db $P/pmc_sq/p_results.db $P/pmc_busy/p_results.db $P/pmc_wait/p_results.db $P/pmc_wait2/p_results.db \
     $P/pmc_sq_algos/p_results.db $P/pmc_busy_algos/p_results.db $P/pmc_wait_algos/p_results.db $P/pmc_wait2_algos/p_results.db; } > $D/${TAG}_pmc.md
ls $P/pmc_tcc_*/p_results.db > /dev/null 2>&1 && { echo "# Round ${TAG#r} — L2 (TCC) request mix of the 4K vuchar3 box5x5 kernel, one counter per rocprofv3 pass (tools/run_kernels.py box 16)"; echo
  python tools/prof_summary.py $P/pmc_tcc_*/p_results.db; } > $D/${TAG}_tcc.md
tail -1 $P/bench_s20_under_rocprof.json > $D/${TAG}_bench_s20_under_rocprof.json
python tools/make_traffic_json.py $P/pmc_fetch/p_results.db $P/pmc_write/p_results.db $D/${TAG}_traffic.json > /dev/null
python tools/make_issue_json.py $D/${TAG}_issue.json $P/pmc_busy/p_results.db $P/pmc_wait/p_results.db $P/pmc_sq/p_results.db $P/pmc_busy_algos/p_results.db $P/pmc_wait_algos/p_results.db $P/pmc_sq_algos/p_results.db > /dev/null
echo $TAG > $D/CURRENT   # the round whose summaries describe this tree: what bench.py's static roofline quotes read (bench_pyrlk.py: profile_round)
