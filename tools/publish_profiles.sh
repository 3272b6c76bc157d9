#!/bin/bash
# Summarise gpurun_out/prof_<tag> (written by tools/profile_round.sh on the GPU box) into the tracked profiles/ directory.
set -u
TAG=${1:-r01}
P=gpurun_out/prof_$TAG
{ echo "# Round ${TAG#r} — rocprofv3 --kernel-trace --stats of \`python bench.py\` (defaults: 2000 steps, 200 warm-up) and of tools/run_algos.py"; echo
  echo "Produced by tools/profile_round.sh on one MI355X; summarised from the rocpd sqlite outputs by tools/prof_summary.py."
  echo "bench.py's own JSON line from the same (profiled) run: profiles/${TAG}_bench_under_rocprof.json — its roofline.avg_launch_us is the"
  echo "timed region only; the trace average below also contains the warm-up launches and the untimed graph-upload replay."; echo
  python tools/prof_summary.py $P/kt/bench_results.db $P/kt_algos/algos_results.db; } > profiles/${TAG}_bench_kernel_trace.md
{ echo "# Round ${TAG#r} — PMC passes (separate runs, rocprofv3 --kernel-trace --pmc …) over tools/run_kernels.py / tools/run_algos.py"; echo
  echo "FETCH_SIZE / WRITE_SIZE are KB per dispatch; FETCH_SIZE under-reports wide reads by 2x on gfx950 (MI355X_MICROARCH.md), corrected in ${TAG}_traffic.json."; echo
  python tools/prof_summary.py $P/pmc_fetch/p_results.db $P/pmc_write/p_results.db $P/pmc_sq/p_results.db $P/pmc_sq_algos/p_results.db; } > profiles/${TAG}_pmc.md
tail -1 $P/bench_under_rocprof.json > profiles/${TAG}_bench_under_rocprof.json
python tools/make_traffic_json.py $P/pmc_fetch/p_results.db $P/pmc_write/p_results.db profiles/${TAG}_traffic.json > /dev/null
