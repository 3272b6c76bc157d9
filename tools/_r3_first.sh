#!/bin/bash
# round 3, first GPU call: gpu tests, bench at the driver's flags and defaults, kernel traces
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench s20 rc=$?"
timeout 400 python bench.py --no-cpu > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt20 -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/bench_s20_under_rocprof.json 2> $O/bench_s20_under_rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_algos -o algos -- python $R/tools/run_algos.py > $O/run_algos.log 2>&1
cd $R
python tools/prof_summary.py $O/kt20/bench_results.db $O/kt_algos/algos_results.db > $O/kernel_trace.md 2>&1
rm -rf $O/kt20 $O/kt_algos
tail -c 600 $O/bench_s20.json
