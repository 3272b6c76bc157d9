#!/bin/bash
# Round profile: rocprofv3 kernel trace of bench.py itself + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) over the
# headline kernels.  Run on the GPU box from the repo root: bash tools/profile_round.sh <tag>; outputs under gpurun_out/prof_<tag>.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $R/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_algos -o algos -- python $R/tools/run_algos.py > $OUT/run_algos.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python $R/tools/run_kernels.py all 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python $R/tools/run_kernels.py all 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_sq -o p -- python $R/tools/run_kernels.py all 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_sq_algos -o p -- python $R/tools/run_algos.py > /dev/null 2>&1
cd $R
find $OUT -name "*_results.db" | sort
