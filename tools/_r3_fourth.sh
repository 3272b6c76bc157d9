#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 120 tests/cpp/_build/device_lambda_test time 2>&1 | grep "box\|ok\|block" > $O/lambda.log; cat $O/lambda.log
timeout 200 python tools/fast_time.py > $O/fast_time.log 2>&1; tail -6 $O/fast_time.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt_algos -o algos -- python $R/tools/run_algos.py > $O/run_algos.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/kt_ve -o ve -- $R/benchmarks/video_extruder_bench > $O/ve.log 2>&1
cd $R
python tools/timeline.py $O/kt_algos/algos_results.db sdof_reset_kernel 40 > $O/timeline_sdof.md 2>&1
python tools/timeline.py $O/kt_ve/ve_results.db sdof_reset_kernel 70 > $O/timeline_ve.md 2>&1
tail -3 $O/ve.log
rm -rf $O/kt_algos $O/kt_ve
