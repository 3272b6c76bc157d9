#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_sdof.py tests/test_gpu_video_extruder.py tests/test_gpu_multi_rank.py tests/test_gpu_strips.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 120 tests/cpp/_build/device_lambda_test time 2>&1 | grep "box\|ok\|block" > $O/lambda.log; cat $O/lambda.log
timeout 200 python tools/time_flow.py > $O/time_flow.log 2>&1; tail -8 $O/time_flow.log
timeout 200 benchmarks/video_extruder_bench > $O/ve.log 2>&1; tail -1 $O/ve.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt_algos -o algos -- python $R/tools/run_algos.py > $O/run_algos.log 2>&1
cd $R
python tools/timeline.py $O/kt_algos/algos_results.db sdof_reset_kernel 24 > $O/timeline_sdof.md 2>&1; cat $O/timeline_sdof.md
rm -rf $O/kt_algos
