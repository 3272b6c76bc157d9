#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_algos.py tests/test_gpu_video_steps.py tests/test_cpp_api.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 120 tests/cpp/_build/device_lambda_test time 2>&1 | grep "box\|ok\|block" > $O/lambda.log; cat $O/lambda.log
timeout 300 python tools/tune_pyrlk.py > $O/tune_pyrlk.log 2>&1; tail -12 $O/tune_pyrlk.log
timeout 300 python tools/time_ingest_pyr.py > $O/ingest_pyr.log 2>&1; cat $O/ingest_pyr.log
