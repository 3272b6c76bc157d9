#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_algos.py tests/test_gpu_video_steps.py tests/test_cpp_api.py tests/test_gpu_reference_unit_tests.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
for n in 0 1 4; do echo "VPP_PW_NBH_NPX=$n"; VPP_PW_NBH_NPX=$n timeout 120 tests/cpp/_build/device_lambda_test time 2>&1 | grep "box\|ok"; done > $O/lambda.log 2>&1; cat $O/lambda.log
timeout 300 python tools/time_ingest_pyr.py > $O/ingest_pyr.log 2>&1; cat $O/ingest_pyr.log
timeout 200 python tools/fast_time.py > $O/fast_time.log 2>&1; tail -6 $O/fast_time.log
