#!/bin/bash
# round 3, second GPU call: new tests (FAST raw rows, fused ingest + pyramid, prebuilt C++ programs), lambda / block_wise timings, PMC of the algorithm kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_algos.py tests/test_gpu_video_steps.py tests/test_cpp_api.py tests/test_gpu_video_extruder.py tests/test_gpu_reference_unit_tests.py tests/test_gpu_strips.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 120 tests/cpp/_build/device_lambda_test time > $O/lambda.log 2>&1; cat $O/lambda.log
timeout 200 python tools/time_ingest_pyr.py > $O/ingest_pyr.log 2>&1; cat $O/ingest_pyr.log
timeout 200 python tools/fast_time.py > $O/fast_time.log 2>&1; tail -8 $O/fast_time.log
cd /tmp && export TMPDIR=/tmp
pmc() { local d=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$d -o p -- python $R/tools/run_algos.py > $O/$d.log 2>&1 || echo "pass $d failed" >> $O/failed.txt; }
pmc pmc_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pmc pmc_busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
pmc pmc_wait SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
pmc pmc_wait2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA
cd $R
python tools/prof_summary.py $O/pmc_sq/p_results.db $O/pmc_busy/p_results.db $O/pmc_wait/p_results.db $O/pmc_wait2/p_results.db > $O/pmc_algos.md 2>&1
python tools/make_issue_json.py $O/issue.json $O/pmc_sq/p_results.db $O/pmc_busy/p_results.db $O/pmc_wait/p_results.db > /dev/null 2>&1
rm -rf $O/pmc_sq $O/pmc_busy $O/pmc_wait $O/pmc_wait2
