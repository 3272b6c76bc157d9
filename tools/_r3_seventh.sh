#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_sdof.py tests/test_gpu_video_extruder.py tests/test_gpu_multi_rank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 200 python tools/time_flow.py > $O/time_flow.log 2>&1; tail -10 $O/time_flow.log
timeout 200 benchmarks/video_extruder_bench > $O/ve.log 2>&1; tail -1 $O/ve.log | cut -c1-300; tail -1 $O/ve.log | grep -o '"per_update_ms.*'
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt_algos -o algos -- python $R/tools/run_algos.py > $O/run_algos.log 2>&1
cd $R
python tools/timeline.py $O/kt_algos/algos_results.db sdof_reset_kernel 20 > $O/timeline_sdof.md 2>&1; cat $O/timeline_sdof.md
rm -rf $O/kt_algos
