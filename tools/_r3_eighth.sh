#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_algos.py tests/test_gpu_video_steps.py tests/test_gpu_sdof.py tests/test_gpu_video_extruder.py tests/test_gpu_multi_rank.py tests/test_golden.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python tools/time_ingest_pyr.py > $O/ingest_pyr.log 2>&1; cat $O/ingest_pyr.log
timeout 200 python tools/rounds_ab.py > $O/flow.log 2>&1; tail -2 $O/flow.log
timeout 200 benchmarks/video_extruder_bench > $O/ve.log 2>&1; tail -1 $O/ve.log | grep -o '"ms_per_update_median_steady[^,]*'; tail -1 $O/ve.log | grep -o '"per_update_ms.*'
