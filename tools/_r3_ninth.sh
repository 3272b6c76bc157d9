#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_cpp_api.py tests/test_gpu_video_extruder.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench rc=$?"
tail -c 1500 $O/bench_s20.json
