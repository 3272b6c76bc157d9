timeout 900 python -m pytest tests/test_gpu_algos.py tests/test_gpu_strips.py tests/test_gpu_edges.py tests/test_golden.py tests/test_gpu_video_extruder.py -x -q 2>&1 | tail -8
timeout 120 python tools/fast_time.py 2>&1 | tail -12
