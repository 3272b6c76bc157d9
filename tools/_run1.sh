set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_sdof.py tests/test_gpu_video_extruder.py tests/test_gpu_video_steps.py -x -q 2>&1 | tail -8
mkdir -p $R/gpurun_out/kt_algos_r3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_algos_r3 -o algos -- python $R/tools/run_algos.py > $R/gpurun_out/kt_algos_r3/run_algos.log 2>&1
cd $R
tail -12 gpurun_out/kt_algos_r3/run_algos.log
python tools/prof_summary.py $(find gpurun_out/kt_algos_r3 -name "*_results.db" | head -1) > gpurun_out/kt_algos_r3/summary.md 2>&1
grep -i "sdof\|pyramid\|kernel |" gpurun_out/kt_algos_r3/summary.md | head -30
rm -f $(find gpurun_out/kt_algos_r3 -name "*_results.db")
