#!/bin/bash
# round 6, third GPU pass: FAST-9 RAW in one launch (parity + timing A/B), pyrLK batch rate, the timed-graph repro, which bench leg aborts under the profiler
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_algos.py tests/test_gpu_edges.py tests/test_gpu_video_steps.py tests/test_gpu_video_extruder.py -m gpu -x -q -k "fast or extruder or detect or keypoint" > gpurun_out/gputests3.log 2>&1; echo "gpu tests exit $?"; tail -5 gpurun_out/gputests3.log
echo "== fast9 timing, fused write on (default)"; timeout 120 python tools/fast_time.py 2>&1 | grep -v amdgpu.ids
echo "== fast9 timing, fused write off"; timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, runpy
sys.path.insert(0, ".")
from vpp_amd import capi
capi.lib().vpp_set_tuning(b"fast9.raw_fused", 0)
runpy.run_path("tools/fast_time.py", run_name="__main__")
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_r06_c.json 2> gpurun_out/bench_r06_c.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail_n1.json"))
fb = d.get("pyrlk", {}).get("frame_pair_batches", {})
print({k: v for k, v in fb.items() if k != "how"})
print("fast9", json.dumps(d.get("pyrlk", {}).get("fast9_4k", {}))[:600])
PY
MODES=timed bash tools/capture_repro.sh 6
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 ${PASSES:-10}); do
  rm -rf $R/gpurun_out/kt_rp
  VPP_BENCH_FAULTHANDLER=1 timeout -k 10 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_rp -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $R/gpurun_out/bt_$i.json 2> $R/gpurun_out/bt_$i.err
  rc=$?; echo "profiled bench pass $i with faulthandler: exit $rc"
  if [ $rc -ne 0 ]; then grep -v "incomplete dispatches" $R/gpurun_out/bt_$i.err | grep -A40 -m1 -E "Fatal Python|free\(\)|corrupt|malloc" | head -80; cp $R/gpurun_out/bt_$i.err $R/gpurun_out/abort_faulthandler.txt; break; fi
  rm -f $R/gpurun_out/bt_$i.err $R/gpurun_out/bt_$i.json
done
rm -rf $R/gpurun_out/kt_rp
