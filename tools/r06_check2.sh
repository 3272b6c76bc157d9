#!/bin/bash
# round 6, second GPU pass: pyrLK frame-pair batches (parity + rate), the C++ programs, and the aborting thread's backtrace of a profiled bench pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_algos.py tests/test_gpu_multi_rank.py tests/test_cpp_api.py tests/test_gpu_core.py -m gpu -x -q -k "pyrlk or cpp or frame_pair or device or held_back or capture or recorded" > gpurun_out/gputests2.log 2>&1; echo "gpu tests exit $?"; tail -5 gpurun_out/gputests2.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_r06_b.json 2> gpurun_out/bench_r06_b.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail_n1.json"))
print(json.dumps(d.get("pyrlk", {}).get("frame_pair_batches"), indent=0))
print({k: d.get("pyrlk", {}).get("sweep", {}).get(k) for k in ("1250", "10000")})
PY
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 ${PASSES:-12}); do
  rm -rf $R/gpurun_out/kt_rp
  timeout -k 10 120 rocprofv3 --preload $R/tools/libabort_bt.so --disable-signal-handlers true --kernel-trace --stats -d $R/gpurun_out/kt_rp -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $R/gpurun_out/bt_$i.json 2> $R/gpurun_out/bt_$i.err
  rc=$?; echo "profiled bench pass $i with abort_bt: exit $rc"
  if [ $rc -ne 0 ]; then grep -E "abort_bt|free\(\)|corrupt|malloc" $R/gpurun_out/bt_$i.err | head -60; cp $R/gpurun_out/bt_$i.err $R/gpurun_out/abort_backtrace.txt; break; fi
  rm -f $R/gpurun_out/bt_$i.err $R/gpurun_out/bt_$i.json
done
rm -rf $R/gpurun_out/kt_rp
