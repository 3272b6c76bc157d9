#!/bin/bash
# round 6, first GPU pass: the GPU suite, the HIP-only capture repro, six profiled bench passes (round 5: 4 of 6 aborted), one clean bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests exit $?"; tail -3 gpurun_out/gputests.log
bash tools/capture_repro.sh 6
cd /tmp && export TMPDIR=/tmp
ok=0
for i in 1 2 3 4 5 6; do
  rm -rf $R/gpurun_out/kt_rp
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_rp -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $R/gpurun_out/bench_rp_$i.json 2> $R/gpurun_out/bench_rp_$i.err
  rc=$?; echo "bench under rocprofv3 pass $i: exit $rc, $(wc -c < $R/gpurun_out/bench_rp_$i.json) bytes of json, $(grep -c -iE 'free\(\)|corrupt|abort' $R/gpurun_out/bench_rp_$i.err) abort lines"
  [ $rc -eq 0 ] && ok=$((ok+1))
done
echo "bench under rocprofv3 --kernel-trace: $ok of 6 passes completed" | tee $R/gpurun_out/bench_rp_summary.txt
rm -rf $R/gpurun_out/kt_rp
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06_a.json 2> gpurun_out/bench_r06_a.err; echo "bench exit $?"; cat gpurun_out/bench_r06_a.json | cut -c1-1500
