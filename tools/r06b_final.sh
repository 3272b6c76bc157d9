#!/bin/bash
# round 6, second session: the driver's bench command, the 2-rank dry run on one GPU, the round's profiles on the final kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06_s20.json 2> gpurun_out/bench_r06_s20.err; echo "bench exit $?"; wc -c gpurun_out/bench_r06_s20.json
VPP_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_r06_n2_one_gpu_dry_run.json 2> gpurun_out/bench_r06_n2.err; echo "2-rank dry run exit $?"; tail -c 600 gpurun_out/bench_r06_n2_one_gpu_dry_run.json
PASS_LIMIT=200 timeout 2400 bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1; echo "profile_round exit $?"; tail -5 gpurun_out/profile_round_r06.log; ls gpurun_out/pub_r06 2>/dev/null
