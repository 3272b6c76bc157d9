set -u
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.json 2> gpurun_out/bench_s20.err; tail -c 400 gpurun_out/bench_s20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_s20.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "steps"): print(k, d[k])
r = d["roofline"]; print("roofline frac", r["frac"], "sustained", r["frac_sustained"], "avg_launch_us", r["avg_launch_us"], "traffic", r["traffic"])
print("per-frame", d["box5x5_one_launch_per_frame"])
print("add4k", d["add4k"])
print("cpu", d["cpu_baseline"])
PY
