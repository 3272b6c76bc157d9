set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_gpu_core.py -x -q -k "batch or very_start" 2>&1 | tail -5
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_s20.json 2> gpurun_out/bench_s20.err; tail -c 300 gpurun_out/bench_s20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_s20.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "steps"): print(k, d[k])
r = d["roofline"]; print("roofline frac", r["frac"], "sustained", r["frac_sustained"], "avg_launch_us", r["avg_launch_us"], "sustained us", r["avg_launch_us_sustained"])
print("per-frame", d["box5x5_one_launch_per_frame"])
print("add4k", d["add4k"])
print("regions", d["config"]["timed_regions"])
for k in ("pyrlk",):
    p = d.get(k, {})
    print({kk: (vv if not isinstance(vv, dict) else "...") for kk, vv in p.items()})
    print("sdof", p.get("semi_dense_flow_4k")); print("ve", p.get("video_extruder_4k")); print("fast", p.get("fast9_4k"))
PY
timeout 300 python bench.py --no-cpu > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "steps"): print(k, d[k])
r = d["roofline"]; print("roofline frac", r["frac"], "sustained", r["frac_sustained"], "avg_launch_us", r["avg_launch_us"])
print("per-frame", d["box5x5_one_launch_per_frame"]); print("add4k", d["add4k"])
PY
