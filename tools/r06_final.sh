#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_final.log 2>&1; echo "gpu tests exit $?"; grep -E "passed|failed" gpurun_out/gputests_final.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench exit $?"; wc -c gpurun_out/bench_final.json; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 1), "frac", r["frac"], "traffic_source", r.get("traffic_source"), "add", r["add4k"]["frac"], "per_frame", r["per_frame_call"]["frac"], "checked", d["checked"])
print("lambda", r.get("lambda_call", {}).get("int_5x5"))
print("legs", json.dumps(r["legs"])[:900])
PY
