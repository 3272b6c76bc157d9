set -u
for sets in 8 16 32 64; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --sets $sets > gpurun_out/bench_sets$sets.json 2> gpurun_out/bench_sets$sets.err
python - $sets <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/bench_sets{sys.argv[1]}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("sets", sys.argv[1], "value", round(d["value"], 1), "frac", round(r["frac"], 4), "sustained", round(r["frac_sustained"], 4), "us/launch", round(r["avg_launch_us"], 2),
      "| per-frame frac", round(d["box5x5_one_launch_per_frame"]["frac_in_region"], 4), round(d["box5x5_one_launch_per_frame"]["frac_sustained"], 4),
      "| add frac", round(d["add4k"]["roofline"]["frac"], 4), round(d["add4k"]["roofline"]["frac_sustained"], 4), "add Gpx/s", round(d["add4k"]["gpixels_per_s"], 1))
PY
done
