#!/bin/bash
# kernel trace of the driver's own bench command (K = 20): per-launch durations of the box kernel in launch order
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kt_s20
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --regions 3 > $OUT/bench.json 2> $OUT/bench.err
cd $R
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
box = [r for r in rows if 'box_u8_wide' in r['Kernel_Name']]
d = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in box]
d.sort()
print('box launches', len(d))
tail = d[-60:]
print('last 60 durations us:', [round((e - s) / 1e3, 2) for s, e in tail])
print('gaps us:', [round((tail[i + 1][0] - tail[i][1]) / 1e3, 2) for i in range(len(tail) - 1)])
PY
