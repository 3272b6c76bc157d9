#!/bin/bash
# per-dispatch timeline of one steady (no re-detection) video_extruder_update at 4K from a rocprofv3 kernel trace of benchmarks/video_extruder_bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ve_tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- $R/benchmarks/video_extruder_bench 2160 3840 14 > $OUT/run.log 2>&1
cd $R
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fin = [i for i, r in enumerate(rows) if 've_finish_kernel<true>' in r['Kernel_Name'] or 've_finish_kernel<(bool)1>' in r['Kernel_Name']]
import os
if os.environ.get("DETECT"):
    ff = [i for i, r in enumerate(rows) if "ve_finish_kernel<false>" in r["Kernel_Name"]]
    a = ff[1]; b = [i for i in fin if i > a][0] - 1   # from a re-detection frame's finish kernel to the end of the next steady update's flow
else:
    a, b = fin[1] + 1, fin[2]   # the dispatches of the third steady update
t0 = int(rows[a]['Start_Timestamp']); prev = None
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return (n[:n.index('(')] if '(' in n else n)[:56]
print('| # | kernel | start us | dur us | gap us | grid |\n|---|---|---|---|---|---|')
for k, r in enumerate(rows[a:b + 1]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = '' if prev is None else f'{(s - prev) / 1e3:.1f}'
    print(f"| {k} | {short(r['Kernel_Name'])} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap} | {r.get('Grid_Size_X', r.get('Grid_Size', ''))} |")
    prev = e
PY
