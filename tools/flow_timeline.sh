#!/bin/bash
# per-dispatch timeline of one 4K frame pair of vpp_semi_dense_optical_flow (the last of a few calls) from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/flow_tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/tools/rounds_ab.py > $OUT/run.log 2>&1
cd $R
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if '__amd_rocclr' not in r['Kernel_Name'] and 'at::native' not in r['Kernel_Name']]   # (the harness's own tensor copies between the calls)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# rounds_ab runs winsize 9 (11 calls) then winsize 7; take the 10th call of winsize 9: a call starts with the pyramid pair's launch (the read-back rides in the last sweep's
# launch since round 6, so it no longer ends a call in the trace)
py = [i for i, r in enumerate(rows) if 'pyramid_swar3_pair' in r['Kernel_Name']]
start = py[9]; end = py[10] - 1
t0 = int(rows[start]['Start_Timestamp']); prev = None
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return (n[:n.index('(')] if '(' in n else n)[:48]
print('| # | kernel | start us | dur us | gap us | grid |\n|---|---|---|---|---|---|')
for k, r in enumerate(rows[start:end + 1]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = '' if prev is None else f'{(s - prev) / 1e3:.1f}'
    print(f"| {k} | {short(r['Kernel_Name'])} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap} | {r.get('Grid_Size_X', r.get('Grid_Size', ''))} |")
    prev = e
PY
tail -3 $OUT/run.log
