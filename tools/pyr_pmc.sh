#!/bin/bash
# issue / wait counters of the packed pyramid kernel (one 4K pyramid per launch): rocprofv3 --pmc passes of tools/run_pyr_once.py, per tuning preset
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pyr_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  tag=$(echo "$t" | tr -c 'a-zA-Z0-9\n' '_')
  i=0
  for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    VPP_TUNE="$t" timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c -d $OUT/${tag}_$i -o p --output-format csv -- python $R/tools/run_pyr_once.py > $OUT/${tag}_$i.log 2>&1 || echo "$t pass $i failed"
  done
  python - "$OUT" "$tag" "$t" <<'PY'
import csv, glob, sys, collections
out, tag, t = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pyramid_swar3" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"== {t}")
for k in sorted(acc): print(f"  {k}: {sum(acc[k]) / len(acc[k]):.0f}  (n={len(acc[k])})")
PY
done
