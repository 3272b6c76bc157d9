"""profiles/<round>_issue.json from the SQ passes of tools/profile_round.sh: per kernel symbol, the share of the chip's SIMD issue cycles
spent on VALU / LDS instructions while the kernel runs.
  SQ_ACTIVE_INST_VALU / _LDS count quad-cycles summed over all SIMDs (MI355X_MICROARCH.md: x4 = cycles); SQ_BUSY_CYCLES is summed over
  its 32 counter instances (one per shader engine half), so SQ_BUSY_CYCLES / 32 = the kernel's duration in cycles.
  valu_issue_frac = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (SQ_BUSY_CYCLES / 32)
usage: make_issue_json.py <out.json> <results.db> [...]   (any number of pass databases; counters are taken where they are found)"""
import json, sqlite3, sys
from collections import defaultdict

SIMDS, BUSY_INSTANCES = 1024, 32


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[: n.index("(")] if "(" in n else n


acc = defaultdict(lambda: defaultdict(list))
for db in sys.argv[2:]:
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select k.name, p.counter_name, p.value from counters_collection p join kernels k on p.dispatch_id = k.dispatch_id").fetchall()
    except Exception:  # noqa: BLE001
        continue
    for name, c, v in rows:
        acc[short(name)][c].append(v)
out = {}
for k, cs in sorted(acc.items()):
    if "at::native" in k or "rocclr" in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "SQ_ACTIVE_INST_VALU" not in m or "SQ_BUSY_CYCLES" not in m or m["SQ_BUSY_CYCLES"] <= 0:
        continue
    cyc = m["SQ_BUSY_CYCLES"] / BUSY_INSTANCES
    e = {"kernel_cycles": cyc, "active_inst_valu_quad_cycles": m["SQ_ACTIVE_INST_VALU"], "valu_issue_frac": m["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cyc}
    if "SQ_ACTIVE_INST_LDS" in m:
        e["active_inst_lds_quad_cycles"] = m["SQ_ACTIVE_INST_LDS"]; e["lds_issue_frac"] = m["SQ_ACTIVE_INST_LDS"] * 4 / SIMDS / cyc
    if "SQ_INSTS_VALU" in m:
        e["insts_valu"] = m["SQ_INSTS_VALU"]
    if "SQ_WAVES" in m:
        e["waves"] = m["SQ_WAVES"]
    out[k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
