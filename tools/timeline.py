"""Per-dispatch timeline from a rocprofv3 --kernel-trace rocpd database: the last N dispatches whose kernel name matches a prefix list, with
start offsets, durations and the gaps between consecutive dispatches.  usage: timeline.py <results.db> <first-kernel-substring> [count]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
end = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = con.execute(f"select name, {start}, {end}, grid_x, workgroup_x from kernels order by {start}").fetchall()
key = sys.argv[2]; count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
idx = [i for i, r in enumerate(rows) if key in r[0]]
if not idx: sys.exit("no dispatch matches " + key)
i0 = idx[-1]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[: n.index("(")] if "(" in n else n
t0 = rows[i0][1]; prev_end = None
print("| # | kernel | start us | duration us | gap before us | grid | wg |\n|---|---|---|---|---|---|---|")
for k, r in enumerate(rows[i0:i0 + count]):
    gap = "" if prev_end is None else f"{(r[1] - prev_end) / 1e3:.2f}"
    print(f"| {k} | {short(r[0])[:70]} | {(r[1] - t0) / 1e3:.2f} | {(r[2] - r[1]) / 1e3:.2f} | {gap} | {r[3]} | {r[4]} |")
    prev_end = r[2]
