"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel duration stats (kernel trace) and per-kernel mean PMC values.
usage: prof_summary.py <results.db> [...]   -> markdown tables on stdout"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[: n.index("(")] if "(" in n else n


import os
for path in sys.argv[1:]:
    if not os.path.exists(path):
        continue   # a pass that was not taken this round
    con = sqlite3.connect(path)
    rows = con.execute("select name, duration, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels").fetchall()
    # one row per (kernel symbol, grid size): launches of one symbol that carry different amounts of work (1 ... 64 frames per launch, the sweeps of the three
    # scales) must not be averaged together — the row of the timed instance is then comparable with bench.py's roofline.avg_launch_us
    agg = defaultdict(list)
    meta = {}
    for name, dur, vg, sg, lds, gx, wx in rows:
        agg[(short(name), gx)].append(dur); meta[(short(name), gx)] = (vg, sg, lds, gx, wx)
    print(f"### {path}\n\n| kernel | grid (threads) | wg | calls | avg us | min us | max us | vgpr | lds B |\n|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if "at::native" in k[0]: continue
        m = meta[k]
        print(f"| {k[0]} | {m[3]} | {m[4]} | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {m[0]} | {m[2]} |")
    try:
        pm = con.execute("select k.name, p.counter_name, p.value from counters_collection p join kernels k on p.dispatch_id = k.dispatch_id").fetchall()
    except Exception as e:  # noqa
        try:
            cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
            print("counters_collection columns:", cols)
        except Exception as e2:  # noqa
            print("no counters:", e2)
        pm = []
    if pm:
        acc = defaultdict(lambda: defaultdict(list))
        for name, c, v in pm:
            acc[short(name)][c].append(v)
        print("\n| kernel | counter | mean per dispatch |\n|---|---|---|")
        for k, cs in acc.items():
            if "at::native" in k: continue
            for c, v in sorted(cs.items()):
                print(f"| {k} | {c} | {sum(v)/len(v):.4g} |")
    print()
