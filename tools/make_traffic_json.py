"""profiles/<round>_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes).
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read (MI355X_MICROARCH.md §HBM):
it is doubled here.  usage: make_traffic_json.py <fetch.db> <write.db> <out.json>"""
import json, sqlite3, sys
from collections import defaultdict


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    acc = defaultdict(list)
    for name, c, v in con.execute("select k.name, p.counter_name, p.value from counters_collection p join kernels k on p.dispatch_id = k.dispatch_id"):
        if c == counter:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            acc[n[: n.index("(")] if "(" in n else n].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "at::native" in k or "rocclr" in k:
        continue
    f, w = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
    out[k] = {"fetch_bytes_corrected_x2": f, "write_bytes": w, "hbm_bytes_per_launch": f + w}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
