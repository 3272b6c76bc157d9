"""GPU tuning sweep for K1 (add) and K2 (box5x5): prints avg kernel time per variant (hipGraph of 200 launches,
rotating buffers > 256 MiB)."""
import os
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi

lib = capi.lib(); capi.check(lib.vpp_init(0))
NR, NC = 2160, 3840
npx = NR * NC


def time_graph(launch, steps=200):
    for i in range(10):
        launch(i, capi.stream_ptr())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = capi.stream_ptr()
        for i in range(steps):
            launch(i, cs)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    return best  # us


res = {}
src_h = rand_image(NR, NC, vi.U8, 3, border=2, seed=3, align=16)
nsets = 8
srcs = [DeviceImage.from_host(src_h) for _ in range(nsets)]
dsts = [DeviceImage(NR, NC, vi.U8, 3, 0, 16) for _ in range(nsets)]
sd, dd = [s.desc for s in srcs], [d.desc for d in dsts]
for impl, rows_list in ((1, (1, 2, 4, 8)), (0, (8, 16))):
    lib.vpp_set_tuning(b"box.impl", impl)
    for rows in rows_list:
        for nt in (0, 1):
            lib.vpp_set_tuning(b"box.nt", nt); lib.vpp_set_tuning(b"box.rows", rows)
            us = time_graph(lambda i, s: lib.vpp_box_filter(P(dd[i % nsets]), P(sd[i % nsets]), 5, 5, s))
            res[f"box impl={impl} rows={rows} nt={nt}"] = (us, 6 * npx / us / 1e3)
lib.vpp_set_tuning(b"box.impl", -1)
lib.vpp_set_tuning(b"box.rows", -1); lib.vpp_set_tuning(b"box.nt", -1)
lib.vpp_set_tuning(b"box.force_generic", 1)
us = time_graph(lambda i, s: lib.vpp_box_filter(P(dd[i % nsets]), P(sd[i % nsets]), 5, 5, s), 50)
res["box generic-LDS"] = (us, 6 * npx / us / 1e3)
lib.vpp_set_tuning(b"box.force_generic", 0)
# same-buffer (MALL-resident) variant for reference
us = time_graph(lambda i, s: lib.vpp_box_filter(P(dd[0]), P(sd[0]), 5, 5, s))
res["box default, single buffer (MALL)"] = (us, 6 * npx / us / 1e3)
del srcs, dsts

nadd = 4
b_h = rand_image(NR, NC, vi.I32, seed=2, lo=0, hi=2**30 - 1)
A = [DeviceImage(NR, NC, vi.I32) for _ in range(nadd)]
B = [DeviceImage.from_host(b_h) for _ in range(nadd)]
C = [DeviceImage.from_host(b_h) for _ in range(nadd)]
ad, bd, cd = [x.desc for x in A], [x.desc for x in B], [x.desc for x in C]
for unroll in (1, 2, 4, 8):
    for nt in (0, 1):
        lib.vpp_set_tuning(b"add.unroll", unroll); lib.vpp_set_tuning(b"add.nt", nt)
        us = time_graph(lambda i, s: lib.vpp_pixelwise_binary(0, P(ad[i % nadd]), P(bd[i % nadd]), P(cd[i % nadd]), s))
        res[f"add unroll={unroll} nt={nt}"] = (us, 12 * npx / us / 1e3)
# torch reference points: copy (8 B/px... 2x33MB) and add
x, y, z = [torch.empty(NR * NC, dtype=torch.int32, device="cuda") for _ in range(3)]
us = time_graph(lambda i, s: torch.add(x, y, out=z), 100)
res["torch.add int32 (same buffers)"] = (us, 12 * npx / us / 1e3)
for k, (us, gbs) in res.items():
    print(f"{k:40s} {us:9.2f} us  {gbs:9.1f} GB/s  {gbs/80:.1f}% of 8 TB/s")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tune_core.json"), "w"), indent=1)
