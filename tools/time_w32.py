"""4K 32-bit box 5x5 (the reference benchmark's own element type) and frame ingest: hipGraph timings over rotating buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
lib = capi.lib(); capi.check(lib.vpp_init(0))
def time_graph(launch, steps=300):
    for i in range(10): launch(i, capi.stream_ptr())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = capi.stream_ptr()
        for i in range(steps): launch(i, cs)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    return best
NR, NC, ns = 2160, 3840, 6
for dtype, name in ((vi.I32, "int"), (vi.F32, "float")):
    src_h = rand_image(NR, NC, dtype, 1, border=2, seed=3, align=16, lo=0 if dtype == vi.I32 else None, hi=999 if dtype == vi.I32 else None)
    srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]; dsts = [DeviceImage(NR, NC, dtype, 1, 0, 16) for _ in range(ns)]
    sd, dd = [s.desc for s in srcs], [d.desc for d in dsts]
    for rows in (2, 4):
        for wpb in (4, 2, 1):
            lib.vpp_set_tuning(b"box.rows32", rows); lib.vpp_set_tuning(b"box.waves_per_block", wpb)
            us = time_graph(lambda i, s: lib.vpp_box_filter(P(dd[i % ns]), P(sd[i % ns]), 5, 5, s))
            print(f"box 5x5 {name} 4K rows {rows} waves/block {wpb}: {us:6.2f} us  {8 * NR * NC / us / 1e6:5.2f} TB/s  frac {8 * NR * NC / us / 1e6 / 8:.3f}")
lib.vpp_set_tuning(b"box.rows32", -1); lib.vpp_set_tuning(b"box.waves_per_block", -1)
rgb_h = rand_image(NR, NC, vi.U8, 3, border=0, seed=6)
nin = 8
rgbs = [DeviceImage.from_host(rgb_h) for _ in range(nin)]; grays = [DeviceImage(NR, NC, vi.U8, 1, 3, 32) for _ in range(nin)]
us = time_graph(lambda i, s: lib.vpp_rgb_to_graylevel(P(grays[i % nin].desc), P(rgbs[i % nin].desc), 1, s))
print(f"ingest vuchar3 -> gray + border 3: {us:.2f} us  frac {4 * NR * NC / us / 1e6 / 8:.3f}")
# size scaling of the ingest kernel (fixed launch cost vs stream rate)
for (nr, nc, nb) in ((1080, 1920, 16), (4320, 7680, 3)):
    h = rand_image(nr, nc, vi.U8, 3, border=0, seed=6)
    rg = [DeviceImage.from_host(h) for _ in range(nb)]; gr = [DeviceImage(nr, nc, vi.U8, 1, 3, 32) for _ in range(nb)]
    us = time_graph(lambda i, s: lib.vpp_rgb_to_graylevel(P(gr[i % nb].desc), P(rg[i % nb].desc), 1, s), steps=100)
    print(f"ingest {nc}x{nr}: {us:.2f} us  frac {4 * nr * nc / us / 1e6 / 8:.3f}")
