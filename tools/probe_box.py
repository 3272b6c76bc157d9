"""Calibration probes for K2: the streaming kernel with its arithmetic result discarded (same loads / stores) and plain
device copies of the same byte counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
lib = capi.lib(); capi.check(lib.vpp_init(0))
NR, NC = 2160, 3840; npx = NR * NC
def time_graph(launch, steps=200):
    for i in range(10): launch(i, capi.stream_ptr())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = capi.stream_ptr()
        for i in range(steps): launch(i, cs)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    return best
BORDER = int(sys.argv[1]) if len(sys.argv) > 1 else 2   # 2 = the reference benchmark's border (end row blocks take the guarded loads)
src_h = rand_image(NR, NC, vi.U8, 3, border=BORDER, seed=3, align=16)
ns = 8
srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]; dsts = [DeviceImage(NR, NC, vi.U8, 3, 0, 16) for _ in range(ns)]
sd, dd = [s.desc for s in srcs], [d.desc for d in dsts]
for probe in (0, 1):
    for rows in (1, 2, 4):
        for wpb in (1, 2, 4):
            lib.vpp_set_tuning(b"box.probe", probe); lib.vpp_set_tuning(b"box.rows", rows); lib.vpp_set_tuning(b"box.waves_per_block", wpb)
            us = time_graph(lambda i, s: lib.vpp_box_filter(P(dd[i % ns]), P(sd[i % ns]), 5, 5, s))
            print(f"border {BORDER} stream probe={probe} rows={rows} waves/block={wpb}: {us:.2f} us")
a = [torch.empty(25_000_000, dtype=torch.uint8, device="cuda") for _ in range(ns)]; b = [torch.empty(25_000_000, dtype=torch.uint8, device="cuda") for _ in range(ns)]
us = time_graph(lambda i, s: b[i % ns].copy_(a[i % ns]))
print(f"torch copy 25 MB -> 25 MB: {us:.2f} us  ({50e6/us/1e6:.2f} TB/s)")
x = [torch.empty(100_000_000, dtype=torch.uint8, device="cuda") for _ in range(4)]; y = [torch.empty(100_000_000, dtype=torch.uint8, device="cuda") for _ in range(4)]
us = time_graph(lambda i, s: y[i % 4].copy_(x[i % 4]), 100)
print(f"torch copy 100 MB -> 100 MB: {us:.2f} us  ({200e6/us/1e6:.2f} TB/s)")

# a 1:1 read/write stream of the box's byte counts through the runtime's own device-to-device copy
import ctypes
ha = [torch.empty(25_000_000, dtype=torch.uint8, device="cuda") for _ in range(ns)]; hb = [torch.empty(25_000_000, dtype=torch.uint8, device="cuda") for _ in range(ns)]
us = time_graph(lambda i, s: lib.vpp_memcpy_d2d(ctypes.c_void_p(hb[i % ns].data_ptr()), ctypes.c_void_p(ha[i % ns].data_ptr()), 25_000_000, s))
print(f"hipMemcpyDtoDAsync 25 MB -> 25 MB: {us:.2f} us  ({50e6/us/1e6:.2f} TB/s)")
