"""How does a pure streaming kernel (K1 add, int32) scale with bytes per launch?  Calibrates what fraction of 8 TB/s a kernel
of a given size can reach at all on this box (fixed ramp / drain cost per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vpp_amd.synth import P, DeviceImage
from vpp_amd import capi, image as vi
lib = capi.lib(); capi.check(lib.vpp_init(0))
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
for nr, nc in ((540, 3840), (1080, 3840), (2160, 1920), (2160, 3840), (4320, 3840), (4320, 7680)):
    mb = nr * nc * 12 / 1e6
    ns = max(2, int(600 / mb) + 1)
    A = [DeviceImage(nr, nc, vi.I32) for _ in range(ns)]; B = [DeviceImage(nr, nc, vi.I32) for _ in range(ns)]; C = [DeviceImage(nr, nc, vi.I32) for _ in range(ns)]
    ad, bd, cd = [x.desc for x in A], [x.desc for x in B], [x.desc for x in C]
    us = time_graph(lambda i, s: lib.vpp_pixelwise_binary(0, P(ad[i % ns]), P(bd[i % ns]), P(cd[i % ns]), s), 100)
    print(f"add {nr}x{nc} int32: {mb:7.1f} MB/launch  {us:7.2f} us  {mb/us/1e0*1e-3*1e3/1e3:6.2f} TB/s  {mb/us/8:5.1f}% of 8 TB/s".replace("TB/s  ", "TB/s "))
    del A, B, C
