"""vpp_blockwise_maxima_filter at 4K (u8 / u16 / f32, block 10 and 32), per tuning knob blockwise_maxima.rows (hipGraph of 100 calls over rotating images)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
lib = capi.lib(); capi.check(lib.vpp_init(0))
def time_graph(launch, steps=100):
    for i in range(3): launch(i, capi.stream_ptr())
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
NR, NC = 2160, 3840
for dt, name, sz in ((vi.U8, "u8", 1), (vi.U16, "u16", 2), (vi.F32, "f32", 4)):
    ims = [DeviceImage.from_host(rand_image(NR, NC, dt, 1, seed=8 + k)) for k in range(20 if sz < 4 else 8)]
    for bs in (10, 32):
        out = []
        for v in (0, 1, 0, 1):
            lib.vpp_set_tuning(b"blockwise_maxima.rows", v)
            us = time_graph(lambda i, s: lib.vpp_blockwise_maxima_filter(P(ims[i % len(ims)].desc), bs, s))
            out.append(f"rows={v}: {us:.2f} us ({NR * NC * 2 * sz / us / 1e3:.0f} GB/s)")
        print(f"{name} 4K bs {bs}: " + "  ".join(out), flush=True)
lib.vpp_set_tuning(b"blockwise_maxima.rows", -1)
