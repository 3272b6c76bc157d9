"""hipGraph timings of the pyramid builders (fused one-launch kernels vs the per-level chain) at 1080p and 4K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, texture
from vpp_amd import capi, image as vi, pyr
lib = capi.lib(); capi.check(lib.vpp_init(0))
def time_graph(launch, steps=200):
    for i in range(5): launch(capi.stream_ptr())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = capi.stream_ptr()
        for i in range(steps): launch(cs)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    return best
for (nr, nc, border) in ((1080, 1920, 3), (2160, 3840, 18)):
    f = np.clip(np.rint(texture(nr, nc, seed=5)), 0, 255).astype(np.uint8)
    d = DeviceImage.from_host(u8_image(f))
    for fused in (1, 0):
        lib.vpp_set_tuning(b"pyr.fused", fused)
        lv = [DeviceImage(a, b, vi.U8, 1, border) for a, b in pyr.level_dims(nr, nc, 3)]
        gl = [DeviceImage(a, b, vi.F32, 2, border) for a, b in pyr.level_dims(nr, nc, 3)]
        dl, dg = vi.desc_array(lv), vi.desc_array(gl)
        t1 = time_graph(lambda s: lib.vpp_pyramid_build(dl, 3, P(d.desc), s))
        t2 = time_graph(lambda s: lib.vpp_scharr_pyramid_build(dg, 3, P(lv[0].desc), s))
        print(f"{nr}x{nc} border {border} fused={fused}: u8 pyramid {t1:.2f} us, scharr + gradient pyramid {t2:.2f} us")
lib.vpp_set_tuning(b"pyr.fused", -1)
