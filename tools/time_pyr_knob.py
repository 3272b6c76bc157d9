"""vpp_pyramid_build (u8, 3 levels) at 4K / 1080p by border and by a tuning knob (hipGraph of 200 builds): python tools/time_pyr_knob.py pyr.near 0 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, texture
from vpp_amd import capi, image as vi, pyr
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]   # A/B of several builds on one box
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
knob = sys.argv[1].encode(); values = [int(x) for x in sys.argv[2:]]
for (nr, nc, border) in ((2160, 3840, 3), (2160, 3840, 18), (1080, 1920, 3), (1080, 1920, 18)):
    f = np.clip(np.rint(texture(nr, nc, seed=5)), 0, 255).astype(np.uint8)
    d = DeviceImage.from_host(u8_image(f))
    lv = [DeviceImage(a, b, vi.U8, 1, border) for a, b in pyr.level_dims(nr, nc, 3)]
    dl = vi.desc_array(lv)
    out = []
    for v in values * 2:
        lib.vpp_set_tuning(knob, v)
        out.append(f"{knob.decode()}={v}: {time_graph(lambda s: lib.vpp_pyramid_build(dl, 3, P(d.desc), s)):.2f} us")
    print(f"{nr}x{nc} border {border}: " + "  ".join(out), flush=True)
lib.vpp_set_tuning(knob, -1)
