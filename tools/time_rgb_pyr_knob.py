"""vpp_rgb_pyramid_build (4K / 1080p rgb frame -> gray 3-level pyramid, one launch) by a tuning knob (hipGraph of 200 builds): python tools/time_rgb_pyr_knob.py pyr.gray_row_tiles 0 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, DeviceImage, rand_image
from vpp_amd import capi, image as vi, pyr
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]
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
for (nr, nc, ch, border) in ((2160, 3840, 3, 4), (2160, 3840, 4, 4), (1080, 1920, 3, 4)):
    d = DeviceImage.from_host(rand_image(nr, nc, vi.U8, ch, border=0, seed=35, align=32))
    lv = [DeviceImage(a, b, vi.U8, 1, border) for a, b in pyr.level_dims(nr, nc, 3)]
    dl = vi.desc_array(lv)
    out = []
    for v in values * 2:
        lib.vpp_set_tuning(knob, v)
        out.append(f"{knob.decode()}={v}: {time_graph(lambda s: lib.vpp_rgb_pyramid_build(dl, 3, P(d.desc), s)):.2f} us")
    print(f"{nr}x{nc} x{ch} border {border}: " + "  ".join(out), flush=True)
lib.vpp_set_tuning(knob, -1)
