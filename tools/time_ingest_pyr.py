"""hipGraph timings of the frame ingest followed by the image pyramid (two launches) against vpp_rgb_pyramid_build (one launch), 4K and 1080p,
rotating over enough frames that the sources come from HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi, pyr
lib = capi.lib(); capi.check(lib.vpp_init(0))
def time_graph(launch, steps=200):
    for i in range(5): launch(i, capi.stream_ptr())
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
for (nr, nc, border, nset, sw) in ((2160, 3840, 3, 12, 1), (2160, 3840, 18, 12, 1), (1080, 1920, 3, 40, 1), (2160, 3840, 18, 12, 0), (1080, 1920, 3, 40, 0)):
    lib.vpp_set_tuning(b"pyr.swar", sw)
    rgb_h = rand_image(nr, nc, vi.U8, 3, border=0, seed=6)
    rgbs = [DeviceImage.from_host(rgb_h) for _ in range(nset)]
    grays = [DeviceImage(nr, nc, vi.U8, 1, border, 32) for _ in range(nset)]
    lvs = [[DeviceImage(a, b, vi.U8, 1, border) for a, b in pyr.level_dims(nr, nc, 3)] for _ in range(nset)]
    dls = [vi.desc_array(l) for l in lvs]
    def chain(i, s):
        k = i % nset
        lib.vpp_rgb_to_graylevel(P(grays[k].desc), P(rgbs[k].desc), 1, s)
        lib.vpp_pyramid_build(dls[k], 3, P(grays[k].desc), s)
    def fused(i, s):
        k = i % nset
        lib.vpp_rgb_pyramid_build(dls[k], 3, P(rgbs[k].desc), s)
    def ingest(i, s):
        k = i % nset
        lib.vpp_rgb_to_graylevel(P(grays[k].desc), P(rgbs[k].desc), 1, s)
    def pyronly(i, s):
        k = i % nset
        lib.vpp_pyramid_build(dls[k], 3, P(grays[k].desc), s)
    print(f"{nr}x{nc} border {border} packed={sw}: ingest {time_graph(ingest):.2f} us, pyramid {time_graph(pyronly):.2f} us, ingest + pyramid {time_graph(chain):.2f} us, fused vpp_rgb_pyramid_build {time_graph(fused):.2f} us", flush=True)
