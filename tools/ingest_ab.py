"""A/B of the frame-ingest launch geometry (vpp_rgb_to_graylevel, 4K vuchar3 -> uchar + mirror border 3): threads per workgroup (ingest.block; default 64 for a
single frame, 256 for launches of 4 frames and more).  Per variant: every byte against the oracle once, then event-timed launch graphs of 1024 per-frame calls
over 64 frame sets — folded into 64-frame launches at record time, and as one launch per call.  (Round 5 also measured a wave-cooperative kernel here —
coalesced 16-byte loads turned around through LDS: 5.89 / 9.79 us against 5.85 / 9.19 for the better block size of the lane-per-chunk kernel; not kept.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, rand_image, DeviceImage, HostImage
from vpp_amd import capi, image as vi
from oracle import binding
lib = capi.lib(); capi.check(lib.vpp_init(0)); orc = binding.load(omp=True)
NR, NC, nin = 2160, 3840, 64
rgb_h = rand_image(NR, NC, vi.U8, 3, border=0, seed=6)
want = HostImage(NR, NC, vi.U8, 1, 3); orc.orc_rgb_to_graylevel(P(want.desc), P(rgb_h.desc), 1)
rgbs = [DeviceImage.from_host(rgb_h) for _ in range(nin)]; grays = [DeviceImage(NR, NC, vi.U8, 1, 3, 32) for _ in range(nin)]
side = torch.cuda.Stream(); sp = ctypes.c_void_p(side.cuda_stream)


def graph_us(ncalls):
    gh = ctypes.c_void_p()
    capi.check(lib.vpp_graph_begin(sp))
    for i in range(ncalls):
        capi.check(lib.vpp_rgb_to_graylevel(P(grays[i % nin].desc), P(rgbs[i % nin].desc), 1, sp))
    capi.check(lib.vpp_graph_end(sp, 1, ctypes.byref(gh)))
    ts = []
    for _ in range(5):
        capi.check(lib.vpp_graph_launch(gh, sp)); torch.cuda.synchronize()
        ms = ctypes.c_float(0); capi.check(lib.vpp_graph_elapsed_ms(gh, ctypes.byref(ms))); ts.append(ms.value)
    lib.vpp_graph_destroy(gh)
    return sorted(ts[1:])[1] * 1e3 / ncalls


for block in (0, 64, 128, 256):
    lib.vpp_set_tuning(b"ingest.block", block if block else -1)
    for g in grays[:2]:
        g.store.zero_()
    capi.check(lib.vpp_rgb_to_graylevel(P(grays[0].desc), P(rgbs[0].desc), 1, capi.stream_ptr()))
    capi.check(lib.vpp_rgb_to_graylevel_batch(vi.desc_array(grays[1:3]), vi.desc_array(rgbs[1:3]), 2, 1, capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    ok = all(np.array_equal(grays[k].download().view(with_border=True), want.view(with_border=True)) for k in range(3))
    lib.vpp_set_tuning(b"ingest.coalesce", 1); lib.vpp_set_tuning(b"launch.capture_width", -1)
    rec = graph_us(1024)
    lib.vpp_set_tuning(b"ingest.coalesce", 0); lib.vpp_set_tuning(b"launch.capture_width", 1)
    one = graph_us(1024)
    lib.vpp_set_tuning(b"ingest.coalesce", -1); lib.vpp_set_tuning(b"launch.capture_width", -1)
    b = NR * NC * 4
    print(f"block {block or 'default'}: {'ok' if ok else 'MISMATCH'}  recorded (64-frame launches) {rec:.2f} us = {b / rec / 8e6:.3f}   one launch per call {one:.2f} us = {b / one / 8e6:.3f}", flush=True)
