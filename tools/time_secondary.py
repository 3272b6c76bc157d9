"""Kernel times of the secondary paths at 4K / 1080p (hipGraph replay, rotating buffers where it matters): generic box filters,
border fill, pyramid level, Scharr, copy.  Prints us per launch and the algorithmic GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]  # A/B timing of two builds in ONE gpurun call (boxes differ by ~10 %)
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
NR, NC = 2160, 3840
ns = 6
def box(dtype, ch, R, C, border, es):
    src_h = rand_image(NR, NC, dtype, ch, border=border, seed=3, align=16, lo=0 if dtype != vi.F32 else None, hi=999 if dtype == vi.I32 else None)
    srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]; dsts = [DeviceImage(NR, NC, dtype, ch, 0, 16) for _ in range(ns)]
    sd, dd = [s.desc for s in srcs], [d.desc for d in dsts]
    us = time_graph(lambda i, s: lib.vpp_box_filter(P(dd[i % ns]), P(sd[i % ns]), R, C, s))
    b = 2 * NR * NC * es
    print(f"box {R}x{C} dtype {dtype} x{ch}: {us:8.2f} us  {b / us / 1e3:8.1f} GB/s algorithmic")
    return srcs, dsts
for rows in (1, 2, 4, 8):
    lib.vpp_set_tuning(b"box.rows32", rows); print("rows32", rows, end=": "); box(vi.I32, 1, 5, 5, 2, 4)       # the reference's own benchmark type (box_5x5_filter.cc)
lib.vpp_set_tuning(b"box.rows32", -1)
box(vi.U8, 1, 5, 5, 2, 1)
box(vi.U8, 4, 5, 5, 2, 4)
for rows in (1, 2, 4):
    lib.vpp_set_tuning(b"box.rows32", rows); print("rows32", rows, end=": "); box(vi.F32, 1, 5, 5, 2, 4)
lib.vpp_set_tuning(b"box.rows32", -1)
box(vi.U8, 3, 3, 3, 1, 3)
box(vi.U8, 3, 7, 7, 3, 3)
# pixel_wise on 8-bit images (packed-byte arithmetic): 3 x 24.9 MB per launch
for op, name in ((0, 'add'), (3, 'min'), (5, 'absdiff')):
    A = [DeviceImage(NR, NC, vi.U8, 3, 0, 16) for _ in range(ns)]; B = [DeviceImage.from_host(rand_image(NR, NC, vi.U8, 3, seed=1, align=16)) for _ in range(2 * ns)]
    us = time_graph(lambda i, s: lib.vpp_pixelwise_binary(op, P(A[i % ns].desc), P(B[2 * (i % ns)].desc), P(B[2 * (i % ns) + 1].desc), s)); print(f"pixel_wise {name} vuchar3 4K: {us:.2f} us  ({3 * NR * NC * 3 / us / 1e3:.1f} GB/s)")
# border fill, pyramid level, scharr at 1080p / 4K
for (nr, nc) in ((1080, 1920), (2160, 3840)):
    im = rand_image(nr, nc, vi.U8, 1, border=3, seed=4)
    d = [DeviceImage.from_host(im) for _ in range(ns)]
    us = time_graph(lambda i, s: lib.vpp_fill_border(P(d[i % ns].desc), 0, None, s)); print(f"fill_border_mirror u8 {nr}x{nc} b3: {us:.2f} us")
    nxt = [DeviceImage(1 + nr // 2, 1 + nc // 2, vi.U8, 1, 3) for _ in range(ns)]
    us = time_graph(lambda i, s: lib.vpp_pyr_down(P(nxt[i % ns].desc), P(d[i % ns].desc), s)); print(f"pyr_down u8 {nr}x{nc}: {us:.2f} us  ({nr * nc * 1.25 / us / 1e3:.1f} GB/s)")
    g = [DeviceImage(nr, nc, vi.F32, 2, 3) for _ in range(ns)]
    us = time_graph(lambda i, s: lib.vpp_scharr(P(g[i % ns].desc), P(d[i % ns].desc), s)); print(f"scharr f32 {nr}x{nc}: {us:.2f} us  ({nr * nc * 9 / us / 1e3:.1f} GB/s)")
    c = [DeviceImage(nr, nc, vi.U8, 1, 3) for _ in range(ns)]
    us = time_graph(lambda i, s: lib.vpp_copy(P(c[i % ns].desc), P(d[i % ns].desc), 0, s)); print(f"copy u8 {nr}x{nc}: {us:.2f} us  ({nr * nc * 2 / us / 1e3:.1f} GB/s)")
    gn = [DeviceImage(1 + nr // 2, 1 + nc // 2, vi.F32, 2, 3) for _ in range(ns)]
    us = time_graph(lambda i, s: lib.vpp_pyr_down(P(gn[i % ns].desc), P(g[i % ns].desc), s)); print(f"pyr_down f32x2 {nr}x{nc}: {us:.2f} us  ({nr * nc * 8 * 1.25 / us / 1e3:.1f} GB/s)")
# dense FAST flags and the blockwise maxima filter at 4K
from vpp_amd.synth import rects_image, u8_image
import numpy as np
fim = u8_image(rects_image(NR, NC, seed=31), border=3)
fim.view(with_border=True)[..., 0] = np.pad(fim.view()[..., 0], 3, mode="symmetric")
fd = [DeviceImage.from_host(fim) for _ in range(ns)]; fo = [DeviceImage(NR, NC, vi.U8, 1) for _ in range(ns)]
us = time_graph(lambda i, s: lib.vpp_fast9_dense(P(fo[i % ns].desc), P(fd[i % ns].desc), 20, s)); print(f"fast9_dense u8 4K: {us:.2f} us  ({NR * NC * 2 / us / 1e3:.1f} GB/s, {NR * NC / us / 1e3:.1f} Gpx/s)")
us = time_graph(lambda i, s: lib.vpp_blockwise_maxima_filter(P(fd[i % ns].desc), 10, s)); print(f"blockwise_maxima_filter u8 4K bs 10: {us:.2f} us  ({NR * NC * 2 / us / 1e3:.1f} GB/s)")
lb = [DeviceImage(NR, NC, vi.U8, 1) for _ in range(ns)]
us = time_graph(lambda i, s: lib.vpp_lbp_transform(P(lb[i % ns].desc), P(fd[i % ns].desc), s)); print(f"lbp_transform u8 4K: {us:.2f} us  ({NR * NC * 2 / us / 1e3:.1f} GB/s)")
for (R, C, b) in ((3, 3, 1), (7, 5, 3), (3, 5, 2)):
    src_h = rand_image(NR, NC, vi.U8, 3, border=b, seed=3, align=16)
    srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]; dsts = [DeviceImage(NR, NC, vi.U8, 3, 0, 16) for _ in range(ns)]
    for g in (0, 1):
        lib.vpp_set_tuning(b"box.force_generic", g)
        us = time_graph(lambda i, s: lib.vpp_box_filter(P(dsts[i % ns].desc), P(srcs[i % ns].desc), R, C, s)); print(f"box {R}x{C} vuchar3 4K {'generic' if g else 'streamed'}: {us:.2f} us ({2 * NR * NC * 3 / us / 1e3:.0f} GB/s)")
    lib.vpp_set_tuning(b"box.force_generic", 0)
for dt, nm in ((vi.I32, "int"), (vi.F32, "float")):
    src_h = rand_image(NR, NC, dt, 1, border=1, seed=3, align=16, lo=0 if dt == vi.I32 else None, hi=999 if dt == vi.I32 else None)
    srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]; dsts = [DeviceImage(NR, NC, dt, 1, 0, 16) for _ in range(ns)]
    for g in (0, 1):
        lib.vpp_set_tuning(b"box.force_generic", g)
        us = time_graph(lambda i, s: lib.vpp_box_filter(P(dsts[i % ns].desc), P(srcs[i % ns].desc), 3, 3, s)); print(f"box 3x3 {nm} 4K {'generic' if g else 'streamed'}: {us:.2f} us ({2 * NR * NC * 4 / us / 1e3:.0f} GB/s)")
    lib.vpp_set_tuning(b"box.force_generic", 0)
