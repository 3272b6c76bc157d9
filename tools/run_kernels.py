"""Launch a few box / add kernels (for rocprofv3 runs).  usage: run_kernels.py [box|add|box32|all] [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
which = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
NR, NC = 2160, 3840
if which in ("box", "all"):
    src_h = rand_image(NR, NC, vi.U8, 3, border=2, seed=3, align=16)
    ns = 64
    srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]
    dsts = [DeviceImage(NR, NC, vi.U8, 3, 0, 16) for _ in range(ns)]
    for s in srcs: capi.check(lib.vpp_fill_border(P(s.desc), 0, None, st))
    for i in range(n): capi.check(lib.vpp_box_filter(P(dsts[i % ns].desc), P(srcs[i % ns].desc), 5, 5, st))
    for i in range(max(2, n // ns)): capi.check(lib.vpp_box_filter_batch(vi.desc_array(dsts), vi.desc_array(srcs), ns, 5, 5, st))   # bench.py's headline launch: 64 frames
    torch.cuda.synchronize()
if which in ("add", "all"):
    b_h = rand_image(NR, NC, vi.I32, seed=2, lo=0, hi=2**30 - 1)
    ns = 16
    A = [DeviceImage(NR, NC, vi.I32) for _ in range(ns)]; B = [DeviceImage.from_host(b_h) for _ in range(ns)]; C = [DeviceImage.from_host(b_h) for _ in range(ns)]
    for i in range(n): capi.check(lib.vpp_pixelwise_binary(0, P(A[i % ns].desc), P(B[i % ns].desc), P(C[i % ns].desc), st))
    for i in range(max(2, n // ns)): capi.check(lib.vpp_pixelwise_binary_batch(0, vi.desc_array(A), vi.desc_array(B), vi.desc_array(C), ns, st))   # bench.py's add leg: 16 triples
    torch.cuda.synchronize()
if which in ("box32", "all"):   # the reference benchmark's own element type (box_5x5_filter.cc) and the frame ingest
    src_h = rand_image(NR, NC, vi.I32, 1, border=2, seed=3, align=16, lo=0, hi=999)
    ns = 5
    srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]
    dsts = [DeviceImage(NR, NC, vi.I32, 1, 0, 16) for _ in range(ns)]
    for i in range(n): capi.check(lib.vpp_box_filter(P(dsts[i % ns].desc), P(srcs[i % ns].desc), 5, 5, st))
    rgb_h = rand_image(NR, NC, vi.U8, 3, border=0, seed=6)
    rgbs = [DeviceImage.from_host(rgb_h) for _ in range(8)]; grays = [DeviceImage(NR, NC, vi.U8, 1, 3, 32) for _ in range(8)]
    for i in range(n): capi.check(lib.vpp_rgb_to_graylevel(P(grays[i % 8].desc), P(rgbs[i % 8].desc), 1, st))
    torch.cuda.synchronize()
