"""Launch FAST-9 / semi-dense flow / pyrLK a few times (for rocprofv3 kernel traces)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, rects_image, fast9_bench_frame
from vpp_amd.synth import flow_scene, texture, translate
from vpp_amd import pyr, image as vi
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
im = u8_image(fast9_bench_frame(), border=3)
im.view(with_border=True)[..., 0] = np.pad(im.view()[..., 0], 3, mode="symmetric")
d = DeviceImage.from_host(im)
cap = 3000000
rc = torch.zeros((cap, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros(cap, dtype=torch.int32, device="cuda"); n = ctypes.c_int(0)
for mode in (0, 1, 2):
    for _ in range(5):
        lib.vpp_fast9_detect(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st)
f1, f2, kps = flow_scene(2160, 3840, spacing=10)
d1, d2 = DeviceImage.from_host(u8_image(f1, border=3)), DeviceImage.from_host(u8_image(f2, border=3))
dk = torch.from_numpy(kps).cuda(); m = len(kps)
gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
import time
lib.vpp_set_tuning(b"sdof.stats", 1)
st4 = (ctypes.c_uint * 4)()
lib.vpp_debug_sdof_round_stats(st4, 1)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    capi.check(lib.vpp_semi_dense_optical_flow(P(d1.desc), P(d2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
    torch.cuda.synchronize(); print("sdof 4K", m, "keypoints:", (time.perf_counter() - t0) * 1e3, "ms")
    lib.vpp_debug_sdof_round_stats(st4, 1); print("  propagation: %d rounds, %d jobs, %d evaluated, %d changes" % tuple(st4))

# pyrLK, BASELINE configs[3]: 1080p, 3 levels, 10 000 keypoints, 7x7
NR, NC, L, B = 1080, 1920, 3, 3
tex = texture(NR, NC, seed=5)
g1 = np.clip(np.rint(tex), 0, 255).astype(np.uint8); g2 = np.clip(np.rint(translate(tex, 1.5, -2.25)), 0, 255).astype(np.uint8)
q1, q2 = DeviceImage.from_host(u8_image(g1)), DeviceImage.from_host(u8_image(g2))
kps_h = pyr.make_keypoints(pyr.grid_keypoints(NR, NC, 10000, margin=32))
k0 = torch.from_numpy(kps_h.view(np.uint8).reshape(-1).copy()).cuda()
for i in range(6):
    p1 = pyr.device_pyramid(lib, q1, L, B); p2 = pyr.device_pyramid(lib, q2, L, B); gr = pyr.device_grad_pyramid(lib, p1[0], L, B, vi.F32)
    k = k0.clone()
    capi.check(lib.vpp_pyrlk_match(vi.desc_array(p1), vi.desc_array(gr), vi.desc_array(p2), L, V(k.data_ptr()), 10000, 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30,
                                   ctypes.c_float(0.01), 0, None, st))
torch.cuda.synchronize()
# round 4: the latency floor (1 250 keypoints = one of 8 ranks' share of configs[3]: 64 lanes per keypoint) and the reference's own benchmark
# configuration (11 x 11 window, 4 levels: benchmarks/pyrlk_opencv_comparison.cc:47,64-65) — kernels of other symbols, so the PMC passes see them separately
kps_s = pyr.make_keypoints(pyr.grid_keypoints(NR, NC, 1250, margin=32))
ks0 = torch.from_numpy(kps_s.view(np.uint8).reshape(-1).copy()).cuda()
p1 = pyr.device_pyramid(lib, q1, L, B); p2 = pyr.device_pyramid(lib, q2, L, B); gr = pyr.device_grad_pyramid(lib, p1[0], L, B, vi.F32)
for i in range(6):
    k = ks0.clone()
    capi.check(lib.vpp_pyrlk_match(vi.desc_array(p1), vi.desc_array(gr), vi.desc_array(p2), L, V(k.data_ptr()), 1250, 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30,
                                   ctypes.c_float(0.01), 0, None, st))
L4, B4 = 4, 8
r1 = pyr.device_pyramid(lib, q1, L4, B4); r2 = pyr.device_pyramid(lib, q2, L4, B4); rg = pyr.device_grad_pyramid(lib, r1[0], L4, B4, vi.F32)
for i in range(6):
    k = k0.clone()
    capi.check(lib.vpp_pyrlk_match(vi.desc_array(r1), vi.desc_array(rg), vi.desc_array(r2), L4, V(k.data_ptr()), 10000, 11, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30,
                                   ctypes.c_float(0.01), 0, None, st))
torch.cuda.synchronize()
# round 6: 8 frame pairs x 1 250 keypoints in ONE launch (vpp_pyrlk_match_batch: a rank's slice of configs[3] over a group of frames) — pyrlk_match_batch_kernel<7, 16>
F = 8
sets = []
for f in range(F):
    texf = texture(NR, NC, seed=60 + f)
    a = DeviceImage.from_host(u8_image(np.clip(np.rint(texf), 0, 255).astype(np.uint8)))
    b = DeviceImage.from_host(u8_image(np.clip(np.rint(translate(texf, 1.5 - 0.1 * f, -2.25 + 0.2 * f)), 0, 255).astype(np.uint8)))
    pa = pyr.device_pyramid(lib, a, L, B); sets.append((pa, pyr.device_grad_pyramid(lib, pa[0], L, B, vi.F32), pyr.device_pyramid(lib, b, L, B)))
dP = vi.desc_array([l_ for q in sets for l_ in q[0]]); dG = vi.desc_array([l_ for q in sets for l_ in q[1]]); dN = vi.desc_array([l_ for q in sets for l_ in q[2]])
for i in range(6):
    ks = [ks0.clone() for _ in range(F)]
    capi.check(lib.vpp_pyrlk_match_batch(dP, dG, dN, F, L, (ctypes.c_void_p * F)(*[k_.data_ptr() for k_ in ks]), (ctypes.c_int * F)(*([1250] * F)), 7, ctypes.c_float(1e-4),
                                         ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, st))
torch.cuda.synchronize()
