"""Launch FAST-9 / semi-dense flow / pyrLK a few times (for rocprofv3 kernel traces)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, rects_image
from test_gpu_sdof import flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
im = u8_image(rects_image(2160, 3840, seed=4), border=3)
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
lib.vpp_debug_sdof_stats(st4, 1)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    capi.check(lib.vpp_semi_dense_optical_flow(P(d1.desc), P(d2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
    torch.cuda.synchronize(); print("sdof 4K", m, "keypoints:", (time.perf_counter() - t0) * 1e3, "ms")
    lib.vpp_debug_sdof_stats(st4, 1); print("  sweep stats: visited %d, jacobi applied %d, slow path %d, slow changed %d" % tuple(st4))
