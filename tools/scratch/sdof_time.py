import ctypes, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from util import P, u8_image, DeviceImage
from test_gpu_sdof import flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
f1, f2, kps = flow_scene(2160, 3840, spacing=10)
d1, d2 = DeviceImage.from_host(u8_image(f1, border=3)), DeviceImage.from_host(u8_image(f2, border=3))
dk = torch.from_numpy(kps).cuda(); m = len(kps)
gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
for skip in (0, 1):
    lib.vpp_set_tuning(b"sdof.skip_slow", skip)
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        capi.check(lib.vpp_semi_dense_optical_flow(P(d1.desc), P(d2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
        torch.cuda.synchronize(); print("skip_slow", skip, "sdof 4K", m, "keypoints:", (time.perf_counter() - t0) * 1e3, "ms")
