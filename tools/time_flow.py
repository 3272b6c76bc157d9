"""Time vpp_semi_dense_optical_flow on the 4K bench scene (BASELINE configs[4] defaults: winsize 9, 3 scales, 2 sweeps, patch 5)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage
from vpp_amd import capi
from test_gpu_sdof import flow_scene
V = ctypes.c_void_p
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]  # A/B timing of two builds in ONE gpurun call (boxes differ by ~10 %)
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
s1, s2, sk = flow_scene(2160, 3840, spacing=10)
e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
m = len(sk); dk = torch.from_numpy(sk).cuda()
gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
for ws in (9, 7):
    ts = []
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, ws, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"winsize {ws}: {min(ts[2:]) * 1e3:.3f} ms per 4K frame pair ({m} keypoints), checksum {int(gp.sum())} {int(gd.sum())}")
# the strip-sharded form on one GPU (SURVEY 8e bullet 2): claim + descent on `nstrips` streams, gather / broadcast as device copies
for nstrips in (1, 2, 4, 8):
    ts = []
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        capi.check(lib.vpp_semi_dense_optical_flow_strips(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, nstrips, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"winsize 9, {nstrips} strip(s): {min(ts[2:]) * 1e3:.3f} ms per 4K frame pair")
