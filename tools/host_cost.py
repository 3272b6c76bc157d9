"""Host-side cost of one flow call (time until the call returns, the device still busy) against the device-side time per call in a back-to-back loop."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
s1, s2, sk = flow_scene(2160, 3840, spacing=10)
e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
m = len(sk); dk = torch.from_numpy(sk).cuda()
gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
call = lambda: capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
for _ in range(5): call()
torch.cuda.synchronize()
N = 50
host = []
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); call(); host.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
host.sort()
print(f"host time per call: median {host[N // 2] * 1e6:.1f} us, min {host[0] * 1e6:.1f} us; {N} calls queued in {(t1 - t0) * 1e3:.2f} ms, done after {(t2 - t0) * 1e3:.2f} ms = {(t2 - t0) / N * 1e6:.1f} us per call back to back")
