"""Round log of the fused sweeps of one 4K flow call (tuning sdof.stats = 1): per sweep launch its start, the rounds' job counts and when each was published
(100 MHz wall clock), and how many workgroups stayed."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
spacing = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s1, s2, sk = flow_scene(2160, 3840, spacing=spacing)
e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
m = len(sk); dk = torch.from_numpy(sk).cuda()
gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
call = lambda: capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
for _ in range(3): call()
lib.vpp_set_tuning(b"sdof.stats", 1)
log = (ctypes.c_ulonglong * 512)(); n = ctypes.c_uint(0)
lib.vpp_debug_sdof_sweep_log(log, ctypes.byref(n), 1)
call()
lib.vpp_debug_sdof_sweep_log(log, ctypes.byref(n), 1)
lib.vpp_set_tuning(b"sdof.stats", -1)
ent = sorted(((e & 0xFFFFFFFF), e >> 56, (e >> 32) & 0xFFFFFF) for e in list(log)[:min(n.value, 512)])
t0 = ent[0][0] if ent else 0
for t, rnd, cnt in ent:
    what = {255: f"launch start, {cnt} workgroups", 254: f"  {cnt} workgroups run the rounds"}.get(rnd, f"  round {rnd}: {cnt} jobs")
    print(f"{(t - t0) * 0.01:8.2f} us  {what}")
