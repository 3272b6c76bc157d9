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
log = (ctypes.c_ulonglong * 4096)(); n = ctypes.c_uint(0)
lib.vpp_debug_sdof_sweep_log(log, ctypes.byref(n), 1)
call()
lib.vpp_debug_sdof_sweep_log(log, ctypes.byref(n), 1)
lib.vpp_set_tuning(b"sdof.stats", -1)
ent = sorted(((e & 0xFFFFFFFF), e >> 56, (e >> 32) & 0xFFFFFF) for e in list(log)[:min(n.value, 4096)])
t0 = ent[0][0] if ent else 0
verbose = len(sys.argv) > 2
cls, done = [], []
for t, rnd, cnt in ent:
    if rnd == 253: cls.append((t, cnt)); continue
    if rnd == 252: done.append((t, cnt)); continue
    if rnd == 251:
        if verbose: print(f"{(t - t0) * 0.01:8.2f} us      workgroup ticket {cnt >> 8} finished its share of round {cnt & 255}")
        continue
    if rnd in (255, 1) and (cls or done):
        if cls: print(f"            {len(cls)} workgroups with candidates ({sum(c for _, c in cls)} in all, max {max(c for _, c in cls)}); classified between {(cls[0][0] - t0) * 0.01:.2f} and {(cls[-1][0] - t0) * 0.01:.2f} us")
        if done:
            print(f"            round 0 done on them between {(done[0][0] - t0) * 0.01:.2f} and {(done[-1][0] - t0) * 0.01:.2f} us; the last five: " + ", ".join(f"{(t_ - t0) * 0.01:.2f} us ({c} cand.)" for t_, c in done[-5:]))
        cls, done = [], []
    what = {255: f"launch start, {cnt} workgroups", 254: f"  {cnt} workgroups run the rounds", 250: "  skipped: the previous sweep of this scale found no candidate"}.get(rnd, f"  round {rnd}: {cnt} jobs")
    print(f"{(t - t0) * 0.01:8.2f} us  {what}")
