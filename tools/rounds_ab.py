"""The 4K bench scene's flow call with its round statistics (tuning sdof.stats = 1)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
for kv in os.environ.get("VPP_TUNE", "").split(","):   # tuning knobs for this run: VPP_TUNE="name=value,..."
    if "=" in kv: lib.vpp_set_tuning(kv.split("=")[0].encode(), int(kv.split("=")[1]))
s1, s2, sk = flow_scene(2160, 3840, spacing=10)
e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
m = len(sk); dk = torch.from_numpy(sk).cuda()
gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
st4 = (ctypes.c_uint * 4)()
for ws in (9, 7):
    lib.vpp_set_tuning(b"sdof.stats", 0)
    ts = []
    for it in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, ws, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lib.vpp_set_tuning(b"sdof.stats", 1); lib.vpp_debug_sdof_round_stats(st4, 1)
    capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, ws, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
    lib.vpp_debug_sdof_round_stats(st4, 1)
    print(f"winsize {ws}: {min(ts[2:]) * 1e3:.3f} ms  rounds {st4[0]} jobs {st4[1]} evaluated {st4[2]} changes {st4[3]}  checksum {int(gp.sum())} {int(gd.sum())}", flush=True)
