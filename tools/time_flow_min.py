"""The 4K (and 1080p) bench scenes' flow call, 60 timed calls each: min / median per call.  VPP_AMD_LIB selects the build (A/B of several builds on ONE box)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
out = []
for shape, spacing in (((2160, 3840), 10), ((1080, 1920), 10), ((2160, 3840), 5)):   # the last one: every cell of the finest scale claimed, long queues in the sweeps
    s1, s2, sk = flow_scene(*shape, spacing=spacing)
    e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
    m = len(sk); dk = torch.from_numpy(sk).cuda()
    gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(70):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    capi.check(lib.vpp_sync(st))
    ts = sorted(ts[10:])
    out.append(f"{shape[0]}p/{spacing} min {ts[0] * 1e3:.4f} median {ts[len(ts) // 2] * 1e3:.4f} ms (checksum {int(gp.sum())} {int(gd.sum())})")
print("  ".join(out), flush=True)
