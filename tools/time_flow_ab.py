"""A/B of the propagation implementations of vpp_semi_dense_optical_flow on the 4K bench scene in ONE process / one box:
sdof.propagate = 1 (lock-step wavefront on one workgroup) vs 0 (Jacobi rounds to the fixed point, the default)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
for shape, spacing in (((2160, 3840), 10), ((2160, 3840), 5), ((1080, 1920), 10)):
    s1, s2, sk = flow_scene(*shape, spacing=spacing)
    e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
    m = len(sk); dk = torch.from_numpy(sk).cuda()
    gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
    sums = {}
    for mode in (1, 0, 1, 0):
        lib.vpp_set_tuning(b"sdof.propagate", mode)
        ts = []
        for it in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        sums[mode] = (int(gp.sum()), int(gd.sum()), int(gv.sum()))
        print(f"{shape} spacing {spacing} ({m} kps) propagate={mode}: min {min(ts[2:]) * 1e3:.3f} ms  median {sorted(ts[2:])[len(ts[2:]) // 2] * 1e3:.3f} ms  checksum {sums[mode]}", flush=True)
    print("  identical:", sums[0] == sums[1])
    if hasattr(lib, "vpp_debug_sdof_round_stats"):
        lib.vpp_set_tuning(b"sdof.propagate", 0); lib.vpp_set_tuning(b"sdof.stats", 1)
        out = (ctypes.c_uint * 4)(); lib.vpp_debug_sdof_round_stats(out, 1)
        capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
        lib.vpp_debug_sdof_round_stats(out, 1); lib.vpp_set_tuning(b"sdof.stats", 0)
        print("  rounds, jobs, evaluated, changes:", list(out))
lib.vpp_set_tuning(b"sdof.propagate", -1)
