"""A/B of one tuning knob of vpp_semi_dense_optical_flow on the 4K bench scenes in ONE process / one box, values interleaved:
    python tools/flow_knob_ab.py sdof.descent_tpb 64 256"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]  # A/B of two builds in one call
knob = sys.argv[1].encode(); values = [int(x) for x in sys.argv[2:]]
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
for shape, spacing in (((2160, 3840), 10), ((2160, 3840), 5), ((1080, 1920), 10)):
    s1, s2, sk = flow_scene(*shape, spacing=spacing)
    e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
    m = len(sk); dk = torch.from_numpy(sk).cuda()
    gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
    sums = {}
    for v in values * 3:
        lib.vpp_set_tuning(knob, v)
        ts = []
        for it in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        rc = lib.vpp_sync(st)   # (reports a device-side protocol that gave up)
        if rc: print("  vpp_sync:", rc, lib.vpp_last_error().decode(), flush=True)
        sums.setdefault(v, set()).add((int(gp.sum()), int(gd.sum()), int(gv.sum())))
        print(f"{shape} spacing {spacing} ({m} kps) {knob.decode()}={v}: min {min(ts[2:]) * 1e3:.3f} ms  median {sorted(ts[2:])[len(ts[2:]) // 2] * 1e3:.3f} ms", flush=True)
    print("  identical:", len(set.union(*sums.values())) == 1)
lib.vpp_set_tuning(knob, -1)
