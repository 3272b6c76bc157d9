"""One 4K frame pair of vpp_semi_dense_optical_flow: back-to-back asynchronous calls against replays of a recorded launch graph (1 and 4 calls per graph)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0))
s = V(); capi.check(lib.vpp_stream_create(ctypes.byref(s)))
for shape, spacing in (((2160, 3840), 10), ((1080, 1920), 10)):
    s1, s2, sk = flow_scene(*shape, spacing=spacing)
    e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
    m = len(sk); dk = torch.from_numpy(sk).cuda()
    gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
    call = lambda: capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), s))
    for _ in range(3): call()
    capi.check(lib.vpp_sync(s))
    ref = (int(gp.sum()), int(gd.sum()), int(gv.sum()))
    t0 = time.perf_counter()
    for _ in range(20): call()
    capi.check(lib.vpp_sync(s)); dt_async = (time.perf_counter() - t0) / 20
    out = [f"{shape} {m} kps: async calls {dt_async * 1e3:.3f} ms"]
    for per in (1, 4):
        g = ctypes.c_void_p()
        capi.check(lib.vpp_graph_begin(s))
        for _ in range(per): call()
        capi.check(lib.vpp_graph_end(s, 1, ctypes.byref(g)))
        for _ in range(3): capi.check(lib.vpp_graph_launch(g, s))
        capi.check(lib.vpp_sync(s))
        t0 = time.perf_counter()
        reps = 20 // per
        for _ in range(reps): capi.check(lib.vpp_graph_launch(g, s))
        capi.check(lib.vpp_sync(s)); dt = (time.perf_counter() - t0) / (reps * per)
        ms = ctypes.c_float(); capi.check(lib.vpp_graph_elapsed_ms(g, ctypes.byref(ms)))
        out.append(f"graph of {per}: {dt * 1e3:.3f} ms per pair (device clock of the last replay {ms.value / per:.3f})")
        assert ref == (int(gp.sum()), int(gd.sum()), int(gv.sum()))
        lib.vpp_graph_destroy(g)
    print(";  ".join(out), flush=True)
