"""Debug: which part of the recorded flow call misbehaves when replayed after an eager call of the same layout."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from vpp_amd import capi
from vpp_amd.synth import P, flow_scene, u8_image, DeviceImage
from oracle import binding
lib = capi.lib(); capi.check(lib.vpp_init(0)); orc = binding.load(omp=False)
fa1, fa2, ka = flow_scene(240, 320, seed=51, spacing=5)
i1, i2 = u8_image(fa1, border=3), u8_image(fa2, border=3)
n = len(ka)
want = (np.zeros((n, 2), np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint8))
orc.orc_semi_dense_optical_flow(P(i1.desc), P(i2.desc), ka.ctypes.data_as(ctypes.c_void_p), n, 9, 3, 0, 2, 5, want[0].ctypes.data_as(ctypes.c_void_p), want[1].ctypes.data_as(ctypes.c_void_p), want[2].ctypes.data_as(ctypes.c_void_p))
d1, d2, dk = DeviceImage.from_host(i1), DeviceImage.from_host(i2), torch.from_numpy(ka).cuda()
out = (torch.zeros((n, 2), dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda"))
for knobs in ({}, {b"sdof.self_cleaning_owner": 0}, {b"sdof.claim_up_front": 0}, {b"sdof.reset_up_front": 0}, {b"sdof.propagate": 1}):
    for k, v in knobs.items():
        lib.vpp_set_tuning(k, v)
    st = torch.cuda.Stream(); sp = ctypes.c_void_p(st.cuda_stream)
    def call():
        capi.check(lib.vpp_semi_dense_optical_flow(P(d1.desc), P(d2.desc), ctypes.c_void_p(dk.data_ptr()), n, 9, 3, 0, 2, 5, ctypes.c_void_p(out[0].data_ptr()), ctypes.c_void_p(out[1].data_ptr()), ctypes.c_void_p(out[2].data_ptr()), sp))
    def bad():
        capi.check(lib.vpp_sync(sp))
        r = [int((g.cpu().numpy() != w).sum()) for g, w in zip(out, want)]
        for g in out: g.zero_()
        torch.cuda.synchronize()
        return r
    torch.cuda.synchronize()
    call(); r0 = bad()
    g = ctypes.c_void_p(); capi.check(lib.vpp_graph_begin(sp)); call(); capi.check(lib.vpp_graph_end(sp, 0, ctypes.byref(g)))
    rep = lambda: capi.check(lib.vpp_graph_launch(g, sp))
    seq = []
    rep(); seq.append(("replay", bad()))
    rep(); seq.append(("replay", bad()))
    call(); seq.append(("eager", bad()))
    rep(); seq.append(("replay after eager", bad()))
    rep(); seq.append(("replay", bad()))
    call(); call(); seq.append(("eager x2", bad()))
    rep(); rep(); seq.append(("replay x2", bad()))
    print(knobs, "first eager", r0, seq, flush=True)
    lib.vpp_graph_destroy(g)
    for k in knobs:
        lib.vpp_set_tuning(k, -1)
