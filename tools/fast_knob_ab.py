"""A/B of one tuning knob of vpp_fast9_detect (raw mode and the others) on a 4K frame, values interleaved, synchronous and graph-recorded asynchronous calls:
    python tools/fast_knob_ab.py fast9.write_rows_threads 256 128"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, rects_image, fast9_bench_frame
from vpp_amd import capi
V = ctypes.c_void_p
knob = sys.argv[1].encode(); values = [int(x) for x in sys.argv[2:]]
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
im = u8_image(fast9_bench_frame(), border=3)
im.view(with_border=True)[..., 0] = np.pad(im.view()[..., 0], 3, mode="symmetric")
d = DeviceImage.from_host(im)
cap = 3000000
rc = torch.zeros((cap, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros(cap, dtype=torch.int32, device="cuda"); n = ctypes.c_int(0)
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream(); sp = V(side.cuda_stream)
sums = {}
for v in values * 3:
    lib.vpp_set_tuning(knob, v)
    for mode in (0,):
        for _ in range(5):
            capi.check(lib.vpp_fast9_detect(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            lib.vpp_fast9_detect(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st)
        torch.cuda.synchronize(); sync_ms = (time.perf_counter() - t0) / 100 * 1e3
        sums.setdefault(v, set()).add((n.value, int(rc[:n.value].sum()), int(sc[:n.value].sum())))
        call = lambda: capi.check(lib.vpp_fast9_detect_async(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, V(cnt.data_ptr()), sp))
        call(); capi.check(lib.vpp_sync(sp))
        g = V(); capi.check(lib.vpp_graph_begin(sp))
        for _ in range(50): call()
        capi.check(lib.vpp_graph_end(sp, 0, ctypes.byref(g)))
        capi.check(lib.vpp_graph_launch(g, sp)); capi.check(lib.vpp_sync(sp))
        t0 = time.perf_counter()
        for _ in range(4): capi.check(lib.vpp_graph_launch(g, sp))
        capi.check(lib.vpp_sync(sp)); async_ms = (time.perf_counter() - t0) / 200 * 1e3
        lib.vpp_graph_destroy(g)
        print(f"{knob.decode()}={v} mode {mode}: synchronous call {sync_ms:.4f} ms, asynchronous (recorded) {async_ms:.4f} ms, n {n.value}", flush=True)
print("identical:", len(set.union(*sums.values())) == 1)
lib.vpp_set_tuning(knob, -1)
