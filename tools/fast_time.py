"""Time vpp_fast9_detect (RAW / LOCAL_MAXIMA / BLOCKWISE) on a 4K frame; also the rocprofv3 target for the K7-K9 kernels."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, rects_image, fast9_bench_frame
from vpp_amd import capi
V = ctypes.c_void_p
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
for kv in sys.argv[1:]:   # tuning knobs: name=value (e.g. fast9.raw_fused=0)
    k, v = kv.split("="); lib.vpp_set_tuning(k.encode(), int(v)); print("tuning", k, v)
im = u8_image(fast9_bench_frame(), border=3)
im.view(with_border=True)[..., 0] = np.pad(im.view()[..., 0], 3, mode="symmetric")
d = DeviceImage.from_host(im)
cap = 3000000
rc = torch.zeros((cap, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros(cap, dtype=torch.int32, device="cuda"); n = ctypes.c_int(0)
for mode in (0, 1, 2):
    for _ in range(5):
        capi.check(lib.vpp_fast9_detect(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        lib.vpp_fast9_detect(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st)
    torch.cuda.synchronize(); print("mode", mode, "n", n.value, (time.perf_counter() - t0) / 50 * 1e3, "ms/call")
# the asynchronous form, 50 calls recorded into one launch graph (no host round trip per call): device time per detection
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream(); sp = V(side.cuda_stream)
for mode in (0, 1, 2):
    call = lambda: capi.check(lib.vpp_fast9_detect_async(P(d.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, V(cnt.data_ptr()), sp))
    call(); capi.check(lib.vpp_sync(sp))
    g = V(); capi.check(lib.vpp_graph_begin(sp))
    for _ in range(50): call()
    capi.check(lib.vpp_graph_end(sp, 1, ctypes.byref(g)))
    for _ in range(3): capi.check(lib.vpp_graph_launch(g, sp)); capi.check(lib.vpp_sync(sp))
    ms = ctypes.c_float(0); capi.check(lib.vpp_graph_elapsed_ms(g, ctypes.byref(ms)))
    print("async mode", mode, "n", int(cnt.item()), ms.value / 50, "ms/call in a 50-call graph")
    lib.vpp_graph_destroy(g)
