import sys, time, ctypes
sys.path.insert(0, '/root/repo')
import torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
lib = capi.lib(); capi.check(lib.vpp_init(0))
NR, NC = 2160, 3840
src_h = rand_image(NR, NC, vi.U8, 3, border=2, seed=3, align=16)
ns = 8
srcs = [DeviceImage.from_host(src_h) for _ in range(ns)]; dsts = [DeviceImage(NR, NC, vi.U8, 3, 0, 16) for _ in range(ns)]
side = torch.cuda.Stream(); sp = ctypes.c_void_p(side.cuda_stream)
K = 20
gh = ctypes.c_void_p()
torch.cuda.synchronize()
capi.check(lib.vpp_graph_begin(sp))
for i in range(K): lib.vpp_box_filter(P(dsts[i % ns].desc), P(srcs[i % ns].desc), 5, 5, sp)
capi.check(lib.vpp_graph_end(sp, 1, ctypes.byref(gh)))
for _ in range(200): lib.vpp_graph_launch(gh, sp)
torch.cuda.synchronize()
def region(how):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.vpp_graph_launch(gh, sp)
    if how == "sync": torch.cuda.synchronize()
    elif how == "stream_sync": side.synchronize()
    elif how == "vpp_sync": lib.vpp_sync(sp)
    else:
        while not side.query(): pass
    return (time.perf_counter() - t0) / K * 1e6
for how in ("sync", "stream_sync", "vpp_sync", "spin"):
    v = sorted(region(how) for _ in range(15))
    print(f"{how:12s}: median {v[7]:.2f} us/step, best {v[0]:.2f}")
