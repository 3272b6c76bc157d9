"""Single-GPU timing of the strip-sharded image-space phases (SURVEY 8e bullet 2): FAST-9 on a 4K frame as one launch set vs 2 / 4 row strips
on their own streams with the halo copy in front (the device-to-device form of vpp_halo_exchange)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, HostImage, rects_image
from vpp_amd import capi, image as vi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0))
img = rects_image(2160, 3840, seed=4)
def detect(d, stream, bufs):
    rc, sc, n = bufs
    capi.check(lib.vpp_fast9_detect(P(d.desc), 20, None, 2, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), rc.shape[0], P(n), stream))
def bufs(cap=600000):
    return torch.zeros((cap, 2), dtype=torch.int32, device="cuda"), torch.zeros(cap, dtype=torch.int32, device="cuda"), ctypes.c_int(0)
for nstrips in (1, 2, 4):
    bounds = [2160 * k // nstrips // 20 * 20 for k in range(nstrips)] + [2160]
    strips, streams, bb = [], [torch.cuda.Stream() for _ in range(nstrips)], [bufs() for _ in range(nstrips)]
    for k in range(nstrips):
        h = HostImage(bounds[k + 1] - bounds[k], 3840, vi.U8, 1, 3)
        h.view()[..., 0] = img[bounds[k]:bounds[k + 1]]
        strips.append(DeviceImage.from_host(h))
    def step():
        st0 = ctypes.c_void_p(streams[0].cuda_stream)
        for s in strips: lib.vpp_fill_border(P(s.desc), 0, None, st0)
        for k in range(nstrips - 1): lib.vpp_halo_copy(P(strips[k].desc), P(strips[k + 1].desc), 3, st0)
        ev = torch.cuda.Event(); ev.record(streams[0])
        for k in range(nstrips):
            streams[k].wait_event(ev)
            detect(strips[k], ctypes.c_void_p(streams[k].cuda_stream), bb[k])   # vpp_fast9_detect waits for its own count
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize()
    print(f"FAST-9 blockwise(10) 4K, {nstrips} strip(s): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per frame, {sum(b[2].value for b in bb)} keypoints")
