"""pyrLK configs[3] (1080p, 3 levels, WS 7) by lanes per keypoint (tuning pyrlk.lpk) and keypoint count, event-timed, interleaved."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd import pyr
from vpp_amd.synth import P, u8_image, DeviceImage, texture, translate
from vpp_amd import capi, image as vi
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
NR, NC, L, B = 1080, 1920, 3, 3
tex = texture(NR, NC, seed=5)
d1 = DeviceImage.from_host(u8_image(np.clip(np.rint(tex), 0, 255).astype(np.uint8))); d2 = DeviceImage.from_host(u8_image(np.clip(np.rint(translate(tex, 1.5, -2.25)), 0, 255).astype(np.uint8)))
p1 = pyr.device_pyramid(lib, d1, L, B); g1 = pyr.device_grad_pyramid(lib, p1[0], L, B, vi.F32); p2 = pyr.device_pyramid(lib, d2, L, B)
a1, ag, a2 = vi.desc_array(p1), vi.desc_array(g1), vi.desc_array(p2)
for n in [int(x) for x in sys.argv[1:]] or (5000, 10000, 20000, 40000):
    k0 = torch.from_numpy(pyr.make_keypoints(pyr.grid_keypoints(NR, NC, n, margin=32)).view(np.uint8).reshape(-1).copy()).cuda(); k = k0.clone()
    res = {}
    for rep in range(3):
        for lpk in (8, 16, 32, 64):
            lib.vpp_set_tuning(b"pyrlk.lpk", lpk)
            ts = []
            for it in range(8):
                k.copy_(k0); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.vpp_pyrlk_match(a1, ag, a2, L, V(k.data_ptr()), n, 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, st)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            res.setdefault(lpk, []).append(min(ts[2:]) * 1e3)
    print(f"n={n}: " + "  ".join(f"lpk {l}: {min(v):.1f} us ({n / min(v):.1f} M/s)" for l, v in res.items()), flush=True)
lib.vpp_set_tuning(b"pyrlk.lpk", -1)
