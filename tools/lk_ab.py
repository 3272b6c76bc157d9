import ctypes, os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd import pyr
from vpp_amd.synth import P, u8_image, DeviceImage, texture, translate
from vpp_amd import capi, image as vi
if os.environ.get("VPP_AMD_LIB"): capi.LIB_PATH = os.environ["VPP_AMD_LIB"]
V = ctypes.c_void_p
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
NR, NC, L, B = 1080, 1920, 3, 3
tex = texture(NR, NC, seed=5)
d1 = DeviceImage.from_host(u8_image(np.clip(np.rint(tex), 0, 255).astype(np.uint8))); d2 = DeviceImage.from_host(u8_image(np.clip(np.rint(translate(tex, 1.5, -2.25)), 0, 255).astype(np.uint8)))
p1 = pyr.device_pyramid(lib, d1, L, B); g1 = pyr.device_grad_pyramid(lib, p1[0], L, B, vi.F32); p2 = pyr.device_pyramid(lib, d2, L, B)
a1, ag, a2 = vi.desc_array(p1), vi.desc_array(g1), vi.desc_array(p2)
out = []
for n in (10000, 400000):
    k0 = torch.from_numpy(pyr.make_keypoints(pyr.grid_keypoints(NR, NC, n, margin=32)).view(np.uint8).reshape(-1).copy()).cuda(); k = k0.clone()
    ts = []
    for it in range(12):
        k.copy_(k0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.vpp_pyrlk_match(a1, ag, a2, L, V(k.data_ptr()), n, 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, st)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    out.append(f"n={n}: {min(ts[2:])*1e3:.1f} us crc {int(k.to(torch.int32).sum())}")
print("  ".join(out))
