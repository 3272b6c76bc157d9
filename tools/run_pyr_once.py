"""vpp_pyramid_build (u8, 3 levels, 4K, border 3), 10 plain launches: the target of tools/pyr_pmc.sh (rocprofv3 --pmc)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, texture
from vpp_amd import capi, image as vi, pyr
lib = capi.lib(); capi.check(lib.vpp_init(0))
nr, nc, border = 2160, 3840, 3
f = np.clip(np.rint(texture(nr, nc, seed=5)), 0, 255).astype(np.uint8)
d = DeviceImage.from_host(u8_image(f))
lv = [DeviceImage(a, b, vi.U8, 1, border) for a, b in pyr.level_dims(nr, nc, 3)]
dl = vi.desc_array(lv)
for i in range(10): lib.vpp_pyramid_build(dl, 3, P(d.desc), capi.stream_ptr())
torch.cuda.synchronize()
