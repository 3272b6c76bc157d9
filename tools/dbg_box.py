import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, rand_image, DeviceImage
from vpp_amd import capi, image as vi
from oracle import binding
lib = capi.lib(); capi.check(lib.vpp_init(0)); orc = binding.load()
for (shape, ch, g) in [((67, 131), 3, 0), ((67, 131), 3, 1), ((40, 64), 1, 0), ((40, 64), 4, 0)]:
    src = rand_image(*shape, vi.U8, ch, border=2, seed=3, align=16, fill_border=True)
    want = src.like(border=0)
    orc.orc_box_filter(P(want.desc), P(src.desc), 5, 5)
    lib.vpp_set_tuning(b"box.force_generic", g)
    dsrc = DeviceImage.from_host(src); ddst = DeviceImage.from_host(src.like(border=0))
    capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), 5, 5, capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
    got = ddst.download().view().reshape(shape[0], -1).astype(int); w = want.view().reshape(shape[0], -1).astype(int)
    bad = got != w
    print("shape", shape, "ch", ch, "generic", g, "bad", bad.sum(), "of", bad.size)
    if bad.any():
        print(" bad rows:", np.unique(np.nonzero(bad)[0])[:40])
        print(" bad byte cols mod 16 hist:", np.bincount(np.nonzero(bad)[1] % 16, minlength=16))
        r = np.nonzero(bad)[0][0]
        print(" row", r, "got ", got[r, :32]); print(" row", r, "want", w[r, :32])
        print(" diff", (got - w)[r, :48])
