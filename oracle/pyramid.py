"""pyramid2d<V> / gradient pyramid construction on the CPU oracle (test infrastructure, like everything under oracle/):
used by the tests and by bench.py's cpu_baseline leg."""
import ctypes

from vpp_amd import image as vi
from vpp_amd.image import HostImage
from vpp_amd.pyr import level_dims

P = ctypes.byref


def host_pyramid(orc, img, nlevels, border):
    """pyramid2d<V>(img, nlevels, 2, _border = border): copy into level 0 + propagate_level0 (pyramid.hh:146-198)."""
    levels = [HostImage(nr, nc, img.dtype, img.channels, border) for nr, nc in level_dims(img.nrows, img.ncols, nlevels)]
    assert orc.orc_copy(P(levels[0].desc), P(img.desc), 0) == 0
    assert orc.orc_fill_border(P(levels[0].desc), 0, None) == 0
    for l in range(1, nlevels):
        assert orc.orc_pyr_down(P(levels[l].desc), P(levels[l - 1].desc)) == 0
    return levels


def host_grad_pyramid(orc, level0, nlevels, border, dtype=vi.F32):
    """scharr(pyr[0], grad[0]); grad.propagate_level0() (pyrlk_opencv_comparison.cc:56-60, lucas_kanade.hpp:152-157)."""
    levels = [HostImage(nr, nc, dtype, 2, border) for nr, nc in level_dims(level0.nrows, level0.ncols, nlevels)]
    assert orc.orc_scharr(P(levels[0].desc), P(level0.desc)) == 0
    assert orc.orc_fill_border(P(levels[0].desc), 0, None) == 0
    for l in range(1, nlevels):
        assert orc.orc_pyr_down(P(levels[l].desc), P(levels[l - 1].desc)) == 0
    return levels


