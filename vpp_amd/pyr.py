"""Pyramid construction over the C ABI mirroring pyramid2d<V> (vpp/core/pyramid.hh:126-198), and keypoint record helpers
(plumbing for bench.py, the tools and the tests)."""
import ctypes

import numpy as np

from vpp_amd import capi
from vpp_amd import image as vi
from vpp_amd.image import DeviceImage

P = ctypes.byref

def level_dims(nr, nc, nlevels):
    dims = []
    for _ in range(nlevels):
        dims.append((nr, nc))
        nr, nc = 1 + nr // 2, 1 + nc // 2  # pyramid.hh:140,154 with factor 2
    return dims


def device_pyramid(lib, dimg, nlevels, border, st=None):
    """pyramid2d<V>(img, nlevels, 2, _border = border) through vpp_pyramid_build (one launch for u8 x1 pyramids of 2-3 levels)."""
    st = st or capi.stream_ptr()
    levels = [DeviceImage(nr, nc, dimg.dtype, dimg.channels, border) for nr, nc in level_dims(dimg.nrows, dimg.ncols, nlevels)]
    capi.check(lib.vpp_pyramid_build(vi.desc_array(levels), nlevels, P(dimg.desc), st))
    return levels


def device_grad_pyramid(lib, level0, nlevels, border, dtype=vi.F32, st=None):
    """scharr(level0) + fill_border_mirror + propagate_level0 through vpp_scharr_pyramid_build (one launch for 2-3 levels)."""
    st = st or capi.stream_ptr()
    levels = [DeviceImage(nr, nc, dtype, 2, border) for nr, nc in level_dims(level0.nrows, level0.ncols, nlevels)]
    capi.check(lib.vpp_scharr_pyramid_build(vi.desc_array(levels), nlevels, P(level0.desc), st))
    return levels


class Keypoint(ctypes.Structure):
    _fields_ = [("pos_r", ctypes.c_float), ("pos_c", ctypes.c_float), ("vel_r", ctypes.c_float), ("vel_c", ctypes.c_float), ("age", ctypes.c_int32)]


KP_DTYPE = np.dtype([("pos_r", "<f4"), ("pos_c", "<f4"), ("vel_r", "<f4"), ("vel_c", "<f4"), ("age", "<i4")])


def make_keypoints(pos):
    """keypoint<float>(pos): velocity 0, age 1 (keypoint_container.hh:16-18)."""
    k = np.zeros(len(pos), KP_DTYPE)
    k["pos_r"], k["pos_c"], k["age"] = pos[:, 0], pos[:, 1], 1
    return k


def grid_keypoints(nr, nc, n, margin=32, seed=5):
    """n keypoints on a jittered grid, >= margin px from every edge (BASELINE config 4)."""
    rng = np.random.default_rng(seed)
    gr = int(np.ceil(np.sqrt(n * nr / nc)))
    gc = int(np.ceil(n / gr))
    rr = np.linspace(margin + 2, nr - margin - 3, gr)
    cc = np.linspace(margin + 2, nc - margin - 3, gc)
    g = np.stack(np.meshgrid(rr, cc, indexing="ij"), -1).reshape(-1, 2)[:n]
    g = g + rng.uniform(-1.5, 1.5, size=g.shape)
    return g.astype(np.float32)
