"""Row-strip sharding of the image-space phases (SURVEY 8e bullet 2) on ONE GPU: the frame is cut into strips, each strip is an ordinary
bordered image on its own stream, the halo rows travel with vpp_halo_copy (the one-process form of vpp_halo_exchange), and the stencil
kernels run on the strips.  The union of the strip results must be IDENTICAL to the full-frame result (set, order and scores)."""
import ctypes

import numpy as np
import pytest
import torch

from util import P, u8_image, rand_image, rects_image, DeviceImage, HostImage
from test_gpu_algos import gpu_detect
from vpp_amd import capi, image as vi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    l = capi.lib()
    capi.check(l.vpp_init(0))
    return l


def make_strips(lib, frame, bounds, border, channels=1):
    """frame: (nr, nc[, ch]) uint8; strips = bordered device images of rows [bounds[k], bounds[k+1]); true-edge borders mirrored, inner halos exchanged"""
    strips = []
    for k in range(len(bounds) - 1):
        h = HostImage(bounds[k + 1] - bounds[k], frame.shape[1], vi.U8, channels, border)
        h.view()[...] = frame[bounds[k]:bounds[k + 1]].reshape(h.view().shape)
        strips.append(DeviceImage.from_host(h))
    st = capi.stream_ptr()
    for s in strips:
        capi.check(lib.vpp_fill_border(P(s.desc), 0, None, st))           # every border mirrored first (right at the frame's edges) ...
    for k in range(len(strips) - 1):
        capi.check(lib.vpp_halo_copy(P(strips[k].desc), P(strips[k + 1].desc), border, st))   # ... then the inner edges get the neighbour's rows
    return strips


@pytest.mark.parametrize("bounds", [(0, 540, 1080), (0, 200, 540, 1080), (0, 20, 1080)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fast9_on_strips_equals_the_full_frame(lib, bounds, mode):
    """FAST-9 on row strips: raw and blockwise(10) (strip boundaries on multiples of the block size) need the 3 rows around a pixel; a local
    maximum also compares with the SCORES of the row above / below, so for that mode the strips carry a 4-row halo and the detection runs on
    the view [r0 - 1, r1 + 1) of the strip (border 3 inside the 4 exchanged rows); keypoints of the two extra rows are dropped.  All three
    modes are then identical to the full frame in set, order and scores — no exception at the inner edges."""
    img = rects_image(1080, 1920, seed=4)
    full = u8_image(img, border=3)
    dfull = DeviceImage.from_host(full)
    capi.check(lib.vpp_fill_border(P(dfull.desc), 0, None, capi.stream_ptr()))
    want_rc, want_sc = gpu_detect(lib, dfull, 20, mode=mode, bs=10)
    halo = 4 if mode == 1 else 3
    strips = make_strips(lib, img, bounds, halo)
    got_rc, got_sc = [], []
    for k, s in enumerate(strips):
        up, down = (1 if k > 0 else 0, 1 if k + 1 < len(strips) else 0) if mode == 1 else (0, 0)
        view = s
        if mode == 1:   # a descriptor over the same memory: one more row at each inner edge, border 3 (the reference's sub-image semantics, imageNd.hpp:325-341)
            class View:
                pass
            view = View()
            d = s.desc
            view.desc = vi.ImageDesc(d.first_pixel - up * d.pitch, d.nrows + up + down, d.ncols, d.pitch, 3, d.dtype, d.channels)
        rc, sc = gpu_detect(lib, view, 20, mode=mode, bs=10)
        rc = rc.copy(); rc[:, 0] += bounds[k] - up
        own = (rc[:, 0] >= bounds[k]) & (rc[:, 0] < bounds[k + 1])
        got_rc.append(rc[own]); got_sc.append(sc[own])
    got_rc, got_sc = np.concatenate(got_rc), np.concatenate(got_sc)
    np.testing.assert_array_equal(got_rc, want_rc)
    np.testing.assert_array_equal(got_sc, want_sc)
    assert len(want_rc) > 1000


def test_box5x5_on_strips_equals_the_full_frame(lib):
    """box_nbh2d 5x5 on vuchar3 (2-row halo), 4K, four strips."""
    src = rand_image(2160, 3840, vi.U8, 3, border=2, seed=3, align=16)
    dsrc = DeviceImage.from_host(src)
    capi.check(lib.vpp_fill_border(P(dsrc.desc), 0, None, capi.stream_ptr()))
    dfull = DeviceImage(2160, 3840, vi.U8, 3, 0, 16)
    capi.check(lib.vpp_box_filter(P(dfull.desc), P(dsrc.desc), 5, 5, capi.stream_ptr()))
    bounds = (0, 500, 1080, 1700, 2160)
    strips = make_strips(lib, src.view(), bounds, 2, channels=3)
    outs = []
    for s in strips:
        o = DeviceImage(s.nrows, 3840, vi.U8, 3, 0, 16)
        capi.check(lib.vpp_box_filter(P(o.desc), P(s.desc), 5, 5, capi.stream_ptr()))
        outs.append(o)
    torch.cuda.synchronize()
    got = np.concatenate([o.download().view() for o in outs])
    np.testing.assert_array_equal(got, dfull.download().view())
