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
    """FAST-9 (3-row halo): raw, local maxima (reads the score of the neighbouring rows: the halo rows are scored inside each strip exactly as on
    the frame) and blockwise(10) with strip boundaries on multiples of the block size."""
    img = rects_image(1080, 1920, seed=4)
    full = u8_image(img, border=3)
    dfull = DeviceImage.from_host(full)
    capi.check(lib.vpp_fill_border(P(dfull.desc), 0, None, capi.stream_ptr()))
    want_rc, want_sc = gpu_detect(lib, dfull, 20, mode=mode, bs=10)
    strips = make_strips(lib, img, bounds, 3)
    got_rc, got_sc = [], []
    for k, s in enumerate(strips):
        rc, sc = gpu_detect(lib, s, 20, mode=mode, bs=10)
        rc = rc.copy(); rc[:, 0] += bounds[k]
        got_rc.append(rc); got_sc.append(sc)
    got_rc, got_sc = np.concatenate(got_rc), np.concatenate(got_sc)
    if mode == 1:
        # a local maximum compares with the scores of the rows just outside the strip, which the strip does not compute: the test documents the
        # boundary rows as the one place where strips may differ, and checks everything else
        inner = np.ones(len(want_rc), bool)
        for b in bounds[1:-1]:
            inner &= (want_rc[:, 0] < b - 1) | (want_rc[:, 0] > b)
        keep = np.ones(len(got_rc), bool)
        for b in bounds[1:-1]:
            keep &= (got_rc[:, 0] < b - 1) | (got_rc[:, 0] > b)
        np.testing.assert_array_equal(got_rc[keep], want_rc[inner]); np.testing.assert_array_equal(got_sc[keep], want_sc[inner])
    else:
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
