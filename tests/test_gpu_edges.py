"""Edge cases through the C ABI: tiny / ragged images, empty keypoint sets, dead keypoints, out-of-domain tracks,
duplicate keypoints per cell, frames without corners — all against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

import pyr
from test_gpu_algos import gpu_detect, _run_pyrlk_both
from test_gpu_sdof import run_both
from test_oracle_algos import run_detect, lk_scene
from util import P, rand_image, HostImage, DeviceImage, u8_image, rects_image
from vpp_amd import capi, image as vi

pytestmark = pytest.mark.gpu
V = ctypes.c_void_p


@pytest.mark.parametrize("shape", [(1, 1), (1, 17), (17, 1), (2, 3), (3, 5000)])
def test_tiny_and_ragged_pixelwise_and_box(lib, orc, shape):
    for dtype, ch in ((vi.I32, 1), (vi.U8, 3)):
        b = rand_image(*shape, dtype, ch, seed=1, lo=0 if dtype == vi.I32 else None, hi=1000 if dtype == vi.I32 else None)
        c = rand_image(*shape, dtype, ch, seed=2, lo=0 if dtype == vi.I32 else None, hi=1000 if dtype == vi.I32 else None)
        want = b.like()
        orc.orc_pixelwise_binary(0, P(want.desc), P(b.desc), P(c.desc))
        db, dc, da = DeviceImage.from_host(b), DeviceImage.from_host(c), DeviceImage.from_host(b.like())
        capi.check(lib.vpp_pixelwise_binary(0, P(da.desc), P(db.desc), P(dc.desc), capi.stream_ptr()))
        np.testing.assert_array_equal(da.download().raw, want.raw)
    for border in (2, 4):
        src = rand_image(*shape, vi.U8, 3, border=border, seed=3, fill_border=True)
        want = src.like(border=0)
        orc.orc_box_filter(P(want.desc), P(src.desc), 5, 5)
        ds, dd = DeviceImage.from_host(src), DeviceImage.from_host(want.like())
        capi.check(lib.vpp_box_filter(P(dd.desc), P(ds.desc), 5, 5, capi.stream_ptr()))
        np.testing.assert_array_equal(dd.download().raw, want.raw)


@pytest.mark.parametrize("shape", [(3, 3), (4, 7), (5, 4), (65, 66)])
def test_small_pyramids(lib, orc, shape):
    img = rand_image(*shape, vi.U8, 1, seed=4)
    # border 2 <= the coarsest level's size: a mirror border wider than the image reads border pixels whose value depends on
    # the fill order in the reference (fill.hh:60-83), which no caller relies on
    hp = pyr.host_pyramid(orc, img, 2, 2)
    dp = pyr.device_pyramid(lib, DeviceImage.from_host(img), 2, 2)
    for h, d in zip(hp, dp):
        np.testing.assert_array_equal(d.download().raw, h.raw)


def test_fast9_degenerate_frames(lib, orc):
    flat = u8_image(np.full((70, 130), 77, np.uint8), border=3)
    orc.orc_fill_border(P(flat.desc), 0, None)
    for mode in (0, 1, 2):
        rc, sc = gpu_detect(lib, DeviceImage.from_host(flat), 20, mode=mode)
        assert len(rc) == 0 and len(sc) == 0
    # th = 0 on noise: almost every pixel is a corner; tiny image smaller than one tile; block size larger than the image
    rng = np.random.default_rng(0)
    noisy = u8_image(rng.integers(0, 256, size=(9, 11)).astype(np.uint8), border=3)
    orc.orc_fill_border(P(noisy.desc), 0, None)
    for mode, bs in ((0, 10), (1, 10), (2, 3), (2, 50)):
        for th in (0, 5, 255):
            w_rc, w_sc = run_detect(orc, noisy, th, mode=mode, bs=bs)
            g_rc, g_sc = gpu_detect(lib, DeviceImage.from_host(noisy), th, mode=mode, bs=bs)
            np.testing.assert_array_equal(g_rc, w_rc); np.testing.assert_array_equal(g_sc, w_sc)


def test_pyrlk_empty_dead_and_escaping_keypoints(lib, orc):
    f1, f2, kps = lk_scene(120, 160, 40)
    # no keypoints at all: nothing happens
    st = lib.vpp_pyrlk_match(None, None, None, 0, None, 0, 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, None)
    assert st in (capi.OK, capi.ERR_INVALID_ARG)
    kps["age"][:] = 0                                   # every keypoint dead: all records untouched
    got, want, _, _ = _run_pyrlk_both(lib, orc, f1, f2, kps)
    np.testing.assert_array_equal(got.view(np.uint8), kps.view(np.uint8))
    _, _, kps = lk_scene(120, 160, 40)
    kps["pos_r"][0], kps["pos_c"][0] = 0.2, 0.3        # window mostly outside: partially valid offsets, may be removed
    kps["pos_r"][1], kps["pos_c"][1] = 119.6, 159.9
    kps["pos_r"][2], kps["pos_c"][2] = 60.0, 0.0
    got, want, gd, wd = _run_pyrlk_both(lib, orc, f1, f2, kps, B=6)
    np.testing.assert_array_equal(got["age"], want["age"])
    alive = want["age"] > 0
    for f in ("pos_r", "pos_c", "vel_r", "vel_c"):
        np.testing.assert_allclose(got[f][alive], want[f][alive], rtol=1e-4, atol=1e-4 if f.startswith("pos") else 0.0)


def test_sdof_single_and_duplicate_keypoints(lib, orc):
    from test_gpu_sdof import flow_scene
    f1, f2, _ = flow_scene(60, 80)
    for kps in (np.array([[30, 40]], np.int32),
                np.array([[30, 40], [31, 41], [30, 40], [0, 0], [59, 79], [59, 0]], np.int32)):  # several per cell + the four corners
        got, want = run_both(lib, orc, f1, f2, kps, 7, 3, 0, 2, 5)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)


def test_8k_frames_box_add_ingest(lib):
    """Twice the BASELINE frame size in each direction (7680 x 4320): index arithmetic and grid sizes beyond the bench shapes.
    Checked through properties: sampled pixels against numpy, row sums, and the fused ingest against its two steps' definition."""
    nr, nc = 4320, 7680
    rng = np.random.default_rng(3)
    src = rand_image(nr, nc, vi.U8, 3, border=2, seed=41, fill_border=True)
    dsrc = DeviceImage.from_host(src); ddst = DeviceImage(nr, nc, vi.U8, 3)
    capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), 5, 5, capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    out = ddst.download().view()
    s = src.view(with_border=True).astype(np.int64)
    for r, c in [(0, 0), (nr - 1, nc - 1), (0, nc - 1), (nr - 1, 0)] + [(int(rng.integers(nr)), int(rng.integers(nc))) for _ in range(200)]:
        np.testing.assert_array_equal(out[r, c], s[r:r + 5, c:c + 5].sum(axis=(0, 1)) // 25)
    # int32 add: A = B + C everywhere (image_add.cc:21-28)
    b = rand_image(nr, nc, vi.I32, seed=5, lo=0, hi=2**30 - 1)
    db = DeviceImage.from_host(b); da = DeviceImage(nr, nc, vi.I32)
    capi.check(lib.vpp_pixelwise_binary(0, P(da.desc), P(db.desc), P(db.desc), capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    np.testing.assert_array_equal(da.download().view(), b.view() * 2)
    # frame ingest: gray = (r + g + b) / 3, border = gray of the mirrored pixel
    dg = DeviceImage(nr, nc, vi.U8, 1, border=3)
    rgb = rand_image(nr, nc, vi.U8, 3, border=0, seed=43)
    drgb = DeviceImage.from_host(rgb)
    capi.check(lib.vpp_rgb_to_graylevel(P(dg.desc), P(drgb.desc), 1, capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    want = (rgb.view().astype(np.int32).sum(axis=2) // 3).astype(np.uint8)
    np.testing.assert_array_equal(dg.download().view(with_border=True)[..., 0], np.pad(want, 3, mode="symmetric"))


def test_8k_fast9_matches_oracle(lib, orc):
    """FAST-9 on a 7680 x 4320 frame (four times the BASELINE frame), raw and blockwise, bit-exact against the oracle."""
    im = u8_image(rects_image(4320, 7680, seed=9), border=3)
    im.view(with_border=True)[..., 0] = np.pad(im.view()[..., 0], 3, mode="symmetric")
    d = DeviceImage.from_host(im)
    for mode in (0, 2):
        want_rc, want_sc = run_detect(orc, im, 25, mode=mode, bs=10, compat=0, cap=8000000)
        got_rc, got_sc = gpu_detect(lib, d, 25, mode=mode, bs=10, compat=0, cap=8000000)
        assert len(want_rc) > 10000
        np.testing.assert_array_equal(got_rc, want_rc)
        np.testing.assert_array_equal(got_sc, want_sc)
