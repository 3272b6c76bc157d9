"""CPU tests: the oracle restatement against independent numpy formulations and the reference's own
known-answer checks (benchmarks/image_add.cc:21-28, benchmarks/box_5x5_filter2.cc:26-41, tests/border.cc)."""
import ctypes

import numpy as np
import pytest

from util import P, rand_image, HostImage
from vpp_amd import image as vi


def test_layout_matches_survey_appendix_c():
    # SURVEY.md Appendix C, computed from imageNd.hpp:151-196
    assert vi.layout(1080, 1920, 4, 0, 16) == (7680, 8294400, 0)
    assert vi.layout(2160, 3840, 3, 2, 16) == (11552, 24998528, 23120)
    assert vi.layout(2160, 3840, 3, 2, 32) == (11584, 25067776, 23200)
    assert vi.layout(2160, 3840, 1, 3, 32) == (3904, 8456064, 11744)
    assert vi.layout(1080, 1920, 8, 3, 16)[0] == 15424
    assert vi.layout(2160, 3840, 1, 18, 32) == (3904, 8573184, 70304)


@pytest.mark.parametrize("dtype,ch", [(vi.I32, 1), (vi.U8, 3), (vi.F32, 2), (vi.I16, 1)])
def test_add_checker(orc, dtype, ch):
    lo, hi = (0, 2**30 - 1) if dtype == vi.I32 else (None, None)
    b = rand_image(37, 53, dtype, ch, seed=1, lo=lo, hi=hi)
    c = rand_image(37, 53, dtype, ch, seed=2, lo=lo, hi=hi)
    a = b.like()
    assert orc.orc_pixelwise_binary(0, P(a.desc), P(b.desc), P(c.desc)) == 0
    want = (b.view() + c.view()).astype(a.view().dtype)  # numpy wraps like the compiled reference
    np.testing.assert_array_equal(a.view(), want)


def test_add_i32_wraps(orc):
    b = rand_image(8, 8, vi.I32, seed=1, lo=2**31 - 100, hi=2**31 - 1)
    c = rand_image(8, 8, vi.I32, seed=2, lo=200, hi=300)
    a = b.like()
    orc.orc_pixelwise_binary(0, P(a.desc), P(b.desc), P(c.desc))
    want = (b.view().astype(np.int64) + c.view()).astype(np.int32)
    np.testing.assert_array_equal(a.view(), want)


@pytest.mark.parametrize("mode", [0, 1])
def test_fill_border_vs_numpy_pad(orc, mode):
    im = rand_image(9, 13, vi.U8, 3, border=4, seed=3)
    orc.orc_fill_border(P(im.desc), mode, None)
    want = np.pad(im.view(), ((4, 4), (4, 4), (0, 0)), mode="symmetric" if mode == 0 else "edge")
    np.testing.assert_array_equal(im.view(with_border=True), want)


def test_fill_border_value_matches_reference_test(orc):
    # tests/border.cc:11-40: border filled with a value, interior untouched
    im = rand_image(5, 6, vi.I32, 1, border=2, seed=3)
    inner = im.view().copy()
    v = ctypes.c_int32(7)
    orc.orc_fill_border(P(im.desc), 2, P(v))
    full = im.view(with_border=True)
    np.testing.assert_array_equal(im.view(), inner)
    mask = np.ones(full.shape, bool)
    mask[2:-2, 2:-2] = False
    assert (full[mask] == 7).all()


@pytest.mark.parametrize("dtype,ch,R,C", [(vi.U8, 3, 5, 5), (vi.I32, 1, 5, 5), (vi.U8, 1, 3, 3), (vi.F32, 1, 5, 5), (vi.U8, 4, 5, 5)])
def test_box_filter_vs_numpy(orc, dtype, ch, R, C):
    lo, hi = (0, 999) if dtype == vi.I32 else (None, None)  # box_5x5_filter.cc:191 uses rand() % 1000
    src = rand_image(31, 45, dtype, ch, border=2, seed=3, lo=lo, hi=hi, fill_border=True)
    dst = src.like(border=0)
    assert orc.orc_box_filter(P(dst.desc), P(src.desc), R, C) == 0
    full = src.view(with_border=True)
    acc = np.zeros(dst.view().shape, dtype=np.float32 if dtype == vi.F32 else np.int64)
    b = 2
    for dr in range(-(R // 2), R // 2 + 1):
        for dc in range(-(C // 2), C // 2 + 1):
            acc = acc + full[b + dr:b + dr + 31, b + dc:b + dc + 45]
    if dtype == vi.F32:
        want = acc / np.float32(R * C)
        np.testing.assert_array_equal(dst.view(), want.astype(np.float32))
    else:
        want = (np.trunc(acc / (R * C))).astype(dst.view().dtype)  # C++ truncating division
        np.testing.assert_array_equal(dst.view(), want)


def _np_lowpass_down(img):
    """independent numpy statement of pyramid.hh:12-81 for integer images (img: HxWxC int array, mirror borders)."""
    k = np.array([1, 4, 6, 4, 1])
    p = np.pad(img.astype(np.int64), ((0, 0), (2, 2), (0, 0)), mode="symmetric")
    h = sum(k[i] * p[:, i:i + img.shape[1]] for i in range(5))
    h = np.trunc(h / 16).astype(np.int64)
    p = np.pad(h, ((2, 2), (0, 0), (0, 0)), mode="symmetric")
    v = sum(k[i] * p[i:i + img.shape[0]] for i in range(5))
    v = np.trunc(v / 16).astype(np.int64)
    nr, nc = 1 + img.shape[0] // 2, 1 + img.shape[1] // 2
    out = np.zeros((nr, nc, img.shape[2]), np.int64)  # Q4: rows/cols past the end read the zero-filled border
    sub = v[::2, ::2]
    out[:sub.shape[0], :sub.shape[1]] = sub
    return out


@pytest.mark.parametrize("shape", [(40, 56), (41, 57), (40, 57)])
@pytest.mark.parametrize("dtype,ch", [(vi.U8, 1), (vi.I32, 2)])
def test_pyr_down_vs_numpy(orc, shape, dtype, ch):
    lo, hi = (-500, 500) if dtype == vi.I32 else (None, None)
    prev = rand_image(shape[0], shape[1], dtype, ch, border=3, seed=7, lo=lo, hi=hi)
    orc.orc_fill_border(P(prev.desc), 0, None)
    nxt = HostImage(1 + shape[0] // 2, 1 + shape[1] // 2, dtype, ch, border=3)
    assert orc.orc_pyr_down(P(nxt.desc), P(prev.desc)) == 0
    want = _np_lowpass_down(prev.view())
    np.testing.assert_array_equal(nxt.view(), want.astype(nxt.view().dtype))
    np.testing.assert_array_equal(nxt.view(with_border=True), np.pad(nxt.view(), ((3, 3), (3, 3), (0, 0)), mode="symmetric"))


@pytest.mark.parametrize("odt", [vi.F32, vi.I32])
def test_scharr_vs_numpy(orc, odt):
    src = rand_image(33, 47, vi.U8, 1, border=1, seed=9, fill_border=True)
    out = HostImage(33, 47, odt, 2)
    assert orc.orc_scharr(P(out.desc), P(src.desc)) == 0
    f = src.view(with_border=True)[..., 0].astype(np.int64)
    r1, r2, r3 = f[:-2], f[1:-1], f[2:]
    g0 = (3 * r3[:, :-2] + 10 * r3[:, 1:-1] + 3 * r3[:, 2:]) - (3 * r1[:, :-2] + 10 * r1[:, 1:-1] + 3 * r1[:, 2:])
    g1 = (3 * r1[:, 2:] + 10 * r2[:, 2:] + 3 * r3[:, 2:]) - (3 * r1[:, :-2] + 10 * r2[:, :-2] + 3 * r3[:, :-2])
    if odt == vi.F32:
        np.testing.assert_array_equal(out.view()[..., 0], (g0 / 32.0).astype(np.float32))
        np.testing.assert_array_equal(out.view()[..., 1], (g1 / 32.0).astype(np.float32))
    else:
        np.testing.assert_array_equal(out.view()[..., 0], np.trunc(g0 / 32.0).astype(np.int32))
        np.testing.assert_array_equal(out.view()[..., 1], np.trunc(g1 / 32.0).astype(np.int32))


def test_linear_interpolate_reference_test(orc):
    # tests/imageNd.cc:87-107: 2x2 image (0,10;20,30), border 1; interpolate(0.5,0.5) == int((10+20+30)/4.f) == 15
    im = HostImage(2, 2, vi.U8, 1, border=1)
    im.view()[..., 0] = [[0, 10], [20, 30]]
    out = (ctypes.c_float * 2)()
    assert orc.orc_linear_interpolate(P(im.desc), ctypes.c_float(0.5), ctypes.c_float(0.5), out) == 0
    assert out[0] == float(int((10 + 20 + 30) / 4.0))
