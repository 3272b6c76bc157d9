"""rgb_to_graylevel / frame ingest and the video_extruder re-detection mask: oracle restatement against the reference's own
test (tests/colorspace_conversions.cc), against numpy, and — where oracle/_ref exists — against the reference headers."""
import ctypes

import numpy as np
import pytest

from util import P, rand_image, HostImage
from vpp_amd import image as vi


def test_reference_unit_test_restated(orc):
    """tests/colorspace_conversions.cc: i1(p) = (i, i, i) with a wrapping uchar counter -> gray == i."""
    src = HostImage(100, 100, vi.U8, 3)
    v = src.view()
    v[...] = (np.arange(100 * 100, dtype=np.uint32) & 255).astype(np.uint8).reshape(100, 100, 1)
    dst = HostImage(100, 100, vi.U8, 1)
    assert orc.orc_rgb_to_graylevel(P(dst.desc), P(src.desc), 0) == 0
    np.testing.assert_array_equal(dst.view()[..., 0], v[..., 0])


@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("border", [0, 3])
def test_gray_is_integer_mean_of_three(orc, ch, border):
    src = rand_image(37, 53, vi.U8, ch, border=border, seed=11, fill_border=True)
    dst = HostImage(37, 53, vi.U8, 1, border)
    assert orc.orc_rgb_to_graylevel(P(dst.desc), P(src.desc), 0) == 0
    s = src.view(with_border=True).astype(np.int32)
    np.testing.assert_array_equal(dst.view(with_border=True)[..., 0], ((s[..., 0] + s[..., 1] + s[..., 2]) // 3).astype(np.uint8))


def test_ingest_border_is_the_mirror_of_the_gray_interior(orc):
    src = rand_image(29, 41, vi.U8, 3, border=0, seed=12)
    dst = HostImage(29, 41, vi.U8, 1, 3)
    assert orc.orc_rgb_to_graylevel(P(dst.desc), P(src.desc), 1) == 0
    inner = dst.view()[..., 0]
    np.testing.assert_array_equal(dst.view(with_border=True)[..., 0], np.pad(inner, 3, mode="symmetric"))


@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("shape,border,mirror", [((37, 53), 3, 0), ((37, 53), 0, 0), ((40, 64), 3, 1), ((19, 23), 5, 1), ((100, 100), 2, 1)])
def test_oracle_is_the_reference(orc, ref, ch, shape, border, mirror):
    src = rand_image(*shape, vi.U8, ch, border=0 if mirror else border, seed=13, fill_border=True)
    a, b = HostImage(*shape, vi.U8, 1, border), HostImage(*shape, vi.U8, 1, border)
    assert ref.ref_rgb_to_graylevel(P(a.desc), P(src.desc), mirror) == 0
    assert orc.orc_rgb_to_graylevel(P(b.desc), P(src.desc), mirror) == 0
    np.testing.assert_array_equal(a.view(with_border=True), b.view(with_border=True))


def test_keypoint_mask(orc):
    m = HostImage(60, 80, vi.U8, 1, 10)
    rc = np.array([[0, 0], [30, 40], [59, 79], [12, 75]], np.int32)
    assert orc.orc_keypoint_mask(P(m.desc), rc.ctypes.data_as(ctypes.c_void_p), len(rc), 10) == 0
    want = np.ones((80, 100), np.uint8)
    for r, c in rc:
        want[r + 10 - 10:r + 10 + 10, c + 10 - 10:c + 10 + 10] = 0
    np.testing.assert_array_equal(m.view(with_border=True)[..., 0], want)


def test_lbp_reference_unit_test_restated(orc):
    """tests/lbp.cc: the 3x3 pattern around V(1,1) = 1 gives 0b10101110."""
    v = HostImage(3, 3, vi.U8, 1, 1)
    a = v.view()[..., 0]
    a[...] = [[0, 2, 2], [2, 1, 0], [2, 0, 2]]
    out = HostImage(3, 3, vi.U8, 1)
    assert orc.orc_lbp_transform(P(out.desc), P(v.desc)) == 0
    assert out.view()[1, 1, 0] == 0b10101110


@pytest.mark.parametrize("shape,border", [((3, 3), 1), ((37, 53), 1), ((40, 64), 3), ((1, 7), 2)])
def test_lbp_oracle_is_the_reference(orc, ref, shape, border):
    src = rand_image(*shape, vi.U8, 1, border=border, seed=14, fill_border=True)
    a, b = HostImage(*shape, vi.U8, 1), HostImage(*shape, vi.U8, 1)
    assert ref.ref_lbp_transform(P(a.desc), P(src.desc)) == 0
    assert orc.orc_lbp_transform(P(b.desc), P(src.desc)) == 0
    np.testing.assert_array_equal(a.view(), b.view())
    s = src.view(with_border=True)[..., 0].astype(np.int32)
    bb = border
    c = s[bb:bb + shape[0], bb:bb + shape[1]]
    want = np.zeros(shape, np.int32)
    for k, (dr, dc) in enumerate([(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]):
        want += (s[bb + dr:bb + dr + shape[0], bb + dc:bb + dc + shape[1]] > c) << k
    np.testing.assert_array_equal(b.view()[..., 0], want.astype(np.uint8))


@pytest.mark.parametrize("shape,th,border", [((30, 41), 20, 3), ((64, 64), 5, 3), ((7, 9), 0, 4), ((50, 33), -2, 3), ((40, 70), 300, 3)])
def test_fast9_dense_oracle_is_the_reference(orc, ref, shape, th, border):
    """FAST_internals::fast_detector9(A, B, th) (fast.hpp:511-551): restatement == the reference template, u8 and int outputs."""
    from util import rects_image, u8_image
    src = u8_image(rects_image(*shape, seed=shape[0] + th), border=border)
    src.view(with_border=True)[..., 0] = np.pad(src.view()[..., 0], border, mode="symmetric")
    for dt in (vi.U8, vi.I32):
        a, b = HostImage(*shape, dt, 1), HostImage(*shape, dt, 1)
        assert ref.ref_fast9_dense(P(a.desc), P(src.desc), th) == 0
        assert orc.orc_fast9_dense(P(b.desc), P(src.desc), th) == 0
        np.testing.assert_array_equal(a.view(), b.view())
        assert set(np.unique(b.view())) <= {0, 1}
    small = HostImage(*shape, vi.U8, 1, 2)
    assert orc.orc_fast9_dense(P(b.desc), P(small.desc), th) != 0  # the ring needs a border of 3


@pytest.mark.parametrize("dtype", [vi.U8, vi.I32, vi.F32, vi.I16])
@pytest.mark.parametrize("shape,bs", [((20, 30), 10), ((23, 31), 10), ((9, 9), 4), ((5, 40), 7), ((16, 16), 1)])
def test_blockwise_maxima_filter_oracle(orc, dtype, shape, bs):
    """fast.hpp:577-614 against an independent numpy statement: per block, zero everything but the first strict maximum > 0."""
    img = rand_image(*shape, dtype, 1, border=1, seed=bs + shape[1], lo=-3 if dtype in (vi.I32, vi.F32, vi.I16) else 0, hi=6)
    a = img.view()[..., 0].copy()
    assert orc.orc_blockwise_maxima_filter(P(img.desc), bs) == 0
    want = np.zeros_like(a)
    for r in range(0, shape[0], bs):
        for c in range(0, shape[1], bs):
            blk = a[r:r + bs, c:c + bs]
            if blk.max() > 0:
                k = int(np.argmax(blk))  # first occurrence of the maximum in row-major order
                want[r + k // blk.shape[1], c + k % blk.shape[1]] = blk.max()
    np.testing.assert_array_equal(img.view()[..., 0], want)
