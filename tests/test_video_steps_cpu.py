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
