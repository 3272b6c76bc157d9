"""Pins the oracle restatement against the REAL reference: matt-42/vpp's own headers compiled unmodified against Eigen/iod
stand-ins (oracle/ref -> oracle/_ref/libvpp_ref.so, SURVEY.md §8c route A).  Skipped where the library was never built
(no /root/reference); the golden fixtures under tests/golden/ carry the same evidence everywhere (test_golden.py)."""
import ctypes

import numpy as np
import pytest

import pyr
from test_oracle_algos import run_detect, pyrlk_cc_fixture, lk_scene
from test_gpu_sdof import flow_scene
from util import P, rand_image, HostImage, rects_image, u8_image
from vpp_amd import image as vi


@pytest.mark.parametrize("dtype,ch", [(vi.I32, 1), (vi.U8, 3), (vi.F32, 1)])
def test_add(orc, ref, dtype, ch):
    lo, hi = (0, 2**30 - 1) if dtype == vi.I32 else (None, None)
    b, c = rand_image(37, 53, dtype, ch, seed=1, lo=lo, hi=hi), rand_image(37, 53, dtype, ch, seed=2, lo=lo, hi=hi)
    a1, a2 = b.like(), b.like()
    assert ref.ref_pixelwise_add(P(a1.desc), P(b.desc), P(c.desc)) == 0
    assert orc.orc_pixelwise_binary(0, P(a2.desc), P(b.desc), P(c.desc)) == 0
    np.testing.assert_array_equal(a1.raw, a2.raw)


@pytest.mark.parametrize("dtype,ch", [(vi.I32, 1), (vi.U8, 3), (vi.U8, 1)])
def test_box5x5(orc, ref, dtype, ch):
    lo, hi = (0, 999) if dtype == vi.I32 else (None, None)
    src = rand_image(61, 83, dtype, ch, border=2, seed=3, lo=lo, hi=hi, fill_border=True)
    d1, d2 = src.like(border=0), src.like(border=0)
    assert ref.ref_box_filter5x5(P(d1.desc), P(src.desc)) == 0
    assert orc.orc_box_filter(P(d2.desc), P(src.desc), 5, 5) == 0
    np.testing.assert_array_equal(d1.raw, d2.raw)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("dtype,ch,border", [(vi.U8, 1, 3), (vi.U8, 3, 2), (vi.I32, 2, 4), (vi.F32, 2, 3)])
def test_fill_border(orc, ref, mode, dtype, ch, border):
    a = rand_image(19, 23, dtype, ch, border=border, seed=5)
    b = a.like(); b.raw[:] = a.raw
    val = (ctypes.c_uint8 * 16)(*range(1, 17))
    assert ref.ref_fill_border(P(a.desc), mode, val) == 0
    assert orc.orc_fill_border(P(b.desc), mode, val) == 0
    np.testing.assert_array_equal(a.raw, b.raw)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape,border", [((3, 23), 5), ((19, 2), 4), ((2, 3), 7), ((1, 1), 3)])
def test_fill_border_wider_than_the_image(orc, ref, mode, shape, border):
    """border > nrows or > ncols: mirrored sources lie in other border regions, so the reference's region order (fill.hh:56-82)
    shows; the pre-existing border bytes are random so that stale reads show too."""
    a = rand_image(*shape, vi.U8, 1, border=border, seed=7, fill_border=True)
    b = a.like(); b.raw[:] = a.raw
    assert ref.ref_fill_border(P(a.desc), mode, None) == 0
    assert orc.orc_fill_border(P(b.desc), mode, None) == 0
    np.testing.assert_array_equal(a.raw, b.raw)


@pytest.mark.parametrize("shape", [(40, 56), (41, 57), (135, 240)])
@pytest.mark.parametrize("dtype,ch", [(vi.U8, 1), (vi.I32, 2), (vi.F32, 2)])
def test_pyramid(orc, ref, shape, dtype, ch):
    """pyramid2d<V>(img, 3, 2, _border = 3) of the reference == copy + mirror + orc_pyr_down chain (incl. SURVEY Q4 cells)."""
    lo, hi = (-500, 500) if dtype == vi.I32 else (None, None)
    img = rand_image(*shape, dtype, ch, seed=7, lo=lo, hi=hi)
    want = [HostImage(nr, nc, dtype, ch, 3) for nr, nc in pyr.level_dims(*shape, 3)]
    assert ref.ref_pyramid(P(img.desc), 3, 3, vi.desc_array(want)) == 0
    got = pyr.host_pyramid(orc, img, 3, 3)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g.view(with_border=True).view(np.uint8), w.view(with_border=True).view(np.uint8))


@pytest.mark.parametrize("odt", [vi.F32, vi.I32])
def test_scharr(orc, ref, odt):
    src = rand_image(33, 47, vi.U8, 1, border=1, seed=9, fill_border=True)
    o1, o2 = HostImage(33, 47, odt, 2), HostImage(33, 47, odt, 2)
    assert ref.ref_scharr(P(o1.desc), P(src.desc)) == 0
    assert orc.orc_scharr(P(o2.desc), P(src.desc)) == 0
    np.testing.assert_array_equal(o1.raw, o2.raw)


def ref_detect(ref, im, th, mask=None, mode=0, bs=10, cap=200000):
    rc = np.zeros((cap, 2), np.int32); sc = np.zeros(cap, np.int32); n = ctypes.c_int(0)
    assert ref.ref_fast9(P(im.desc), th, P(mask.desc) if mask is not None else None, mode, bs, rc.ctypes.data_as(ctypes.c_void_p),
                         sc.ctypes.data_as(ctypes.c_void_p), cap, P(n)) == 0
    return rc[:n.value].copy(), sc[:n.value].copy()


@pytest.mark.parametrize("shape,seed", [((60, 90), 4), ((128, 160), 5), ((200, 333), 6)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fast9_reference_mode_is_the_reference(orc, ref, shape, seed, mode):
    """fast9() of the reference (AVX2 fast_detector9_simd incl. its a4/a12 quirk) == oracle compat=REFERENCE, all modes + scores."""
    im = u8_image(rects_image(*shape, seed=seed), border=3)
    orc.orc_fill_border(P(im.desc), 0, None)
    want_rc, want_sc = ref_detect(ref, im, 20, mode=mode)
    got_rc, got_sc = run_detect(orc, im, 20, mode=mode, compat=0)
    assert len(want_rc) > 0
    np.testing.assert_array_equal(got_rc, want_rc)
    np.testing.assert_array_equal(got_sc, want_sc)


@pytest.mark.parametrize("mval", [255, 1, 16, 2])
def test_fast9_mask(orc, ref, mval):
    im = u8_image(rects_image(96, 128, seed=14), border=3)
    orc.orc_fill_border(P(im.desc), 0, None)
    mask = HostImage(96, 128, vi.U8, 1, border=10)
    mask.view()[...] = mval
    mask.view()[20:60, 32:96] = 0
    for mode in (0, 2):
        want_rc, want_sc = ref_detect(ref, im, 10, mask=mask, mode=mode)
        got_rc, got_sc = run_detect(orc, im, 10, mask=mask, mode=mode)
        np.testing.assert_array_equal(got_rc, want_rc)
        np.testing.assert_array_equal(got_sc, want_sc)


def test_fast9_corrected_mode_is_the_reference_scalar_detector(orc, ref):
    """compat=CORRECTED == the reference's dense scalar fast_detector9(A,B,th) (fast.hpp:512-551)."""
    im = u8_image(rects_image(48, 64, seed=8), border=3)
    orc.orc_fill_border(P(im.desc), 0, None)
    rc, _ = run_detect(orc, im, 20, compat=1)
    got = set(map(tuple, rc))
    for r in range(0, 48, 1):
        for c in range(0, 64, 7):
            assert bool(ref.ref_is_fast9_keypoint(P(im.desc), r, c, 20)) == ((r, c) in got)


def test_fast9_border_exception(ref):
    im = HostImage(20, 20, vi.U8, 1, border=2)
    n = ctypes.c_int(0)
    assert ref.ref_fast9(P(im.desc), 20, None, 0, 10, None, None, 0, P(n)) == 2


def test_lucas_kanade_golden_through_the_reference(orc, ref):
    """tests/pyrlk.cc run by the reference itself, and the oracle bit-identical to it."""
    f1, f2 = pyrlk_cc_fixture()
    i1, i2 = u8_image(f1), u8_image(f2)
    pts = np.array([[50, 50], [48, 51], [52, 49], [50.5, 49.25]], np.float32)
    want = np.zeros((4, 2), np.float32); wd = np.zeros(4, np.float32)
    assert ref.ref_lucas_kanade(P(i1.desc), P(i2.desc), pts.ctypes.data_as(ctypes.c_void_p), 4, 5, 2, 50, ctypes.c_double(0.001), ctypes.c_double(0.01),
                                want.ctypes.data_as(ctypes.c_void_p), wd.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.linalg.norm(want[0] - [2, 2]) < 0.05  # the reference's own assertion (tests/pyrlk.cc:48-49)
    hp1, hp2 = pyr.host_pyramid(orc, i1, 2, 2), pyr.host_pyramid(orc, i2, 2, 2)
    hg = pyr.host_grad_pyramid(orc, hp1[0], 2, 2, vi.I32)
    got = np.zeros((4, 2), np.float32); gd = np.zeros(4, np.float32)
    orc.orc_lucas_kanade(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), 2, pts.ctypes.data_as(ctypes.c_void_p), None, 4, 5, 0, 50, 0,
                         got.ctypes.data_as(ctypes.c_void_p), gd.ctypes.data_as(ctypes.c_void_p))
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(gd.view(np.uint32), wd.view(np.uint32))


@pytest.mark.parametrize("ws", [5, 7, 9])
def test_pyrlk_match(orc, ref, ws):
    f1, f2, kps = lk_scene(240, 320, 400)
    kps["age"][::13] = 0
    i1, i2 = u8_image(f1), u8_image(f2)
    want = kps.copy()
    assert ref.ref_pyrlk_match(P(i1.desc), P(i2.desc), 3, 5, want.ctypes.data_as(ctypes.c_void_p), len(kps), ws, ctypes.c_float(1e-4),
                               ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0) == 0
    hp1, hp2 = pyr.host_pyramid(orc, i1, 3, 5), pyr.host_pyramid(orc, i2, 3, 5)
    hg = pyr.host_grad_pyramid(orc, hp1[0], 3, 5, vi.F32)
    got = kps.copy()
    orc.orc_pyrlk_match(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), 3, got.ctypes.data_as(ctypes.c_void_p), len(kps), ws,
                        ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None)
    np.testing.assert_array_equal(got["age"], want["age"])
    for f in ("pos_r", "pos_c", "vel_r", "vel_c"):
        np.testing.assert_array_equal(got[f].view(np.uint32), want[f].view(np.uint32))


@pytest.mark.parametrize("shape,ws,nscales,min_scale,prop,patch", [((120, 160), 9, 3, 0, 2, 5), ((121, 163), 7, 4, 0, 2, 5), ((120, 160), 9, 3, 1, 3, 5), ((96, 128), 5, 2, 0, 0, 3)])
def test_semi_dense_optical_flow(orc, ref, shape, ws, nscales, min_scale, prop, patch):
    f1, f2, kps = flow_scene(*shape)
    rng = np.random.default_rng(0)
    kps = np.concatenate([np.stack([rng.integers(0, shape[0], 300), rng.integers(0, shape[1], 300)], 1).astype(np.int32), kps])
    i1, i2 = u8_image(f1, border=3), u8_image(f2, border=3)
    n = len(kps)
    outs = []
    for fn in (ref.ref_semi_dense_optical_flow, orc.orc_semi_dense_optical_flow):
        p = np.zeros((n, 2), np.int32); d = np.zeros(n, np.int32); v = np.zeros(n, np.uint8)
        assert fn(P(i1.desc), P(i2.desc), kps.ctypes.data_as(ctypes.c_void_p), n, ws, nscales, min_scale, prop, patch,
                  p.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)) == 0
        outs.append((p, d, v))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("dtype,kind", [(vi.U8, "sparse"), (vi.U8, "dense"), (vi.I32, "signed"), (vi.F32, "ramp"), (vi.U8, "plateau")])
def test_local_maxima_filter(orc, ref, dtype, kind):
    """local_maxima_filter in its serial raster-order form (fast.hpp:555-575): score-like sparse images, dense noise (long dependency chains),
    signed values (zeroing a negative RAISES it), monotone ramps (every pixel waits for its left neighbour), plateaus (ties are not maxima)."""
    rng = np.random.default_rng(17)
    shape = (37, 53)
    im = HostImage(*shape, dtype, 1, 1)
    v = im.view(with_border=True)[..., 0]
    if kind == "sparse":
        v[...] = np.where(rng.random(v.shape) < 0.15, rng.integers(1, 255, v.shape), 0)
    elif kind == "dense":
        v[...] = rng.integers(0, 6, v.shape)
    elif kind == "signed":
        v[...] = rng.integers(-5, 6, v.shape)
    elif kind == "ramp":
        v[...] = (np.arange(v.shape[1])[None, ::-1] * 3.0 + np.arange(v.shape[0])[::-1][:, None] * 0.5 + rng.random(v.shape) * 0.2).astype(np.float32)
    else:
        v[...] = 7; v[10:20, 10:30] = 9; v[12, 14] = 11
    a = im; b = im.like(); b.raw[:] = a.raw
    assert ref.ref_local_maxima_filter(P(a.desc)) == 0
    assert orc.orc_local_maxima_filter(P(b.desc)) == 0
    np.testing.assert_array_equal(a.raw, b.raw)


def _min_ev_boundary(track_alive, lo=1e-4, hi=1e9):
    """Smallest float32 threshold at which `track_alive(th)` turns False (the keypoint is kept at lo and removed at hi): bisection over the
    float's bit pattern, so the answer is exact to the ulp."""
    lo_b, hi_b = int(np.float32(lo).view(np.uint32)), int(np.float32(hi).view(np.uint32))
    assert track_alive(np.float32(lo)) and not track_alive(np.float32(hi))
    while hi_b - lo_b > 1:
        mid = (lo_b + hi_b) // 2
        if track_alive(np.array([mid], np.uint32).view(np.float32)[0]):
            lo_b = mid
        else:
            hi_b = mid
    return np.array([hi_b], np.uint32).view(np.float32)[0]


def test_min_ev_gate_at_the_threshold(orc, ref):
    """lk.hh:75-81 removes a keypoint when min |eigenvalue| of its structure tensor is below `_min_ev`.  The reference takes the eigenvalues
    from Eigen's general solver (third-party, absent here); the shim and the oracle use the closed form of the symmetric 2x2 case.  For a
    handful of keypoints the exact float threshold at which the ORACLE starts to remove the keypoint is found by bisection, and the
    reference's own code over the shim is shown to flip at the same ulp (kept one ulp below, removed at and above)."""
    f1, f2, kps = lk_scene(120, 160, 40)
    i1, i2 = u8_image(f1), u8_image(f2)
    hp1, hp2 = pyr.host_pyramid(orc, i1, 3, 5), pyr.host_pyramid(orc, i2, 3, 5)
    hg = pyr.host_grad_pyramid(orc, hp1[0], 3, 5, vi.F32)

    def orc_alive(k, th):
        one = kps[k:k + 1].copy()
        orc.orc_pyrlk_match(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), 3, one.ctypes.data_as(ctypes.c_void_p), 1, 7,
                            ctypes.c_float(float(th)), ctypes.c_float(1e9), 30, ctypes.c_float(0.01), 0, None)
        return bool(one["age"][0] > 0)

    def ref_alive(k, th):
        one = kps[k:k + 1].copy()
        assert ref.ref_pyrlk_match(P(i1.desc), P(i2.desc), 3, 5, one.ctypes.data_as(ctypes.c_void_p), 1, 7, ctypes.c_float(float(th)), ctypes.c_float(1e9), 30, ctypes.c_float(0.01), 0) == 0
        return bool(one["age"][0] > 0)

    found = 0
    for k in range(0, 40, 7):
        if not orc_alive(k, 1e-4):
            continue
        t = _min_ev_boundary(lambda th: orc_alive(k, th))
        below, above = np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(np.inf))
        assert orc_alive(k, below) and not orc_alive(k, t) and not orc_alive(k, above)
        assert ref_alive(k, below) and not ref_alive(k, t) and not ref_alive(k, above), (k, t)
        found += 1
    assert found >= 4
