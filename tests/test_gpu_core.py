"""GPU parity tests (through the C ABI) for pixel_wise add, box_nbh2d mean, borders, copy/fill.
Bit-exact against the CPU oracle on the same seeded inputs; full-size cases use the benchmarks' inline checkers
(benchmarks/image_add.cc:21-28, box_5x5_filter2.cc:26-41) restated in numpy on a sampled subset."""
import ctypes

import numpy as np
import pytest

from util import P, rand_image, HostImage, DeviceImage
from vpp_amd import image as vi
from vpp_amd import capi

pytestmark = pytest.mark.gpu


def _sync(lib):
    capi.check(lib.vpp_sync(capi.stream_ptr()))


@pytest.mark.parametrize("op", range(6))
@pytest.mark.parametrize("dtype,ch,shape,border,align", [
    (vi.I32, 1, (1080, 1920), 0, 32),     # BASELINE config 1 shape (flat path)
    (vi.I32, 1, (61, 77), 0, 32),         # pitched rows, ragged
    (vi.U8, 3, (64, 100), 2, 16),         # pitched rows with border
    (vi.F32, 2, (33, 47), 1, 32),
    (vi.I16, 1, (17, 19), 0, 2),          # unaligned -> scalar kernel
    (vi.U8, 1, (5, 7), 0, 1),
    (vi.I8, 3, (64, 100), 0, 16),         # packed signed bytes
    (vi.U16, 1, (33, 64), 0, 32),         # packed 16-bit
    (vi.I16, 2, (40, 56), 1, 32),
    (vi.U8, 3, (270, 480), 0, 16),        # flat path, packed bytes
])
def test_pixelwise_binary_matches_oracle(lib, orc, op, dtype, ch, shape, border, align):
    lo, hi = (0, 2**30 - 1) if dtype == vi.I32 and op != 2 else (None, None)
    b = rand_image(*shape, dtype, ch, border=border, seed=1, lo=lo, hi=hi, align=align)
    c = rand_image(*shape, dtype, ch, border=border, seed=2, lo=lo, hi=hi, align=align)
    want = b.like()
    assert orc.orc_pixelwise_binary(op, P(want.desc), P(b.desc), P(c.desc)) == 0
    db, dc, da = DeviceImage.from_host(b), DeviceImage.from_host(c), DeviceImage.from_host(b.like())
    capi.check(lib.vpp_pixelwise_binary(op, P(da.desc), P(db.desc), P(dc.desc), capi.stream_ptr()))
    _sync(lib)
    got = da.download()
    np.testing.assert_array_equal(got.raw, want.raw)  # bit-exact, and nothing outside the domain was touched


def test_add_4k_checker(lib):
    """BASELINE config 1': 4K int32, inline checker A == B + C everywhere (image_add.cc:21-28)."""
    shape = (2160, 3840)
    b = rand_image(*shape, vi.I32, seed=2, lo=0, hi=2**30 - 1)
    c = rand_image(*shape, vi.I32, seed=3, lo=0, hi=2**30 - 1)
    db, dc = DeviceImage.from_host(b), DeviceImage.from_host(c)
    da = DeviceImage(*shape, vi.I32)
    for unroll in (1, 2, 4, 8):
        for nt in (0, 1):
            lib.vpp_set_tuning(b"add.unroll", unroll); lib.vpp_set_tuning(b"add.nt", nt)
            da.store.zero_()
            capi.check(lib.vpp_pixelwise_binary(0, P(da.desc), P(db.desc), P(dc.desc), capi.stream_ptr()))
            _sync(lib)
            np.testing.assert_array_equal(da.download().view(), b.view() + c.view())
    lib.vpp_set_tuning(b"add.unroll", -1); lib.vpp_set_tuning(b"add.nt", -1)


@pytest.mark.parametrize("dtype,ch,R,C,shape,border,align", [
    (vi.U8, 3, 5, 5, (67, 131), 2, 16),     # fast path, ragged row end, border exactly 2
    (vi.U8, 3, 5, 5, (64, 1024), 2, 32),
    (vi.U8, 3, 5, 5, (40, 5000), 3, 32),    # > one block wide
    (vi.U8, 1, 5, 5, (50, 300), 2, 32),
    (vi.U8, 2, 5, 5, (50, 300), 2, 32),
    (vi.U8, 4, 5, 5, (50, 300), 5, 32),
    (vi.U8, 3, 3, 3, (31, 45), 1, 32),      # generic kernel
    (vi.I32, 1, 5, 5, (100, 200), 2, 32),   # the reference benchmark's own element type (32-bit streaming kernel)
    (vi.I32, 1, 5, 5, (67, 131), 2, 16),    # ragged row end, border exactly 2
    (vi.I32, 1, 5, 5, (33, 1000), 3, 32),   # several strips
    (vi.U32, 1, 5, 5, (50, 249), 2, 32),
    (vi.I32, 2, 5, 5, (20, 30), 2, 32),     # vint2 -> generic kernel
    (vi.F32, 1, 5, 5, (64, 64), 2, 32),      # float: taps in the reference's order (32-bit streaming kernel)
    (vi.F32, 1, 5, 5, (67, 131), 2, 16),
    (vi.F32, 1, 5, 5, (33, 1000), 3, 32),
    (vi.F32, 2, 5, 5, (20, 30), 2, 32),      # vfloat2 -> generic kernel
    (vi.U8, 3, 7, 5, (30, 40), 3, 32),
    (vi.I16, 2, 3, 5, (30, 40), 2, 32),
])
def test_box_filter_matches_oracle(lib, orc, dtype, ch, R, C, shape, border, align):
    lo, hi = (0, 999) if dtype in (vi.I32, vi.U32) else (None, None)
    src = rand_image(*shape, dtype, ch, border=border, seed=3, lo=lo, hi=hi, align=align, fill_border=True)
    want = src.like(border=0)
    assert orc.orc_box_filter(P(want.desc), P(src.desc), R, C) == 0
    dsrc = DeviceImage.from_host(src)
    fast = dtype == vi.U8 and R == 5 and C == 5
    w32 = dtype in (vi.I32, vi.U32, vi.F32) and ch == 1 and R == 5 and C == 5
    for impl, rows in (((2, 2), (1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (0, 8), (0, 16), (0, 32)) if fast else ((1, 1), (1, 2), (1, 4), (1, 8)) if w32 else ((1, 2),)):
        lib.vpp_set_tuning(b"box.impl", impl); lib.vpp_set_tuning(b"box.rows", rows); lib.vpp_set_tuning(b"box.rows32", rows)
        ddst = DeviceImage.from_host(src.like(border=0))
        capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), R, C, capi.stream_ptr()))
        _sync(lib)
        got = ddst.download()
        if dtype == vi.F32:
            np.testing.assert_array_equal(got.view().view(np.uint32), want.view().view(np.uint32))
        else:
            np.testing.assert_array_equal(got.view(), want.view())
    lib.vpp_set_tuning(b"box.rows", -1); lib.vpp_set_tuning(b"box.impl", -1); lib.vpp_set_tuning(b"box.rows32", -1)


@pytest.mark.parametrize("R,C", [(3, 3), (3, 5), (5, 3), (7, 3), (7, 5), (3, 7), (5, 7), (7, 7)])
@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_box_u8_streamed_windows_match_oracle(lib, orc, R, C, ch):
    """Odd windows up to 7 x 7 on 8-bit images go through the streaming kernel when (C/2) * channels <= 8 halo bytes (else the
    generic LDS kernel): bit-exact either way; border exactly the reach and larger, ragged widths, all-255 images (largest sums)."""
    reach = max(R, C) // 2
    for shape, border, align in (((37, 61), reach, 16), ((23, 1111), reach + 2, 32), ((9, 16), reach, 32), ((2, 3), reach + 1, 16)):
        src = rand_image(*shape, vi.U8, ch, border=border, seed=R * 10 + C + ch, align=align, fill_border=True)
        if shape == (9, 16):
            src.raw[...] = 255
        want = src.like(border=0)
        assert orc.orc_box_filter(P(want.desc), P(src.desc), R, C) == 0
        dsrc = DeviceImage.from_host(src); ddst = DeviceImage.from_host(src.like(border=0))
        capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), R, C, capi.stream_ptr()))
        _sync(lib)
        np.testing.assert_array_equal(ddst.download().raw, want.raw)   # and nothing outside the domain was written


@pytest.mark.parametrize("R,C", [(3, 3), (3, 5), (5, 3), (7, 3), (7, 5), (5, 7)])
@pytest.mark.parametrize("dtype", [vi.I32, vi.U32, vi.F32])
def test_box_32bit_streamed_windows_match_oracle(lib, orc, dtype, R, C):
    """32-bit single-channel images: windows up to 7 rows x 5 columns stream (5 x 7 falls to the generic kernel); integers exact,
    floats bit-identical (taps added in the reference's row-major order)."""
    reach = max(R, C) // 2
    for shape, border, align in (((37, 61), reach, 16), ((23, 1111), reach + 2, 32), ((2, 3), reach + 1, 16), ((40, 256), reach, 32)):
        lo, hi = (0, 999) if dtype != vi.F32 else (None, None)
        src = rand_image(*shape, dtype, 1, border=border, seed=R * 10 + C, lo=lo, hi=hi, align=align, fill_border=True)
        want = src.like(border=0)
        assert orc.orc_box_filter(P(want.desc), P(src.desc), R, C) == 0
        dsrc = DeviceImage.from_host(src); ddst = DeviceImage.from_host(src.like(border=0))
        capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), R, C, capi.stream_ptr()))
        _sync(lib)
        np.testing.assert_array_equal(ddst.download().raw.view(np.uint8), want.raw.view(np.uint8))


@pytest.mark.parametrize("dtype", [vi.I32, vi.F32])
def test_box_32bit_row_bands_match_oracle(lib, orc, dtype):
    """Sources of 2 GiB and more leave the 32-bit kernel as row bands (one buffer descriptor addresses < 4 GiB); forced here on a
    small image: bands of 8 and of 2 rows, ragged last band, every window the streaming kernel serves."""
    lo, hi = (0, 999) if dtype != vi.F32 else (None, None)
    src = rand_image(45, 300, dtype, 1, border=3, seed=21, lo=lo, hi=hi, align=16, fill_border=True)
    dsrc = DeviceImage.from_host(src)
    try:
        for band in (8, 2):
            lib.vpp_set_tuning(b"box.band_rows32", band)
            for R, C in ((5, 5), (3, 3), (7, 5)):
                want = src.like(border=0)
                assert orc.orc_box_filter(P(want.desc), P(src.desc), R, C) == 0
                ddst = DeviceImage.from_host(src.like(border=0))
                capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), R, C, capi.stream_ptr()))
                _sync(lib)
                np.testing.assert_array_equal(ddst.download().raw.view(np.uint8), want.raw.view(np.uint8))
    finally:
        lib.vpp_set_tuning(b"box.band_rows32", -1)


def test_box_u8_windows_4k_streamed_equal_generic(lib):
    """BASELINE frame size: the streamed 3x3 / 7x7 / 5x3 results equal the generic kernel's, and one pixel equals the numpy mean."""
    src = rand_image(2160, 3840, vi.U8, 3, border=3, seed=12, fill_border=True)
    dsrc = DeviceImage.from_host(src)
    s = src.view(with_border=True).astype(np.int64)
    for R, C in ((3, 3), (5, 3), (7, 5)):
        outs = []
        for g in (0, 1):
            lib.vpp_set_tuning(b"box.force_generic", g)
            d = DeviceImage.from_host(src.like(border=0))
            capi.check(lib.vpp_box_filter(P(d.desc), P(dsrc.desc), R, C, capi.stream_ptr()))
            _sync(lib)
            outs.append(d.download().view().copy())
        lib.vpp_set_tuning(b"box.force_generic", 0)
        np.testing.assert_array_equal(outs[0], outs[1])
        r, c = 1234, 2345
        win = s[3 + r - R // 2:3 + r + R // 2 + 1, 3 + c - C // 2:3 + c + C // 2 + 1]
        np.testing.assert_array_equal(outs[0][r, c], win.sum(axis=(0, 1)) // (R * C))


def test_box_int32_4k_matches_generic(lib):
    """The reference's own benchmark type at 4K (box_5x5_filter.cc:187-191: values % 1000): streaming kernel == generic kernel."""
    src = rand_image(2160, 3840, vi.I32, 1, border=2, seed=12, lo=0, hi=999, fill_border=True, align=16)
    dsrc = DeviceImage.from_host(src)
    outs = []
    for g in (0, 1):
        lib.vpp_set_tuning(b"box.force_generic", g)
        d = DeviceImage(2160, 3840, vi.I32, 1, 0, 16)
        capi.check(lib.vpp_box_filter(P(d.desc), P(dsrc.desc), 5, 5, capi.stream_ptr()))
        _sync(lib)
        outs.append(d.download().view().copy())
    lib.vpp_set_tuning(b"box.force_generic", 0)
    np.testing.assert_array_equal(outs[0], outs[1])
    s = src.view(with_border=True)[..., 0].astype(np.int64)
    r, c = 1000, 2000  # one inline check against numpy (box_5x5_filter.cc:26-41)
    assert outs[0][r, c, 0] == int(s[r:r + 5, c:c + 5].sum()) // 25


def test_box_fast_equals_generic_on_device(lib):
    src = rand_image(200, 777, vi.U8, 3, border=2, seed=11, fill_border=True)
    dsrc = DeviceImage.from_host(src)
    outs = []
    for g in (0, 1):
        lib.vpp_set_tuning(b"box.force_generic", g)
        d = DeviceImage.from_host(src.like(border=0))
        capi.check(lib.vpp_box_filter(P(d.desc), P(dsrc.desc), 5, 5, capi.stream_ptr()))
        _sync(lib)
        outs.append(d.download().view().copy())
    lib.vpp_set_tuning(b"box.force_generic", 0)
    np.testing.assert_array_equal(outs[0], outs[1])


def test_box_4k_vuchar3_checker(lib, orc):
    """BASELINE config 2 at full size: interior checker of box_5x5_filter2.cc:26-41 on sampled pixels + oracle on a band."""
    nr, nc = 2160, 3840
    src = rand_image(nr, nc, vi.U8, 3, border=2, seed=3, align=16)
    orc.orc_fill_border(P(src.desc), 0, None)
    dsrc = DeviceImage.from_host(src)
    ddst = DeviceImage(nr, nc, vi.U8, 3, 0, 16)
    capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), 5, 5, capi.stream_ptr()))
    _sync(lib)
    got = ddst.download().view()
    full = src.view(with_border=True).astype(np.int64)
    rng = np.random.default_rng(0)
    rs = np.concatenate([rng.integers(0, nr, 4000), [0, 0, nr - 1, nr - 1]])
    cs = np.concatenate([rng.integers(0, nc, 4000), [0, nc - 1, 0, nc - 1]])
    for r, c in zip(rs, cs):
        want = full[r:r + 5, c:c + 5].sum(axis=(0, 1)) // 25
        assert (got[r, c] == want).all(), (r, c)
    # whole-image check against the oracle
    want = src.like(border=0)
    orc.orc_box_filter(P(want.desc), P(src.desc), 5, 5)
    np.testing.assert_array_equal(got, want.view())


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("dtype,ch,shape,border", [(vi.U8, 1, (9, 13), 3), (vi.U8, 3, (40, 50), 2), (vi.F32, 2, (30, 20), 5), (vi.I32, 1, (12, 11), 10)])
def test_fill_border_matches_oracle(lib, orc, mode, dtype, ch, shape, border):
    im = rand_image(*shape, dtype, ch, border=border, seed=5)
    val = (ctypes.c_uint8 * 16)(*range(1, 17))
    dim = DeviceImage.from_host(im)
    assert orc.orc_fill_border(P(im.desc), mode, val) == 0
    capi.check(lib.vpp_fill_border(P(dim.desc), mode, val, capi.stream_ptr()))
    _sync(lib)
    np.testing.assert_array_equal(dim.download().raw, im.raw)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape,border", [((3, 23), 5), ((19, 2), 4), ((2, 3), 7), ((1, 1), 3), ((4, 4), 18)])
def test_fill_border_wider_than_the_image_matches_oracle(lib, orc, mode, shape, border):
    """border > nrows / ncols: the mirror reads other border regions, the eight regions run in the reference's order (fill.hh:56-82);
    random pre-existing border bytes so that a stale read would show (oracle pinned to the reference in test_ref_pins_oracle.py)."""
    im = rand_image(*shape, vi.U8, 1, border=border, seed=7, fill_border=True)
    dim = DeviceImage.from_host(im)
    assert orc.orc_fill_border(P(im.desc), mode, None) == 0
    capi.check(lib.vpp_fill_border(P(dim.desc), mode, None, capi.stream_ptr()))
    _sync(lib)
    np.testing.assert_array_equal(dim.download().raw, im.raw)


def test_copy_and_fill(lib, orc):
    src = rand_image(33, 21, vi.U8, 3, border=2, seed=6, fill_border=True)
    for wb in (0, 1):
        dst = src.like(border=4)
        want = src.like(border=4)
        orc.orc_copy(P(want.desc), P(src.desc), wb)
        ds, dd = DeviceImage.from_host(src), DeviceImage.from_host(dst)
        capi.check(lib.vpp_copy(P(dd.desc), P(ds.desc), wb, capi.stream_ptr()))
        _sync(lib)
        np.testing.assert_array_equal(dd.download().raw, want.raw)
    val = (ctypes.c_uint8 * 3)(9, 8, 7)
    for wb in (0, 1):
        want = src.like()
        orc.orc_fill(P(want.desc), val, wb)
        dd = DeviceImage.from_host(src.like())
        capi.check(lib.vpp_fill(P(dd.desc), val, wb, capi.stream_ptr()))
        _sync(lib)
        np.testing.assert_array_equal(dd.download().raw, want.raw)


def test_error_statuses(lib):
    a = DeviceImage(8, 8, vi.U8, 3, border=1)
    b = DeviceImage(8, 8, vi.U8, 3, border=0)
    assert lib.vpp_box_filter(P(b.desc), P(a.desc), 5, 5, None) == capi.ERR_BORDER_TOO_SMALL
    assert b"border" in lib.vpp_last_error()
    c = DeviceImage(8, 9, vi.U8, 3)
    assert lib.vpp_pixelwise_binary(0, P(b.desc), P(b.desc), P(c.desc), None) == capi.ERR_INVALID_ARG


@pytest.mark.parametrize("dtype,ch,R,C", [(vi.U8, 3, 5, 5), (vi.U8, 1, 3, 3), (vi.U8, 4, 7, 5), (vi.I32, 1, 5, 5), (vi.F32, 1, 5, 5), (vi.I32, 1, 3, 3)])
def test_box_source_at_the_very_start_of_an_allocation(lib, orc, dtype, ch, R, C):
    """A source whose buffer begins exactly where its device allocation begins (a fresh torch segment, border exactly the window reach):
    the descriptor kernels start their first chunk in the aligned 16-byte granule that holds the first row's left border (see box.hip,
    fits_descriptor) — inside the allocation's first page, never in front of it.  Same result as the oracle, no fault."""
    import torch
    border = max(R, C) // 2
    lo, hi = (0, 999) if dtype == vi.I32 else (None, None)
    src = rand_image(301, 1777, dtype, ch, border=border, seed=9, lo=lo, hi=hi, align=32, fill_border=True)
    want = src.like(border=0)
    assert orc.orc_box_filter(P(want.desc), P(src.desc), R, C) == 0
    dsrc = DeviceImage(301, 1777, dtype, ch, border=border, align=32)
    dsrc.store = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")   # its own segment: data_ptr() is the allocation's first byte
    dsrc.shift = 0
    dsrc.upload(src)
    ddst = DeviceImage.from_host(src.like(border=0))
    capi.check(lib.vpp_box_filter(P(ddst.desc), P(dsrc.desc), R, C, capi.stream_ptr()))
    _sync(lib)
    got = ddst.download()
    if dtype == vi.F32:
        np.testing.assert_array_equal(got.view().view(np.uint32), want.view().view(np.uint32))
    else:
        np.testing.assert_array_equal(got.view(), want.view())


@pytest.mark.parametrize("shape,ch,border,n", [((67, 131), 3, 2, 5), ((270, 480), 3, 2, 8), ((64, 1024), 1, 3, 17), ((40, 5000), 4, 2, 3), ((31, 45), 3, 2, 2)])
def test_box_filter_batch_equals_the_oracle_per_frame(lib, orc, shape, ch, border, n):
    """vpp_box_filter_batch: n frames of one geometry in ONE launch (up to kBoxBatchMax = 64 frames per launch; the split of larger batches is
    exercised at 4K by test_box_filter_batch_4k_at_the_benchmarked_geometry) — every frame bit-identical to the oracle's box filter of that
    frame; and with mixed geometries (falls back to single calls)."""
    srcs = [rand_image(*shape, vi.U8, ch, border=border, seed=20 + k, align=16, fill_border=True) for k in range(n)]
    wants = []
    for s in srcs:
        w = s.like(border=0); assert orc.orc_box_filter(P(w.desc), P(s.desc), 5, 5) == 0; wants.append(w)
    dsrc = [DeviceImage.from_host(s) for s in srcs]
    ddst = [DeviceImage.from_host(s.like(border=0)) for s in srcs]
    capi.check(lib.vpp_box_filter_batch(vi.desc_array(ddst), vi.desc_array(dsrc), n, 5, 5, capi.stream_ptr()))
    _sync(lib)
    for d, w in zip(ddst, wants):
        np.testing.assert_array_equal(d.download().view(), w.view())
    # mixed geometries: one frame of another size in the batch
    odd = rand_image(shape[0] + 3, shape[1] + 5, vi.U8, ch, border=border, seed=99, align=16, fill_border=True)
    wodd = odd.like(border=0); assert orc.orc_box_filter(P(wodd.desc), P(odd.desc), 5, 5) == 0
    dsrc2 = dsrc[:2] + [DeviceImage.from_host(odd)]
    ddst2 = [DeviceImage.from_host(srcs[0].like(border=0)), DeviceImage.from_host(srcs[1].like(border=0)), DeviceImage.from_host(wodd.like())]
    capi.check(lib.vpp_box_filter_batch(vi.desc_array(ddst2), vi.desc_array(dsrc2), 3, 5, 5, capi.stream_ptr()))
    _sync(lib)
    for d, w in zip(ddst2, [wants[0], wants[1], wodd]):
        np.testing.assert_array_equal(d.download().view(), w.view())


@pytest.mark.parametrize("shape,n", [((135, 240), 4), ((1080, 1920), 3), ((33, 52), 18)])
def test_pixelwise_binary_batch_equals_the_oracle_per_triple(lib, orc, shape, n):
    for op in (0, 1, 4):   # add, sub (batched kernel), max (single calls)
        bs = [rand_image(*shape, vi.I32, seed=30 + k, lo=0, hi=2**30) for k in range(n)]
        cs = [rand_image(*shape, vi.I32, seed=60 + k, lo=0, hi=2**30) for k in range(n)]
        wants = []
        for b, c in zip(bs, cs):
            a = b.like(); assert orc.orc_pixelwise_binary(op, P(a.desc), P(b.desc), P(c.desc)) == 0; wants.append(a)
        db, dc, da = [DeviceImage.from_host(x) for x in bs], [DeviceImage.from_host(x) for x in cs], [DeviceImage.from_host(x.like()) for x in bs]
        capi.check(lib.vpp_pixelwise_binary_batch(op, vi.desc_array(da), vi.desc_array(db), vi.desc_array(dc), n, capi.stream_ptr()))
        _sync(lib)
        for d, w in zip(da, wants):
            np.testing.assert_array_equal(d.download().view(), w.view())


def _device_equal(dimg, want_dev):
    """Whole allocation (pixels, padding, everything) of a DeviceImage against an uploaded expectation, compared in HBM."""
    import torch
    return bool(torch.equal(dimg.store[dimg.shift:dimg.shift + dimg.alloc_bytes], want_dev.store[want_dev.shift:want_dev.shift + want_dev.alloc_bytes]))


def test_box_filter_batch_4k_at_the_benchmarked_geometry(lib, orc):
    """The instance bench.py times: vpp_box_filter_batch on 64 distinct 3840x2160 vuchar3 frames in ONE launch
    (box_u8_wide_kernel<3,5,5,6,4,...>, 6 rows per wave, one XCD remap over the whole 64-frame block grid), then 65 frames (64 + 1: the
    second launch carries a single frame through the same instance) and 130 frames (64 + 64 + 2: the n > kBoxBatchMax split) — every byte
    of every frame against the oracle (the per-frame expectation is computed by the oracle and compared in HBM); and the per-frame call
    form on the same frames.  The inline checker of benchmarks/box_5x5_filter2.cc:26-41 restated on sampled pixels of one frame."""
    from oracle import binding
    omp = binding.load(omp=True)     # the same restatement built with OpenMP (integer arithmetic: identical results, checked on frame 0 below)
    nr, nc, N = 2160, 3840, 130
    base = rand_image(nr, nc, vi.U8, 3, border=2, seed=3, align=16)
    srcs, wants = [], []
    for k in range(N):
        h = base.like()
        h.view()[...] = base.view() ^ np.uint8((k * 37 + 1) & 255)     # 130 distinct frames
        orc.orc_fill_border(P(h.desc), 0, None)
        w = h.like(border=0)
        assert omp.orc_box_filter(P(w.desc), P(h.desc), 5, 5) == 0
        if k == 0:
            w1 = h.like(border=0)
            assert orc.orc_box_filter(P(w1.desc), P(h.desc), 5, 5) == 0
            np.testing.assert_array_equal(w.raw, w1.raw)
            full = h.view(with_border=True).astype(np.int64)
            rng = np.random.default_rng(0)
            for r, c in zip(rng.integers(0, nr, 500), rng.integers(0, nc, 500)):
                assert (w.view()[r, c] == full[r:r + 5, c:c + 5].sum(axis=(0, 1)) // 25).all()
        srcs.append(DeviceImage.from_host(h)); wants.append(DeviceImage.from_host(w))
    dsts = [DeviceImage(nr, nc, vi.U8, 3, 0, 16) for _ in range(N)]
    for n in (64, 65, 130):
        for d in dsts:
            d.store.zero_()
        capi.check(lib.vpp_box_filter_batch(vi.desc_array(dsts[:n]), vi.desc_array(srcs[:n]), n, 5, 5, capi.stream_ptr()))
        _sync(lib)
        bad = [k for k in range(n) if not _device_equal(dsts[k], wants[k])]
        assert not bad, (n, bad[:8])
        assert all(int(d.store.count_nonzero()) == 0 for d in dsts[n:]), n     # nothing beyond the batch was written
    # the reference's call form on the same frames: one call per frame
    for d in dsts[:8]:
        d.store.zero_()
    for k in range(8):
        capi.check(lib.vpp_box_filter(P(dsts[k].desc), P(srcs[k].desc), 5, 5, capi.stream_ptr()))
    _sync(lib)
    assert all(_device_equal(dsts[k], wants[k]) for k in range(8))


def test_pixelwise_binary_batch_4k_at_the_benchmarked_geometry(lib, orc):
    """The instance bench.py times: vpp_pixelwise_binary_batch(add) on 16 triples of 3840x2160 int images in ONE launch
    (binary_flat_batch_kernel), and 17 triples (16 + 1: the split) — every element of every triple against the oracle."""
    nr, nc, N = 2160, 3840, 17
    bs = [rand_image(nr, nc, vi.I32, seed=100 + k, lo=0, hi=2**30 - 1) for k in range(N)]
    cs = [rand_image(nr, nc, vi.I32, seed=200 + k, lo=0, hi=2**30 - 1) for k in range(N)]
    wants = []
    for b, c in zip(bs, cs):
        a = b.like(); assert orc.orc_pixelwise_binary(0, P(a.desc), P(b.desc), P(c.desc)) == 0
        wants.append(DeviceImage.from_host(a))
    np.testing.assert_array_equal(wants[3].download().view(), bs[3].view() + cs[3].view())   # benchmarks/image_add.cc:21-28
    db, dc = [DeviceImage.from_host(x) for x in bs], [DeviceImage.from_host(x) for x in cs]
    da = [DeviceImage(nr, nc, vi.I32) for _ in range(N)]
    for n in (16, 17):
        for d in da:
            d.store.zero_()
        capi.check(lib.vpp_pixelwise_binary_batch(0, vi.desc_array(da[:n]), vi.desc_array(db[:n]), vi.desc_array(dc[:n]), n, capi.stream_ptr()))
        _sync(lib)
        bad = [k for k in range(n) if not _device_equal(da[k], wants[k])]
        assert not bad, (n, bad)
        assert all(int(d.store.count_nonzero()) == 0 for d in da[n:])


def test_batches_whose_frames_feed_each_other_run_in_sequence(lib, orc):
    """The batch entry points promise the results of n calls made one after the other.  A chain — frame k's result is frame k + 1's source (the
    frames of a ring) — only has those results in sequence: such a batch must not go out as one launch (box.hip / pixelwise.hip fall back to the
    n calls); same for two results that overlap."""
    # box: I[k + 1] = box5x5(I[k]), all images bordered (the border of a result keeps its old bytes, as with n single calls)
    n = 5
    ims = [rand_image(96, 256, vi.U8, 3, border=2, seed=70 + k, align=16, fill_border=True) for k in range(n + 1)]
    dev = [DeviceImage.from_host(x) for x in ims]
    for k in range(n):
        assert orc.orc_box_filter(P(ims[k + 1].desc), P(ims[k].desc), 5, 5) == 0
    capi.check(lib.vpp_box_filter_batch(vi.desc_array(dev[1:]), vi.desc_array(dev[:n]), n, 5, 5, capi.stream_ptr()))
    _sync(lib)
    for k in range(n + 1):
        np.testing.assert_array_equal(dev[k].download().raw, ims[k].raw, err_msg=f"box chain, image {k}")
    # add: A[k + 1] = A[k] + C[k]
    A = [rand_image(64, 128, vi.I32, seed=80 + k, lo=0, hi=2**20) for k in range(n + 1)]
    C = [rand_image(64, 128, vi.I32, seed=90 + k, lo=0, hi=2**20) for k in range(n)]
    dA, dC = [DeviceImage.from_host(x) for x in A], [DeviceImage.from_host(x) for x in C]
    for k in range(n):
        assert orc.orc_pixelwise_binary(0, P(A[k + 1].desc), P(A[k].desc), P(C[k].desc)) == 0
    capi.check(lib.vpp_pixelwise_binary_batch(0, vi.desc_array(dA[1:]), vi.desc_array(dA[:n]), vi.desc_array(dC), n, capi.stream_ptr()))
    _sync(lib)
    for k in range(n + 1):
        np.testing.assert_array_equal(dA[k].download().raw, A[k].raw, err_msg=f"add chain, image {k}")
    # in place per triple (A[k] += C[k]) is each call's own business and stays one launch: results as the single calls
    B = [rand_image(64, 128, vi.I32, seed=95 + k, lo=0, hi=2**20) for k in range(n)]
    dB = [DeviceImage.from_host(x) for x in B]
    capi.check(lib.vpp_pixelwise_binary_batch(0, vi.desc_array(dB), vi.desc_array(dB), vi.desc_array(dC), n, capi.stream_ptr()))
    _sync(lib)
    for k in range(n):
        np.testing.assert_array_equal(dB[k].download().view(), B[k].view() + C[k].view())


def _kernel_nodes(lib, graph):
    n = ctypes.c_int()
    capi.check(lib.vpp_debug_graph_kernel_nodes(graph, ctypes.byref(n)))
    return n.value


def test_recorded_per_frame_calls_are_batched_and_keep_their_data_flow(lib, orc):
    """Recorded per-frame calls (box.hip, common.hpp): vpp_box_filter calls on a stream recorded through vpp_graph_begin are held back in the thread's window and
    a window that closes records ONE batched node — no node of the graph under capture is ever edited (rounds 4-5 did, with hipGraphKernelNodeSetParams).  A
    sequence with 70 independent frames, frames that read results of the node already recorded (they join the open window), a frame that overwrites a source the
    first batch read (WAR against a recorded node: joins), a frame that reads a PENDING result (closes the window) and one that reads that frame's result (again):
    results equal the calls made one after the other (the oracle applies them in sequence), and the graph has the 4 kernel nodes the data flow needs instead of
    75 (64 | 6 + 3 | 1 | 1; round 5's node editing needed 5).  With the knobs off: 75 nodes, same results."""
    shape, ch = (48, 200), 3
    rng_ims = lambda: [rand_image(*shape, vi.U8, ch, border=2, seed=300 + k, align=16, fill_border=True) for k in range(210)]
    seq = [(100 + k, k) for k in range(70)] + [(200, 103), (201, 104), (5, 150), (202, 5), (203, 202)]
    host = rng_ims()
    for d, s_ in seq:
        assert orc.orc_box_filter(P(host[d].desc), P(host[s_].desc), 5, 5) == 0
    st = torch_stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    for knobs, want_nodes in (({}, 4), ({b"box.coalesce": 0, b"launch.capture_width": 1}, 75)):
        for k, v in knobs.items():
            lib.vpp_set_tuning(k, v)
        try:
            dev = [DeviceImage.from_host(x) for x in rng_ims()]
            import torch
            torch.cuda.synchronize()
            graph = ctypes.c_void_p()
            capi.check(lib.vpp_graph_begin(sp))
            for d, s_ in seq:
                capi.check(lib.vpp_box_filter(P(dev[d].desc), P(dev[s_].desc), 5, 5, sp))
            capi.check(lib.vpp_graph_end(sp, 0, ctypes.byref(graph)))
            assert lib.vpp_deferred_pending() == 0          # vpp_graph_end recorded the last window
            assert _kernel_nodes(lib, graph) == want_nodes
            capi.check(lib.vpp_graph_launch(graph, sp)); capi.check(lib.vpp_sync(sp))
            for k in range(210):
                np.testing.assert_array_equal(dev[k].download().raw, host[k].raw, err_msg=f"image {k} ({knobs})")
            capi.check(lib.vpp_graph_destroy(graph))
        finally:
            for k in knobs:
                lib.vpp_set_tuning(k, -1)


def torch_stream():
    import torch
    return torch.cuda.Stream()


def test_recorded_adds_are_batched(lib, orc):
    """The same for `A = B + C` on flat int images: 20 recorded calls are 2 kernel nodes (16 + 4 triples), a call that reads an earlier sum gets its own."""
    import torch
    n = 20
    B = [rand_image(64, 128, vi.I32, seed=400 + k, lo=0, hi=2**20) for k in range(n)]
    C = [rand_image(64, 128, vi.I32, seed=440 + k, lo=0, hi=2**20) for k in range(n)]
    dB, dC = [DeviceImage.from_host(x) for x in B], [DeviceImage.from_host(x) for x in C]
    dA = [DeviceImage(64, 128, vi.I32) for _ in range(n + 1)]
    st = torch_stream(); sp = ctypes.c_void_p(st.cuda_stream)
    torch.cuda.synchronize()
    graph = ctypes.c_void_p()
    capi.check(lib.vpp_graph_begin(sp))
    for k in range(n):
        capi.check(lib.vpp_pixelwise_binary(0, P(dA[k].desc), P(dB[k].desc), P(dC[k].desc), sp))
    capi.check(lib.vpp_pixelwise_binary(0, P(dA[n].desc), P(dA[3].desc), P(dA[17].desc), sp))     # reads two earlier sums
    capi.check(lib.vpp_graph_end(sp, 0, ctypes.byref(graph)))
    assert _kernel_nodes(lib, graph) == 3
    capi.check(lib.vpp_graph_launch(graph, sp)); capi.check(lib.vpp_sync(sp))
    for k in range(n):
        np.testing.assert_array_equal(dA[k].download().view(), B[k].view() + C[k].view())
    np.testing.assert_array_equal(dA[n].download().view(), B[3].view() + C[3].view() + B[17].view() + C[17].view())
    capi.check(lib.vpp_graph_destroy(graph))


def test_deferred_per_frame_calls_through_the_c_abi(lib, orc):
    """vpp_box_filter_deferred / vpp_pixelwise_binary_deferred / vpp_rgb_to_graylevel_deferred (include/vpp_amd.h, round 5): the reference's one-frame-per-call form
    without a launch per call.  70 frames: a window of 64 goes out by itself, the rest at vpp_flush; a frame that reads a pending result does not join it; argument
    errors are reported at the call; any other entry point (here vpp_fill_border, vpp_sync) launches the window first; every result against the oracle."""
    lib.vpp_deferred_flushes.restype = ctypes.c_ulonglong
    st = capi.stream_ptr()
    n, nr, nc = 70, 64, 176
    srcs, wants = [], []
    for k in range(n):
        h = rand_image(nr, nc, vi.U8, 3, border=2, seed=500 + k, align=16, fill_border=True)
        w = h.like(border=0); assert orc.orc_box_filter(P(w.desc), P(h.desc), 5, 5) == 0
        srcs.append(DeviceImage.from_host(h)); wants.append(w)
    dsts = [DeviceImage(nr, nc, vi.U8, 3, 0, 16) for _ in range(n)]
    _sync(lib)
    assert lib.vpp_deferred_pending() == 0
    f0 = lib.vpp_deferred_flushes()
    for k in range(n):
        capi.check(lib.vpp_box_filter_deferred(P(dsts[k].desc), P(srcs[k].desc), 5, 5, st))
        assert lib.vpp_deferred_pending() == (k + 1) % 64
    assert lib.vpp_deferred_flushes() == f0 + 1
    capi.check(lib.vpp_flush(st)); _sync(lib)
    assert lib.vpp_deferred_pending() == 0 and lib.vpp_deferred_flushes() == f0 + 2
    for d, w in zip(dsts, wants):
        np.testing.assert_array_equal(d.download().view(), w.view())
    # errors at the call, nothing joins the window
    bad = DeviceImage(nr, nc, vi.U8, 3, 1, 16)   # border 1 < 2
    assert lib.vpp_box_filter_deferred(P(dsts[0].desc), P(bad.desc), 5, 5, st) == capi.ERR_BORDER_TOO_SMALL and lib.vpp_deferred_pending() == 0
    assert lib.vpp_box_filter_deferred(P(dsts[0].desc), P(dsts[0].desc), 5, 5, st) != 0
    # data flow: the second call reads the first one's pending result -> the first is launched, the second opens a new window; another entry point flushes it
    a, b, c, e = (rand_image(135, 240, vi.I32, seed=600 + q, lo=0, hi=2**29) for q in range(4))
    da, db, dc, de = (DeviceImage.from_host(x) for x in (a, b, c, e))
    capi.check(lib.vpp_pixelwise_binary_deferred(0, P(da.desc), P(db.desc), P(dc.desc), st)); assert lib.vpp_deferred_pending() == 1
    capi.check(lib.vpp_pixelwise_binary_deferred(0, P(de.desc), P(da.desc), P(db.desc), st)); assert lib.vpp_deferred_pending() == 1
    capi.check(lib.vpp_fill_border(P(srcs[0].desc), 0, None, st)); assert lib.vpp_deferred_pending() == 0     # anything else this thread queues comes after the window
    _sync(lib)
    np.testing.assert_array_equal(da.download().view(), b.view() + c.view())
    np.testing.assert_array_equal(de.download().view(), b.view() + c.view() + b.view())
    # the frame ingest, 5 frames, against the oracle; vpp_sync alone launches and waits
    frames = [rand_image(72, 160, vi.U8, 3, border=0, seed=700 + k) for k in range(5)]
    dfr = [DeviceImage.from_host(f) for f in frames]; dg = [DeviceImage(72, 160, vi.U8, 1, 3, 32) for _ in range(5)]
    for k in range(5):
        capi.check(lib.vpp_rgb_to_graylevel_deferred(P(dg[k].desc), P(dfr[k].desc), 1, st))
    assert lib.vpp_deferred_pending() == 5
    capi.check(lib.vpp_sync(st)); assert lib.vpp_deferred_pending() == 0
    for k in range(5):
        want = HostImage(72, 160, vi.U8, 1, 3); assert orc.orc_rgb_to_graylevel(P(want.desc), P(frames[k].desc), 1) == 0
        np.testing.assert_array_equal(dg[k].download().view(with_border=True), want.view(with_border=True))


def test_held_back_calls_cross_host_threads(lib, orc):
    """The window is per host thread, the ORDER is per stream (round 6): frames one thread has held back are launched by whatever any other thread queues on that
    stream afterwards (here the main thread's vpp_sync — the first thread has returned from its calls and handed the stream over, it has neither flushed nor
    ended), and a thread that ends with frames held back launches them (it used to drop them silently)."""
    import threading
    import torch
    st = torch_stream(); sp = ctypes.c_void_p(st.cuda_stream)
    n, nr, nc = 15, 64, 176
    srcs, wants = [], []
    for k in range(n):
        h = rand_image(nr, nc, vi.U8, 3, border=2, seed=900 + k, align=16, fill_border=True)
        w = h.like(border=0); assert orc.orc_box_filter(P(w.desc), P(h.desc), 5, 5) == 0
        srcs.append(DeviceImage.from_host(h)); wants.append(w)
    dsts = [DeviceImage(nr, nc, vi.U8, 3, 0, 16) for _ in range(n)]
    capi.check(lib.vpp_sync(sp)); torch.cuda.synchronize()
    handed_over, go_on = threading.Event(), threading.Event()
    seen = {}

    def worker():
        try:
            torch.cuda.set_device(0)
            for k in range(10):
                capi.check(lib.vpp_box_filter_deferred(P(dsts[k].desc), P(srcs[k].desc), 5, 5, sp))
            seen["pending_at_hand_over"] = lib.vpp_deferred_pending()
            handed_over.set()
            assert go_on.wait(60)
            seen["pending_after_foreign_sync"] = lib.vpp_deferred_pending()
            for k in range(10, n):
                capi.check(lib.vpp_box_filter_deferred(P(dsts[k].desc), P(srcs[k].desc), 5, 5, sp))
            seen["pending_at_exit"] = lib.vpp_deferred_pending()
        except BaseException as e:   # noqa: BLE001 — reported by the main thread
            seen["error"] = e
            handed_over.set()

    t = threading.Thread(target=worker); t.start()
    assert handed_over.wait(60) and "error" not in seen, seen.get("error")
    assert lib.vpp_deferred_pending() == 0                      # (the main thread's own window)
    capi.check(lib.vpp_sync(sp))                                 # launches the worker's 10 frames, then waits for them
    for k in range(10):
        np.testing.assert_array_equal(dsts[k].download().view(), wants[k].view(), err_msg=f"frame {k}: held back by the worker, synchronised by the main thread")
    go_on.set(); t.join(60)
    assert not t.is_alive() and "error" not in seen, seen.get("error")
    assert seen == {"pending_at_hand_over": 10, "pending_after_foreign_sync": 0, "pending_at_exit": 5}
    capi.check(lib.vpp_sync(sp))                                 # the worker has ended: its last 5 frames were launched when it did
    for k in range(10, n):
        np.testing.assert_array_equal(dsts[k].download().view(), wants[k].view(), err_msg=f"frame {k}: held back when its thread ended")


def test_held_back_calls_of_several_threads_on_one_stream(lib, orc):
    """Four host threads hold back frames on ONE stream at the same time and each of them also queues other work (vpp_fill_border, vpp_sync) on it, which launches
    whatever window waits on that stream — its own or another thread's, while that thread may be appending to it.  Every thread's frames are its own, so every
    result is defined whatever the interleaving: all of them against the oracle, over several rounds."""
    import threading
    import torch
    st = torch_stream(); sp = ctypes.c_void_p(st.cuda_stream)
    T, per, rounds, nr, nc = 4, 9, 6, 48, 176
    srcs, dsts, wants = {}, {}, {}
    for t in range(T):
        for k in range(per):
            h = rand_image(nr, nc, vi.U8, 3, border=2, seed=1200 + t * per + k, align=16, fill_border=True)
            w = h.like(border=0); assert orc.orc_box_filter(P(w.desc), P(h.desc), 5, 5) == 0
            srcs[t, k] = DeviceImage.from_host(h); dsts[t, k] = DeviceImage(nr, nc, vi.U8, 3, 0, 16); wants[t, k] = w
    scratch = [DeviceImage.from_host(rand_image(32, 64, vi.U8, 1, border=3, seed=1300 + t)) for t in range(T)]
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(T)

    def worker(t):
        try:
            torch.cuda.set_device(0)
            start.wait(30)
            for r in range(rounds):
                for k in range(per):
                    capi.check(lib.vpp_box_filter_deferred(P(dsts[t, k].desc), P(srcs[t, k].desc), 5, 5, sp))
                    if (k + t + r) % 4 == 0:
                        capi.check(lib.vpp_fill_border(P(scratch[t].desc), 0, None, sp))     # another entry point on the shared stream: launches the windows that wait on it
                if (r + t) % 2 == 0:
                    capi.check(lib.vpp_sync(sp))
        except BaseException as e:   # noqa: BLE001
            errors.append((t, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errors, errors
    capi.check(lib.vpp_sync(sp))          # (every worker has ended: whatever it still held back was launched when it did)
    for key, d in dsts.items():
        np.testing.assert_array_equal(d.download().view(), wants[key].view(), err_msg=f"thread {key[0]} frame {key[1]}")


def test_a_capture_the_library_did_not_begin_records_every_call_at_once(lib, orc):
    """vpp_graph_end closes the last held-back window of a recorded stream; a capture begun by other means (here torch's) ends where the library cannot see it, so
    there nothing may be held back: plain and deferred per-frame calls record their own node each, and the replayed graph carries all of them."""
    import torch
    n, nr, nc = 6, 48, 176
    srcs, wants = [], []
    for k in range(n):
        h = rand_image(nr, nc, vi.U8, 3, border=2, seed=950 + k, align=16, fill_border=True)
        w = h.like(border=0); assert orc.orc_box_filter(P(w.desc), P(h.desc), 5, 5) == 0
        srcs.append(DeviceImage.from_host(h)); wants.append(w)
    dsts = [DeviceImage(nr, nc, vi.U8, 3, 0, 16) for _ in range(n)]
    st = torch_stream(); sp = ctypes.c_void_p(st.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        for k in range(n):
            f = lib.vpp_box_filter if k % 2 == 0 else lib.vpp_box_filter_deferred
            capi.check(f(P(dsts[k].desc), P(srcs[k].desc), 5, 5, sp))
            assert lib.vpp_deferred_pending() == 0
    for d in dsts:
        assert not d.download().view().any()        # recorded, not run
    g.replay(); torch.cuda.synchronize()
    for d, w in zip(dsts, wants):
        np.testing.assert_array_equal(d.download().view(), w.view())
