"""GPU parity (through the C ABI) of the steps either side of the algorithms in the reference's video loop: rgb_to_graylevel,
the fused frame ingest, and video_extruder's re-detection mask.  Bit-exact against the CPU oracle; 4K through properties."""
import ctypes

import numpy as np
import pytest
import torch

from util import P, rand_image, HostImage, DeviceImage
from vpp_amd import image as vi
from vpp_amd import capi

pytestmark = pytest.mark.gpu
V = ctypes.c_void_p


@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("shape,sborder,dborder,mirror,align", [
    ((37, 53), 3, 3, 0, 32), ((37, 53), 0, 0, 0, 32), ((64, 128), 2, 1, 0, 16), ((1, 1), 0, 0, 0, 16), ((3, 5), 2, 2, 0, 1),
    ((40, 64), 0, 3, 1, 32), ((19, 23), 0, 5, 1, 16), ((100, 100), 2, 2, 1, 32), ((7, 9), 0, 7, 1, 4), ((270, 481), 0, 3, 1, 32),
])
def test_rgb_to_graylevel_matches_oracle(lib, orc, ch, shape, sborder, dborder, mirror, align):
    src = rand_image(*shape, vi.U8, ch, border=sborder, seed=31, fill_border=True, align=align)
    want = HostImage(*shape, vi.U8, 1, dborder, align); want.raw[:] = 0xA5
    assert orc.orc_rgb_to_graylevel(P(want.desc), P(src.desc), mirror) == 0
    got = HostImage(*shape, vi.U8, 1, dborder, align); got.raw[:] = 0xA5
    ds, dd = DeviceImage.from_host(src), DeviceImage.from_host(got)
    capi.check(lib.vpp_rgb_to_graylevel(P(dd.desc), P(ds.desc), mirror, capi.stream_ptr()))
    np.testing.assert_array_equal(dd.download().raw, want.raw)  # bit-exact, nothing outside the mapped region touched


def test_ingest_4k_properties(lib):
    """4K vuchar3 -> gray, border 3: interior = integer mean, border = symmetric mirror of the interior."""
    src = rand_image(2160, 3840, vi.U8, 3, border=0, seed=32)
    ds, dd = DeviceImage.from_host(src), DeviceImage(2160, 3840, vi.U8, 1, 3)
    capi.check(lib.vpp_rgb_to_graylevel(P(dd.desc), P(ds.desc), 1, capi.stream_ptr()))
    g = dd.download().view(with_border=True)[..., 0]
    s = src.view().astype(np.int32)
    inner = ((s[..., 0] + s[..., 1] + s[..., 2]) // 3).astype(np.uint8)
    np.testing.assert_array_equal(g, np.pad(inner, 3, mode="symmetric"))


@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("shape,nlevels,border,align", [((40, 56), 3, 3, 32), ((41, 57), 3, 3, 32), ((35, 33), 2, 3, 16), ((18, 300), 3, 3, 32), ((271, 481), 3, 14, 32),
                                                          ((64, 130), 3, 3, 1), ((1080, 1920), 3, 3, 32), ((2160, 3840), 3, 18, 32), ((96, 96), 4, 3, 32), ((50, 70), 1, 2, 32)])
def test_rgb_pyramid_build_equals_the_ingest_followed_by_the_pyramid(lib, orc, ch, shape, nlevels, border, align):
    """vpp_rgb_pyramid_build (frame ingest fused with the image pyramid, one launch) leaves in every level, borders included, exactly what the
    oracle's rgb_to_graylevel(mirror) + pyramid chain leaves; frames cut by every tile edge, unaligned pitches, 4 levels / 1 level (chain fallback)."""
    import pyr
    if shape == (2160, 3840) and ch == 4:
        pytest.skip("4K once")
    rgb = rand_image(*shape, vi.U8, ch, border=0, seed=35, align=align)
    gray = HostImage(*shape, vi.U8, 1, border, 32)
    assert orc.orc_rgb_to_graylevel(P(gray.desc), P(rgb.desc), 1) == 0
    want = pyr.host_pyramid(orc, gray, nlevels, border)
    levels = [DeviceImage(nr, nc, vi.U8, 1, border) for nr, nc in pyr.level_dims(shape[0], shape[1], nlevels)]
    capi.check(lib.vpp_rgb_pyramid_build(vi.desc_array(levels), nlevels, P(DeviceImage.from_host(rgb).desc), capi.stream_ptr()))
    for h, d in zip(want, levels):
        np.testing.assert_array_equal(d.download().raw, h.raw)
    assert lib.vpp_rgb_pyramid_build(vi.desc_array(levels), nlevels, P(levels[0].desc), None) != 0   # the frame must be x3 / x4


def test_rgb_to_graylevel_rejects_bad_arguments(lib):
    a, b = DeviceImage(8, 8, vi.U8, 3), DeviceImage(8, 8, vi.U8, 1)
    assert lib.vpp_rgb_to_graylevel(P(a.desc), P(a.desc), 0, None) != 0            # dst must be x1
    assert lib.vpp_rgb_to_graylevel(P(b.desc), P(b.desc), 0, None) != 0            # src must be x3 / x4
    c = DeviceImage(9, 8, vi.U8, 1)
    assert lib.vpp_rgb_to_graylevel(P(c.desc), P(a.desc), 0, None) != 0            # domain mismatch
    d = DeviceImage(2, 2, vi.U8, 1, 3)
    assert lib.vpp_rgb_to_graylevel(P(d.desc), P(DeviceImage(2, 2, vi.U8, 3).desc), 1, None) != 0  # mirror border > image
    # the batch entry point answers as the n calls would: a frame k > 0 of another element type is not folded into frame 0's launch (round-4 advisor finding)
    srcs = [DeviceImage(8, 16, vi.U8, 3) for _ in range(3)]
    dsts = [DeviceImage(8, 16, vi.U8, 1), DeviceImage(8, 16, vi.U8, 1), DeviceImage(8, 16, vi.U16, 1)]   # same domain and pitch (32), u16 pixels
    single = lib.vpp_rgb_to_graylevel(P(dsts[2].desc), P(srcs[2].desc), 0, None)
    assert single != 0
    assert lib.vpp_rgb_to_graylevel_batch(vi.desc_array(dsts), vi.desc_array(srcs), 3, 0, None) == single
    inpl = [DeviceImage(8, 16, vi.U8, 3) for _ in range(2)]
    assert lib.vpp_rgb_to_graylevel_batch(vi.desc_array(inpl), vi.desc_array(inpl), 2, 0, None) != 0   # in place, frame 0 included


@pytest.mark.parametrize("shape,border,spacing,n", [((60, 80), 10, 10, 4), ((2160, 3840), 10, 10, 80000), ((50, 50), 3, 5, 40), ((20, 20), 10, 10, 0)])
def test_keypoint_mask_matches_oracle(lib, orc, shape, border, spacing, n):
    rng = np.random.default_rng(33)
    rc = np.stack([rng.integers(0, shape[0], n), rng.integers(0, shape[1], n)], 1).astype(np.int32)
    want = HostImage(*shape, vi.U8, 1, border)
    assert orc.orc_keypoint_mask(P(want.desc), rc.ctypes.data_as(V), n, spacing) == 0
    dm = DeviceImage(*shape, vi.U8, 1, border)
    drc = torch.from_numpy(rc.reshape(-1).copy()).cuda() if n else None
    capi.check(lib.vpp_keypoint_mask(P(dm.desc), V(drc.data_ptr()) if n else None, n, spacing, capi.stream_ptr()))
    np.testing.assert_array_equal(dm.download().view(with_border=True), want.view(with_border=True))


@pytest.mark.parametrize("shape,border,align", [((3, 3), 1, 32), ((37, 53), 1, 32), ((40, 64), 3, 16), ((129, 1001), 3, 32), ((64, 130), 3, 1), ((1, 7), 2, 32), ((2160, 3840), 3, 32)])
def test_lbp_transform_matches_oracle(lib, orc, shape, border, align):
    src = rand_image(*shape, vi.U8, 1, border=border, seed=34, fill_border=True, align=align)
    want = HostImage(*shape, vi.U8, 1, 0, align)
    assert orc.orc_lbp_transform(P(want.desc), P(src.desc)) == 0
    ds, dd = DeviceImage.from_host(src), DeviceImage(*shape, vi.U8, 1, 0, align)
    capi.check(lib.vpp_lbp_transform(P(dd.desc), P(ds.desc), capi.stream_ptr()))
    np.testing.assert_array_equal(dd.download().view(), want.view())
    assert lib.vpp_lbp_transform(P(dd.desc), P(DeviceImage(*shape, vi.U8, 1, 0).desc), None) != 0   # border 0: refused


def test_allgather_tracks_single_rank(lib):
    """The C-ABI exchange step on a 1-rank communicator: the all-gather of 20-byte keypoint records returns the shard itself.
    (World sizes > 1 are covered on CPU with gloo in test_multi_gpu_cpu.py and run on the 8-GPU node by bench.py --gpus N.)"""
    import pyr
    idbuf = (ctypes.c_char * 128)()
    capi.check(lib.vpp_comm_unique_id(idbuf))
    comm = ctypes.c_void_p()
    capi.check(lib.vpp_comm_init(ctypes.byref(comm), 1, idbuf, 0))
    kps = pyr.make_keypoints(pyr.grid_keypoints(480, 640, 1000, margin=16))
    shard = torch.from_numpy(kps.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.zeros_like(shard)
    capi.check(lib.vpp_allgather_tracks(comm, V(shard.data_ptr()), len(kps), V(out.data_ptr()), capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    assert torch.equal(out, shard)
    capi.check(lib.vpp_comm_destroy(comm))
    assert lib.vpp_comm_init(ctypes.byref(comm), 1, idbuf, 3) != 0   # rank out of range


def test_runtime_pools(lib):
    """vpp_malloc / vpp_free keep freed blocks by size class; vpp_malloc_host hands out pinned, device-visible staging memory."""
    p1, p2 = ctypes.c_void_p(), ctypes.c_void_p()
    capi.check(lib.vpp_malloc(1000000, ctypes.byref(p1)))
    capi.check(lib.vpp_free(p1))
    capi.check(lib.vpp_malloc(1000001, ctypes.byref(p2)))           # same 4 KiB size class: the cached block comes back
    assert p1.value == p2.value
    capi.check(lib.vpp_free(p2))
    capi.check(lib.vpp_release_cached_memory())
    h = ctypes.c_void_p()
    capi.check(lib.vpp_malloc_host(4096, ctypes.byref(h)))
    host = (ctypes.c_uint8 * 4096).from_address(h.value)
    src = torch.arange(4096, dtype=torch.int32, device="cuda").to(torch.uint8)
    capi.check(lib.vpp_memcpy_d2d(h, V(src.data_ptr()), 4096, capi.stream_ptr()))   # a device-side copy INTO the pinned block
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    assert bytes(host[:16]) == bytes(range(16))
    capi.check(lib.vpp_free_host(h))
    assert lib.vpp_free_host(ctypes.c_void_p(12345)) != 0          # not one of ours


def _merge_model(pos, age, spacing, nr, nc):
    """video_extruder.hpp:60-84 restated literally: the serial champion loop over an (nr/s) x (nc/s) index image."""
    idx = {}
    age = age.copy()
    removed = np.zeros(len(age), np.uint8)
    for i in range(len(age)):
        cell = (int(pos[i, 0]) // spacing, int(pos[i, 1]) // spacing)
        if cell in idx:
            o = idx[cell]
            other_age = int(age[o])
            if other_age < age[i]:
                removed[o] = 1; age[o] = 0; idx[cell] = i
            if other_age > age[i]:
                removed[i] = 1; age[i] = 0
        else:
            idx[cell] = i
    return removed


@pytest.mark.parametrize("n,nr,nc,spacing,max_age", [(4000, 90, 130, 10, 4), (20000, 211, 317, 7, 30), (3000, 40, 40, 10, 2), (1, 50, 50, 10, 3), (5000, 64, 64, 64, 6)])
def test_keypoint_merge_matches_the_serial_champion_loop(lib, orc, n, nr, nc, spacing, max_age):
    rng = np.random.default_rng(n + spacing)
    prev = np.stack([rng.integers(0, nr, n), rng.integers(0, nc, n)], 1).astype(np.int32)
    moved = (prev + rng.integers(-12, 13, size=prev.shape)).astype(np.int32)
    matched = (rng.random(n) < 0.8).astype(np.uint8)
    moved[matched == 0] = prev[matched == 0]                      # the read-back kernel leaves unmatched keypoints where they were
    age_prev = rng.integers(0, max_age + 1, n).astype(np.int32)   # 0 = a keypoint that died earlier and is still in the container
    inside = (moved[:, 0] >= 0) & (moved[:, 0] < nr) & (moved[:, 1] >= 0) & (moved[:, 1] < nc)
    pos = np.where(((matched == 1) & inside)[:, None], moved, prev)
    age = np.where(matched == 1, np.where(inside, age_prev + 1, 0), age_prev).astype(np.int32)
    want = np.zeros(n, np.uint8)
    assert orc.orc_keypoint_merge(np.ascontiguousarray(pos, np.int32).ctypes.data_as(ctypes.c_void_p), age.ctypes.data_as(ctypes.c_void_p), n, nr, nc, spacing,
                                  want.ctypes.data_as(ctypes.c_void_p)) == 0               # the oracle's literal restatement of video_extruder.hpp:60-84
    np.testing.assert_array_equal(want, _merge_model(pos, age, spacing, nr, nc))           # ... and an independent Python model of the same loop
    assert n < 10 or 0 < want.sum() < n
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dm, dp, dmt, da = t(moved), t(prev), t(matched), t(age_prev)
    out = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    for _ in range(2):  # twice: the scratch list heads are reset on every call
        capi.check(lib.vpp_keypoint_merge(ctypes.c_void_p(dm.data_ptr()), ctypes.c_void_p(dp.data_ptr()), ctypes.c_void_p(dmt.data_ptr()),
                                          ctypes.c_void_p(da.data_ptr()), n, nr, nc, spacing, ctypes.c_void_p(out.data_ptr()), capi.stream_ptr()))
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    assert lib.vpp_keypoint_merge(None, None, None, None, 5, nr, nc, spacing, None, None) == capi.ERR_INVALID_ARG


@pytest.mark.parametrize("shape,th,border", [((30, 41), 20, 3), ((64, 64), 5, 3), ((7, 9), 0, 4), ((50, 33), -2, 3), ((129, 200), 12, 5), ((1, 1), 3, 3)])
def test_fast9_dense_matches_oracle(lib, orc, shape, th, border):
    from util import rects_image, u8_image
    src = u8_image(rects_image(*shape, seed=shape[0] + th), border=border)
    src.view(with_border=True)[..., 0] = np.pad(src.view()[..., 0], border, mode="symmetric")
    ds = DeviceImage.from_host(src)
    for dt in (vi.U8, vi.I32):
        want = HostImage(*shape, dt, 1)
        assert orc.orc_fast9_dense(P(want.desc), P(src.desc), th) == 0
        out = DeviceImage(*shape, dt, 1, border=1)
        capi.check(lib.vpp_fast9_dense(P(out.desc), P(ds.desc), th, capi.stream_ptr()))
        np.testing.assert_array_equal(out.download().view(), want.view())
    small = DeviceImage(*shape, vi.U8, 1, border=2)
    assert lib.vpp_fast9_dense(P(out.desc), P(small.desc), th, None) == capi.ERR_BORDER_TOO_SMALL


def test_fast9_dense_4k_is_the_corrected_ring_detector(lib):
    """Size-independent property at the BASELINE frame size: the dense flags are exactly the keypoints vpp_fast9_detect reports
    on the true ring (compat = corrected, raw mode, no mask) — two independent kernels, one definition (fast.hpp:84-112)."""
    from util import rects_image, u8_image
    from test_gpu_algos import gpu_detect
    src = u8_image(rects_image(2160, 3840, seed=31), border=3)
    src.view(with_border=True)[..., 0] = np.pad(src.view()[..., 0], 3, mode="symmetric")
    ds = DeviceImage.from_host(src)
    out = DeviceImage(2160, 3840, vi.U8, 1)
    capi.check(lib.vpp_fast9_dense(P(out.desc), P(ds.desc), 20, capi.stream_ptr()))
    flags = out.download().view()[..., 0]
    rc, _ = gpu_detect(lib, ds, 20, mode=0, bs=10, compat=1, cap=4000000)
    assert len(rc) > 10000
    want = np.zeros((2160, 3840), np.uint8)
    want[rc[:, 0], rc[:, 1]] = 1
    np.testing.assert_array_equal(flags, want)


@pytest.mark.parametrize("dtype", [vi.U8, vi.I8, vi.U16, vi.I16, vi.I32, vi.U32, vi.F32])
# (the last two: several workgroup spans per row of blocks, blocks cut by the right / bottom edge — blockwise_maxima_rows_kernel; block sizes 1 and 300 on 4-byte pixels keep one lane per block)
@pytest.mark.parametrize("shape,bs", [((20, 30), 10), ((23, 31), 10), ((9, 9), 4), ((5, 40), 7), ((16, 16), 1), ((70, 300), 300), ((37, 2100), 10), ((41, 1300), 33)])
def test_blockwise_maxima_filter_matches_oracle(lib, orc, dtype, shape, bs):
    signed = dtype in (vi.I8, vi.I16, vi.I32, vi.F32)
    img = rand_image(*shape, dtype, 1, border=2, seed=bs + shape[1], lo=-3 if signed else 0, hi=6, fill_border=True)
    d = DeviceImage.from_host(img)
    assert orc.orc_blockwise_maxima_filter(P(img.desc), bs) == 0
    capi.check(lib.vpp_blockwise_maxima_filter(P(d.desc), bs, capi.stream_ptr()))
    np.testing.assert_array_equal(d.download().view(with_border=True), img.view(with_border=True))  # the border is not touched


def test_blockwise_maxima_filter_4k_properties(lib):
    img = rand_image(2160, 3840, vi.U8, 1, seed=8)
    a = img.view()[..., 0].copy()
    d = DeviceImage.from_host(img)
    capi.check(lib.vpp_blockwise_maxima_filter(P(d.desc), 10, capi.stream_ptr()))
    once = d.download().view()[..., 0].copy()
    blocks = once.reshape(216, 10, 384, 10)
    assert ((blocks != 0).sum(axis=(1, 3)) <= 1).all()                                          # at most one survivor per block
    np.testing.assert_array_equal(blocks.max(axis=(1, 3)), a.reshape(216, 10, 384, 10).max(axis=(1, 3)))  # and it is the block maximum
    capi.check(lib.vpp_blockwise_maxima_filter(P(d.desc), 10, capi.stream_ptr()))
    np.testing.assert_array_equal(d.download().view()[..., 0], once)                            # idempotent


class _VeParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("detector_th", "keypoint_spacing", "detector_period", "max_trajectory_length", "nscales", "winsize", "propagation")]


def _tracker_state(lib, ve):
    n, fid = ctypes.c_int(), ctypes.c_int()
    capi.check(lib.vpp_video_extruder_count(ve, ctypes.byref(n), ctypes.byref(fid)))
    pos = np.zeros((n.value, 2), np.int32); vel = np.zeros((n.value, 2), np.int32); age = np.zeros(n.value, np.int32); ln = np.zeros(n.value, np.int32)
    capi.check(lib.vpp_video_extruder_keypoints(ve, pos.ctypes.data_as(V), vel.ctypes.data_as(V), age.ctypes.data_as(V), n.value, capi.stream_ptr()))
    capi.check(lib.vpp_video_extruder_trajectories(ve, ln.ctypes.data_as(V), None, None, None, None, n.value, capi.stream_ptr()))
    return fid.value, pos, vel, age, ln


@pytest.mark.parametrize("shape,winsize,nscales", [((120, 200), 9, 3), ((97, 131), 7, 2), ((64, 96), 5, 4)])
def test_one_frame_per_call_equals_the_two_frame_update(lib, shape, winsize, nscales):
    """vpp_video_extruder_push_frame (gray and colour frames in HBM) and vpp_video_extruder_push_host_frame (tight and pitched frames in host memory)
    leave the tracker in the state of vpp_video_extruder_step on the mirror-bordered gray frames: every position, velocity, age and trajectory
    length after 7 frames, re-detection every 3rd.  (The two-frame call is pinned against the reference by tests/cpp/video_extruder_parity.cc.)"""
    from vpp_amd.synth import texture, translate, rects_image
    nr, nc = shape
    T = 7
    base = texture(nr + 40, nc + 40, seed=9, sigma=1.5)
    rect = rects_image(nr + 40, nc + 40, seed=4).astype(np.float64)
    frames = []
    for t in range(T):
        f = 0.6 * translate(base, 0.9 * t, -1.3 * t) + 0.4 * translate(rect, 0.9 * t, -1.3 * t)
        frames.append(np.clip(np.rint(f[20:20 + nr, 20:20 + nc]), 0, 255).astype(np.uint8))
    par = _VeParams(10, 10, 3, 15, nscales, winsize, 2)
    lib.vpp_video_extruder_create.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def run(feed):
        ve = V()
        capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), nr, nc, 15))
        try:
            feed(ve)
            return _tracker_state(lib, ve)
        finally:
            lib.vpp_video_extruder_destroy(ve)

    def bordered(f):   # gray frame with a mirror-filled border of 3, in HBM
        h = HostImage(nr, nc, vi.U8, 1, 3)
        h.view()[..., 0] = f
        d = DeviceImage.from_host(h)
        capi.check(lib.vpp_fill_border(P(d.desc), 0, None, capi.stream_ptr()))   # VPP_BORDER_MIRROR
        return d

    def two_frame(ve):
        d = [bordered(f) for f in frames]
        for t in range(1, T):
            capi.check(lib.vpp_video_extruder_step(ve, P(d[t - 1].desc), P(d[t].desc), ctypes.byref(par), capi.stream_ptr()))

    def colour(f, ch):   # (g + d, g, g - d): the integer mean is g
        g = f.astype(np.int32); d = np.minimum(np.minimum(g, 255 - g), 11)
        h = HostImage(nr, nc, vi.U8, ch, 0)
        v = h.view(); v[..., 0] = g + d; v[..., 1] = g; v[..., 2] = g - d
        if ch == 4:
            v[..., 3] = 200
        return h

    def push_device(ch):
        def feed(ve):
            for f in frames:
                if ch == 1:
                    h = HostImage(nr, nc, vi.U8, 1, 0); h.view()[..., 0] = f
                else:
                    h = colour(f, ch)
                capi.check(lib.vpp_video_extruder_push_frame(ve, P(DeviceImage.from_host(h).desc), ctypes.byref(par), capi.stream_ptr()))
        return feed

    def push_host(ch, pad):
        def feed(ve):
            keep = []
            for f in frames:
                if ch == 1:
                    rows = f[..., None]
                else:
                    rows = colour(f, ch).view()
                buf = np.full((nr, nc * ch + pad), 0x5A, np.uint8)   # rows `pad` bytes apart from tight
                buf[:, :nc * ch] = rows.reshape(nr, nc * ch)
                keep.append(buf)
                desc = vi.ImageDesc(buf.ctypes.data, nr, nc, buf.shape[1], 0, vi.U8, ch)
                capi.check(lib.vpp_video_extruder_push_host_frame(ve, ctypes.byref(desc), ctypes.byref(par), capi.stream_ptr()))
                buf[:] = 0   # the call has read the buffer: overwriting it must not change anything
        return feed

    def push_host_two_buffers(ve):   # the two-buffer protocol: refill a buffer only after wait(back = 1) has vouched for it
        bufs = [np.zeros((nr, nc), np.uint8), np.zeros((nr, nc), np.uint8)]
        for t, f in enumerate(frames):
            capi.check(lib.vpp_video_extruder_wait_host_frame(ve, 1))
            bufs[t & 1][:] = f
            desc = vi.ImageDesc(bufs[t & 1].ctypes.data, nr, nc, nc, 0, vi.U8, 1)
            capi.check(lib.vpp_video_extruder_push_host_frame_nowait(ve, ctypes.byref(desc), ctypes.byref(par), capi.stream_ptr()))
        capi.check(lib.vpp_video_extruder_wait_host_frame(ve, 0))

    want = run(two_frame)
    assert want[0] == T - 2 and len(want[1]) > 20 and (want[3] > 0).sum() > 10 and (want[2] != 0).any()
    for name, feed in (("gray hbm", push_device(1)), ("rgb hbm", push_device(3)), ("rgba hbm", push_device(4)), ("gray host tight", push_host(1, 0)),
                       ("gray host pitched", push_host(1, 13)), ("rgb host pitched", push_host(3, 7)), ("rgba host tight", push_host(4, 0)), ("gray host, two buffers", push_host_two_buffers)):
        got = run(feed)
        assert got[0] == want[0], name
        for g, w in zip(got[1:], want[1:]):
            np.testing.assert_array_equal(g, w, err_msg=name)


def test_push_frame_rejects_bad_frames(lib):
    ve = V()
    lib.vpp_video_extruder_create.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), 64, 96, 15))
    par = _VeParams(10, 10, 5, 15, 3, 9, 2)
    try:
        assert lib.vpp_video_extruder_push_frame(ve, P(DeviceImage(64, 97, vi.U8, 1).desc), ctypes.byref(par), None) != 0      # another domain
        assert lib.vpp_video_extruder_push_frame(ve, P(DeviceImage(64, 96, vi.U8, 2).desc), ctypes.byref(par), None) != 0      # 2 channels
        assert lib.vpp_video_extruder_push_frame(ve, P(DeviceImage(64, 96, vi.F32, 1).desc), ctypes.byref(par), None) != 0     # not 8-bit
        buf = np.zeros((64, 96), np.uint8)
        short = vi.ImageDesc(buf.ctypes.data, 64, 96, 95, 0, vi.U8, 1)
        assert lib.vpp_video_extruder_push_host_frame(ve, ctypes.byref(short), ctypes.byref(par), None) != 0                   # pitch below a row
        null = vi.ImageDesc(None, 64, 96, 96, 0, vi.U8, 1)
        assert lib.vpp_video_extruder_push_host_frame(ve, ctypes.byref(null), ctypes.byref(par), None) != 0
        bad = _VeParams(10, 10, 5, 15, 9, 9, 2)                                                                                   # 9 scales
        assert lib.vpp_video_extruder_push_frame(ve, P(DeviceImage(64, 96, vi.U8, 1).desc), ctypes.byref(bad), None) != 0
        n, fid = ctypes.c_int(), ctypes.c_int()
        capi.check(lib.vpp_video_extruder_count(ve, ctypes.byref(n), ctypes.byref(fid)))
        assert (n.value, fid.value) == (0, -1)   # nothing ran
    finally:
        lib.vpp_video_extruder_destroy(ve)


def test_push_frame_restarts_when_the_pyramids_change_shape(lib):
    """A change of nscales / winsize between two pushes re-carves the tracker's pyramids: the frame after the change only becomes `prev` (documented in
    include/vpp_amd.h) — the state equals the two-frame updates with that one update left out."""
    from vpp_amd.synth import texture, translate, rects_image
    nr, nc, T = 96, 160, 7
    base = texture(nr + 40, nc + 40, seed=3, sigma=1.5); rect = rects_image(nr + 40, nc + 40, seed=8).astype(np.float64)
    frames = [np.clip(np.rint((0.6 * translate(base, 1.1 * t, 0.7 * t) + 0.4 * translate(rect, 1.1 * t, 0.7 * t))[20:20 + nr, 20:20 + nc]), 0, 255).astype(np.uint8) for t in range(T)]
    par_a, par_b = _VeParams(10, 10, 2, 15, 3, 9, 2), _VeParams(10, 10, 2, 15, 2, 7, 2)
    lib.vpp_video_extruder_create.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    switch = 4   # frames 0..3 with par_a, frames 4.. with par_b: the update (3 -> 4) does not run

    def bordered(f):
        h = HostImage(nr, nc, vi.U8, 1, 3); h.view()[..., 0] = f
        d = DeviceImage.from_host(h)
        capi.check(lib.vpp_fill_border(P(d.desc), 0, None, capi.stream_ptr()))
        return d
    states = []
    for mode in ("two_frame", "push"):
        ve = V(); capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), nr, nc, 15))
        try:
            if mode == "two_frame":
                d = [bordered(f) for f in frames]
                for t in range(1, T):
                    if t == switch:
                        continue
                    capi.check(lib.vpp_video_extruder_step(ve, P(d[t - 1].desc), P(d[t].desc), ctypes.byref(par_a if t < switch else par_b), capi.stream_ptr()))
            else:
                for t, f in enumerate(frames):
                    h = HostImage(nr, nc, vi.U8, 1, 0); h.view()[..., 0] = f
                    capi.check(lib.vpp_video_extruder_push_frame(ve, P(DeviceImage.from_host(h).desc), ctypes.byref(par_a if t < switch else par_b), capi.stream_ptr()))
            states.append(_tracker_state(lib, ve))
        finally:
            lib.vpp_video_extruder_destroy(ve)
    assert states[0][0] == T - 3 and len(states[0][1]) > 20
    assert states[1][0] == states[0][0]
    for g, w in zip(states[1][1:], states[0][1:]):
        np.testing.assert_array_equal(g, w)


def test_a_refused_update_does_not_advance_the_tracker(lib):
    """An update that is refused (frame2 without the 3-pixel border FAST needs, fast.hpp:937-938; a frame of another type) leaves frame_id and the
    container untouched — frame_id % detector_period decides which frames re-detect, so a clock that ran ahead would shift every later detection —
    and the sequence continued after the refusal ends in the state of the sequence without it."""
    from vpp_amd.synth import texture, translate, rects_image
    nr, nc, T = 96, 160, 6
    base = texture(nr + 40, nc + 40, seed=5, sigma=1.5); rect = rects_image(nr + 40, nc + 40, seed=6).astype(np.float64)
    frames = [np.clip(np.rint((0.6 * translate(base, 0.8 * t, -0.9 * t) + 0.4 * translate(rect, 0.8 * t, -0.9 * t))[20:20 + nr, 20:20 + nc]), 0, 255).astype(np.uint8) for t in range(T)]
    par = _VeParams(10, 10, 2, 15, 3, 9, 2)
    lib.vpp_video_extruder_create.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def bordered(f, border=3):
        h = HostImage(nr, nc, vi.U8, 1, border); h.view()[..., 0] = f
        d = DeviceImage.from_host(h)
        capi.check(lib.vpp_fill_border(P(d.desc), 0, None, capi.stream_ptr()))
        return d
    states = []
    for refuse in (False, True):
        ve = V(); capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), nr, nc, 15))
        try:
            d = [bordered(f) for f in frames]
            for t in range(1, T):
                if refuse and t in (1, 3):   # before the first update (empty container) and in the middle of the sequence
                    n0, f0 = ctypes.c_int(), ctypes.c_int()
                    capi.check(lib.vpp_video_extruder_count(ve, ctypes.byref(n0), ctypes.byref(f0)))
                    assert lib.vpp_video_extruder_step(ve, P(d[t - 1].desc), P(bordered(frames[t], 2).desc), ctypes.byref(par), capi.stream_ptr()) == capi.ERR_BORDER_TOO_SMALL
                    assert lib.vpp_video_extruder_step(ve, P(d[t - 1].desc), P(DeviceImage(nr, nc, vi.U8, 3, 3).desc), ctypes.byref(par), capi.stream_ptr()) == capi.ERR_UNSUPPORTED
                    n1, f1 = ctypes.c_int(), ctypes.c_int()
                    capi.check(lib.vpp_video_extruder_count(ve, ctypes.byref(n1), ctypes.byref(f1)))
                    assert (n1.value, f1.value) == (n0.value, f0.value)
                capi.check(lib.vpp_video_extruder_step(ve, P(d[t - 1].desc), P(d[t].desc), ctypes.byref(par), capi.stream_ptr()))
            states.append(_tracker_state(lib, ve))
        finally:
            lib.vpp_video_extruder_destroy(ve)
    assert states[0][0] == T - 2 and len(states[0][1]) > 20
    assert states[1][0] == states[0][0]
    for g, w in zip(states[1][1:], states[0][1:]):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("ch,mirror,n", [(3, 1, 5), (4, 1, 3), (3, 0, 70)])
def test_rgb_to_graylevel_batch_and_recorded_calls(lib, orc, ch, mirror, n):
    """vpp_rgb_to_graylevel_batch: n frames of one geometry in one launch (70 > the 64 a launch carries: two launches), every frame against the oracle; the same
    frames as per-frame calls recorded into a launch graph are recorded as batched nodes (the held-back window of common.hpp) with the same bytes; a
    frame of another geometry in the batch sends it out as the calls in sequence."""
    shape, border = (45, 150), 3
    srcs = [rand_image(*shape, vi.U8, ch, border=0 if mirror else border, seed=500 + k, fill_border=True) for k in range(n)]
    wants = []
    for s_ in srcs:
        w = HostImage(*shape, vi.U8, 1, border); w.raw[:] = 0x5A
        assert orc.orc_rgb_to_graylevel(P(w.desc), P(s_.desc), mirror) == 0
        wants.append(w)
    dsrc = [DeviceImage.from_host(x) for x in srcs]

    def fresh():
        out = []
        for _ in range(n):
            h = HostImage(*shape, vi.U8, 1, border); h.raw[:] = 0x5A
            out.append(DeviceImage.from_host(h))
        return out
    ddst = fresh()
    capi.check(lib.vpp_rgb_to_graylevel_batch(vi.desc_array(ddst), vi.desc_array(dsrc), n, mirror, capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    for d, w in zip(ddst, wants):
        np.testing.assert_array_equal(d.download().raw, w.raw)
    # recorded per-frame calls
    ddst = fresh()
    st = torch.cuda.Stream(); sp = ctypes.c_void_p(st.cuda_stream)
    torch.cuda.synchronize()
    graph = ctypes.c_void_p()
    capi.check(lib.vpp_graph_begin(sp))
    for k in range(n):
        capi.check(lib.vpp_rgb_to_graylevel(P(ddst[k].desc), P(dsrc[k].desc), mirror, sp))
    capi.check(lib.vpp_graph_end(sp, 0, ctypes.byref(graph)))
    nodes = ctypes.c_int()
    capi.check(lib.vpp_debug_graph_kernel_nodes(graph, ctypes.byref(nodes)))
    assert nodes.value == (n + 63) // 64
    capi.check(lib.vpp_graph_launch(graph, sp)); capi.check(lib.vpp_sync(sp))
    for d, w in zip(ddst, wants):
        np.testing.assert_array_equal(d.download().raw, w.raw)
    capi.check(lib.vpp_graph_destroy(graph))
    # mixed geometries
    odd = rand_image(shape[0] + 2, shape[1], vi.U8, ch, border=0 if mirror else border, seed=599, fill_border=True)
    wodd = HostImage(shape[0] + 2, shape[1], vi.U8, 1, border); wodd.raw[:] = 0x5A
    assert orc.orc_rgb_to_graylevel(P(wodd.desc), P(odd.desc), mirror) == 0
    h = HostImage(shape[0] + 2, shape[1], vi.U8, 1, border); h.raw[:] = 0x5A
    d2 = fresh()[:2] + [DeviceImage.from_host(h)]
    s2 = dsrc[:2] + [DeviceImage.from_host(odd)]
    capi.check(lib.vpp_rgb_to_graylevel_batch(vi.desc_array(d2), vi.desc_array(s2), 3, mirror, capi.stream_ptr()))
    capi.check(lib.vpp_sync(capi.stream_ptr()))
    for d, w in zip(d2, [wants[0], wants[1], wodd]):
        np.testing.assert_array_equal(d.download().raw, w.raw)
