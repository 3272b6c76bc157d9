"""Golden vectors produced by the reference itself (tests/golden/make_golden.py): the oracle must reproduce them on CPU
and the HIP path (through the C ABI) on the GPU.  The input CRC guards against the seeded generators drifting."""
import ctypes
import os

import numpy as np
import pytest

import golden_cases as gc
import pyr
from util import P, DeviceImage, HostImage
from vpp_amd import image as vi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = ctypes.c_void_p


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


# ---------------- CPU: oracle == reference golden ----------------
def test_oracle_box_add(orc):
    src, dst = gc.box_case(); g = load("box")
    assert gc.crc(src.raw) == g["in_crc"]
    orc.orc_box_filter(P(dst.desc), P(src.desc), 5, 5)
    np.testing.assert_array_equal(dst.view(), g["out"])
    b, c, a = gc.add_case(); g = load("add")
    assert gc.crc(b.raw, c.raw) == g["in_crc"]
    orc.orc_pixelwise_binary(0, P(a.desc), P(b.desc), P(c.desc))
    np.testing.assert_array_equal(a.view(), g["out"])


def test_oracle_pyramid(orc):
    img, _ = gc.pyramid_case(); g = load("pyramid")
    assert gc.crc(img.raw) == g["in_crc"]
    for i, l in enumerate(pyr.host_pyramid(orc, img, 3, 3)):
        np.testing.assert_array_equal(l.view(with_border=True), g[f"l{i}"])


def test_oracle_fast9(orc):
    from test_oracle_algos import run_detect
    im = gc.fast_case(); g = load("fast9")
    assert gc.crc(im.raw) == g["in_crc"]
    for mode in (0, 1, 2):
        rc, sc = run_detect(orc, im, 20, mode=mode, compat=0)
        np.testing.assert_array_equal(rc, g[f"rc{mode}"]); np.testing.assert_array_equal(sc, g[f"sc{mode}"])


def test_oracle_lk(orc):
    i1, i2, kps = gc.pyrlk_case(); g = load("pyrlk")
    assert gc.crc(i1.raw, i2.raw, kps.view(np.uint8)) == g["in_crc"]
    hp1, hp2 = pyr.host_pyramid(orc, i1, 3, 5), pyr.host_pyramid(orc, i2, 3, 5)
    hg = pyr.host_grad_pyramid(orc, hp1[0], 3, 5, vi.F32)
    k = kps.copy()
    orc.orc_pyrlk_match(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), 3, k.ctypes.data_as(V), len(k), 7, ctypes.c_float(1e-4),
                        ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None)
    np.testing.assert_array_equal(k.view(np.uint8), g["kps"])
    j1, j2, pts = gc.lk_golden_case(); g = load("lucas_kanade")
    assert np.linalg.norm(g["flow"][0] - [2, 2]) < 0.05  # tests/pyrlk.cc:48-49, asserted on the reference's own output
    hp1, hp2 = pyr.host_pyramid(orc, j1, 2, 2), pyr.host_pyramid(orc, j2, 2, 2)
    hg = pyr.host_grad_pyramid(orc, hp1[0], 2, 2, vi.I32)
    flow = np.zeros((len(pts), 2), np.float32); dist = np.zeros(len(pts), np.float32)
    orc.orc_lucas_kanade(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), 2, pts.ctypes.data_as(V), None, len(pts), 5, 0, 50, 0,
                         flow.ctypes.data_as(V), dist.ctypes.data_as(V))
    np.testing.assert_array_equal(flow.view(np.uint32), g["flow"].view(np.uint32))
    np.testing.assert_array_equal(dist.view(np.uint32), g["dist"].view(np.uint32))


def test_oracle_sdof(orc):
    s1, s2, sk, par = gc.sdof_case(); g = load("sdof")
    assert gc.crc(s1.raw, s2.raw, sk) == g["in_crc"]
    p = np.zeros((len(sk), 2), np.int32); d = np.zeros(len(sk), np.int32); v = np.zeros(len(sk), np.uint8)
    orc.orc_semi_dense_optical_flow(P(s1.desc), P(s2.desc), sk.ctypes.data_as(V), len(sk), *par, p.ctypes.data_as(V), d.ctypes.data_as(V), v.ctypes.data_as(V))
    np.testing.assert_array_equal(p, g["pos"]); np.testing.assert_array_equal(d, g["dist"]); np.testing.assert_array_equal(v, g["valid"])


def test_oracle_ingest(orc):
    rgb, g1, rgba, g2 = gc.ingest_case(); g = load("ingest")
    assert gc.crc(rgb.raw, rgba.raw) == g["in_crc"]
    orc.orc_rgb_to_graylevel(P(g1.desc), P(rgb.desc), 1); orc.orc_rgb_to_graylevel(P(g2.desc), P(rgba.desc), 0)
    np.testing.assert_array_equal(g1.view(with_border=True), g["gray_mirror"]); np.testing.assert_array_equal(g2.view(with_border=True), g["gray_rgba"])


def test_oracle_fast_dense(orc):
    fim, ths = gc.fast_dense_case(); g = load("fast_dense")
    assert gc.crc(fim.raw) == g["in_crc"]
    for th in ths:
        for dt in (vi.U8, vi.I32):
            o = HostImage(fim.nrows, fim.ncols, dt, 1)
            assert orc.orc_fast9_dense(P(o.desc), P(fim.desc), th) == 0
            np.testing.assert_array_equal(np.packbits(o.view()[..., 0].astype(bool), axis=1), g["th_%d" % th])
    assert 0 < np.unpackbits(g["th_20"]).sum() < np.unpackbits(g["th_0"]).sum()


# ---------------- GPU: HIP path == reference golden ----------------
@pytest.mark.gpu
def test_gpu_matches_reference_golden(lib):
    import torch
    from vpp_amd import capi
    from test_gpu_algos import gpu_detect
    st = capi.stream_ptr()
    src, dst = gc.box_case()
    dd, ds = DeviceImage.from_host(dst), DeviceImage.from_host(src)  # keep every device buffer alive until the results are read
    capi.check(lib.vpp_box_filter(P(dd.desc), P(ds.desc), 5, 5, st))
    np.testing.assert_array_equal(dd.download().view(), load("box")["out"])
    b, c, a = gc.add_case()
    da, db, dc = DeviceImage.from_host(a), DeviceImage.from_host(b), DeviceImage.from_host(c)
    capi.check(lib.vpp_pixelwise_binary(0, P(da.desc), P(db.desc), P(dc.desc), st))
    np.testing.assert_array_equal(da.download().view(), load("add")["out"])
    img, _ = gc.pyramid_case(); g = load("pyramid")
    for i, l in enumerate(pyr.device_pyramid(lib, DeviceImage.from_host(img), 3, 3)):
        np.testing.assert_array_equal(l.download().view(with_border=True), g[f"l{i}"])
    im = gc.fast_case(); g = load("fast9")
    dim = DeviceImage.from_host(im)
    for mode in (0, 1, 2):
        rc, sc = gpu_detect(lib, dim, 20, mode=mode, compat=0)
        np.testing.assert_array_equal(rc, g[f"rc{mode}"]); np.testing.assert_array_equal(sc, g[f"sc{mode}"])
    i1, i2, kps = gc.pyrlk_case(); g = load("pyrlk")
    dp1 = pyr.device_pyramid(lib, DeviceImage.from_host(i1), 3, 5); dp2 = pyr.device_pyramid(lib, DeviceImage.from_host(i2), 3, 5)
    dg = pyr.device_grad_pyramid(lib, dp1[0], 3, 5, vi.F32)
    dk = torch.from_numpy(kps.view(np.uint8).reshape(-1).copy()).cuda()
    capi.check(lib.vpp_pyrlk_match(vi.desc_array(dp1), vi.desc_array(dg), vi.desc_array(dp2), 3, V(dk.data_ptr()), len(kps), 7, ctypes.c_float(1e-4),
                                   ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, st))
    got = dk.cpu().numpy().view(pyr.KP_DTYPE); want = g["kps"].view(pyr.KP_DTYPE)
    np.testing.assert_array_equal(got["age"], want["age"])
    for f in ("pos_r", "pos_c", "vel_r", "vel_c"):
        np.testing.assert_allclose(got[f], want[f], rtol=1e-4, atol=1e-4 if f.startswith("pos") else 0.0)  # north-star tolerance: relative on the displacements
        assert (got[f].view(np.uint32) == want[f].view(np.uint32)).mean() > 0.999, f                             # observed: bit-identical
    j1, j2, pts = gc.lk_golden_case(); g = load("lucas_kanade")
    dp1 = pyr.device_pyramid(lib, DeviceImage.from_host(j1), 2, 2); dp2 = pyr.device_pyramid(lib, DeviceImage.from_host(j2), 2, 2)
    dg = pyr.device_grad_pyramid(lib, dp1[0], 2, 2, vi.I32)
    dpts = torch.from_numpy(pts).cuda(); flow = torch.zeros((len(pts), 2), device="cuda"); dist = torch.zeros(len(pts), device="cuda")
    capi.check(lib.vpp_lucas_kanade(vi.desc_array(dp1), vi.desc_array(dg), vi.desc_array(dp2), 2, V(dpts.data_ptr()), None, len(pts), 5, 0, 50, 0,
                                    V(flow.data_ptr()), V(dist.data_ptr()), st))
    np.testing.assert_allclose(flow.cpu().numpy(), g["flow"], rtol=1e-4, atol=1e-5)
    s1, s2, sk, par = gc.sdof_case(); g = load("sdof")
    dk = torch.from_numpy(sk).cuda(); n = len(sk)
    gp = torch.zeros((n, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(n, dtype=torch.int32, device="cuda"); gv = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ds1, ds2 = DeviceImage.from_host(s1), DeviceImage.from_host(s2)
    capi.check(lib.vpp_semi_dense_optical_flow(P(ds1.desc), P(ds2.desc), V(dk.data_ptr()), n, *par,
                                               V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
    np.testing.assert_array_equal(gp.cpu().numpy(), g["pos"]); np.testing.assert_array_equal(gd.cpu().numpy(), g["dist"]); np.testing.assert_array_equal(gv.cpu().numpy(), g["valid"])
    fim, ths = gc.fast_dense_case(); g = load("fast_dense")
    dfim = DeviceImage.from_host(fim)
    for th in ths:
        for dt in (vi.U8, vi.I32):
            do = DeviceImage(fim.nrows, fim.ncols, dt, 1)
            capi.check(lib.vpp_fast9_dense(P(do.desc), P(dfim.desc), th, st))
            np.testing.assert_array_equal(np.packbits(do.download().view()[..., 0].astype(bool), axis=1), g["th_%d" % th])
    rgb, g1, rgba, g2 = gc.ingest_case(); g = load("ingest")
    drgb, dg1, drgba, dg2 = DeviceImage.from_host(rgb), DeviceImage.from_host(g1), DeviceImage.from_host(rgba), DeviceImage.from_host(g2)
    capi.check(lib.vpp_rgb_to_graylevel(P(dg1.desc), P(drgb.desc), 1, st)); capi.check(lib.vpp_rgb_to_graylevel(P(dg2.desc), P(drgba.desc), 0, st))
    np.testing.assert_array_equal(dg1.download().view(with_border=True), g["gray_mirror"])
    np.testing.assert_array_equal(dg2.download().view(with_border=True), g["gray_rgba"])


# ---------------- the tracker: video_extruder_update over a 7-frame sequence, fixture from the reference's own headers ----------------
def test_video_extruder_fixture_belongs_to_the_seeded_sequence():
    frames, par = gc.video_extruder_case(); g = load("video_extruder")
    assert gc.crc(*[f.raw for f in frames]) == g["in_crc"]
    assert int(g["frame_id"]) == len(frames) - 2 and len(g["state"]) == len(g["traj_len"]) > 50
    assert (g["state"][:, 4] > 0).sum() > 30 and (g["state"][:, 2:4] != 0).any()   # alive keypoints, moving ones


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["two_frame", "push_frame", "push_host_rgb"])
def test_gpu_video_extruder_matches_reference_golden(lib, entry):
    """vpp_video_extruder_step on the bordered gray frames, vpp_video_extruder_push_frame on the bare gray frames in HBM and vpp_video_extruder_push_host_frame
    on colour frames in host memory whose integer mean is the gray sequence: every position, velocity, age and trajectory length of the reference."""
    from vpp_amd import capi

    class VeParams(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("detector_th", "keypoint_spacing", "detector_period", "max_trajectory_length", "nscales", "winsize", "propagation")]
    frames, par = gc.video_extruder_case(); g = load("video_extruder")
    nr, nc = frames[0].nrows, frames[0].ncols
    p = VeParams(*par)
    lib.vpp_video_extruder_create.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    ve = V(); capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), nr, nc, 15))
    keep = []
    try:
        for t, f in enumerate(frames):
            if entry == "two_frame":
                keep.append(DeviceImage.from_host(f))
                if t:
                    capi.check(lib.vpp_video_extruder_step(ve, P(keep[t - 1].desc), P(keep[t].desc), ctypes.byref(p), capi.stream_ptr()))
            elif entry == "push_frame":
                h = HostImage(nr, nc, vi.U8, 1, 0); h.view()[...] = f.view()
                keep.append(DeviceImage.from_host(h))
                capi.check(lib.vpp_video_extruder_push_frame(ve, P(keep[-1].desc), ctypes.byref(p), capi.stream_ptr()))
            else:
                gray = f.view()[..., 0].astype(np.int32); d = np.minimum(np.minimum(gray, 255 - gray), 13)
                buf = np.zeros((nr, nc * 3 + 5), np.uint8)   # rows 5 bytes apart from tight
                buf[:, :nc * 3] = np.stack([gray + d, gray, gray - d], -1).astype(np.uint8).reshape(nr, nc * 3)
                keep.append(buf)
                desc = vi.ImageDesc(buf.ctypes.data, nr, nc, buf.shape[1], 0, vi.U8, 3)
                capi.check(lib.vpp_video_extruder_push_host_frame(ve, ctypes.byref(desc), ctypes.byref(p), capi.stream_ptr()))
        n, fid = ctypes.c_int(), ctypes.c_int()
        capi.check(lib.vpp_video_extruder_count(ve, ctypes.byref(n), ctypes.byref(fid)))
        assert (n.value, fid.value) == (len(g["state"]), int(g["frame_id"]))
        pos = np.zeros((n.value, 2), np.int32); vel = np.zeros((n.value, 2), np.int32); age = np.zeros(n.value, np.int32); ln = np.zeros(n.value, np.int32)
        capi.check(lib.vpp_video_extruder_keypoints(ve, pos.ctypes.data_as(V), vel.ctypes.data_as(V), age.ctypes.data_as(V), n.value, capi.stream_ptr()))
        capi.check(lib.vpp_video_extruder_trajectories(ve, ln.ctypes.data_as(V), None, None, None, None, n.value, capi.stream_ptr()))
        np.testing.assert_array_equal(np.concatenate([pos, vel, age[:, None]], 1), g["state"])
        np.testing.assert_array_equal(ln, g["traj_len"])
    finally:
        lib.vpp_video_extruder_destroy(ve)
