"""One-off randomized parity sweep (GPU vs the CPU oracle) over shapes / parameters the fixed tests do not enumerate:
FAST-9 (all modes, masks, thresholds), semi-dense flow (window sizes, scales, sweeps, patch sizes), box filters, rgb->gray, u8 pyramids (gray and fused rgb ingest), tracker sequences (two-frame update vs one frame per call).
usage: python tools/stress_parity.py [n_cases] [seed]      (needs a GPU; exits non-zero on the first mismatch)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, rand_image, HostImage, DeviceImage, u8_image, rects_image
from test_gpu_sdof import flow_scene, run_both, SWEEP_IMPLS, set_sweep_impl
from test_oracle_algos import run_detect
from test_gpu_algos import gpu_detect
from vpp_amd import capi, image as vi
from oracle import binding
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = capi.lib(); capi.check(lib.vpp_init(0)); orc = binding.load()
V = ctypes.c_void_p
bad = 0
for case in range(N):
    # FAST-9
    nr, nc = int(rng.integers(8, 300)), int(rng.integers(8, 400))
    img = rects_image(nr, nc, seed=int(rng.integers(1 << 30)))
    im = u8_image(img, border=3)
    im.view(with_border=True)[..., 0] = np.pad(im.view()[..., 0], 3, mode="symmetric")
    th, mode, bs, compat = int(rng.integers(1, 60)), int(rng.integers(0, 3)), int(rng.integers(2, 24)), int(rng.integers(0, 2))
    mask = None
    if rng.integers(0, 2):
        mask = u8_image(rng.choice(np.array([0, 1, 16, 17, 255], np.uint8), size=(nr, nc)), border=0)
    want = run_detect(orc, im, th, mode=mode, bs=bs, compat=compat, mask=mask)
    dim = DeviceImage.from_host(im); dmask = DeviceImage.from_host(mask) if mask is not None else None
    got = gpu_detect(lib, dim, th, mode=mode, bs=bs, compat=compat, mask=dmask)
    ok = all(np.array_equal(g, w) for g, w in zip(got, want))
    print(f"fast9 {nr}x{nc} th={th} mode={mode} bs={bs} compat={compat} mask={mask is not None}: n={len(want[0])} {'ok' if ok else 'MISMATCH'}"); bad += not ok
    # semi-dense flow
    ws = int(rng.choice([5, 7, 9, 11])); nscales = int(rng.integers(1, 4)); min_scale = int(rng.integers(0, nscales)); prop = int(rng.integers(0, 4)); patch = int(rng.choice([3, 5, 7]))
    shape = (int(rng.integers(60, 200)), int(rng.integers(60, 260)))
    f1, f2, kps = flow_scene(*shape, seed=int(rng.integers(1 << 30)), spacing=int(rng.integers(3, 9)))
    for impl in SWEEP_IMPLS:   # one launch per sweep, the same with every workgroup staying for the rounds, two launches per sweep, the one-workgroup wavefront
        set_sweep_impl(lib, impl)
        got, want = run_both(lib, orc, f1, f2, kps, ws, nscales, min_scale, prop, patch)
        ok = all(np.array_equal(g, w) for g, w in zip(got, want))
        print(f"sdof {shape} ws={ws} nscales={nscales} min={min_scale} prop={prop} patch={patch} impl={impl}: valid={int(want[2].sum())} {'ok' if ok else 'MISMATCH'}"); bad += not ok
    set_sweep_impl(lib, None)
    # box 5x5 on u8 x ch and int32, rgb->gray ingest
    ch = int(rng.integers(1, 5)); border = int(rng.integers(2, 5)); nr, nc = int(rng.integers(1, 200)), int(rng.integers(1, 1500))
    for dtype, c in ((vi.U8, ch), (vi.I32, 1)):
        src = rand_image(nr, nc, dtype, c, border=border, seed=int(rng.integers(1 << 30)), lo=0 if dtype == vi.I32 else None, hi=999 if dtype == vi.I32 else None, align=16, fill_border=True)
        want = src.like(border=0); orc.orc_box_filter(P(want.desc), P(src.desc), 5, 5)
        ds, dd = DeviceImage.from_host(src), DeviceImage.from_host(src.like(border=0))
        capi.check(lib.vpp_box_filter(P(dd.desc), P(ds.desc), 5, 5, capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
        ok = np.array_equal(dd.download().view(), want.view())
        print(f"box5x5 dtype={dtype} x{c} {nr}x{nc} border={border}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    gb = int(rng.integers(0, 6)); rgb = rand_image(nr, nc, vi.U8, int(rng.choice([3, 4])), border=0, seed=int(rng.integers(1 << 30)))
    if gb <= nr and gb <= nc:
        want = HostImage(nr, nc, vi.U8, 1, gb); orc.orc_rgb_to_graylevel(P(want.desc), P(rgb.desc), 1)
        dr, dg = DeviceImage.from_host(rgb), DeviceImage(nr, nc, vi.U8, 1, gb)
        capi.check(lib.vpp_rgb_to_graylevel(P(dg.desc), P(dr.desc), 1, capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
        ok = np.array_equal(dg.download().view(with_border=True), want.view(with_border=True))
        print(f"ingest {nr}x{nc} border={gb}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    # box windows other than 5x5 (streamed up to 7x7 / 7x5, generic beyond), all element types
    R, C = int(rng.choice([1, 3, 5, 7, 9])), int(rng.choice([1, 3, 5, 7, 9]))
    dtype = int(rng.choice([vi.U8, vi.I8, vi.U16, vi.I16, vi.I32, vi.U32, vi.F32])); c = int(rng.integers(1, 5)) if dtype == vi.U8 else int(rng.integers(1, 3))
    reach = max(R, C) // 2; border = reach + int(rng.integers(0, 3)); nr, nc = int(rng.integers(1, 120)), int(rng.integers(1, 900))
    wide = dtype in (vi.I32, vi.U32)
    src = rand_image(nr, nc, dtype, c, border=border, seed=int(rng.integers(1 << 30)), lo=0 if wide else None, hi=999 if wide else None, align=int(rng.choice([16, 32])), fill_border=True)
    want = src.like(border=0)
    if orc.orc_box_filter(P(want.desc), P(src.desc), R, C) == 0:
        ds, dd = DeviceImage.from_host(src), DeviceImage.from_host(src.like(border=0))
        capi.check(lib.vpp_box_filter(P(dd.desc), P(ds.desc), R, C, capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
        ok = np.array_equal(dd.download().raw.view(np.uint8), want.raw.view(np.uint8))
        print(f"box {R}x{C} dtype={dtype} x{c} {nr}x{nc} border={border}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    # dense FAST flags and the blockwise maxima filter
    nr, nc = int(rng.integers(1, 200)), int(rng.integers(1, 300)); th = int(rng.integers(-5, 80)); fb = int(rng.integers(3, 6))
    fim = u8_image(rects_image(nr, nc, seed=int(rng.integers(1 << 30))), border=fb)
    fim.view(with_border=True)[..., 0] = np.pad(fim.view()[..., 0], fb, mode="symmetric")
    odt = int(rng.choice([vi.U8, vi.I32]))
    want = HostImage(nr, nc, odt, 1); orc.orc_fast9_dense(P(want.desc), P(fim.desc), th)
    dfi, dfo = DeviceImage.from_host(fim), DeviceImage(nr, nc, odt, 1)
    capi.check(lib.vpp_fast9_dense(P(dfo.desc), P(dfi.desc), th, capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
    ok = np.array_equal(dfo.download().view(), want.view())
    print(f"fast9_dense {nr}x{nc} th={th} out={odt}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    bdt = int(rng.choice([vi.U8, vi.I16, vi.I32, vi.F32])); bs = int(rng.integers(1, 40))
    bim = rand_image(nr, nc, bdt, 1, border=1, seed=int(rng.integers(1 << 30)), lo=-3 if bdt != vi.U8 else 0, hi=9)
    dbi = DeviceImage.from_host(bim); orc.orc_blockwise_maxima_filter(P(bim.desc), bs)
    capi.check(lib.vpp_blockwise_maxima_filter(P(dbi.desc), bs, capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
    ok = np.array_equal(dbi.download().raw.view(np.uint8), bim.raw.view(np.uint8))
    print(f"blockwise_maxima {nr}x{nc} dtype={bdt} bs={bs}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    # video_extruder's merge against the serial champion loop
    from test_gpu_video_steps import _merge_model
    n, mr, mc, sp = int(rng.integers(1, 6000)), int(rng.integers(8, 300)), int(rng.integers(8, 300)), int(rng.integers(1, 40))
    prev = np.stack([rng.integers(0, mr, n), rng.integers(0, mc, n)], 1).astype(np.int32)
    moved = (prev + rng.integers(-9, 10, size=prev.shape)).astype(np.int32); matched = (rng.random(n) < 0.8).astype(np.uint8); moved[matched == 0] = prev[matched == 0]
    age_prev = rng.integers(0, int(rng.integers(1, 12)) + 1, n).astype(np.int32)
    inside = (moved[:, 0] >= 0) & (moved[:, 0] < mr) & (moved[:, 1] >= 0) & (moved[:, 1] < mc)
    pos = np.where(((matched == 1) & inside)[:, None], moved, prev); age = np.where(matched == 1, np.where(inside, age_prev + 1, 0), age_prev).astype(np.int32)
    want = _merge_model(pos, age, sp, mr, mc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dm, dp, dmt, da = t(moved), t(prev), t(matched), t(age_prev); out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    capi.check(lib.vpp_keypoint_merge(V(dm.data_ptr()), V(dp.data_ptr()), V(dmt.data_ptr()), V(da.data_ptr()), n, mr, mc, sp, V(out.data_ptr()), capi.stream_ptr()))
    ok = np.array_equal(out.cpu().numpy(), want)
    print(f"merge n={n} {mr}x{mc} spacing={sp}: removed={int(want.sum())} {'ok' if ok else 'MISMATCH'}"); bad += not ok
    # u8 pyramids of 3 levels (the packed kernel away from the edges, the tile kernel on them) from a gray frame and from an rgb frame (fused ingest)
    import pyr as tpyr
    nr, nc, pb = int(rng.integers(100, 700)), int(rng.integers(100, 1100)), int(rng.integers(2, 21))   # the low-pass reads 2 px beyond a level
    g = rand_image(nr, nc, vi.U8, 1, border=0, seed=int(rng.integers(1 << 30)), align=int(rng.choice([1, 16, 32])))
    want = tpyr.host_pyramid(orc, g, 3, pb)
    got = tpyr.device_pyramid(lib, DeviceImage.from_host(g), 3, pb); capi.check(lib.vpp_sync(capi.stream_ptr()))
    ok = all(np.array_equal(d.download().raw, h.raw) for d, h in zip(got, want))
    print(f"pyramid {nr}x{nc} border={pb}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    chn = int(rng.choice([3, 4]))
    rgbp = rand_image(nr, nc, vi.U8, chn, border=0, seed=int(rng.integers(1 << 30)), align=int(rng.choice([1, 16, 32])))
    gray = HostImage(nr, nc, vi.U8, 1, pb, 32); orc.orc_rgb_to_graylevel(P(gray.desc), P(rgbp.desc), 1)
    want = tpyr.host_pyramid(orc, gray, 3, pb)
    lv = [DeviceImage(a, b, vi.U8, 1, pb) for a, b in tpyr.level_dims(nr, nc, 3)]
    capi.check(lib.vpp_rgb_pyramid_build(vi.desc_array(lv), 3, P(DeviceImage.from_host(rgbp).desc), capi.stream_ptr())); capi.check(lib.vpp_sync(capi.stream_ptr()))
    ok = all(np.array_equal(d.download().raw, h.raw) for d, h in zip(lv, want))
    print(f"rgb pyramid x{chn} {nr}x{nc} border={pb}: {'ok' if ok else 'MISMATCH'}"); bad += not ok
    # the tracker: the two-frame update on mirror-bordered gray frames against one frame per call (frames in HBM / pitched host frames, gray or colour),
    # random geometry and options; every position, velocity, age and trajectory length after 6 frames
    from vpp_amd.synth import texture, translate
    class VeParams(ctypes.Structure):
        _fields_ = [(n_, ctypes.c_int32) for n_ in ("detector_th", "keypoint_spacing", "detector_period", "max_trajectory_length", "nscales", "winsize", "propagation")]
    nr, nc = int(rng.integers(40, 260)), int(rng.integers(48, 360))
    par = VeParams(int(rng.integers(5, 25)), int(rng.integers(4, 14)), int(rng.integers(1, 5)), int(rng.integers(2, 12)), int(rng.integers(1, 4)), int(rng.choice([5, 7, 9, 11])), int(rng.integers(0, 3)))
    while min(nr, nc) >> (par.nscales - 1) < 5 * 2:   # every flow-map level needs one patch at least
        par.nscales -= 1
    tex = texture(nr + 40, nc + 40, seed=int(rng.integers(1 << 30)), sigma=1.5); rc = rects_image(nr + 40, nc + 40, seed=int(rng.integers(1 << 30))).astype(np.float64)
    dr, dc = float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2))
    fr = [np.clip(np.rint((0.6 * translate(tex, dr * t, dc * t) + 0.4 * translate(rc, dr * t, dc * t))[20:20 + nr, 20:20 + nc]), 0, 255).astype(np.uint8) for t in range(6)]
    lib.vpp_video_extruder_create.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    def ve_state(ve):
        n_, fid = ctypes.c_int(), ctypes.c_int()
        capi.check(lib.vpp_video_extruder_count(ve, ctypes.byref(n_), ctypes.byref(fid)))
        pos = np.zeros((n_.value, 2), np.int32); vel = np.zeros((n_.value, 2), np.int32); age = np.zeros(n_.value, np.int32); ln = np.zeros(n_.value, np.int32)
        capi.check(lib.vpp_video_extruder_keypoints(ve, pos.ctypes.data_as(V), vel.ctypes.data_as(V), age.ctypes.data_as(V), n_.value, capi.stream_ptr()))
        capi.check(lib.vpp_video_extruder_trajectories(ve, ln.ctypes.data_as(V), None, None, None, None, n_.value, capi.stream_ptr()))
        return fid.value, pos, vel, age, ln
    chn = int(rng.choice([1, 3, 4])); host = bool(rng.integers(0, 2)); pad = int(rng.integers(0, 20))
    states = []
    for mode in (0, 1):
        ve = V(); capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), nr, nc, 15))
        keep = []
        for t, f in enumerate(fr):
            if mode == 0:
                h = HostImage(nr, nc, vi.U8, 1, 3); h.view()[..., 0] = f
                d = DeviceImage.from_host(h); capi.check(lib.vpp_fill_border(P(d.desc), 0, None, capi.stream_ptr())); keep.append(d)
                if t: capi.check(lib.vpp_video_extruder_step(ve, P(keep[t - 1].desc), P(d.desc), ctypes.byref(par), capi.stream_ptr()))
                continue
            g = f.astype(np.int32); dd = np.minimum(np.minimum(g, 255 - g), 9)
            px = f[..., None] if chn == 1 else np.stack([g + dd, g, g - dd] + ([np.full_like(g, 9)] if chn == 4 else []), -1).astype(np.uint8)
            if host:
                buf = np.zeros((nr, nc * chn + pad), np.uint8); buf[:, :nc * chn] = px.reshape(nr, nc * chn); keep.append(buf)
                desc = vi.ImageDesc(buf.ctypes.data, nr, nc, buf.shape[1], 0, vi.U8, chn)
                capi.check(lib.vpp_video_extruder_push_host_frame(ve, ctypes.byref(desc), ctypes.byref(par), capi.stream_ptr()))
            else:
                h = HostImage(nr, nc, vi.U8, chn, 0); h.view()[...] = px; d = DeviceImage.from_host(h); keep.append(d)
                capi.check(lib.vpp_video_extruder_push_frame(ve, P(d.desc), ctypes.byref(par), capi.stream_ptr()))
        states.append(ve_state(ve)); lib.vpp_video_extruder_destroy(ve)
    ok = states[0][0] == states[1][0] and all(np.array_equal(a_, b_) for a_, b_ in zip(states[0][1:], states[1][1:]))
    print(f"tracker {nr}x{nc} x{chn} {'host' if host else 'hbm'} th={par.detector_th} spacing={par.keypoint_spacing} period={par.detector_period} scales={par.nscales} ws={par.winsize} "
          f"prop={par.propagation}: {len(states[0][1])} entries, {int((states[0][3] > 0).sum())} alive {'ok' if ok else 'MISMATCH'}"); bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
