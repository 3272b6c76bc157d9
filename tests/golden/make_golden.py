"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libvpp_ref.so = matt-42/vpp's own headers).
Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
The inputs are regenerated from seeds by tests/golden_cases.py; each fixture stores a CRC of its inputs + the reference outputs."""
import ctypes
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_cases as gc  # noqa: E402
from oracle import binding  # noqa: E402
from util import P, HostImage  # noqa: E402
from vpp_amd import image as vi  # noqa: E402

ref = binding.load_ref()
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
V = ctypes.c_void_p


def crc(*arrs):
    c = 0
    for a in arrs:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


out = {}
# box5x5 vuchar3 + int add
src, dst = gc.box_case()
assert ref.ref_box_filter5x5(P(dst.desc), P(src.desc)) == 0
out["box"] = dict(in_crc=crc(src.raw), out=dst.view().copy())
b, c, a = gc.add_case()
assert ref.ref_pixelwise_add(P(a.desc), P(b.desc), P(c.desc)) == 0
out["add"] = dict(in_crc=crc(b.raw, c.raw), out=a.view().copy())
# pyramids (u8 and vfloat2 gradient)
img, levels = gc.pyramid_case()
assert ref.ref_pyramid(P(img.desc), len(levels), 3, vi.desc_array(levels)) == 0
out["pyramid"] = dict(in_crc=crc(img.raw), **{f"l{i}": l.view(with_border=True).copy() for i, l in enumerate(levels)})
# FAST9 reference mode, three modes
im = gc.fast_case()
d = dict(in_crc=crc(im.raw))
for mode in (0, 1, 2):
    rc = np.zeros((50000, 2), np.int32); sc = np.zeros(50000, np.int32); n = ctypes.c_int(0)
    assert ref.ref_fast9(P(im.desc), 20, None, mode, 10, rc.ctypes.data_as(V), sc.ctypes.data_as(V), 50000, P(n)) == 0
    d[f"rc{mode}"] = rc[:n.value].copy(); d[f"sc{mode}"] = sc[:n.value].copy()
out["fast9"] = d
# pyrlk_match 7x7 and lucas_kanade golden
i1, i2, kps = gc.pyrlk_case()
k = kps.copy()
assert ref.ref_pyrlk_match(P(i1.desc), P(i2.desc), 3, 5, k.ctypes.data_as(V), len(k), 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0) == 0
out["pyrlk"] = dict(in_crc=crc(i1.raw, i2.raw, kps.view(np.uint8)), kps=k.view(np.uint8).copy())
j1, j2, pts = gc.lk_golden_case()
flow = np.zeros((len(pts), 2), np.float32); dist = np.zeros(len(pts), np.float32)
assert ref.ref_lucas_kanade(P(j1.desc), P(j2.desc), pts.ctypes.data_as(V), len(pts), 5, 2, 50, ctypes.c_double(0.001), ctypes.c_double(0.01), flow.ctypes.data_as(V), dist.ctypes.data_as(V)) == 0
out["lucas_kanade"] = dict(in_crc=crc(j1.raw, j2.raw, pts), flow=flow, dist=dist)
# semi-dense flow
s1, s2, sk, par = gc.sdof_case()
p = np.zeros((len(sk), 2), np.int32); dd = np.zeros(len(sk), np.int32); vv = np.zeros(len(sk), np.uint8)
assert ref.ref_semi_dense_optical_flow(P(s1.desc), P(s2.desc), sk.ctypes.data_as(V), len(sk), *par, p.ctypes.data_as(V), dd.ctypes.data_as(V), vv.ctypes.data_as(V)) == 0
out["sdof"] = dict(in_crc=crc(s1.raw, s2.raw, sk), pos=p, dist=dd, valid=vv)

# frame ingest (clone + mirror + rgb_to_graylevel) and the plain 4-channel rgb_to_graylevel
rgb, g1, rgba, g2 = gc.ingest_case()
assert ref.ref_rgb_to_graylevel(P(g1.desc), P(rgb.desc), 1) == 0
assert ref.ref_rgb_to_graylevel(P(g2.desc), P(rgba.desc), 0) == 0
out["ingest"] = dict(in_crc=crc(rgb.raw, rgba.raw), gray_mirror=g1.view(with_border=True).copy(), gray_rgba=g2.view(with_border=True).copy())

# dense FAST_internals::fast_detector9(A, B, th)
fim, ths = gc.fast_dense_case()
dense = {}
for th in ths:
    o8, o32 = HostImage(fim.nrows, fim.ncols, vi.U8, 1), HostImage(fim.nrows, fim.ncols, vi.I32, 1)
    assert ref.ref_fast9_dense(P(o8.desc), P(fim.desc), th) == 0 and ref.ref_fast9_dense(P(o32.desc), P(fim.desc), th) == 0
    assert (o8.view()[..., 0] == o32.view()[..., 0]).all()
    dense["th_%d" % th] = np.packbits(o8.view()[..., 0].astype(bool), axis=1)
out["fast_dense"] = dict(in_crc=crc(fim.raw), **dense)

# video_extruder_update over a short sequence: the reference's own tracker (oracle/_ref/libvpp_ref_ve.so)
ve_lib = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libvpp_ref_ve.so")
assert os.path.exists(ve_lib), "build oracle/_ref first (make -C oracle ref)"
ve = ctypes.CDLL(ve_lib)
frames, par = gc.video_extruder_case()
cap = 4096
state = np.zeros((cap, 5), np.int32); tlen = np.zeros(cap, np.int32); cnt, fid = ctypes.c_int(0), ctypes.c_int(0)
assert ve.ref_video_extruder_run(vi.desc_array(frames), len(frames), *par, state.ctypes.data_as(V), tlen.ctypes.data_as(V), cap, ctypes.byref(cnt), ctypes.byref(fid)) == 0
out["video_extruder"] = dict(in_crc=crc(*[f.raw for f in frames]), state=state[:cnt.value].copy(), traj_len=tlen[:cnt.value].copy(), frame_id=np.int32(fid.value))

only = set(sys.argv[1:])   # python make_golden.py [names...]: rewrite only these fixtures
for name, d in out.items():
    if only and name not in only:
        continue
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, {k: getattr(v, "shape", v) for k, v in d.items()})
