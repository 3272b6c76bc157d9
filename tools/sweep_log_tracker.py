"""The fused sweeps' event log (tools/sweep_log.py) of one STEADY update of the 4K tracker: texture + rectangles translated by (1, 2) px per frame, the 8th frame's update."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, texture, rects_image
from vpp_amd import capi
V = ctypes.c_void_p
class VeParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("detector_th", "keypoint_spacing", "detector_period", "max_trajectory_length", "nscales", "winsize", "propagation")]
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
NR, NC, NF = 2160, 3840, 9
base = np.clip(0.6 * texture(NR + 2 * NF, NC + 4 * NF, seed=5) + 0.4 * rects_image(NR + 2 * NF, NC + 4 * NF, seed=7).astype(np.float32), 0, 255).astype(np.uint8)
frames = []
for t in range(NF):
    f = u8_image(np.ascontiguousarray(base[NF - t:NF - t + NR, 2 * (NF - t):2 * (NF - t) + NC]), border=3)
    f.view(with_border=True)[..., 0] = np.pad(f.view()[..., 0], 3, mode="symmetric")
    frames.append(DeviceImage.from_host(f))
ve = V(); capi.check(lib.vpp_video_extruder_create(ctypes.byref(ve), NR, NC, 15))
p = VeParams(10, 10, 5, 15, 3, 9, 2)
log = (ctypes.c_ulonglong * 4096)(); n = ctypes.c_uint(0)
for t in range(1, NF):
    if t == NF - 1:
        torch.cuda.synchronize(); lib.vpp_set_tuning(b"sdof.stats", 1); lib.vpp_debug_sdof_sweep_log(log, ctypes.byref(n), 1)
    capi.check(lib.vpp_video_extruder_step(ve, P(frames[t - 1].desc), P(frames[t].desc), ctypes.byref(p), st))
capi.check(lib.vpp_sync(st))
lib.vpp_debug_sdof_sweep_log(log, ctypes.byref(n), 1); lib.vpp_set_tuning(b"sdof.stats", -1)
cnt = ctypes.c_int(); fid = ctypes.c_int(); lib.vpp_video_extruder_count(ve, ctypes.byref(cnt), ctypes.byref(fid)); print("entries", cnt.value, "frame", fid.value)
ent = sorted(((e & 0xFFFFFFFF), e >> 56, (e >> 32) & 0xFFFFFF) for e in list(log)[:min(n.value, 4096)])
t0 = ent[0][0] if ent else 0
cls, done = [], []
for t, rnd, c in ent:
    if rnd == 253: cls.append((t, c)); continue
    if rnd == 252: done.append((t, c)); continue
    if rnd == 251: continue
    if rnd in (255, 1) and (cls or done):
        if cls: print(f"            {len(cls)} workgroups with candidates ({sum(x for _, x in cls)} in all, max {max(x for _, x in cls)}); classified between {(cls[0][0] - t0) * 0.01:.2f} and {(cls[-1][0] - t0) * 0.01:.2f} us")
        if done: print(f"            round 0 done on them between {(done[0][0] - t0) * 0.01:.2f} and {(done[-1][0] - t0) * 0.01:.2f} us; the last three: " + ", ".join(f"{(t_ - t0) * 0.01:.2f} us ({x} cand.)" for t_, x in done[-3:]))
        cls, done = [], []
    what = {255: f"launch start, {c} workgroups", 254: f"  {c} workgroups run the rounds", 250: "  skipped"}.get(rnd, f"  round {rnd}: {c} jobs")
    print(f"{(t - t0) * 0.01:8.2f} us  {what}")
