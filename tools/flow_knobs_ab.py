"""A/B of SETS of tuning knobs of vpp_semi_dense_optical_flow on the 4K bench scene in one process, interleaved:
    python tools/flow_knobs_ab.py "sdof.local_max=100000" "sdof.local_max=32,sdof.helpers=1,sdof.max_stay=160" ..."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vpp_amd.synth import P, u8_image, DeviceImage, flow_scene
from vpp_amd import capi
V = ctypes.c_void_p
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]]
lib = capi.lib(); capi.check(lib.vpp_init(0)); st = capi.stream_ptr()
scenes = (((2160, 3840), 10), ((2160, 3840), 5), ((1080, 1920), 10)) if os.environ.get("ALL_SCENES") else (((2160, 3840), 10),)
for shape, spacing in scenes:
    s1, s2, sk = flow_scene(*shape, spacing=spacing)
    e1, e2 = DeviceImage.from_host(u8_image(s1, border=3)), DeviceImage.from_host(u8_image(s2, border=3))
    m = len(sk); dk = torch.from_numpy(sk).cuda()
    gp = torch.zeros((m, 2), dtype=torch.int32, device="cuda"); gd = torch.zeros(m, dtype=torch.int32, device="cuda"); gv = torch.zeros(m, dtype=torch.uint8, device="cuda")
    sums, best = set(), {}
    for rep in range(4):
        for si, ks in enumerate(sets):
            for k, v in ks.items(): lib.vpp_set_tuning(k.encode(), v)
            ts = []
            for it in range(14):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                capi.check(lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st))
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            rc = lib.vpp_sync(st)
            if rc: print("  vpp_sync:", rc, lib.vpp_last_error().decode(), flush=True)
            sums.add((int(gp.sum()), int(gd.sum()), int(gv.sum())))
            for k in ks: lib.vpp_set_tuning(k.encode(), -1)
            ts = sorted(ts[2:]); best.setdefault(si, []).append((ts[0], ts[len(ts) // 2]))
    for si, ks in enumerate(sets):
        print(f"{shape} spacing {spacing}: {ks}: min {min(b[0] for b in best[si]) * 1e3:.4f} ms, medians " + " ".join(f"{b[1] * 1e3:.4f}" for b in best[si]), flush=True)
    print("  identical:", len(sums) == 1)
