"""pyrLK leg of bench.py (BASELINE.json configs[3]): 1920x1080, 3-level pyramids, 10 000 keypoints, 7x7 window,
min_ev 1e-4, max_err 500, 30 iterations, delta 0.01.  Keypoints are sharded contiguously across ranks
([g*N/G, (g+1)*N/G), pyrlk_match.hh:24 iterates independent keypoints); every rank holds both pyramids; one exchange:
an RCCL all-gather of the 20-byte keypoint records.  Also the FAST-9 4K leg (configs[2], replicas)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def profile_round():
    """The round whose committed rocprofv3 summaries (profiles/<round>_*) describe the kernels of THIS tree: profiles/CURRENT, written by tools/publish_profiles.sh."""
    try:
        return open(os.path.join(ROOT, "profiles", "CURRENT")).read().strip()
    except OSError:
        return None


def issue_roofline(prefix):
    """Per-leg roofline of a kernel that is bound by the SIMDs' instruction issue, not by HBM: the fraction of the chip's
    VALU issue cycles the kernel uses while it runs, from the committed rocprofv3 PMC passes of the CURRENT round only (profiles/<round>_issue.json,
    written by tools/make_issue_json.py from the passes of tools/profile_round.sh; the same numbers are readable in profiles/<round>_pmc.md):
      frac = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs  /  (SQ_BUSY_CYCLES / 32 counter instances)      [both per dispatch]
    A static quote from a committed profile, not measured in this run; keyed by kernel symbol prefix.  When the current round's file has no kernel of this
    name (the symbol changed, or no pass was taken yet) the answer is {"stale": true} — never an older round's entry."""
    import json
    rnd = profile_round()
    path = os.path.join(ROOT, "profiles", f"{rnd}_issue.json")
    if rnd and os.path.exists(path):
        for k, v in json.load(open(path)).items():
            if k.startswith(prefix):
                return {"bound": "valu_issue", "kernel": k, "frac": v["valu_issue_frac"], "lds_issue_frac": v.get("lds_issue_frac"),
                        "valu_wave_instructions": v.get("insts_valu"), "kernel_cycles": v.get("kernel_cycles"), "unit": "fraction of the SIMD issue cycles",
                        "source": os.path.basename(path) + " (committed rocprofv3 --pmc SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_LDS / SQ_INSTS_VALU passes, not measured in this run)"}
    return {"bound": "valu_issue", "stale": True, "frac": None, "wanted": prefix, "note": f"profiles/{rnd}_issue.json has no kernel with this symbol prefix"}


def run(lib, dev, rank, world, timed, barrier, steps=50, warmup=5):
    import torch
    import torch.distributed as dist
    from vpp_amd import pyr
    from vpp_amd.synth import P, u8_image, DeviceImage, texture, translate, rects_image, fast9_bench_frame
    from vpp_amd import capi, image as vi, multi_gpu as mg

    V = ctypes.c_void_p
    st = capi.stream_ptr()
    trace_on = os.environ.get("VPP_BENCH_TRACE", "0") == "1" and rank == 0

    def stage(name):
        if trace_on:
            sys.stderr.write(f"[bench stage] pyrlk leg: {name}\n"); sys.stderr.flush()
    stage("pyrlk match")
    NR, NC, L, B, WS, NK = 1080, 1920, 3, 3, 7, 10000
    tex = texture(NR, NC, seed=5)
    f1 = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    f2 = np.clip(np.rint(translate(tex, 1.5, -2.25)), 0, 255).astype(np.uint8)
    d1, d2 = DeviceImage.from_host(u8_image(f1), dev), DeviceImage.from_host(u8_image(f2), dev)
    kps_h = pyr.make_keypoints(pyr.grid_keypoints(NR, NC, NK, margin=32))
    lo, hi = mg.shard_bounds(NK, rank, world)
    shard0 = torch.from_numpy(kps_h[lo:hi].view(np.uint8).reshape(-1).copy()).to(dev)
    shard = shard0.clone()
    n_local = hi - lo

    p1 = pyr.device_pyramid(lib, d1, L, B)
    g1 = pyr.device_grad_pyramid(lib, p1[0], L, B, vi.F32)
    p2 = pyr.device_pyramid(lib, d2, L, B)
    dp1, dg1, dp2 = vi.desc_array(p1), vi.desc_array(g1), vi.desc_array(p2)
    match = lib.vpp_pyrlk_match

    def step_match(i, stream):
        shard.copy_(shard0, non_blocking=True)  # restore the tracks: pyrlk_match moves them in place
        match(dp1, dg1, dp2, L, V(shard.data_ptr()), n_local, WS, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, stream)
        if world > 1:
            mg.all_gather_records(shard, NK, rank, world)

    wall, ev = timed(step_match, steps, warmup, graph=(world == 1))
    res = {"workload": "pyrlk_match 1920x1080, 3 levels, 10k keypoints, 7x7, min_ev 1e-4, max_err 500, 30 it, delta 0.01",
           "tracks_per_s": NK / (wall / steps), "ms_per_frame": wall / steps * 1e3, "keypoints_per_rank": n_local,
           "exchange": "rccl all_gather of 20-byte keypoint records" if world > 1 else "none (1 GPU)"}

    # the instance the library launches for this many keypoints per rank (pyrlk.hip: 16 lanes per keypoint from 8 000, 32 from 3 500, 64 below)
    lpk = 8 if n_local >= 80000 else (16 if n_local >= 8000 else (32 if n_local >= 3500 else 64))
    res["roofline"] = issue_roofline(f"pyrlk_match_group_kernel<7, {lpk}>")

    # keypoint-count sweep on one GPU (where tracks/s saturates; 1 250 = what one of 8 ranks sees of the 10 k keypoints of configs[3])
    if world == 1:
        sweep = {}
        for nk in (1250, 2500, 5000, 10000, 20000, 40000, 160000):
            kh = pyr.make_keypoints(pyr.grid_keypoints(NR, NC, nk, margin=32))
            k0 = torch.from_numpy(kh.view(np.uint8).reshape(-1).copy()).to(dev)
            k1 = k0.clone()
            nn = len(kh)

            def step_n(i, stream, k0=k0, k1=k1, nn=nn):
                k1.copy_(k0, non_blocking=True)
                match(dp1, dg1, dp2, L, V(k1.data_ptr()), nn, WS, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, stream)
            w, _ = timed(step_n, 20, 3, graph=True)
            sweep[str(nn)] = {"ms": round(w / 20 * 1e3, 4), "tracks_per_s": round(nn / (w / 20))}
        res["sweep"] = sweep
        # what the keypoint-sharded step can reach on G GPUs of one node, read off this GPU's sweep: G ranks x the rate at 10 000 / G keypoints (the all-gather of
        # 200 KB of records comes on top).  The match is latency bound below ~10 k keypoints per GPU — a keypoint runs 93 dependent Gauss-Newton iterations whatever
        # else the chip does — so STRONG scaling of configs[3]'s 10 k keypoints is bounded well below G x; weak scaling (10 k per rank) is what scales linearly.
        res["scaling_bound"] = {str(g): {"keypoints_per_rank": NK // g, "tracks_per_s_upper_bound": g * sweep[str(NK // g)]["tracks_per_s"]} for g in (1, 2, 4, 8) if str(NK // g) in sweep}
        res["scaling_bound_8gpu"] = res["scaling_bound"].get("8", {}).get("tracks_per_s_upper_bound")
        res["scaling_bound"]["note"] = ("strong scaling of 10 000 keypoints over G ranks: G x this GPU's rate at 10 000 / G keypoints, before the all-gather; "
                                        "weak scaling (10 000 keypoints per rank) keeps the 1-GPU rate per rank")

        # ---- frame-pair batches (vpp_pyrlk_match_batch): a rank's slice of configs[3] on 8 GPUs is 1 250 keypoints, whose match costs the latency of ONE keypoint's
        # chain (the sweep above: 1 250 keypoints take what 5 000 do) — strong scaling of one pair is bounded at ~1.5 x on 8 GPUs.  With F DISTINCT frame pairs per
        # launch (own frames, own pyramids, own keypoints) a rank leaves that floor: tracks/s per rank at F = 1 / 4 / 8 / 16 x 1 250 keypoints, and the keypoint-sharded
        # job's projection on 8 ranks (8 x this GPU's rate; the all-gather of F x 200 KB of records comes on top).
        try:
            NSL, FMAX = NK // 8, 16
            rec = NSL * pyr.KP_DTYPE.itemsize
            kall0 = torch.empty(FMAX * rec, dtype=torch.uint8, device=dev)   # all pairs' records in one block: ONE restoring copy per step
            pairs = []
            for f in range(FMAX):
                texf = texture(NR, NC, seed=60 + f)
                a = DeviceImage.from_host(u8_image(np.clip(np.rint(texf), 0, 255).astype(np.uint8)), dev)
                b = DeviceImage.from_host(u8_image(np.clip(np.rint(translate(texf, 1.5 - 0.1 * f, -2.25 + 0.2 * f)), 0, 255).astype(np.uint8)), dev)
                pa = pyr.device_pyramid(lib, a, L, B); ga = pyr.device_grad_pyramid(lib, pa[0], L, B, vi.F32); pb = pyr.device_pyramid(lib, b, L, B)
                kall0[f * rec:(f + 1) * rec].copy_(torch.from_numpy(np.ascontiguousarray(kps_h[f % 8::8][:NSL]).view(np.uint8).reshape(-1).copy()))
                pairs.append((pa, ga, pb))
            kall = kall0.clone()
            batch = lib.vpp_pyrlk_match_batch
            fb = {}
            for F in (1, 4, 8, 16):
                sel = pairs[:F]
                dP = vi.desc_array([l_ for q in sel for l_ in q[0]]); dG = vi.desc_array([l_ for q in sel for l_ in q[1]]); dN = vi.desc_array([l_ for q in sel for l_ in q[2]])
                kp_ptrs = (ctypes.c_void_p * F)(*[kall.data_ptr() + f * rec for f in range(F)]); counts = (ctypes.c_int * F)(*([NSL] * F))
                src, dst = kall0[:F * rec], kall[:F * rec]

                def step_b(i, stream, dP=dP, dG=dG, dN=dN, kp_ptrs=kp_ptrs, counts=counts, F=F, src=src, dst=dst):
                    dst.copy_(src, non_blocking=True)   # restore the tracks: the match moves them in place
                    capi.check(batch(dP, dG, dN, F, L, kp_ptrs, counts, WS, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, stream))
                w, _ = timed(step_b, 20, 3, graph=True)
                rate = F * NSL / (w / 20)
                fb[str(F)] = {"ms_per_launch": round(w / 20 * 1e3, 4), "tracks_per_s_per_rank": round(rate), "projected_8_ranks_tracks_per_s": round(8 * rate)}
            fb["how"] = (f"F distinct 1080p frame pairs x {NSL} keypoints (every 8th keypoint of configs[3]'s 10 k: one of 8 ranks' slice) in ONE launch of pyrlk_match_batch_kernel; "
                         "the restoring copy of the keypoint records is inside the timed region; projected = 8 x this GPU's rate, before the all-gather")
            res["frame_pair_batches"] = fb
        except Exception as e:  # noqa: BLE001
            res["frame_pair_batches"] = {"error": f"{type(e).__name__}: {e}"}

        # ---- the reference's OWN pyrLK benchmark configuration (benchmarks/pyrlk_opencv_comparison.cc:47,64-65): 11 x 11 window, 4 scales, min_ev 1e-4, max_err 500,
        # 30 iterations, delta 0.01 — on the same 1080p scene and 10 000 keypoints; pyramids with a border of 8 (the benchmark's border(3) is narrower than the window's
        # reach: the reference then reads outside its border, this engine clamps — a border the window fits in keeps every tap on the fast, unclamped path)
        try:
            L4, B4, WS11 = 4, 8, 11
            q1 = pyr.device_pyramid(lib, d1, L4, B4); h1 = pyr.device_grad_pyramid(lib, q1[0], L4, B4, vi.F32); q2 = pyr.device_pyramid(lib, d2, L4, B4)
            dq1, dh1, dq2 = vi.desc_array(q1), vi.desc_array(h1), vi.desc_array(q2)
            k0 = torch.from_numpy(kps_h.view(np.uint8).reshape(-1).copy()).to(dev); k1 = k0.clone()

            def step_11(i, stream):
                k1.copy_(k0, non_blocking=True)
                match(dq1, dh1, dq2, L4, V(k1.data_ptr()), NK, WS11, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, stream)
            w11, _ = timed(step_11, 20, 3, graph=True)
            a11 = {"workload": "pyrlk_match 1920x1080, 4 levels, 10k keypoints, 11x11 (benchmarks/pyrlk_opencv_comparison.cc:47,64-65), pyramid border 8",
                   "tracks_per_s": NK / (w11 / 20), "ms_per_frame": w11 / 20 * 1e3, "kernel": "pyrlk_match_group_kernel<11, 32> (round 3: one lane per keypoint, pyrlk_match_kernel<11>)"}
            lib.vpp_set_tuning(b"pyrlk.lpk", 1)   # round 3's path for this window (one lane per keypoint): three launches between an event pair
            step_11(0, st); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(3):
                step_11(i, st)
            e1.record(); torch.cuda.synchronize()
            lib.vpp_set_tuning(b"pyrlk.lpk", -1)
            a11["one_lane_per_keypoint_ms"] = e0.elapsed_time(e1) / 3
            res["authors_config_ws11_4scales"] = a11
        except Exception as e:  # noqa: BLE001
            lib.vpp_set_tuning(b"pyrlk.lpk", -1)
            res["authors_config_ws11_4scales"] = {"error": f"{type(e).__name__}: {e}"}

    if world > 1:
        # the same step without Python on it: one C++ process per GPU (benchmarks/pyrlk_shard_bench.cc), match + RCCL all-gather recorded in a
        # launch graph per rank.  Guarded by a timeout: a harness that cannot initialise its communicator costs at most that.
        exe = os.path.join(ROOT, "benchmarks", "pyrlk_shard_bench")
        if os.path.exists(exe) and os.environ.get("VPP_BENCH_ONE_DEVICE", "0") != "1":
            import json as _json, subprocess as _sp
            uid = f"/tmp/vpp_uid_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
            # F = 1: configs[3] as stated (one pair, 10 k keypoints over the ranks: latency bound per rank); F = 8: eight distinct pairs per step (one
            # vpp_pyrlk_match_batch launch + one all-gather per rank) — the form in which the keypoint-sharded job scales
            for key, nsteps, pairs in (("cpp_harness", "200", "1"), ("cpp_harness_8_frame_pairs", "50", "8")):
                try:
                    out = _sp.run([exe, str(rank), str(world), uid + "_" + pairs, nsteps, str(NK), pairs], capture_output=True, text=True, timeout=180)
                    if rank == 0:
                        res[key] = _json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else {"error": (out.stderr or out.stdout)[-300:], "rc": out.returncode}
                except Exception as e:  # noqa: BLE001
                    if rank == 0:
                        res[key] = {"error": f"{type(e).__name__}: {e}"}
                barrier()
        # BASELINE configs[4] as a tile-sharded step (benchmarks/flow_strip_bench.cc, one C++ process per GPU): RCCL all-gather of the frames'
        # row strips, claim + descent sharded by flow-map row strips with one grouped all-gather of the maps per scale, halo exchange + FAST-9
        # on strips; every rank checks its results against its own single-rank calls.  Reported beside the replicas leg below.
        exe = os.path.join(ROOT, "benchmarks", "flow_strip_bench")
        if os.path.exists(exe) and os.environ.get("VPP_BENCH_ONE_DEVICE", "0") != "1" and 2160 % world == 0:
            import json as _json, subprocess as _sp
            uid = f"/tmp/vpp_uid_flow_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
            try:
                out = _sp.run([exe, str(rank), str(world), uid, "50"], capture_output=True, text=True, timeout=180)
                if rank == 0:
                    res["flow_strips_4k"] = _json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else {"error": (out.stderr or out.stdout)[-300:], "rc": out.returncode}
            except Exception as e:  # noqa: BLE001
                if rank == 0:
                    res["flow_strips_4k"] = {"error": f"{type(e).__name__}: {e}"}
            barrier()
        # weak scaling in keypoints: NK keypoints PER rank (a denser keypoint set on the same frame pair, rank g owning the slice
        # [g*NK, (g+1)*NK) of world*NK), the all-gather carries all world*NK records
        full0 = torch.from_numpy(kps_h.view(np.uint8).reshape(-1).copy()).to(dev)
        full = full0.clone()

        def step_weak(i, stream):
            full.copy_(full0, non_blocking=True)
            match(dp1, dg1, dp2, L, V(full.data_ptr()), NK, WS, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None, stream)
            mg.all_gather_records(full, NK * world, rank, world)

        wwall, _ = timed(step_weak, steps, warmup, graph=False)
        res["weak_scaling"] = {"keypoints_per_rank": NK, "tracks_per_s": NK * world / (wwall / steps), "ms_per_frame": wwall / steps * 1e3,
                               "exchange": f"rccl all_gather of {NK * world} 20-byte records"}

    stage("pyramids")
    # pyramids + gradient of a frame pair (what a caller pays per new frame besides the match)
    dp2_, dg1_ = vi.desc_array(p2), vi.desc_array(g1)

    def step_pyr(i, stream):
        lib.vpp_pyramid_build(dp2_, L, P(d2.desc), stream)            # pyramid2d<uchar>(frame, 3, 2, _border = 3): one launch
        lib.vpp_scharr_pyramid_build(dg1_, L, P(p1[0].desc), stream)  # scharr + gradient pyramid: one launch

    pwall, _ = timed(step_pyr, steps, warmup, graph=True)
    res["pyramids_ms_per_frame"] = pwall / steps * 1e3
    res["tracks_per_s_incl_pyramids"] = NK / ((wall + pwall) / steps)

    stage("fast9")
    # FAST-9 on 4K (replicas): raw and blockwise(10); each call ends with the host read of the keypoint count
    im = u8_image(fast9_bench_frame(), border=3)
    v = im.view(with_border=True)[..., 0]
    v[...] = np.pad(im.view()[..., 0], 3, mode="symmetric")
    dim = DeviceImage.from_host(im, dev)
    cap = 3000000
    rc = torch.zeros((cap, 2), dtype=torch.int32, device=dev); sc = torch.zeros(cap, dtype=torch.int32, device=dev)
    n = ctypes.c_int(0)
    fast = {}
    for name, mode in (("raw", 0), ("blockwise10", 2)):
        for _ in range(3):
            lib.vpp_fast9_detect(P(dim.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st)
        barrier()
        t0 = time.perf_counter()
        it = 20
        for _ in range(it):
            lib.vpp_fast9_detect(P(dim.desc), 20, None, mode, 10, 0, V(rc.data_ptr()), V(sc.data_ptr()), cap, P(n), st)
        barrier()
        dt = (time.perf_counter() - t0) / it
        fast[name] = {"ms": dt * 1e3, "gpixels_per_s": 2160 * 3840 * world / dt / 1e9, "keypoints": n.value}
    fast["raw"]["roofline"] = issue_roofline("fast9_detect2_kernel<true, 0")
    fast["blockwise10"]["roofline"] = issue_roofline("fast9_detect2_kernel<true, 2")
    res["fast9_4k"] = fast

    stage("flow")
    # semi-dense optical flow on one 4K frame pair (BASELINE configs[4] on a single GPU): a keypoint every 10 px
    # (video_extruder keypoint_spacing), winsize 9, 3 scales, propagation 2, patch 5 (video_extruder.hpp:35-41,54)
    from vpp_amd.synth import flow_scene
    s1, s2, sk = flow_scene(2160, 3840, spacing=10)
    e1, e2 = DeviceImage.from_host(u8_image(s1, border=3), dev), DeviceImage.from_host(u8_image(s2, border=3), dev)
    dk = torch.from_numpy(sk).to(dev); m = len(sk)
    gp = torch.zeros((m, 2), dtype=torch.int32, device=dev); gd = torch.zeros(m, dtype=torch.int32, device=dev); gv = torch.zeros(m, dtype=torch.uint8, device=dev)

    def sdof():
        lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(gp.data_ptr()), V(gd.data_ptr()), V(gv.data_ptr()), st)
    for _ in range(2):
        sdof()
    barrier()
    t0 = time.perf_counter()
    it = 5
    for _ in range(it):
        sdof()
    barrier()
    dt = (time.perf_counter() - t0) / it
    res["semi_dense_flow_4k"] = {"ms_per_frame_pair": dt * 1e3, "frame_pairs_per_s": world / dt, "keypoints": m,
                                 "note": "one frame pair per GPU (replicas); serial-order semantics, bit-exact vs the oracle",
                                 "roofline": issue_roofline("sdof_descent_group_kernel<9, false,"), "roofline_sweeps": issue_roofline("sdof_sweep_kernel<9>")}
    stage("flow: concurrent streams")
    # several independent frame pairs in flight on one GPU: each on its own stream (its own scratch: common.hpp Scratch is per stream); a pair is a
    # chain of ~25 short launches (pyramids, claim, descent, classify, propagation rounds), so independent pairs fill each other's launch gaps
    conc = {}
    for k in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        outs = [(torch.zeros((m, 2), dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.uint8, device=dev)) for _ in range(k)]

        def run(reps):
            for _ in range(reps):
                for j in range(k):
                    o = outs[j]
                    lib.vpp_semi_dense_optical_flow(P(e1.desc), P(e2.desc), V(dk.data_ptr()), m, 9, 3, 0, 2, 5, V(o[0].data_ptr()), V(o[1].data_ptr()), V(o[2].data_ptr()),
                                                    ctypes.c_void_p(streams[j].cuda_stream))
        torch.cuda.synchronize()
        run(2)
        barrier()
        t0 = time.perf_counter()
        reps = 4
        run(reps)
        barrier()
        dtk = (time.perf_counter() - t0) / (reps * k)
        same = all(bool(torch.equal(o[0], gp)) and bool(torch.equal(o[1], gd)) and bool(torch.equal(o[2], gv)) for o in outs)
        conc[str(k)] = {"frame_pairs_per_s": world / dtk, "identical_to_the_single_stream_result": same}
    res["semi_dense_flow_4k"]["concurrent_streams"] = conc
    res["semi_dense_flow_4k"]["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")

    stage("ingest")
    # frame ingest: 4K vuchar3 frame -> gray with a mirror-filled border of 3 in one pass; 4 B/px algorithmic.  One call per frame, recorded on one stream:
    # the library holds the calls back and records batched launches (common.hpp: the held-back window); the same calls as one launch each are reported beside it
    from vpp_amd.synth import rand_image
    nin = 64  # 64 x (24.9 + 8.3 MB) = 2.1 GB, all touched by every 64 calls: nothing survives in the Infinity Cache
    rgb_h = rand_image(2160, 3840, vi.U8, 3, border=0, seed=6)
    rgbs = [DeviceImage.from_host(rgb_h, dev) for _ in range(nin)]; grays = [DeviceImage(2160, 3840, vi.U8, 1, 3, 32, dev) for _ in range(nin)]
    rd, gdsc = [x.desc for x in rgbs], [x.desc for x in grays]
    isteps = 1024
    ingest_call = lambda i, s: lib.vpp_rgb_to_graylevel(P(gdsc[i % nin]), P(rd[i % nin]), 1, s)
    iwall, iev = timed(ingest_call, isteps, 64, graph=True, c_graph=True)
    lib.vpp_set_tuning(b"ingest.coalesce", 0); lib.vpp_set_tuning(b"launch.capture_width", 1)
    swall, sev = timed(ingest_call, isteps, 64, graph=True, c_graph=True)
    lib.vpp_set_tuning(b"ingest.coalesce", -1); lib.vpp_set_tuning(b"launch.capture_width", -1)
    ibytes = 2160 * 3840 * 4
    # the same per-frame calls EAGERLY through the deferred window (vpp_rgb_to_graylevel_deferred: what the C++ surface's rgb_to_graylevel calls), no launch graph
    ideferred = None
    try:
        dcall = lib.vpp_rgb_to_graylevel_deferred
        for i in range(2 * nin):
            dcall(P(gdsc[i % nin]), P(rd[i % nin]), 1, st)
        capi.check(lib.vpp_flush(st)); torch.cuda.synchronize()
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0.record()
        for i in range(isteps):
            dcall(P(gdsc[i % nin]), P(rd[i % nin]), 1, st)
        capi.check(lib.vpp_flush(st))
        d1.record(); torch.cuda.synchronize()
        dus = d0.elapsed_time(d1) * 1e3 / isteps
        ideferred = {"us_per_frame": dus, "frac": ibytes / (dus * 1e-6) / 1e9 / 8000.0}
    except AttributeError:
        pass
    res["ingest_4k"] = {"deferred_eager": ideferred,"workload": "vuchar3 3840x2160 -> uchar + mirror border 3 (clone + fill_border_mirror + rgb_to_graylevel fused), one call per frame over 64 frame sets, recorded on one stream",
                        "us_per_frame": iev / isteps * 1e6, "gpixels_per_s": 2160 * 3840 * world / (iwall / isteps) / 1e9,
                        "roofline": {"bound": "hbm", "achieved": ibytes / (iev / isteps) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": ibytes / (iev / isteps) / 1e9 / 8000.0},
                        "one_launch_per_call": {"us_per_frame": sev / isteps * 1e6, "frac": ibytes / (sev / isteps) / 1e9 / 8000.0},
                        "how": "the calls are held back and recorded as 64-frame launches (common.hpp); one_launch_per_call = the same calls with the batching off"}

    stage("ingest + pyramid")
    # ingest fused with the image pyramid it feeds (vpp_rgb_pyramid_build: one launch) against the two-call chain, 4K, border 3, 3 levels
    try:
        npy = 12
        lvs = [[DeviceImage(a, b, vi.U8, 1, 3, 32, dev) for a, b in pyr.level_dims(2160, 3840, 3)] for _ in range(npy)]
        dls = [vi.desc_array(l) for l in lvs]
        psteps = 200
        fw, fev = timed(lambda i, s: lib.vpp_rgb_pyramid_build(dls[i % npy], 3, P(rd[i % nin]), s), psteps, 20, graph=True)

        def chain2(i, s):
            lib.vpp_rgb_to_graylevel(P(gdsc[i % nin]), P(rd[i % nin]), 1, s)
            lib.vpp_pyramid_build(dls[i % npy], 3, P(gdsc[i % nin]), s)
        cw, cev = timed(chain2, psteps, 20, graph=True)
        pw, pev = timed(lambda i, s: lib.vpp_pyramid_build(dls[i % npy], 3, P(gdsc[i % nin]), s), psteps, 20, graph=True)
        res["ingest_pyramid_4k"] = {"workload": "vuchar3 3840x2160 -> 3-level uchar pyramid with mirror border 3",
                                    "fused_us": fev / psteps * 1e6, "ingest_then_pyramid_us": cev / psteps * 1e6, "pyramid_only_us": pev / psteps * 1e6,
                                    "note": "vpp_rgb_pyramid_build (one launch, the gray frame written once) vs vpp_rgb_to_graylevel + vpp_pyramid_build; bit-identical levels"}
    except Exception as e:  # noqa: BLE001
        res["ingest_pyramid_4k"] = {"error": f"{type(e).__name__}: {e}"}

    stage("video_extruder_bench")
    # video_extruder_update on 4K frames through the C++ drop-in surface (benchmarks/video_extruder_bench.cc), rank 0 only
    # BASELINE configs[4] is REPLICAS ONLY: a tracker is sequential in time and 76-78 % of a strip-sharded pair's device time is replicated on
    # every rank (profiles/r06_flow_replicated_share.md), so N GPUs serve N independent video streams — with --gpus N every rank runs its own tracker on its own GPU
    exe = os.path.join(ROOT, "benchmarks", "video_extruder_bench")
    ve = None
    if os.path.exists(exe) and (rank == 0 or world > 1):
        import json, subprocess
        try:
            env = dict(os.environ)
            if world > 1 and os.environ.get("VPP_BENCH_ONE_DEVICE", "0") != "1":
                env["HIP_VISIBLE_DEVICES"] = str(torch.device(dev).index or 0)   # the harness uses device 0 of what it sees
            out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
            ve = json.loads(out.stdout.strip().splitlines()[-1])
            ve.pop("per_update_ms", None)
        except Exception as e:  # noqa: BLE001
            ve = {"error": str(e)}
    if world > 1:
        rates = [None] * world
        dist.all_gather_object(rates, (ve or {}).get("frames_per_s"))
        if rank == 0 and ve is not None:
            ok = [r for r in rates if r]
            ve["replicas"] = {"streams": world, "frames_per_s_total": round(sum(ok), 1), "frames_per_s_per_stream": [round(r, 1) if r else None for r in rates],
                              "scope": "replicas only: one independent 4K video stream (tracker) per GPU, no data-path collective"}
    if rank == 0 and ve is not None:
        res["video_extruder_4k"] = ve
    if rank == 0 and "flow_strips_4k" in res and isinstance(res["flow_strips_4k"], dict):
        res["flow_strips_4k"]["replicated_share_of_device_time"] = {"2_ranks": 0.785, "8_ranks": 0.763, "source": "profiles/r06_flow_replicated_share.md",
                                                                    "scope": "a working exchange, not a scaling design: configs[4] scales as replicas (video_extruder_4k.replicas)"}
    stage("lambda_call_bench")
    # the reference's literal opaque lambdas (benchmarks/box_5x5_filter2.cc:71-81, examples/box_filter.cc:23-32) compiled single-source: us per 4K frame and fraction
    exe = os.path.join(ROOT, "benchmarks", "lambda_call_bench")
    if rank == 0 and os.path.exists(exe):
        import json, subprocess
        try:
            out = subprocess.run([exe, "200"], capture_output=True, text=True, timeout=300)
            res["lambda_call"] = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else {"error": (out.stderr or out.stdout)[-300:], "rc": out.returncode}
        except Exception as e:  # noqa: BLE001
            res["lambda_call"] = {"error": str(e)}
    return res


def cpu_baseline(orc):
    """Oracle (OpenMP build) timed on the host: pyrlk_match on a 1000-keypoint sample of the same scene, and FAST-9 raw on one 4K frame."""
    from vpp_amd import pyr
    from oracle import pyramid as opyr
    from vpp_amd.synth import P, u8_image, texture, translate, rects_image, fast9_bench_frame
    from vpp_amd import image as vi
    V = ctypes.c_void_p
    NR, NC, L, B = 1080, 1920, 3, 3
    tex = texture(NR, NC, seed=5)
    i1 = u8_image(np.clip(np.rint(tex), 0, 255).astype(np.uint8)); i2 = u8_image(np.clip(np.rint(translate(tex, 1.5, -2.25)), 0, 255).astype(np.uint8))
    hp1, hp2 = opyr.host_pyramid(orc, i1, L, B), opyr.host_pyramid(orc, i2, L, B)
    hg = opyr.host_grad_pyramid(orc, hp1[0], L, B, vi.F32)
    kps = pyr.make_keypoints(pyr.grid_keypoints(NR, NC, 10000, margin=32))
    k = kps.copy()
    orc.orc_pyrlk_match(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), L, k.ctypes.data_as(V), len(k), 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None)
    t0 = time.perf_counter(); it = 3
    for _ in range(it):
        k = kps.copy()
        orc.orc_pyrlk_match(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), L, k.ctypes.data_as(V), len(k), 7, ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None)
    dt = (time.perf_counter() - t0) / it
    out = {"pyrlk_tracks_per_s": len(kps) / dt, "pyrlk_sample": f"{it} passes over the same 10k keypoints (pyramids prebuilt)"}
    im = u8_image(fast9_bench_frame(), border=3)
    orc.orc_fill_border(P(im.desc), 0, None)
    rc = np.zeros((3000000, 2), np.int32); n = ctypes.c_int(0)
    t0 = time.perf_counter()
    orc.orc_fast9_detect(P(im.desc), 20, None, 0, 10, 0, rc.ctypes.data_as(V), None, 3000000, P(n))
    out["fast9_raw_gpixels_per_s"] = 2160 * 3840 / (time.perf_counter() - t0) / 1e9
    # semi-dense flow on the same 4K frame pair as the GPU leg, one pass of the serial restatement (the reference's sweeps are
    # sequential by construction and its OpenMP claim loop is racy, SURVEY Q9: one thread is the reference's defined behaviour)
    from vpp_amd.synth import flow_scene
    s1, s2, sk = flow_scene(2160, 3840, spacing=10)
    e1, e2 = u8_image(s1, border=3), u8_image(s2, border=3)
    m = len(sk)
    gp = np.zeros((m, 2), np.int32); gd = np.zeros(m, np.int32); gv = np.zeros(m, np.uint8)
    t0 = time.perf_counter()
    orc.orc_semi_dense_optical_flow(P(e1.desc), P(e2.desc), sk.ctypes.data_as(V), m, 9, 3, 0, 2, 5, gp.ctypes.data_as(V), gd.ctypes.data_as(V), gv.ctypes.data_as(V))
    out["semi_dense_flow_4k_frame_pairs_per_s"] = 1.0 / (time.perf_counter() - t0)
    out["semi_dense_flow_sample"] = "1 pass, 1 thread (serial semantics), same frame pair and keypoints as the GPU leg"
    return out
