"""Host-side: the reference build's (oracle/_ref/libvpp_ref_omp.so) 4K vuchar3 box 5x5 and 4K int add against OMP_NUM_THREADS / binding — what thread count the
cpu_baseline leg of bench.py should be quoted at on a given host.  usage: python tools/cpu_ref_threads.py [child <what>]"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from vpp_amd.synth import P, rand_image
    from vpp_amd import image as vi
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvpp_ref_omp.so"))
    src = rand_image(2160, 3840, vi.U8, 3, border=2, seed=3, align=32)
    dst = src.like(border=0)
    lib.ref_box_filter5x5(P(dst.desc), P(src.desc))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); lib.ref_box_filter5x5(P(dst.desc), P(src.desc)); ts.append(time.perf_counter() - t0)
    b = rand_image(2160, 3840, vi.I32, seed=1, lo=0, hi=2**30 - 1); c = rand_image(2160, 3840, vi.I32, seed=2, lo=0, hi=2**30 - 1); a = b.like()
    lib.ref_pixelwise_add(P(a.desc), P(b.desc), P(c.desc))
    ta = []
    for _ in range(5):
        t0 = time.perf_counter(); lib.ref_pixelwise_add(P(a.desc), P(b.desc), P(c.desc)); ta.append(time.perf_counter() - t0)
    px = 2160 * 3840
    print(f"box min {min(ts) * 1e3:8.2f} ms = {px / min(ts) / 1e9:7.3f} Gpx/s (median {px / sorted(ts)[2] / 1e9:7.3f})   add min {min(ta) * 1e3:7.2f} ms = {px / min(ta) / 1e9:7.2f} Gpx/s", flush=True)
    sys.exit(0)
print("host cpus:", os.cpu_count(), " affinity:", len(os.sched_getaffinity(0)), flush=True)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except OSError:
    pass
for threads in (1, 4, 8, 16, 32, 64, 128, 256):
    if threads > (os.cpu_count() or 1): break
    for extra in ({}, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}, {"OMP_WAIT_POLICY": "passive"}):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), **extra)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
        print(f"threads {threads:4d} {str(extra):58s} {out.stdout.strip() or out.stderr.strip()[-200:]}", flush=True)
