"""A process's HBM copy rate on this box, 1.6 GB -> 1.6 GB (the bench's working set), in three consecutive 2-second windows: is a box / a process fast or slow, and from when?"""
import sys, time, torch
n = 1600 * 1024 * 1024
src = torch.empty(n, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
src.random_(0, 255); torch.cuda.synchronize()
out = []
t_start = time.perf_counter()
for w in range(3):
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 2.0:
        for _ in range(20): dst.copy_(src)
        torch.cuda.synchronize(); k += 20
    dt = time.perf_counter() - t0
    out.append(2 * n * k / dt / 1e12)
print("copy TB/s per 2-s window:", " ".join(f"{x:.3f}" for x in out), " frac of 8 TB/s:", " ".join(f"{x / 8:.3f}" for x in out), flush=True)
