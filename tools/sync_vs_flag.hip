// sync_vs_flag.hip — how much of a synchronous call is the runtime's completion path?  A ~20 us kernel, 300 times: (a) launch + hipStreamSynchronize;
// (b) launch + the host polling a word in pinned host memory that the kernel's LAST block (arrival counters) writes.  Build: make -C tools sync_vs_flag
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void work_kernel(unsigned* data, int iters, unsigned* sub, unsigned* done, volatile unsigned* flag, unsigned seq) {
  unsigned v = data[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
  data[blockIdx.x * 256 + threadIdx.x] = v;
  if (!flag) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nsub = 64, sc = blockIdx.x % nsub, expect = gridDim.x / nsub + (sc < gridDim.x % nsub ? 1u : 0u);
    if (__hip_atomic_fetch_add(&sub[sc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1) {
      __hip_atomic_store(&sub[sc * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((unsigned*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

int main() {
  const int blocks = 4096, iters = 6000, N = 300;
  unsigned *data, *ctl; volatile unsigned* flag;
  CK(hipMalloc(&data, blocks * 256 * 4)); CK(hipMemset(data, 1, blocks * 256 * 4));
  CK(hipMalloc(&ctl, 65 * 32 * 4)); CK(hipMemset(ctl, 0, 65 * 32 * 4));
  CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault)); *flag = 0;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int mode = 0; mode < 4; mode++) {
    double best = 1e9, sum = 0;
    for (int i = 0; i < N + 20; i++) {
      const unsigned seq = (unsigned)(mode * 100000 + i + 1);
      const double t0 = now();
      if (mode % 2 == 0) {
        work_kernel<<<blocks, 256, 0, st>>>(data, iters, ctl, ctl + 64 * 32, nullptr, seq);
        CK(hipStreamSynchronize(st));
      } else {
        work_kernel<<<blocks, 256, 0, st>>>(data, iters, ctl, ctl + 64 * 32, flag, seq);
        while (*flag != seq) _mm_pause();
      }
      const double dt = now() - t0;
      if (i >= 20) { sum += dt; best = dt < best ? dt : best; }
    }
    CK(hipStreamSynchronize(st));
    std::printf("%s: mean %.2f us, min %.2f us per synchronous launch\n", mode % 2 == 0 ? "launch + hipStreamSynchronize      " : "launch + poll a pinned completion word", sum / N * 1e6, best * 1e6);
  }
  // the kernel alone, 100 launches back to back
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 100; i++) work_kernel<<<blocks, 256, 0, st>>>(data, iters, ctl, ctl + 64 * 32, nullptr, 0);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("kernel alone: %.2f us per launch (100 back to back)\n", ms * 10);
  return 0;
}
