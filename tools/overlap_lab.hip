// overlap_lab.hip — how can CONSECUTIVE, INDEPENDENT 4K launches overlap their ramps and tails?  GPU box only; measurement tooling, not product.
// The reference's call form is one frame per call (benchmarks/box_5x5_filter2.cc:43-69); a 50 MB launch pays ~4.6 us of ramp + drain behind a kernel
// boundary, so one launch per frame reaches 47 % of the HBM peak where 64 frames in one launch reach 70-75 % (LABNOTES.md section 5).
// This program times the same 256 per-frame calls of the product library (vpp_box_filter on 64 rotating 4K vuchar3 frame sets = 1.6 GB + 1.6 GB)
//   defaults    recorded with the library's defaults: consecutive unrelated calls fold into one batched node at record time (box.hip, coalesce_frame)
//   serial      one stream, eager / recorded into a graph with launch.capture_width = 1 (every node behind the previous one)
//   width W     recorded with launch.capture_width = W (IndependentCall: calls on unrelated images become sibling nodes, W lanes)
//   anyorder    hipExtAnyOrderLaunch (the AQL packet without its barrier bit), eager and recorded
//   streams K   K streams round robin, eager, joined once at the end
// plus a micro-test that shows whether two packets of ONE queue overlap at all under hipExtAnyOrderLaunch (a spinning kernel of few workgroups),
// and checks one frame's result per mode against the serial result (bit-exact).
//   build: make -C tools overlap_lab        run: tools/overlap_lab
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/vpp_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define VK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s (line %d)\n", #x, r_, vpp_last_error(), __LINE__); exit(1); } } while (0)

static const int NR = 2160, NC = 3840, CH = 3, BORDER = 2, NSETS = 64, NL = 256;

__global__ void spin_kernel(long long cycles, unsigned long long* stamps, int slot) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = (unsigned long long)t0;
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = (unsigned long long)wall_clock64();
}

struct Frames { std::vector<vpp_image_desc> src, dst; };

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  VK(vpp_init(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // ---- frames
  int32_t spitch, dpitch; size_t sbytes, dbytes, sfirst, dfirst;
  VK(vpp_image_layout(NR, NC, CH, BORDER, 16, &spitch, &sbytes, &sfirst));
  VK(vpp_image_layout(NR, NC, CH, 0, 16, &dpitch, &dbytes, &dfirst));
  std::vector<uint8_t> host(sbytes);
  for (size_t i = 0; i < sbytes; i++) host[i] = (uint8_t)((i * 2654435761u) >> 13);
  Frames F;
  for (int k = 0; k < NSETS; k++) {
    void *s = nullptr, *d = nullptr;
    VK(vpp_malloc(sbytes + 256, &s)); VK(vpp_malloc(dbytes + 256, &d));
    CK(hipMemcpy(s, host.data(), sbytes, hipMemcpyHostToDevice));
    CK(hipMemset(d, 0, dbytes));
    F.src.push_back(vpp_image_desc{(uint8_t*)s + sfirst, NR, NC, spitch, BORDER, VPP_U8, CH});
    F.dst.push_back(vpp_image_desc{(uint8_t*)d + dfirst, NR, NC, dpitch, 0, VPP_U8, CH});
  }
  std::vector<uint8_t> want(dbytes), got(dbytes);
  auto call = [&](int i, hipStream_t s) { VK(vpp_box_filter(&F.dst[i % NSETS], &F.src[i % NSETS], 5, 5, s)); };
  call(0, st); CK(hipStreamSynchronize(st));
  CK(hipMemcpy(want.data(), (uint8_t*)F.dst[0].first_pixel - dfirst, dbytes, hipMemcpyDeviceToHost));
  auto check = [&](const char* what) {
    for (int k : {0, 17, 63}) {
      CK(hipMemcpy(got.data(), (uint8_t*)F.dst[k].first_pixel - dfirst, dbytes, hipMemcpyDeviceToHost));
      if (memcmp(got.data(), want.data(), dbytes)) { printf("  !! %s: frame %d differs from the serial result\n", what, k); return; }
    }
    for (int k = 0; k < NSETS; k++) CK(hipMemsetAsync((uint8_t*)F.dst[k].first_pixel - dfirst, 0, dbytes, st));
    CK(hipStreamSynchronize(st));
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* what, double us) { printf("%-46s %7.3f us per frame   %.3f of 8 TB/s\n", what, us, 6.0 * NR * NC / (us * 1e-6) / 8e12); fflush(stdout); };

  auto time_eager = [&](const char* what, int nstreams) {
    std::vector<hipStream_t> ss{st};
    for (int k = 1; k < nstreams; k++) { hipStream_t x; CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking)); ss.push_back(x); }
    std::vector<hipEvent_t> done(nstreams);
    for (auto& e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<double> t;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipEventRecord(e0, st));
      for (int k = 1; k < nstreams; k++) CK(hipStreamWaitEvent(ss[k], e0, 0));
      for (int i = 0; i < NL; i++) call(i, ss[i % nstreams]);
      for (int k = 1; k < nstreams; k++) { CK(hipEventRecord(done[k], ss[k])); CK(hipStreamWaitEvent(st, done[k], 0)); }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) t.push_back(ms * 1e3 / NL);
    }
    report(what, median(t));
    check(what);
    for (int k = 1; k < nstreams; k++) CK(hipStreamDestroy(ss[k]));
    for (auto& e : done) CK(hipEventDestroy(e));
  };
  auto time_graph = [&](const char* what) {
    vpp_graph* g = nullptr;
    call(0, st); CK(hipStreamSynchronize(st));
    VK(vpp_graph_begin(st));
    for (int i = 0; i < NL; i++) call(i, st);
    if (vpp_graph_end(st, 1, &g) != 0) { printf("%-46s graph_end failed: %s\n", what, vpp_last_error()); return; }
    std::vector<double> t;
    for (int rep = 0; rep < 6; rep++) {
      VK(vpp_graph_launch(g, st)); CK(hipStreamSynchronize(st));
      float ms; VK(vpp_graph_elapsed_ms(g, &ms));
      if (rep) t.push_back(ms * 1e3 / NL);
    }
    report(what, median(t));
    check(what);
    VK(vpp_graph_destroy(g));
  };

  // ---- do two packets of one queue overlap at all without the barrier bit?
  {
    unsigned long long* stamps; CK(hipHostMalloc((void**)&stamps, 64 * sizeof(unsigned long long), hipHostMallocDefault));
    for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
      memset(stamps, 0, 64 * sizeof(unsigned long long));
      CK(hipStreamSynchronize(st));
      for (int k = 0; k < 4; k++)
        hipExtLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, st, nullptr, nullptr, (unsigned)flags, (long long)2000000, stamps, k);   // 100 MHz wall clock: 20 ms
      CK(hipStreamSynchronize(st));
      printf("spin x4 on one stream, flags=%d: start offsets (ms) ", flags);
      for (int k = 0; k < 4; k++) printf("%.2f ", (double)(stamps[2 * k] - stamps[0]) / 1e5);
      printf(" (serial = 0 20 40 60; overlapped = all ~0)\n");
    }
    CK(hipHostFree(stamps));
  }

  time_graph("recorded, library defaults (record-time batching)");
  VK(vpp_set_tuning("box.coalesce", 0));   // from here on: one kernel node per call
  VK(vpp_set_tuning("launch.capture_width", 1));
  time_eager("serial, eager, 1 stream", 1);
  time_graph("serial, recorded (capture_width 1)");
  for (int W : {2, 3, 4, 6, 8, 16}) {
    VK(vpp_set_tuning("launch.capture_width", W));
    char name[96]; snprintf(name, sizeof name, "recorded, capture_width %d", W);
    time_graph(name);
  }
  VK(vpp_set_tuning("launch.capture_width", 1));
  VK(vpp_set_tuning("box.anyorder", 1));
  time_eager("anyorder, eager, 1 stream", 1);
  time_graph("anyorder, recorded (capture_width 1)");
  VK(vpp_set_tuning("box.anyorder", -1));
  for (int K : {2, 4, 8}) {
    char name[96]; snprintf(name, sizeof name, "eager, %d streams round robin", K);
    time_eager(name, K);
  }
  // the batch entry for calibration on this box: 4 / 8 / 64 frames per launch
  for (int fpl : {4, 8, 64}) {
    vpp_graph* g = nullptr;
    VK(vpp_graph_begin(st));
    for (int i = 0; i < NL / fpl; i++) VK(vpp_box_filter_batch(&F.dst[(i * fpl) % NSETS], &F.src[(i * fpl) % NSETS], fpl, 5, 5, st));
    VK(vpp_graph_end(st, 1, &g));
    std::vector<double> t;
    for (int rep = 0; rep < 5; rep++) { VK(vpp_graph_launch(g, st)); CK(hipStreamSynchronize(st)); float ms; VK(vpp_graph_elapsed_ms(g, &ms)); if (rep) t.push_back(ms * 1e3 / NL); }
    char name[96]; snprintf(name, sizeof name, "batch entry, %d frames per launch", fpl);
    report(name, median(t));
    VK(vpp_graph_destroy(g));
  }
  return 0;
}
