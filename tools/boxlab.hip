// boxlab.hip — standalone calibration / sweep harness for K2 (box.hip), GPU box only.  Measurement tooling, not product:
// it compiles box.hip into this translation unit with -DVPP_BOX_LAB (every geometry / cache-policy instance of the
// line-aligned kernel), checks each configuration bit-exact against a plain host loop over the same bytes, and times it
// as a hipGraph of back-to-back launches over rotating frame sets (> 256 MiB, so the Infinity Cache cannot serve the stream).
//   build: make -C tools boxlab        run: tools/boxlab [sweep-name | "k=v,k=v" ...]
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>
#include "../vpp_amd/csrc/box.hip"

namespace vpp_amd {
static char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
static std::map<std::string, int> g_tune;
int tuning(const char* name, int dflt) { auto it = g_tune.find(name); return it == g_tune.end() ? dflt : it->second; }
// the library's held-back window (runtime.hip) does not exist in this harness: nothing is ever pending
std::atomic<int> g_defer_pending{0};
thread_local int g_defer_bypass = 0;
int defer_flush_stream(void*) { return 0; }
}  // namespace vpp_amd

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 32 frame sets: 800 MB of sources + 800 MB of results.  (Rounds 1-2 rotated over 8 sets = 200 MB of sources, which the 256 MiB Infinity Cache
// kept resident — the stores are non-temporal and do not allocate — so those timings were of an L3-fed kernel, not an HBM-fed one.)
static const int NR = 2160, NC = 3840, CH = 3, BORDER = 2, NSETS = 32;

__global__ void copy16_kernel(u32x4* __restrict__ d, const u32x4* __restrict__ s, size_t n, int nt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    if (nt) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i); else d[i] = s[i];
  }
}

struct Timing { double best, med; };
template <class F> Timing time_graph(hipStream_t st, F launch, int steps = 200, int reps = 7) {
  for (int i = 0; i < 16; i++) launch(i, st);
  CK(hipStreamSynchronize(st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < steps; i++) launch(i, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> us;
  for (int r = 0; r < reps; r++) {
    CK(hipGraphLaunch(ge, st));  // keeps the queue busy so that the timed replay starts behind running work
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    us.push_back(ms * 1e3 / steps);
  }
  std::sort(us.begin(), us.end());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return Timing{us.front(), us[us.size() / 2]};
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipStream_t st; CK(hipStreamCreate(&st));
  // layout of imageNd::allocate (vpp/core/imageNd.hpp:151-196) for vuchar3, alignment 16
  const int row_bytes = NC * CH;
  auto layout = [&](int border, int& pitch, size_t& bytes, size_t& first) {
    int bs = border * CH, pad = 0;
    if (bs % 16) { pad = 16 - bs % 16; bs += pad; }
    pitch = row_bytes + 2 * bs; if (pitch % 16) pitch += 16 - pitch % 16;
    bytes = (size_t)(NR + 2 * border) * pitch; first = pad + (size_t)border * pitch + border * CH;
  };
  int spitch, dpitch; size_t sbytes, dbytes, sfirst, dfirst;
  layout(BORDER, spitch, sbytes, sfirst); layout(0, dpitch, dbytes, dfirst);
  std::vector<uint8_t> hs(sbytes), want((size_t)NR * row_bytes), got(dbytes);
  std::mt19937 rng(3);
  for (size_t i = 0; i < sbytes; i += 4) { uint32_t v = rng(); memcpy(&hs[i], &v, std::min<size_t>(4, sbytes - i)); }
  {  // host check values: the kernel lambda of benchmarks/box_5x5_filter2.cc:73-80 on interleaved bytes
    const uint8_t* p0 = hs.data() + sfirst;
    std::vector<int> col(row_bytes + 12);
    for (int r = 0; r < NR; r++) {
      for (int x = -6; x < row_bytes + 6; x++) { int s = 0; for (int dr = -2; dr <= 2; dr++) s += p0[(ptrdiff_t)(r + dr) * spitch + x]; col[x + 6] = s; }
      for (int x = 0; x < row_bytes; x++) want[(size_t)r * row_bytes + x] = (uint8_t)((col[x] + col[x + 3] + col[x + 6] + col[x + 9] + col[x + 12]) / 25);
    }
  }
  uint8_t *ds[NSETS], *dd[NSETS];
  vpp_image_desc sd[NSETS], ddsc[NSETS];
  for (int k = 0; k < NSETS; k++) {
    CK(hipMalloc(&ds[k], sbytes)); CK(hipMalloc(&dd[k], dbytes));
    CK(hipMemcpy(ds[k], hs.data(), sbytes, hipMemcpyHostToDevice)); CK(hipMemset(dd[k], 0xEE, dbytes));
    sd[k] = vpp_image_desc{ds[k] + sfirst, NR, NC, spitch, BORDER, VPP_U8, CH};
    ddsc[k] = vpp_image_desc{dd[k] + dfirst, NR, NC, dpitch, 0, VPP_U8, CH};
  }
  printf("# src pitch %d first %zu (base %% 256 = %zu), dst pitch %d\n", spitch, sfirst, (size_t)((uintptr_t)ds[0] % 256), dpitch);
  auto box = [&](int i, hipStream_t s) {
    const int batch = vpp_amd::tuning("box.batch", 1);   // frames per launch (vpp_box_filter_batch); NSETS % batch == 0
    const int k = (i * batch) % NSETS;
    const int rc = batch > 1 ? vpp_box_filter_batch(&ddsc[k], &sd[k], batch, 5, 5, s) : vpp_box_filter(&ddsc[k], &sd[k], 5, 5, s);
    if (rc != VPP_OK) { fprintf(stderr, "box: %s\n", vpp_amd::g_err); exit(1); }
  };
  // clock preheat: ~0.3 s of the default kernel
  { for (int rep = 0; rep < 60; rep++) { for (int i = 0; i < 500; i++) box(i, st); CK(hipStreamSynchronize(st)); } }

  auto run_cfg = [&](const std::string& cfg) {
    vpp_amd::g_tune.clear();
    size_t p = 0;
    while (p < cfg.size()) {
      size_t q = cfg.find(',', p); if (q == std::string::npos) q = cfg.size();
      std::string kv = cfg.substr(p, q - p); size_t eq = kv.find('=');
      if (eq != std::string::npos) vpp_amd::g_tune["box." + kv.substr(0, eq)] = atoi(kv.c_str() + eq + 1);
      p = q + 1;
    }
    CK(hipMemset(dd[0], 0xEE, dbytes));
    box(0, st); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(got.data(), dd[0], dbytes, hipMemcpyDeviceToHost));
    size_t bad = 0;
    const int probe = vpp_amd::tuning("box.probe", 0);
    if (!probe) for (int r = 0; r < NR; r++) bad += memcmp(&got[dfirst + (size_t)r * dpitch], &want[(size_t)r * row_bytes], row_bytes) != 0;
    const int batch = vpp_amd::tuning("box.batch", 1);
    Timing t = time_graph(st, box, std::max(32, 200 / batch));
    t.best /= batch; t.med /= batch;   // per frame
    printf("%-58s best %6.2f us  med %6.2f us  %5.2f TB/s  frac %.3f  %s\n", cfg.c_str(), t.best, t.med, 6.0 * NR * NC / t.med / 1e6, 6.0 * NR * NC / t.med / 1e6 / 8.0,
           probe ? "(probe)" : bad ? "MISMATCH" : "exact");
    if (bad) printf("   !! %zu rows differ\n", bad);
    fflush(stdout);
  };

  std::vector<std::string> cfgs;
  std::string mode = argc > 1 ? argv[1] : "sweep1";
  if (mode == "sweep1") {
    cfgs.push_back("impl=1,rows=2");
    cfgs.push_back("impl=1,rows=2,probe=1");
    for (int order : {0, 1})
      for (int wx : {4, 2, 1})
        for (int rows : {2, 4, 3, 1}) {
          char b[128]; snprintf(b, sizeof b, "impl=2,rows=%d,wx=%d,order=%d,sp=1,ntload=0", rows, wx, order); cfgs.push_back(b);
        }
    for (int sp : {0, 2, 3, 4}) for (int ntl : {0, 1}) { char b[128]; snprintf(b, sizeof b, "impl=2,rows=2,wx=4,order=0,sp=%d,ntload=%d", sp, ntl); cfgs.push_back(b); }
    cfgs.push_back("impl=2,rows=2,wx=4,order=0,sp=1,ntload=1");
    cfgs.push_back("impl=2,rows=2,wx=4,order=0,sp=1,ntload=0,probe=1");
    cfgs.push_back("impl=2,rows=4,wx=4,order=0,sp=1,ntload=0,probe=1");
    cfgs.push_back("impl=1,rows=2");
  } else if (mode == "sweep3") {
    cfgs.push_back("impl=1,rows=2");
    for (int halo : {0, 1})
      for (int wx : {1, 4})
        for (int rows : {2, 3, 4, 5})
          for (int mix : {0, 1})
            for (int occ : {8, 6}) {
              if (rows == 2 && mix) continue;
              if (rows <= 3 && occ != 8) continue;
              char b[160]; snprintf(b, sizeof b, "impl=2,halo=%d,wx=%d,order=%d,rows=%d,mix=%d,occ=%d", halo, wx, wx == 1 ? 1 : 0, rows, mix, occ); cfgs.push_back(b);
            }
    for (int rows : {2, 4}) for (int probe : {1, 3}) { char b[160]; snprintf(b, sizeof b, "impl=2,halo=0,wx=1,order=1,rows=%d,mix=%d,occ=8,probe=%d", rows, rows == 4, probe); cfgs.push_back(b); }
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=4,mix=1,occ=8,probe=1");
    cfgs.push_back("impl=2,halo=0,wx=1,order=1,rows=4,mix=1,occ=8,sp=18");
    cfgs.push_back("impl=1,rows=2");
  } else if (mode == "sweep4") {
    cfgs.push_back("impl=1,rows=2");
    for (int wx : {4, 2})
      for (int order : {0, 1, 2, 3})
        for (int rows : {2, 3}) { char b[160]; snprintf(b, sizeof b, "impl=2,halo=0,wx=%d,order=%d,rows=%d,mix=0,occ=8", wx, order, rows); cfgs.push_back(b); }
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=6");
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=4");
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8,sp=18");
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8,probe=1");
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8,probe=3");
    cfgs.push_back("impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8");
  } else if (mode == "sweep5") {
    cfgs.push_back("impl=2");
    for (int shape : {1, 2, 3, 4, 5, 6}) { char b[160]; snprintf(b, sizeof b, "impl=2,shape=%d,rows=2", shape); cfgs.push_back(b); }
    cfgs.push_back("impl=2,shape=1,rows=3");
    for (int laux : {0, 1, 2, 16})
      for (int sp : {0, 1, 2, 3, 16, 17, 18, 19}) { if (laux == 0 && sp == 2) continue; char b[160]; snprintf(b, sizeof b, "impl=2,sp=%d,laux=%d", sp, laux); cfgs.push_back(b); }
    cfgs.push_back("impl=2");
  } else if (mode == "sweep6") {   // round 3: everything again with the sources really coming from HBM
    for (int batch : {1, 8}) {
      char b[160];
      snprintf(b, sizeof b, "impl=2,batch=%d", batch); cfgs.push_back(b);
      for (int rows : {2, 3, 4}) for (int wx : {4, 1}) { snprintf(b, sizeof b, "impl=2,halo=0,wx=%d,order=%d,rows=%d,mix=0,occ=8,batch=%d", wx, wx == 1 ? 1 : 0, rows, batch); cfgs.push_back(b); }
      for (int order : {1, 2, 3}) { snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=%d,rows=2,mix=0,occ=8,batch=%d", order, batch); cfgs.push_back(b); }
      for (int occ : {6, 4}) { snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=%d,batch=%d", occ, batch); cfgs.push_back(b); }
      for (int probe : {1, 3}) { snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8,probe=%d,batch=%d", probe, batch); cfgs.push_back(b); }
      for (int laux : {1, 2, 16}) { snprintf(b, sizeof b, "impl=2,sp=2,laux=%d,batch=%d", laux, batch); cfgs.push_back(b); }
      for (int sp : {0, 1, 3, 16, 18}) { snprintf(b, sizeof b, "impl=2,sp=%d,laux=0,batch=%d", sp, batch); cfgs.push_back(b); }
      for (int shape : {1, 5, 6}) { snprintf(b, sizeof b, "impl=2,shape=%d,rows=2,batch=%d", shape, batch); cfgs.push_back(b); }
      snprintf(b, sizeof b, "impl=2,halo=1,wx=4,order=0,rows=2,mix=0,occ=8,batch=%d", batch); cfgs.push_back(b);
    }
    cfgs.push_back("impl=1,rows=2");
    cfgs.push_back("impl=0,rows=8");
  } else if (mode == "sweep7") {   // batch of 8, honest HBM: more rows per wave (more unique bytes in flight per wave) against occupancy
    cfgs.push_back("impl=2,batch=8");
    for (int rows : {3, 4, 5, 6})
      for (int occ : {8, 6, 4})
        for (int mix : {0, 1}) { char b[160]; snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=%d,mix=%d,occ=%d,batch=8", rows, mix, occ); cfgs.push_back(b); }
    for (int rows : {3, 4, 6}) { char b[160]; snprintf(b, sizeof b, "impl=2,halo=0,wx=2,order=0,rows=%d,mix=0,occ=%d,batch=8", rows, rows == 3 ? 8 : 4); cfgs.push_back(b); }
    for (int batch : {2, 4, 16}) { char b[160]; snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=3,mix=0,occ=8,batch=%d", batch); cfgs.push_back(b); }
  } else if (mode == "sweep9") {   // which kernel at which batch size
    for (int batch : {1, 2, 4, 8, 16, 32}) {
      char b[160];
      snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8,batch=%d", batch); cfgs.push_back(b);
      snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=3,mix=0,occ=8,batch=%d", batch); cfgs.push_back(b);
      snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=6,mix=0,occ=4,batch=%d", batch); cfgs.push_back(b);
      snprintf(b, sizeof b, "impl=2,halo=0,wx=4,order=0,rows=2,mix=0,occ=8,probe=1,batch=%d", batch); cfgs.push_back(b);
    }
  } else if (mode == "sweep2") {
    for (int wx : {1, 4})
      for (int probe : {0, 1, 2, 3})
        for (int rows : {2, 4}) { char b[128]; snprintf(b, sizeof b, "impl=2,rows=%d,wx=%d,order=%d,sp=1,ntload=0,probe=%d", rows, wx, wx == 1 ? 1 : 0, probe); cfgs.push_back(b); }
  } else {
    for (int i = 1; i < argc; i++) cfgs.push_back(argv[i]);
  }
  for (auto& c : cfgs) run_cfg(c);

  // plain 1:1 streams of the same byte count for calibration
  {
    const size_t n16 = 24883200 / 16;
    for (int nt : {0, 1})
      for (int blocks : {2048, 8192, 16384, 97200}) {
        Timing t = time_graph(st, [&](int i, hipStream_t s) { int k = i % NSETS; copy16_kernel<<<blocks, 256, 0, s>>>((u32x4*)dd[k], (const u32x4*)ds[k], n16, nt); });
        printf("copy 24.9 MB -> 24.9 MB, 16 B/lane grid-stride, %5d blocks, nt=%d: best %6.2f med %6.2f us (%.2f TB/s)\n", blocks, nt, t.best, t.med, 2.0 * 24883200 / t.med / 1e6);
      }
    Timing t = time_graph(st, [&](int i, hipStream_t s) { int k = i % NSETS; CK(hipMemcpyAsync(dd[k], ds[k], 24883200, hipMemcpyDeviceToDevice, s)); });
    printf("hipMemcpyAsync D2D 24.9 MB: best %6.2f med %6.2f us\n", t.best, t.med);
  }
  return 0;
}
