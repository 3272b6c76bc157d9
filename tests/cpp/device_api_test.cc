// The vpp-shaped C++ surface in a -DVPP_AMD_DEVICE build: tagged functors and algorithm front-ends run on the MI355X
// through the C ABI and are checked against the CPU oracle (test infrastructure) on the same inputs.
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <random>

#include <vpp/vpp.hh>
#include <vpp/algorithms/fast_detector/fast.hh>
#include <vpp/algorithms/filters/scharr.hh>
#include <vpp/algorithms/lucas_kanade.hh>
#include <vpp/algorithms/optical_flow.hh>
#include <vpp/algorithms/pyrlk/pyrlk_match.hh>
#include <vpp/algorithms/video_extruder.hh>
#include <vpp/algorithms/lbp/lbp_transform.hh>
#include <vpp/algorithms/lbp/lbp_distance.hh>

#include "../../oracle/oracle.h"

using namespace vpp;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

template <class V> vpp_image_desc host_desc(const image2d<V>& i) {
  typedef pixel_traits<V> PT;
  return vpp_image_desc{(void*)&i(0, 0), i.nrows(), i.ncols(), i.pitch(), i.border(), device::dtype_of<typename PT::component>::value, PT::channels};
}

static std::mt19937 rng(5);
static image2d<unsigned char> texture(int nr, int nc, float dr, float dc) {  // smooth random texture, optionally translated
  static std::vector<float> base;
  const int W = nc + 64, H = nr + 64;
  if (base.empty()) {
    std::vector<float> n(size_t(W) * H);
    for (auto& x : n) x = float(rng() & 0xFFFF) / 65535.f;
    base.assign(n.size(), 0.f);
    for (int pass = 0; pass < 3; pass++) {  // three box blurs ~ gaussian
      for (int r = 2; r < H - 2; r++) for (int c = 2; c < W - 2; c++) { float s = 0; for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) s += n[size_t(r + i) * W + c + j]; base[size_t(r) * W + c] = s / 25; }
      n = base;
    }
    float lo = 1e9, hi = -1e9; for (int r = 8; r < H - 8; r++) for (int c = 8; c < W - 8; c++) { lo = std::min(lo, base[size_t(r) * W + c]); hi = std::max(hi, base[size_t(r) * W + c]); }
    for (auto& x : base) x = (x - lo) / (hi - lo) * 255.f;
  }
  image2d<unsigned char> img(nr, nc, _border = 3, _aligned = 32);  // 32: the reference's AVX2 FAST path uses aligned 256-bit loads (fast.hpp:131,312)
  for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) {
    const float y = r + 32 - dr, x = c + 32 - dc; const int y0 = int(y), x0 = int(x); const float a = y - y0, b = x - x0;
    const float v = (1 - a) * (1 - b) * base[size_t(y0) * W + x0] + a * (1 - b) * base[size_t(y0 + 1) * W + x0] + (1 - a) * b * base[size_t(y0) * W + x0 + 1] + a * b * base[size_t(y0 + 1) * W + x0 + 1];
    img(r, c) = (unsigned char)std::min(255.f, std::max(0.f, std::round(v)));
  }
  return img;
}

static void test_pixel_wise_functors() {
  image2d<int> A(1080, 1920), B(1080, 1920), C(1080, 1920);                  // BASELINE configs[0] shape
  for (auto p : B.domain()) { B(p) = int(rng() >> 2); C(p) = int(rng() >> 2); }
  pixel_wise(A, B, C) | ops::add();                                          // -> vpp_pixelwise_binary on the GPU
  for (auto p : A.domain()) CHECK(A(p) == B(p) + C(p));                      // benchmarks/image_add.cc:21-28
  pixel_wise(A, B, C) | ops::absdiff();
  for (int r = 0; r < 1080; r += 7) for (int c = 0; c < 1920; c += 5) CHECK(A(r, c) == std::abs(B(r, c) - C(r, c)));

  image2d<vuchar3> S(270, 480, _border = 2, _aligned = 16), D(270, 480, _aligned = 16), W(270, 480, _aligned = 16), D2(270, 480, _aligned = 16);
  for (auto p : S.domain()) S(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
  fill_border_mirror(S);                                                     // device
  pixel_wise(D, relative_access(S)) | ops::box_mean<5, 5>();                 // device, LDS-tiled kernel
  pixel_wise(D2, box_nbh2d<vuchar3, 5, 5>(S)) | ops::box_mean<5, 5>();       // legacy spelling, same kernel
  const vpp_image_desc ds = host_desc(S), dw = host_desc(W);
  CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
  for (auto p : D.domain()) CHECK(D(p) == W(p) && D2(p) == W(p));
  // the same functor evaluated by the host engine on an opaque lambda gives the same pixels
  image2d<vuchar3> H(270, 480);
  pixel_wise(H, relative_access(S)) | [](vuchar3& o, auto n) { ops::box_mean<5, 5>()(o, n); };
  for (auto p : D.domain()) CHECK(H(p) == D(p));
}

// Stacks of frames under pixel_wise (vpp/core/pixel_wise.hh): an image3d (slice = frame) and a std::vector<image2d> evaluate the whole stack in ONE
// device launch for the tagged functors (vpp_box_filter_batch / vpp_pixelwise_binary_batch: the entry points the 4K roofline numbers are measured on,
// reached here without touching vpp_amd.h), and frame by frame for opaque lambdas.  Checker: the oracle per frame.
static void test_frame_stacks() {
  std::mt19937 rng(77);   // its own generator: the later tests' scenes are drawn from the global one
  const int n = 9, nr = 96, nc = 200;
  image3d<vuchar3> S(n, nr, nc, _border = 2, _aligned = 16), D(n, nr, nc, _aligned = 16), H(n, nr, nc, _aligned = 16);
  for (int k = 0; k < n; k++) {
    image2d<vuchar3> f = S.slice(k);
    CHECK(&f(0, 0) == &S(k, 0, 0) && &f(nr - 1, nc - 1) == &S(k, nr - 1, nc - 1) && f.border() == 2 && f.pitch() == S.pitch());
    for (auto p : f.domain()) f(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    fill_border_mirror(f);                                                 // device, on the slice's rows of the shared mirror
  }
  pixel_wise(D, relative_access(S)) | ops::box_mean<5, 5>();                 // ONE launch: vpp_box_filter_batch over the 9 frames
  pixel_wise(H, relative_access(S)) | [](vuchar3& o, auto nb) { ops::box_mean<5, 5>()(o, nb); };   // host engine, frame by frame
  for (int k = 0; k < n; k++) {
    image2d<vuchar3> W(nr, nc, _aligned = 16);
    const image2d<vuchar3> sk = S.slice(k);
    const vpp_image_desc ds = host_desc(sk), dw = host_desc(W);
    CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
    for (auto p : W.domain()) CHECK(D(k, p[0], p[1]) == W(p) && H(k, p[0], p[1]) == W(p));
  }
  // frames allocated one by one (a decoder's ring): std::vector<image2d>
  std::vector<image2d<int>> A, B, C;
  for (int k = 0; k < 5; k++) {
    A.emplace_back(135, 240); B.emplace_back(135, 240); C.emplace_back(135, 240);
    for (auto p : B[k].domain()) { B[k](p) = int(rng() >> 3); C[k](p) = int(rng() >> 3); }
  }
  pixel_wise(A, B, C) | ops::add();                                          // ONE launch: vpp_pixelwise_binary_batch
  for (int k = 0; k < 5; k++) for (auto p : A[k].domain()) CHECK(A[k](p) == B[k](p) + C[k](p));
  std::vector<image2d<vuchar3>> VS, VD;
  for (int k = 0; k < 4; k++) {
    VS.emplace_back(64, 128, _border = 2, _aligned = 16); VD.emplace_back(64, 128, _aligned = 16);
    for (auto p : VS[k].domain()) VS[k](p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    fill_border_mirror(VS[k]);
  }
  pixel_wise(VD, relative_access(VS)) | ops::box_mean<5, 5>();
  for (int k = 0; k < 4; k++) {
    image2d<vuchar3> W(64, 128, _aligned = 16);
    const vpp_image_desc ds = host_desc(VS[k]), dw = host_desc(W);
    CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
    for (auto p : W.domain()) CHECK(VD[k](p) == W(p));
  }
  bool thrown = false;
  try { VD.pop_back(); pixel_wise(VD, relative_access(VS)) | ops::box_mean<5, 5>(); } catch (const std::runtime_error&) { thrown = true; }
  CHECK(thrown);                                                             // stacks of different sizes
}

// The reference's call form — ONE frame per call (benchmarks/box_5x5_filter2.cc:43-81, benchmarks/image_add.cc:51-57) — through the deferred window of the
// library (include/vpp_amd.h: vpp_*_deferred; vpp/core/device.hh): the tagged functors do not launch per call, whole windows go out as batched launches, and
// everything a caller can observe (pixels, order, data flow, lifetime) stays that of per-call launches.  Checker: the oracle, frame by frame.
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void test_deferred_per_frame_calls() {
  std::mt19937 rng(91);
  {  // 70 distinct small frames: a window of 64 goes out by itself, the other 6 at the first host access
    const int n = 70, nr = 96, nc = 200;
    std::vector<image2d<vuchar3>> S, D;
    for (int k = 0; k < n; k++) {
      S.emplace_back(nr, nc, _border = 2, _aligned = 16); D.emplace_back(nr, nc, _aligned = 16);
      for (auto p : S[k].domain()) S[k](p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
      fill_border_mirror(S[k]);
      fill(D[k], vuchar3(1, 2, 3));
    }
    for (int k = 0; k < n; k++) (void)D[k].device_desc(true);   // the results' mirrors exist: no upload sits between the calls below
    CHECK(vpp_deferred_pending() == 0);
    const unsigned long long f0 = vpp_deferred_flushes();
    for (int k = 0; k < n; k++) {
      pixel_wise(D[k], relative_access(S[k])) | ops::box_mean<5, 5>();
      CHECK(vpp_deferred_pending() == (k + 1) % 64);
    }
    CHECK(vpp_deferred_flushes() == f0 + 1 && vpp_deferred_pending() == 6);
    const vuchar3 first = D[0](0, 0);                            // a host accessor: everything pending is launched, the mirror comes back
    CHECK(vpp_deferred_pending() == 0 && vpp_deferred_flushes() == f0 + 2);
    (void)first;
    for (int k = 0; k < n; k++) {
      image2d<vuchar3> W(nr, nc, _aligned = 16);
      const vpp_image_desc ds = host_desc(S[k]), dw = host_desc(W);
      CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
      for (auto p : W.domain()) CHECK(D[k](p) == W(p));
    }
  }
  {  // data flow: a call that reads or overwrites a pending result does not join it (the results are those of the calls in sequence)
    image2d<int> A(135, 240), B(135, 240), C(135, 240), E(135, 240), X(135, 240);
    for (auto p : B.domain()) { B(p) = int(rng() >> 3); C(p) = int(rng() >> 3); X(p) = int(rng() >> 4); }
    for (auto* im : {&A, &B, &C, &E, &X}) (void)im->device_desc(false);
    pixel_wise(A, B, C) | ops::add();        // pending
    pixel_wise(E, A, B) | ops::add();        // reads A: the first call is launched, this one opens a new window
    CHECK(vpp_deferred_pending() == 1);
    pixel_wise(X, X, C) | ops::add();        // in place: never deferred (the batched kernel does not serve it), runs behind the window
    CHECK(vpp_deferred_pending() == 0);
    pixel_wise(A, E, X) | ops::sub();        // overwrites A, which E's (already launched) call read
    pixel_wise(E, B, C) | ops::sub();        // overwrites E, which the PENDING call reads: launched first
    CHECK(vpp_deferred_pending() == 1);
    for (auto p : A.domain()) CHECK(E(p) == B(p) - C(p));
    // A = E1 - X' with E1 = (B + C) + B (E's first value) and X' = X + C (what X holds after the in-place call)
    for (auto p : A.domain()) CHECK(A(p) == ((B(p) + C(p)) + B(p)) - X(p));
  }
  {  // a source changed on the host while a call that read it is pending: the pending call saw the old pixels, the next call sees the new ones
    image2d<vuchar3> S(64, 128, _border = 2, _aligned = 16), D1(64, 128, _aligned = 16), D2(64, 128, _aligned = 16), W(64, 128, _aligned = 16);
    for (auto p : S.domain()) S(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    fill_border_mirror(S);
    (void)D1.device_desc(true); (void)D2.device_desc(true);
    { const vpp_image_desc ds = host_desc(S), dw = host_desc(W); CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0); }   // (host_desc reads through the host accessor: S's host copy is current)
    (void)S.device_desc(false);
    pixel_wise(D1, relative_access(S)) | ops::box_mean<5, 5>();
    CHECK(vpp_deferred_pending() == 1);
    S(10, 10) = vuchar3(255, 0, 255);                             // host write: the mirror is stale from here on, the pending call still reads it
    pixel_wise(D2, relative_access(S)) | ops::box_mean<5, 5>();   // uploads S first — behind the pending call
    for (auto p : W.domain()) CHECK(D1(p) == W(p));
    { const vpp_image_desc ds = host_desc(S), dw = host_desc(W); CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0); }
    for (auto p : W.domain()) CHECK(D2(p) == W(p));
  }
  {  // an image that dies while a call on it is pending (its block returns to the pool), and the RAII scope
    image2d<int> A(135, 240), B(135, 240);
    for (auto p : B.domain()) B(p) = int(rng() >> 3);
    (void)A.device_desc(true); (void)B.device_desc(false);
    {
      image2d<int> T(135, 240);
      for (auto p : T.domain()) T(p) = p[0] * 1000 + p[1];
      (void)T.device_desc(false);
      pixel_wise(A, B, T) | ops::add();
      CHECK(vpp_deferred_pending() == 1);
    }                                                              // ~T: vpp_free launches the window before the block can be reused
    CHECK(vpp_deferred_pending() == 0);
    image2d<int> T2(135, 240);                                    // very likely T's block
    fill(T2, -7);
    (void)T2.device_desc(false);
    for (auto p : A.domain()) CHECK(A(p) == B(p) + p[0] * 1000 + p[1]);
    {
      device::batch_scope scope;
      pixel_wise(A, B, T2) | ops::add();
      CHECK(vpp_deferred_pending() == 1);
    }
    CHECK(vpp_deferred_pending() == 0);
    for (auto p : A.domain()) CHECK(A(p) == B(p) - 7);
  }
  {  // rgb_to_graylevel per frame (examples/video_extruder.cc:44-48 form): deferred as well, same pixels as the oracle's
    std::vector<image2d<vuchar3>> F;
    for (int k = 0; k < 5; k++) { F.emplace_back(72, 160, _aligned = 16); for (auto p : F[k].domain()) F[k](p) = vuchar3(rng() & 255, rng() & 255, rng() & 255); (void)F[k].device_desc(false); }
    std::vector<image2d<unsigned char>> G;
    for (int k = 0; k < 5; k++) G.push_back(rgb_to_graylevel_mirror(F[k], 3));
    for (int k = 0; k < 5; k++) {
      image2d<unsigned char> W(72, 160, _border = 3);
      const vpp_image_desc dfr = host_desc(F[k]), dw = host_desc(W);
      CHECK(orc_rgb_to_graylevel(&dw, &dfr, 1) == 0);
      for (auto p : W.domain_with_border()) CHECK(G[k](p) == W(p));
    }
  }
  {  // the measured loop: 64 rotating 4K vuchar3 frame sets, one frame per call (1.6 GB of sources + 1.6 GB of results per pass: nothing survives in the 256 MiB Infinity Cache)
    const int n = 64, nr = 2160, nc = 3840, passes = 8;
    std::vector<image2d<vuchar3>> S, D;
    image2d<vuchar3> base(nr, nc, _border = 2, _aligned = 16);
    for (auto p : base.domain()) base(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    for (int k = 0; k < n; k++) {
      S.emplace_back(nr, nc, _border = 2, _aligned = 16); D.emplace_back(nr, nc, _aligned = 16);
      const unsigned char m = (unsigned char)(k * 37 + 1);
      for (int r = 0; r < nr; r++) { const unsigned char* b = (const unsigned char*)&base(r, 0); unsigned char* o = (unsigned char*)&S[k](r, 0); for (int i = 0; i < nc * 3; i++) o[i] = b[i] ^ m; }
      fill_border_mirror(S[k]);
      (void)D[k].device_desc(true, true);
    }
    device::sync();
    double best = 1e9;
    for (int pass = 0; pass < passes; pass++) {
      const double t0 = now_s();
      for (int k = 0; k < n; k++) pixel_wise(D[k], relative_access(S[k])) | ops::box_mean<5, 5>();   // the reference's loop body
      device::sync();
      best = std::min(best, (now_s() - t0) / n);
    }
    std::printf("4K vuchar3 box 5x5, one frame per call through pixel_wise | ops::box_mean<5,5> over 64 frame sets: %.2f us per frame (%.3f of the HBM peak)\n",
                best * 1e6, 6.0 * nr * nc / best / 8e12);
    CHECK(best * 1e6 < 11.5);   // per-call launches: 11.9-13.5 us by box; the batched rate is 8.3-9.7 us by box (8.91 measured): the bound separates the two on every box
    for (int k = 0; k < n; k += 1) {
      image2d<vuchar3> W(nr, nc, _aligned = 16);
      const vpp_image_desc ds = host_desc(S[k]), dw = host_desc(W);
      CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
      for (int r = 0; r < nr; r++) CHECK(std::memcmp(&D[k](r, 0), &W(r, 0), size_t(nc) * 3) == 0);
    }
  }
  {  // and `A = B + C` on 4K int images (benchmarks/image_add.cc:51-57 as a frame loop), 16 rotating triples
    const int n = 16, nr = 2160, nc = 3840, passes = 8;
    std::vector<image2d<int>> A, B, C;
    for (int k = 0; k < n; k++) {
      A.emplace_back(nr, nc); B.emplace_back(nr, nc); C.emplace_back(nr, nc);
      for (int r = 0; r < nr; r++) { int* b = &B[k](r, 0); int* c = &C[k](r, 0); for (int i = 0; i < nc; i++) { b[i] = (r * 7919 + i * 13 + k) & 0x3fffffff; c[i] = (r * 31 + i * 104729 + 3 * k) & 0x3fffffff; } }
      (void)A[k].device_desc(true, true); (void)B[k].device_desc(false); (void)C[k].device_desc(false);
    }
    device::sync();
    double best = 1e9;
    for (int pass = 0; pass < passes; pass++) {
      const double t0 = now_s();
      for (int k = 0; k < n; k++) pixel_wise(A[k], B[k], C[k]) | ops::add();
      device::sync();
      best = std::min(best, (now_s() - t0) / n);
    }
    std::printf("4K int A = B + C, one triple per call through pixel_wise | ops::add over 16 triples: %.2f us per call (%.3f of the HBM peak)\n", best * 1e6, 12.0 * nr * nc / best / 8e12);
    for (int k = 0; k < n; k++) for (int r = 0; r < nr; r++) { const int *a = &A[k](r, 0), *b = &B[k](r, 0), *c = &C[k](r, 0); for (int i = 0; i < nc; i++) CHECK(a[i] == b[i] + c[i]); }
  }
}

static void test_fast9() {
  image2d<unsigned char> img = texture(240, 320, 0, 0);
  for (int k = 0; k < 60; k++) { int r = rng() % 200, c = rng() % 280, v = rng() & 255; for (int i = 0; i < 20; i++) for (int j = 0; j < 24; j++) img(r + i, c + j) = (unsigned char)v; }
  fill_border_mirror(img);
  const vpp_image_desc d = host_desc(img);
  for (int mode = 0; mode < 3; mode++) {
    std::vector<int> scores;
    std::vector<vint2> kps = mode == 0 ? fast9(img, 20, _scores = &scores) : mode == 1 ? fast9(img, 20, _local_maxima, _scores = &scores) : fast9(img, 20, _blockwise, _block_size = 10, _scores = &scores);
    std::vector<int32_t> rc(400000), sc(200000); int n = 0;
    CHECK(orc_fast9_detect(&d, 20, nullptr, mode, 10, VPP_FAST9_REFERENCE, rc.data(), sc.data(), 200000, &n) == 0);
    CHECK(n > 0 && int(kps.size()) == n);
    for (int i = 0; i < n; i++) CHECK(kps[i][0] == rc[2 * i] && kps[i][1] == rc[2 * i + 1] && scores[i] == sc[i]);
  }
  CHECK(fast9_score(img, 20, vint2(100, 100)) >= 0);
  image2d<unsigned char> small(20, 20, _border = 2);
  bool thrown = false;
  try { fast9(small, 20); } catch (const std::runtime_error& e) { thrown = std::string(e.what()) == "Image need a border of 3px at least for the FAST detector"; }
  CHECK(thrown);                                                             // fast.hpp:937-938
}

static void test_pyrlk() {
  const int nr = 240, nc = 320, L = 3;
  image2d<unsigned char> f1 = texture(nr, nc, 0, 0), f2 = texture(nr, nc, 1.5f, -2.25f);
  pyramid2d<unsigned char> pyr1(f1, L, 2, _border = 5), pyr2(f2, L, 2, _border = 5);   // benchmarks/pyrlk_opencv_comparison.cc:49-60
  pyramid2d<vfloat2> grad(f1.domain(), L, 2, _border = 5);
  scharr(pyr1[0], grad[0]);
  grad.propagate_level0();
  {  // the device-built pyramids equal the oracle's, level by level, borders included
    int r_ = nr, c_ = nc;
    std::vector<image2d<unsigned char>> o1; std::vector<image2d<vfloat2>> og;
    for (int l = 0; l < L; l++) { o1.emplace_back(r_, c_, _border = 5); og.emplace_back(r_, c_, _border = 5); r_ = 1 + r_ / 2; c_ = 1 + c_ / 2; }
    for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) o1[0](r, c) = f1(r, c);
    {  // and the next-frame pyramid
      std::vector<image2d<unsigned char>> o2; int r2 = nr, c2 = nc;
      for (int l = 0; l < L; l++) { o2.emplace_back(r2, c2, _border = 5); r2 = 1 + r2 / 2; c2 = 1 + c2 / 2; }
      for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) o2[0](r, c) = f2(r, c);
      std::vector<vpp_image_desc> O2; for (int l = 0; l < L; l++) O2.push_back(host_desc(o2[l]));
      CHECK(orc_fill_border(&O2[0], 0, nullptr) == 0);
      for (int l = 1; l < L; l++) CHECK(orc_pyr_down(&O2[l], &O2[l - 1]) == 0);
      for (int l = 0; l < L; l++) for (auto p : o2[l].domain_with_border())
        if (!(pyr2[l](p) == o2[l](p))) { std::fprintf(stderr, "pyr2 level %d differs at (%d,%d): %d vs %d\n", l, p[0], p[1], int(pyr2[l](p)), int(o2[l](p))); std::exit(1); }
    }
    std::vector<vpp_image_desc> OP, OG;
    for (int l = 0; l < L; l++) { OP.push_back(host_desc(o1[l])); OG.push_back(host_desc(og[l])); }
    CHECK(orc_fill_border(&OP[0], 0, nullptr) == 0 && orc_scharr(&OG[0], &OP[0]) == 0 && orc_fill_border(&OG[0], 0, nullptr) == 0);
    for (int l = 1; l < L; l++) CHECK(orc_pyr_down(&OP[l], &OP[l - 1]) == 0 && orc_pyr_down(&OG[l], &OG[l - 1]) == 0);
    for (int l = 0; l < L; l++)
      for (auto p : o1[l].domain_with_border()) {
        if (!(pyr1[l](p) == o1[l](p))) { std::fprintf(stderr, "pyr1 level %d differs at (%d,%d): %d vs %d\n", l, p[0], p[1], int(pyr1[l](p)), int(o1[l](p))); std::exit(1); }
        if (!(grad[l](p) == og[l](p))) { std::fprintf(stderr, "grad level %d differs at (%d,%d): (%g,%g) vs (%g,%g)\n", l, p[0], p[1], grad[l](p)[0], grad[l](p)[1], og[l](p)[0], og[l](p)[1]); std::exit(1); }
      }
  }
  pyrlk_keypoint_container kc(f1.domain());
  for (int r = 30; r < nr - 30; r += 11) for (int c = 30; c < nc - 30; c += 13) kc.add(vfloat2(r + 0.25f, c + 0.5f));
  std::vector<vpp_keypoint_f32> want(kc.size());
  for (int i = 0; i < kc.size(); i++) want[i] = vpp_keypoint_f32{kc[i].position[0], kc[i].position[1], 0, 0, 1};
  // oracle on pyramids downloaded from the device (host accessors trigger the download)
  std::vector<vpp_image_desc> P, G, N;
  for (int l = 0; l < L; l++) { P.push_back(host_desc(pyr1[l])); G.push_back(host_desc(grad[l])); N.push_back(host_desc(pyr2[l])); }
  CHECK(orc_pyrlk_match(P.data(), G.data(), N.data(), L, want.data(), int(want.size()), 7, 1e-4f, 500.f, 30, 0.01f, 0, nullptr) == 0);
  pyrlk_match(pyr1, grad, pyr2, kc, lk_match_point_square_win<7>(), 1e-4f, 500.f, 30, 0.01f);
  int alive = 0;
  for (int i = 0; i < kc.size(); i++) {
    CHECK(kc[i].age == want[i].age);
    if (!kc[i].alive()) continue;
    alive++;
    CHECK(std::fabs(kc[i].position[0] - want[i].pos_r) <= 1e-4f * std::fabs(want[i].pos_r) && std::fabs(kc[i].position[1] - want[i].pos_c) <= 1e-4f * std::fabs(want[i].pos_c));
    CHECK(kc.index2d()(cast<vint2>(kc[i].position)) >= 0);
  }
  CHECK(alive > kc.size() * 0.9);  // parity with the oracle is the assertion above; the scene only has to keep most tracks alive
}

// pyrlk_match over several frame pairs in one launch (pyrlk_match.hh: the std::vector overload -> vpp_pyrlk_match_batch): every container against the oracle's
// match of ITS pair (own frames, own translation, own keypoints; one container empty)
static void test_pyrlk_frame_pairs() {
  const int nr = 200, nc = 288, L = 3, F = 4;
  std::vector<pyramid2d<unsigned char>> p1, p2; std::vector<pyramid2d<vfloat2>> gr; std::vector<pyrlk_keypoint_container> kcs;
  std::vector<std::vector<vpp_keypoint_f32>> wants(F);
  for (int f = 0; f < F; f++) {
    image2d<unsigned char> f1 = texture(nr, nc, 0.3f * f, 0.2f * f), f2 = texture(nr, nc, 0.3f * f + 1.5f - 0.5f * f, 0.2f * f - 2.25f + 0.75f * f);
    p1.emplace_back(f1, L, 2, _border = 5); p2.emplace_back(f2, L, 2, _border = 5);
    gr.emplace_back(f1.domain(), L, 2, _border = 5);
    scharr(p1[f][0], gr[f][0]);
    gr[f].propagate_level0();
    kcs.emplace_back(f1.domain());
    if (f != 2) for (int r = 30 + f; r < nr - 30; r += 9 + f) for (int c = 30; c < nc - 30; c += 13 - f) kcs[f].add(vfloat2(r + 0.25f, c + 0.5f));
    wants[f].resize(kcs[f].size());
    for (int i = 0; i < kcs[f].size(); i++) wants[f][i] = vpp_keypoint_f32{kcs[f][i].position[0], kcs[f][i].position[1], 0, 0, 1};
    std::vector<vpp_image_desc> P, G, N;
    for (int l = 0; l < L; l++) { P.push_back(host_desc(p1[f][l])); G.push_back(host_desc(gr[f][l])); N.push_back(host_desc(p2[f][l])); }
    if (!wants[f].empty()) CHECK(orc_pyrlk_match(P.data(), G.data(), N.data(), L, wants[f].data(), int(wants[f].size()), 7, 1e-4f, 500.f, 30, 0.01f, 0, nullptr) == 0);
  }
  pyrlk_match(p1, gr, p2, kcs, lk_match_point_square_win<7>(), 1e-4f, 500.f, 30, 0.01f);
  for (int f = 0; f < F; f++) {
    int alive = 0;
    for (int i = 0; i < kcs[f].size(); i++) {
      CHECK(kcs[f][i].age == wants[f][i].age);
      if (!kcs[f][i].alive()) continue;
      alive++;
      CHECK(kcs[f][i].position[0] == wants[f][i].pos_r && kcs[f][i].position[1] == wants[f][i].pos_c);   // the same float chains as the oracle: bit-identical here
      CHECK(kcs[f].index2d()(cast<vint2>(kcs[f][i].position)) >= 0);
    }
    CHECK(f == 2 ? kcs[f].size() == 0 : alive > kcs[f].size() * 0.8);
  }
}

static void test_lucas_kanade_golden() {  // tests/pyrlk.cc:17-50
  image2d<unsigned char> raw[2] = {image2d<unsigned char>(100, 100), image2d<unsigned char>(100, 100)}, blur[2] = {image2d<unsigned char>(100, 100), image2d<unsigned char>(100, 100)};
  auto gk = [](float s, float* k) { float t = 0; for (int i = 0; i < 9; i++) { k[i] = std::exp(-(i - 4) * (i - 4) / (2 * s * s)); t += k[i]; } for (int i = 0; i < 9; i++) k[i] /= t; };
  float kx[9], ky[9]; gk(3, kx); gk(5, ky);
  for (int f = 0; f < 2; f++) {
    fill(raw[f], 0);
    const int ctr = f ? 52 : 50;
    for (int r = ctr - 2; r <= ctr + 2; r++) for (int c = ctr - 2; c <= ctr + 2; c++) raw[f](r, c) = 255;      // draw::square(_center, _width = 5, _fill = 255)
    auto px = [&](int r, int c) { return float(raw[f](std::min(99, std::max(0, r)), std::min(99, std::max(0, c)))); };  // BORDER_REPLICATE
    image2d<float> h(100, 100);
    for (int r = 0; r < 100; r++) for (int c = 0; c < 100; c++) { double s = 0; for (int i = 0; i < 9; i++) s += kx[i] * px(r, c + i - 4); h(r, c) = float(s); }
    for (int r = 0; r < 100; r++) for (int c = 0; c < 100; c++) { double s = 0; for (int i = 0; i < 9; i++) s += ky[i] * h(std::min(99, std::max(0, r + i - 4)), c); blur[f](r, c) = (unsigned char)std::lround(s); }
  }
  std::vector<vfloat2> keypoints; keypoints.push_back(vfloat2(50, 50));
  int calls = 0;
  lucas_kanade(blur[0], blur[1], _keypoints = keypoints, _niterations = 50, _winsize = 5, _min_ev = 0.001, _delta = 0.01, _nscales = 2,
               _flow = [&](vfloat2 p, vfloat2 f, int) { CHECK(p == vfloat2(50.f, 50.f)); CHECK((f - vfloat2(2.f, 2.f)).norm() < 0.05); calls++; });
  CHECK(calls == 1);
}

static void test_sdof_and_video_extruder() {
  const int nr = 240, nc = 320;
  image2d<unsigned char> f1 = texture(nr, nc, 0, 0), f2 = texture(nr, nc, 3.f, -2.f);
  std::vector<vint2> kps;
  for (int r = 10; r < nr - 10; r += 5) for (int c = 10; c < nc - 10; c += 5) kps.push_back(vint2(r, c));
  std::vector<vint2> got(kps.size(), vint2(-1, -1)); std::vector<int> gd(kps.size(), -1);
  semi_dense_optical_flow(kps, [&](int i, vint2 pos, int d) { got[i] = pos; gd[i] = d; }, f1, f2, _winsize = 9, _patchsize = 5, _propagation = 2, _nscales = 3);
  std::vector<int32_t> wp(kps.size() * 2), wd(kps.size()); std::vector<uint8_t> wv(kps.size());
  const vpp_image_desc d1 = host_desc(f1), d2 = host_desc(f2);
  CHECK(orc_semi_dense_optical_flow(&d1, &d2, (const int32_t*)kps.data(), int(kps.size()), 9, 3, 0, 2, 5, wp.data(), wd.data(), wv.data()) == 0);
  int moved = 0;
  for (size_t i = 0; i < kps.size(); i++) {
    if (wv[i]) { CHECK(got[i] == vint2(wp[2 * i], wp[2 * i + 1]) && gd[i] == wd[i]); moved += (got[i] - kps[i]) == vint2(3, -2); }
    else CHECK(gd[i] == -1);
  }
  CHECK(moved > int(kps.size()) * 0.8);

  // video_extruder over a short synthetic sequence translating by (1,2) px / frame (video_extruder.hpp:24-135)
  video_extruder_ctx ctx = video_extruder_init(make_box2d(nr, nc));
  image2d<unsigned char> prev = texture(nr, nc, 0, 0);
  fill_border_mirror(prev);
  for (int t = 1; t <= 6; t++) {
    image2d<unsigned char> next = texture(nr, nc, 1.f * t, 2.f * t);
    fill_border_mirror(next);
    video_extruder_update(ctx, prev, next, _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _nscales = 3, _winsize = 9);
    prev = next;
  }
  CHECK(ctx.frame_id == 5 && ctx.keypoints.size() > 50 && ctx.trajectories.size() == size_t(ctx.keypoints.size()));
  int good = 0, alive = 0;
  for (int i = 0; i < ctx.keypoints.size(); i++) {
    if (!ctx.keypoints[i].alive() || ctx.trajectories[i].size() < 3) continue;
    alive++;
    good += ctx.keypoints[i].velocity == vint2(1, 2);
  }
  CHECK(alive > 30 && good > alive * 0.8);

  // the same sequence with the views handed out NON-CONST between the updates (what `draw::draw_trajectories(display, ctx.trajectories, 200)` does
  // in examples/video_extruder.cc:57) but not changed: the update compares the host copies with what it downloaded and leaves the device state
  // alone — the run ends in exactly the same state; then a real edit (a keypoint removed on the host) does reach the device
  video_extruder_ctx ctx2 = video_extruder_init(make_box2d(nr, nc));
  prev = texture(nr, nc, 0, 0);
  fill_border_mirror(prev);
  for (int t = 1; t <= 6; t++) {
    image2d<unsigned char> next = texture(nr, nc, 1.f * t, 2.f * t);
    fill_border_mirror(next);
    video_extruder_update(ctx2, prev, next, _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _nscales = 3, _winsize = 9);
    std::vector<keypoint_trajectory>& tr = ctx2.trajectories;   // non-const view, nothing written
    CHECK(tr.size() == size_t(ctx2.keypoints.size()));
    prev = next;
  }
  CHECK(ctx2.frame_id == ctx.frame_id && ctx2.keypoints.size() == ctx.keypoints.size());
  for (int i = 0; i < ctx.keypoints.size(); i++) {
    CHECK(ctx2.keypoints[i].position == ctx.keypoints[i].position && ctx2.keypoints[i].velocity == ctx.keypoints[i].velocity && ctx2.keypoints[i].age == ctx.keypoints[i].age);
    CHECK(ctx2.trajectories[i].size() == ctx.trajectories[i].size() && ctx2.trajectories[i].alive() == ctx.trajectories[i].alive());
    for (int k = 0; k < ctx.trajectories[i].size(); k++) CHECK(ctx2.trajectories[i][k] == ctx.trajectories[i][k]);
  }
  int victim = -1;
  for (int i = 0; i < ctx2.keypoints.size() && victim < 0; i++) if (ctx2.keypoints[i].age >= 3 && ctx2.keypoints[i].velocity == vint2(1, 2)) victim = i;
  CHECK(victim >= 0);
  ctx2.keypoints->remove(victim);   // keypoint_container::remove (keypoint_container.hpp:133-140): age = 0
  {
    image2d<unsigned char> next = texture(nr, nc, 7.f, 14.f);
    fill_border_mirror(next);
    video_extruder_update(ctx2, prev, next, _detector_th = 10, _keypoint_spacing = 10, _detector_period = 50, _nscales = 3, _winsize = 9);
  }
  // the flow runs over every container entry and its callback moves the matched ones, dead or not (video_extruder.hpp:44-53, keypoint_container.hpp:
  // 143-156: age++): the removed keypoint is matched again and comes back with age 0 + 1 — had the edit not reached the device it would be >= 4
  CHECK(ctx2.keypoints[victim].age == 1);
}

#ifdef HAVE_VPP_REF
// the REAL reference (oracle/_ref: matt-42/vpp's own headers) run over the same frame sequence
extern "C" int ref_video_extruder_run(const vpp_image_desc* frames, int nframes, int detector_th, int keypoint_spacing, int detector_period,
                                      int max_trajectory_length, int nscales, int winsize, int propagation, int32_t* out, int32_t* traj_len,
                                      int capacity, int* count, int* frame_id);
static void test_video_extruder_vs_reference() {
  const int nr = 240, nc = 320, T = 9;
  std::vector<image2d<unsigned char>> frames;
  for (int t = 0; t < T; t++) {
    image2d<unsigned char> f = texture(nr, nc, 1.f * t, -2.f * t);
    for (int k = 0; k < 25; k++) {  // moving high-contrast rectangles: plenty of FAST corners
      const int r = 20 + (k * 37) % 180 + t, c = 30 + (k * 53) % 250 - 2 * t, v = (k * 71) & 255;
      for (int i = 0; i < 14; i++) for (int j = 0; j < 18; j++) f(r + i, c + j) = (unsigned char)v;
    }
    fill_border_mirror(f);
    frames.push_back(f);
  }
  std::vector<vpp_image_desc> descs;
  for (auto& f : frames) descs.push_back(host_desc(f));  // host pointers (the accessor downloads the mirrored border)
  std::vector<int32_t> want(5 * 100000), wlen(100000); int wn = 0, wfid = 0;
  std::fprintf(stderr, "video_extruder: calling the reference\n");
  CHECK(ref_video_extruder_run(descs.data(), T, 10, 10, 5, 15, 3, 9, 2, want.data(), wlen.data(), 100000, &wn, &wfid) == 0);
  std::fprintf(stderr, "video_extruder: reference done (%d keypoints)\n", wn);
  video_extruder_ctx ctx = video_extruder_init(make_box2d(nr, nc));
  for (int t = 1; t < T; t++)
    video_extruder_update(ctx, frames[t - 1], frames[t], _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15,
                          _nscales = 3, _winsize = 9, _propagation = 2);
  std::fprintf(stderr, "video_extruder: %d keypoints (reference %d), frame_id %d\n", ctx.keypoints.size(), wn, ctx.frame_id);
  CHECK(ctx.frame_id == wfid && ctx.keypoints.size() == wn && wn > 100);
  int alive = 0;
  for (int i = 0; i < wn; i++) {
    const auto& k = ctx.keypoints[i];
    CHECK(k.position[0] == want[5 * i] && k.position[1] == want[5 * i + 1] && k.velocity[0] == want[5 * i + 2] && k.velocity[1] == want[5 * i + 3] && k.age == want[5 * i + 4]);
    CHECK(ctx.trajectories[i].size() == wlen[i]);
    alive += k.alive();
  }
  CHECK(alive > 50);
}
#endif

static void test_frame_ingest() {
  // examples/video_extruder.cc:46-48: clone with border 3, mirror, gray — stepwise (device kernels) and fused
  image2d<vuchar3> frame(135, 241);
  for (auto p : frame.domain()) frame(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
  auto f = clone(frame, _border = 3);
  fill_border_mirror(f);
  image2d<unsigned char> stepwise = rgb_to_graylevel<unsigned char>(f);      // vpp_rgb_to_graylevel(mirror = 0)
  image2d<unsigned char> fused = rgb_to_graylevel_mirror(frame, 3);          // vpp_rgb_to_graylevel(mirror = 1)
  image2d<unsigned char> want(135, 241, _border = 3);
  const vpp_image_desc dw = host_desc(want), dfr = host_desc(frame);
  CHECK(orc_rgb_to_graylevel(&dw, &dfr, 1) == 0);
  for (int r = -3; r < 138; r++) for (int c = -3; c < 244; c++) CHECK(stepwise(r, c) == want(r, c) && fused(r, c) == want(r, c));
  // the same pixels from the host expression engine on the reference's lambda
  image2d<unsigned char> host(f.domain(), _border = 3);
  pixel_wise(f.domain_with_border(), f, host) | [](vint2, const vuchar3& i, unsigned char& o) { o = (i[0] + i[1] + i[2]) / 3; };
  for (int r = -3; r < 138; r++) for (int c = -3; c < 244; c++) CHECK(host(r, c) == want(r, c));
}

static void test_lbp() {                                                    // tests/lbp.cc
  image2d<unsigned char> V(3, 3, _border = 1), lbp(3, 3);
  V(1, 1) = 1; V(0, 0) = 0; V(0, 1) = 2; V(0, 2) = 2; V(1, 0) = 2; V(1, 2) = 0; V(2, 0) = 2; V(2, 1) = 0; V(2, 2) = 2;
  lbp_transform(V, lbp);
  CHECK(lbp(1, 1) == 0b10101110);
  CHECK(lbp_hamming_distance(0b01010101, 0b01010101) == 0 && lbp_hamming_distance(0b11010101, 0b01010101) == 1 && lbp_hamming_distance(0b11111111, 0b00000000) == 8);
}

// FAST_internals::fast_detector9(A, B, th) (fast.hpp:511-551) and blockwise_maxima_filter (fast.hpp:577-614) through the headers
static void test_dense_fast_and_blockwise_maxima() {
  image2d<unsigned char> A(40, 50, _border = 3);
  fill_with_border(A, (unsigned char)10);
  for (int r = 10; r < 30; r++) for (int c = 15; c < 35; c++) A(r, c) = 200;   // a bright square: its four corners are FAST-9 corners
  image2d<unsigned char> B(A.domain());
  FAST_internals::fast_detector9(A, B, 20);
  CHECK(B(10, 15) == 1 && B(29, 34) == 1 && B(10, 34) == 1 && B(29, 15) == 1);
  CHECK(B(20, 25) == 0 && B(0, 0) == 0 && B(10, 25) == 0);                     // inside, far away, on a straight edge
  auto kps = fast9(A, 20, _fast9_corrected_ring);                             // the sparse detector on the same (true) ring agrees
  int flagged = 0; for (auto p : B.domain()) flagged += B(p);
  CHECK(flagged == int(kps.size()));
  for (auto p : kps) CHECK(B(p) == 1);
  image2d<int> S(20, 20);
  fill(S, 0);
  S(3, 4) = 7; S(5, 5) = 7; S(2, 2) = 3; S(12, 13) = 2; S(15, 3) = -4;
  blockwise_maxima_filter(S, 10);
  CHECK(S(3, 4) == 7 && S(5, 5) == 0 && S(2, 2) == 0 && S(12, 13) == 2 && S(15, 3) == 0);
  // the same per-block step as a block_wise expression (block_wise.hh:26-56 + the tagged functor), against the oracle
  image2d<int> T(37, 53), W(37, 53);
  for (auto p : T.domain()) T(p) = W(p) = (rng() % 5 == 0) ? int(rng() % 50) : 0;
  block_wise(vint2(7, 7), T) | ops::block_maxima();
  const vpp_image_desc dw = host_desc(W);
  CHECK(orc_blockwise_maxima_filter(&dw, 7) == 0);
  for (auto p : T.domain()) CHECK(T(p) == W(p));
}

// antialiasing_lowpass_filter / subsample2 / antialias_subsample2 (pyramid.hh:12-123) against the fused pyramid step
static void test_pyramid_free_functions() {
  image2d<unsigned char> in(37, 52, _border = 2);
  unsigned x = 7u;
  for (auto p : in.domain()) { x = x * 1664525u + 1013904223u; in(p) = (unsigned char)(x >> 24); }
  fill_border_mirror(in);
  pyramid2d<unsigned char> pyr(in, 2, 2, _border = 2);
  image2d<unsigned char> low(in.domain(), _border = 1), half(1 + 37 / 2, 1 + 52 / 2);  // subsample2 reads low(2r, 2c) up to (36, 52): one column into the border, like the reference
  antialiasing_lowpass_filter(in, low);
  subsample2(low, half);   // reads low(2r, 2c): inside the domain for 2r < 37, 2c < 52
  image2d<unsigned char> as2 = antialias_subsample2(in);
  CHECK(as2.nrows() == 19 && as2.ncols() == 27 && as2.border() == 2);
  for (int r = 0; 2 * r < 37; r++) for (int c = 0; 2 * c < 52; c++) { CHECK(half(r, c) == pyr[1](r, c)); CHECK(as2(r, c) == pyr[1](r, c)); }
}

int main() {
  CHECK(vpp_init(0) == 0);
  test_pyramid_free_functions();
  test_dense_fast_and_blockwise_maxima();
  test_lbp();
  test_frame_ingest();
  test_pixel_wise_functors();
  test_frame_stacks();
  test_deferred_per_frame_calls();
  test_fast9();
  test_pyrlk();
  test_pyrlk_frame_pairs();
  test_lucas_kanade_golden();
  test_sdof_and_video_extruder();
#ifdef HAVE_VPP_REF
  test_video_extruder_vs_reference();
#endif
  std::puts("device_api_test ok");
  return 0;
}
