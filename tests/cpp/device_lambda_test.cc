// Single-source build (hipcc -x hip -DVPP_AMD_DEVICE -DVPP_AMD_HIPCC): opaque pixel_wise lambdas — the reference's own call
// form — are the body of the generic gfx950 kernel of vpp/core/pixel_wise_device.hh.  Every case is evaluated twice, on the GPU
// and (same lambda, `_host`) by the host engine of the same headers, and compared bit for bit; the two benchmark bodies are
// pasted unmodified from the reference (benchmarks/image_add.cc:51-57, benchmarks/box_5x5_filter2.cc:71-81).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <vpp/vpp.hh>
#include "../../oracle/oracle.h"   // the CPU oracle: the checker of the 5 x 5 cases below (test infrastructure, never the product path)

using namespace vpp;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static std::mt19937 rng(11);

// ---- benchmarks/image_add.cc:51-57, verbatim ----
void vpp_pixel_wise(image2d<int> A, image2d<int> B, image2d<int> C)
{
  vpp::pixel_wise(A, B, C) | [] (int& a, int& b, int& c)
  {
    a = b + c;
  };
}

// ---- benchmarks/box_5x5_filter2.cc:71-81, verbatim ----
void vpp_pixel_wise(image2d<int> B, image2d<int> A)
{
  vpp::pixel_wise(B, relative_access(A)) | [&] (int& b, auto a)
  {
    int sum = 0;
    for (int i = -2; i <= 2; i++)
    for (int j = -2; j <= 2; j++)
      sum += a(i, j);
    b = sum / 25;
  };
}

template <class V> static bool same_pixels(const image2d<V>& a, const image2d<V>& b) {
  if (!(a.domain() == b.domain())) return false;
  for (int r = 0; r < a.nrows(); r++)
    if (std::memcmp(&a(r, 0), &b(r, 0), size_t(a.ncols()) * sizeof(V))) return false;
  return true;
}

static void test_reference_bodies() {
  for (auto shape : {std::pair<int, int>{1080, 1920}, {37, 61}, {5, 3}, {1, 1}, {64, 4}}) {   // BASELINE configs[0] + ragged / tiny
    image2d<int> A(shape.first, shape.second), B(A.domain()), C(A.domain());
    for (auto p : B.domain()) { B(p) = int(rng() >> 2); C(p) = int(rng() >> 2); }
    fill(A, 0);
    vpp_pixel_wise(A, B, C);
    for (auto p : A.domain()) CHECK(A(p) == B(p) + C(p));                 // benchmarks/image_add.cc:21-28
  }
  for (auto shape : {std::pair<int, int>{270, 480}, {33, 67}, {3, 2}}) {
    image2d<int> A(shape.first, shape.second, _border = 2), B(shape.first, shape.second, _border = 2);
    for (auto p : A.domain_with_border()) A(p) = int(rng() % 1000);       // box_5x5_filter.cc:187-191 value range
    fill(B, 2);
    vpp_pixel_wise(B, A);
    for (auto p : B.domain()) {                                           // benchmarks/box_5x5_filter2.cc:26-41
      int sum = 0;
      for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) sum += A(p[0] + i, p[1] + j);
      CHECK(B(p) == sum / 25);
    }
  }
}

static void test_device_equals_host() {
  // vuchar3 5x5 mean, examples/box_filter.cc:23-32 (accumulate as vint3, divide, cast back)
  {
    image2d<vuchar3> S(131, 203, _border = 2), D(S.domain()), H(S.domain());
    for (auto p : S.domain_with_border()) S(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    auto k = [] (vuchar3& out, auto nbh) {
      vint3 sum = vint3::Zero();
      for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) sum += nbh(i, j).template cast<int>();
      out = (sum / 25).template cast<unsigned char>();
    };
    pixel_wise(D, relative_access(S)) | k;
    pixel_wise(H, relative_access(S))(_host) | k;
    CHECK(same_pixels(D, H));
  }
  // value-returning kernel builds the output image (pixel_wise.hpp:196-211); mixed element sizes (uchar in, int out)
  {
    image2d<unsigned char> U(45, 77);
    for (auto p : U.domain()) U(p) = (unsigned char)(rng() & 255);
    auto k = [] (unsigned char& u) { return int(u) * 3 - 7; };
    image2d<int> D = pixel_wise(U) | k;
    image2d<int> H = pixel_wise(U)(_host) | k;
    CHECK(D.nrows() == 45 && D.ncols() == 77 && same_pixels(D, H));
  }
  // a box2d range hands the coordinates over (tests/pixel_wise.cc:13-27)
  {
    image2d<int> D(31, 50), H(31, 50);
    auto k = [] (vint2 p, int& v) { v = p[0] * 1000 + p[1]; };
    pixel_wise(D.domain(), D) | k;
    pixel_wise(H.domain(), H)(_host) | k;
    CHECK(same_pixels(D, H));
    CHECK(D(30, 49) == 30049);
  }
  // read-modify-write through the reference, float pixels, conditional writes
  {
    image2d<float> A(67, 129), B(A.domain()), A2(A.domain());
    for (auto p : A.domain()) { A(p) = A2(p) = float(rng() % 2000) / 7.f - 100.f; B(p) = float(rng() % 100); }
    auto k = [] (float& a, float& b) { if (a > 0.f) a += b * 0.5f; };
    pixel_wise(A, B) | k;
    pixel_wise(A2, B)(_host) | k;
    CHECK(same_pixels(A, A2));
  }
  // a sub-image view at an odd column is unaligned: one pixel per lane, same result
  {
    image2d<int> A(40, 60), B(40, 60), A2(40, 60);
    for (auto p : B.domain()) B(p) = int(rng() & 0xffff);
    fill(A, -1); fill(A2, -1);
    const box2d win(vint2(3, 5), vint2(30, 51));
    auto sa = A | win, sb = B | win, sa2 = A2 | win;
    auto k = [] (int& a, int& b) { a = b ^ 0x55; };
    pixel_wise(sa, sb) | k;
    pixel_wise(sa2, sb)(_host) | k;
    CHECK(same_pixels(A, A2));
    CHECK(A(2, 5) == -1 && A(3, 4) == -1 && A(3, 5) == (B(3, 5) ^ 0x55));   // nothing outside the window was written
  }
  // legacy box_nbh2d spelling (tests/box_nbh2d.cc:9-18)
  {
    image2d<int> S(20, 30, _border = 1), D(S.domain()), H(S.domain());
    for (auto p : S.domain_with_border()) S(p) = int(rng() % 100);
    auto k = [] (int& o, auto n) { o = n.north() + n.south() + n.east() + n.west() - 4 * n(0, 0); };
    pixel_wise(D, box_nbh2d<int, 3, 3>(S)) | k;
    pixel_wise(H, box_nbh2d<int, 3, 3>(S))(_host) | k;
    CHECK(same_pixels(D, H));
  }
  // a by-value capture needs the `_device` opt-in (a by-reference capture would hold a host address); without it: host
  {
    image2d<int> A(10, 17), A2(10, 17);
    fill(A, 1); fill(A2, 1);
    const int bias = 41;
    pixel_wise(A)(_device) | [=] (int& a) { a += bias; };
    pixel_wise(A2) | [=] (int& a) { a += bias; };
    CHECK(same_pixels(A, A2) && A(9, 16) == 42);
  }
  // aliasing ranges (the same image twice) take the host route: in-place prefix dependences keep the reference's semantics
  {
    image2d<int> A(8, 8);
    fill(A, 1);
    pixel_wise(A, A) | [] (int& a, int& b) { a = b + 1; };
    for (auto p : A.domain()) CHECK(A(p) == 2);
  }
}

// block_wise on the device: one lane per block, the callable sees views (tests/block_wise.cc:101-114 restated on views)
static void test_block_wise_device() {
  {
    image2d<int> img(10, 10, _border = 1), ref(10, 10, _border = 1);
    for (auto* im : {&img, &ref}) { fill_border_with_value(*im, 2); fill(*im, 0); }
    auto k = [] (auto si) { for (int r = 0; r < si.nrows(); r++) for (int c = 0; c < si.ncols(); c++) si(r, c) = 1 + si.nrows() * 10 + si.ncols(); };
    block_wise(vint2(3, 3), img) | k;                                   // device: pwdev::block_view<int>
    block_wise(vint2(3, 3), ref)(_host) | k;                            // host: image2d<int> sub-images
    for (auto p : img.domain_with_border()) CHECK(img(p) == ref(p));    // blocks cover the image, clip at its edge, never touch the border
    CHECK(img(0, 0) == 34 && img(9, 9) == 12 && img(-1, -1) == 2);
  }
  {  // two ranges + a box range: per-block sum of one image into the block's first pixel of another, block origin from the box
    image2d<int> A(37, 53), S(37, 53), S2(37, 53);
    for (auto p : A.domain()) A(p) = int(rng() % 100);
    fill(S, 0); fill(S2, 0);
    auto k = [] (auto b, auto a, auto s) {
      int sum = 0;
      for (int r = 0; r < a.nrows(); r++) for (int c = 0; c < a.ncols(); c++) sum += a(r, c);
      s(0, 0) = sum + b.p1()[0] * 7 + b.p1()[1];
    };
    block_wise(vint2(8, 16), A.domain(), A, S) | k;
    block_wise(vint2(8, 16), A.domain(), A, S2)(_host) | k;
    CHECK(same_pixels(S, S2));
  }
}

// neighbourhoods out of the LDS tile (pixel_wise_device.hh: pixel_wise_tile_kernel, taken under `_nbh_read_only` only): every reach the tile covers (1..4: windows
// up to 9 x 9 read the whole halo) with borders narrower, equal and wider than the reach, and one reach it must leave to the global taps (5: no option), shapes cut
// by the tile edges (16 rows x 64 or 256 pixels), 1-, 3-, 4- and 8-byte pixels, the legacy box_nbh2d spelling, and a sub-image view (unaligned: one pixel per lane)
// — each against the host engine of the same headers AND the global-tap device kernel, bit for bit.
template <class V, class MK> static void tile_case(int nr, int nc, int border, MK make) {
  image2d<V> S(nr, nc, _border = border), D(S.domain()), G(S.domain()), H(S.domain());
  for (auto p : S.domain_with_border()) S(p) = make();
  auto k = [border] (V& out, auto nbh) {   // the extreme taps of the widest legal window + the centre: position-sensitive, so a misplaced tile shows
    out = nbh(0, 0);
    for (int i = -border; i <= border; i += border) for (int j = -border; j <= border; j += border) out = V(out + nbh(i, j) * (i * 3 + j + 7));
  };
  pixel_wise(G, relative_access(S))(_device) | k;                                       // global taps
  if (border <= 4) pixel_wise(D, relative_access(S))(_device, _nbh_read_only) | k;      // LDS tile (byte-sized pixel types; the others keep the global taps)
  else pixel_wise(D, relative_access(S))(_device) | k;
  pixel_wise(H, relative_access(S))(_host) | k;
  if (!same_pixels(D, H) || !same_pixels(G, H)) { std::fprintf(stderr, "tile_case %d x %d border %d, %d-byte pixels\n", nr, nc, border, (int)sizeof(V)); std::exit(1); }
}
// The halo, not the border, bounds what the tile serves: taps two columns away inside a border of 6, and inside a border of 1 on the interior columns of the
// domain (a row tap may never leave the border — the neighbourhood walks the row-pointer table, relative_accessor.hh:18-22 — a column tap may reach whatever the
// row holds), on a view whose first column is odd.
template <class V, class MK> static void tile_reach_case(int nr, int nc, int border, MK make) {
  image2d<V> S(nr, nc, _border = border), D(S.domain()), H(S.domain());
  for (auto p : S.domain_with_border()) S(p) = make();
  fill(D, V(make())); copy(D, H);
  const int m = border >= 2 ? 0 : 3;   // border narrower than the column reach: the columns where every tap stays inside the row
  const box2d win(vint2(0, m), vint2(nr - 1, nc - 1 - m));
  auto k = [] (V& out, auto nbh) { out = V(nbh(-1, -2) + nbh(1, 2) * 3 + nbh(0, -2) * 5 + nbh(-1, 1) * 7); };
  auto sd = D | win, sh = H | win, ss = S | win;
  pixel_wise(sd, relative_access(ss))(_nbh_read_only) | k;
  pixel_wise(sh, relative_access(ss))(_host) | k;
  if (!same_pixels(D, H)) { std::fprintf(stderr, "tile_reach_case %d x %d border %d, %d-byte pixels\n", nr, nc, border, (int)sizeof(V)); std::exit(1); }
}
static void test_neighbourhood_tiles() {
  for (int border : {1, 2, 3, 4, 5})
    for (auto shape : {std::pair<int, int>{16, 256}, {17, 259}, {50, 1030}, {3, 5}, {33, 64}}) {
      tile_case<unsigned char>(shape.first, shape.second, border, [] { return (unsigned char)(rng() & 255); });
      tile_case<int>(shape.first, shape.second, border, [] { return int(rng() % 1000); });
      tile_case<vuchar3>(shape.first, shape.second, border, [] { return vuchar3(rng() & 255, rng() & 255, rng() & 255); });
      tile_case<vfloat2>(shape.first, shape.second, border, [] { return vfloat2(float(rng() % 512), float(rng() % 64)); });
    }
  for (int border : {1, 6})
    for (auto shape : {std::pair<int, int>{48, 512}, {21, 131}}) {
      tile_reach_case<unsigned char>(shape.first, shape.second, border, [] { return (unsigned char)(rng() & 255); });
      tile_reach_case<vuchar3>(shape.first, shape.second, border, [] { return vuchar3(rng() & 255, rng() & 255, rng() & 255); });
    }
  {  // full 5 x 5 sums through the legacy accessor, and on a sub-image whose first column is odd (unaligned view)
    image2d<int> S(70, 300, _border = 2), D(S.domain()), H(S.domain());
    for (auto p : S.domain_with_border()) S(p) = int(rng() % 1000);
    auto k = [] (int& out, auto nbh) { int s = 0; nbh.for_all([&s] (int v) { s += v; }); out = s / 25; };
    pixel_wise(D, box_nbh2d<int, 5, 5>(S)) | k;
    pixel_wise(H, box_nbh2d<int, 5, 5>(S))(_host) | k;
    CHECK(same_pixels(D, H));
    image2d<int> D2(S.domain()), H2(S.domain());
    fill(D2, -1); fill(H2, -1);
    const box2d win(vint2(5, 7), vint2(60, 290));
    auto sd = D2 | win, sh = H2 | win, ss = S | win;
    auto k2 = [] (int& out, auto nbh) { out = nbh(-2, -2) + 2 * nbh(2, 2) - nbh(0, 1); };
    pixel_wise(sd, relative_access(ss)) | k2;
    pixel_wise(sh, relative_access(ss))(_host) | k2;
    CHECK(same_pixels(D2, H2));
  }
}

// A neighbourhood is a reference into the image (relative_accessor.hh:26-33 returns V&; distance_transforms.hh:33-59 writes `sn(0, 0) = ...`): a write through
// it must reach the image on the device exactly as on the host — also for the byte-sized pixel types, which round 4 silently served from a read-only LDS copy
// (advisor finding).  Only the pixel's own position is written, so the parallel evaluation is race-free.
static void test_writes_through_a_neighbourhood() {
  for (auto shape : {std::pair<int, int>{40, 300}, {17, 67}}) {
    {
      image2d<unsigned char> A(shape.first, shape.second, _border = 2), B(shape.first, shape.second, _border = 2);
      for (auto p : A.domain_with_border()) A(p) = (unsigned char)(rng() & 255);
      copy(A, B);
      auto k = [] (auto n) { n(0, 0) = (unsigned char)(n(0, 0) / 2 + 3); };
      pixel_wise(relative_access(A)) | k;
      pixel_wise(relative_access(B))(_host) | k;
      CHECK(same_pixels(A, B));
      CHECK(A(1, 1) == B(1, 1));
    }
    {
      image2d<vuchar3> A(shape.first, shape.second, _border = 2), B(shape.first, shape.second, _border = 2), O(shape.first, shape.second), O2(shape.first, shape.second);
      for (auto p : A.domain_with_border()) A(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
      copy(A, B);
      auto k = [] (vuchar3& o, auto n) { o = n(0, 0); n(0, 0) = vuchar3(o[2], o[0], o[1]); };
      pixel_wise(O, relative_access(A)) | k;
      pixel_wise(O2, relative_access(B))(_host) | k;
      CHECK(same_pixels(A, B) && same_pixels(O, O2));
    }
    {
      image2d<unsigned char> A(shape.first, shape.second, _border = 1), B(shape.first, shape.second, _border = 1);
      for (auto p : A.domain_with_border()) A(p) = (unsigned char)(rng() & 255);
      copy(A, B);
      auto k = [] (auto n) { n(0, 0) = (unsigned char)(255 - n(0, 0)); };
      pixel_wise(box_nbh2d<unsigned char, 3, 3>(A)) | k;
      pixel_wise(box_nbh2d<unsigned char, 3, 3>(B))(_host) | k;
      CHECK(same_pixels(A, B));
    }
  }
}

// The 5 x 5 mean lambdas (benchmarks/box_5x5_filter2.cc:71-81 on int, examples/box_filter.cc:23-32 on vuchar3) against the CPU ORACLE's box filter — not only
// against this product's own host engine: global-tap kernel and LDS-tile kernel, 4K and a ragged small shape.
template <class V> static vpp_image_desc host_desc_of(image2d<V>& im) { (void)im(0, 0); return im.host_desc(); }   // (the host accessor brings the host pixels up to date)
// The register window (4-byte pixels under `_nbh_read_only`): constant-trip window bodies (taps index registers), the widest reach it serves (9 x 9), float pixels,
// ragged widths (the row's last chunk pixel by pixel), heights cut by the 8-row marches and the 32-row workgroups, a view at a 4-pixel-aligned column and one at an
// odd column (unaligned: falls back to the global taps) — against the host engine of the same headers.
template <class V, class MK> static void window_case(int nr, int nc, MK make) {
  image2d<V> S(nr, nc, _border = 4), D(S.domain()), E(S.domain()), H(S.domain()), G(S.domain());
  for (auto p : S.domain_with_border()) S(p) = make();
  auto k5 = [] (V& out, auto nbh) { V s = V(0); for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) s = V(s + nbh(i, j) * V(i * 5 + j + 13)); out = s; };
  auto k9 = [] (V& out, auto nbh) { V s = V(0); for (int i = -4; i <= 4; i += 2) for (int j = -4; j <= 4; j++) s = V(s + nbh(i, j) * V(i * 9 + j + 41)); out = s; };
  pixel_wise(D, relative_access(S))(_nbh_read_only) | k5;
  pixel_wise(H, relative_access(S))(_host) | k5;
  if (!same_pixels(D, H)) { std::fprintf(stderr, "window_case 5x5 %d x %d\n", nr, nc); std::exit(1); }
  pixel_wise(E, relative_access(S))(_nbh_read_only) | k9;
  pixel_wise(G, relative_access(S))(_host) | k9;
  if (!same_pixels(E, G)) { std::fprintf(stderr, "window_case 9x9 %d x %d\n", nr, nc); std::exit(1); }
  if (nr > 20 && nc > 40) {
    for (int c0 : {8, 7}) {
      image2d<V> D2(S.domain()), H2(S.domain());
      fill(D2, V(make())); copy(D2, H2);
      const box2d win(vint2(3, c0), vint2(nr - 4, nc - 6));
      auto sd = D2 | win, sh = H2 | win, ss = S | win;
      pixel_wise(sd, relative_access(ss))(_nbh_read_only) | k5;
      pixel_wise(sh, relative_access(ss))(_host) | k5;
      if (!same_pixels(D2, H2)) { std::fprintf(stderr, "window_case view at column %d, %d x %d\n", c0, nr, nc); std::exit(1); }
    }
  }
}
static void test_register_window() {
  for (auto shape : {std::pair<int, int>{64, 256}, {65, 257}, {33, 1030}, {7, 5}, {1, 1}, {8, 4}, {100, 300}, {31, 255}}) {
    window_case<int>(shape.first, shape.second, [] { return int(rng() % 1000); });
    window_case<float>(shape.first, shape.second, [] { return float(rng() % 512); });
    window_case<unsigned int>(shape.first, shape.second, [] { return (unsigned int)(rng() % 4096); });
  }
}

static void test_box_lambdas_against_the_oracle() {
  for (auto shape : {std::pair<int, int>{2160, 3840}, {37, 61}}) {
    {
      image2d<vuchar3> S(shape.first, shape.second, _border = 2), D(S.domain()), T(S.domain()), W(S.domain());
      for (auto p : S.domain_with_border()) S(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
      auto k3 = [] (vuchar3& out, auto nbh) {
        vint3 sum = vint3::Zero();
        for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) sum += nbh(i, j).template cast<int>();
        out = (sum / 25).template cast<unsigned char>();
      };
      pixel_wise(D, relative_access(S)) | k3;                       // global taps
      pixel_wise(T, relative_access(S))(_nbh_read_only) | k3;       // LDS tile
      vpp_image_desc ds = host_desc_of(S), dw = host_desc_of(W);
      CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
      CHECK(same_pixels(D, W) && same_pixels(T, W));
    }
    {
      image2d<int> S(shape.first, shape.second, _border = 2), D(S.domain()), T(S.domain()), W(S.domain());
      for (auto p : S.domain_with_border()) S(p) = int(rng() % 1000);
      vpp_pixel_wise(D, S);                                         // the reference's benchmark body, verbatim (global taps)
      pixel_wise(T, box_nbh2d<int, 5, 5>(S))(_nbh_read_only) | [] (int& out, auto nbh) { int s = 0; nbh.for_all([&s] (int v) { s += v; }); out = s / 25; };
      vpp_image_desc ds = host_desc_of(S), dw = host_desc_of(W);
      CHECK(orc_box_filter(&dw, &ds, 5, 5) == 0);
      CHECK(same_pixels(D, W) && same_pixels(T, W));
    }
  }
}

// Stream order between the held-back tagged functors and the kernels this translation unit launches itself (advisor, round 5): with every mirror already resident
// no ABI call sits between `pixel_wise | ops::add()` (held back in the thread's window) and the lambda's own hipLaunchKernelGGL, so the lambda used to run BEFORE the
// add whose result it reads.  Both directions, pixel_wise and block_wise, with values that differ per call (a stale read cannot pass), and the `_immediate` option.
static void test_held_back_calls_and_lambdas_keep_their_order() {
  const int nr = 300, nc = 512;
  image2d<int> A(nr, nc), B(A.domain()), C(A.domain()), D(A.domain()), E(A.domain());
  for (auto p : B.domain()) { B(p) = int(rng() % 1000); C(p) = int(rng() % 1000); }
  fill(A, 0); fill(D, 0); fill(E, 0);
  // make every mirror resident and current (state 1 or 2): no upload — hence no ABI call — will sit between the calls below
  pixel_wise(A, B, C) | ops::add();
  pixel_wise(D, A) | [] (int& d, int& a) { d = a; };
  pixel_wise(E, A) | [] (int& e, int& a) { e = a; };
  vpp::device::sync();
  const image2d<int> &cA = A, &cB = B, &cC = C, &cD = D, &cE = E;   // const accessors download a newer mirror and leave it current: the images stay resident
  for (int round = 1; round <= 3; round++) {
    CHECK(A.device_current() && B.device_current() && C.device_current() && D.device_current() && E.device_current());
    // tagged functor feeds a lambda: A = B - C (held back), then D = A * round through the opaque kernel
    pixel_wise(A, B, C) | ops::sub();
    CHECK(vpp_deferred_pending() == 1);
    if (round == 1) pixel_wise(D, A) | [] (int& d, int& a) { d = a * 1; };
    if (round == 2) pixel_wise(D, A) | [] (int& d, int& a) { d = a * 2; };
    if (round == 3) pixel_wise(D, A) | [] (int& d, int& a) { d = a * 3; };
    CHECK(vpp_deferred_pending() == 0);
    // lambda feeds a tagged functor, which feeds a block_wise lambda: E = B + round (lambda), A = E + C (held back), then the first pixel of every 4 x 4 block of E = A's
    if (round == 1) pixel_wise(E, B) | [] (int& e, int& b) { e = b + 1; };
    if (round == 2) pixel_wise(E, B) | [] (int& e, int& b) { e = b + 2; };
    if (round == 3) pixel_wise(E, B) | [] (int& e, int& b) { e = b + 3; };
    pixel_wise(A, E, C) | ops::add();
    CHECK(vpp_deferred_pending() == 1);
    block_wise(vint2(4, 4), E, A) | [] (auto e, auto a) { e(0, 0) = a(0, 0) * 5; };
    CHECK(vpp_deferred_pending() == 0);
    for (int r = 0; r < nr; r += 7) for (int c = 0; c < nc; c += 5) {
      CHECK(cD(r, c) == (cB(r, c) - cC(r, c)) * round);
      CHECK(cA(r, c) == cB(r, c) + round + cC(r, c));
      CHECK(cE(r, c) == ((r % 4 == 0 && c % 4 == 0) ? cA(r, c) * 5 : cB(r, c) + round));
    }
  }
  // `_immediate`: the tagged functor launches at once, nothing is held back
  pixel_wise(A, B, C)(_immediate) | ops::add();
  CHECK(vpp_deferred_pending() == 0);
  image2d<vuchar3> S(64, 160, _border = 2), T(S.domain()), W(S.domain());
  for (auto p : S.domain_with_border()) S(p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
  pixel_wise(T, relative_access(S))(_immediate) | ops::box_mean<5, 5>();
  CHECK(vpp_deferred_pending() == 0);
  pixel_wise(W, relative_access(S)) | ops::box_mean<5, 5>();
  CHECK(vpp_deferred_pending() == 1);
  CHECK(same_pixels(T, W));
  for (auto p : A.domain()) CHECK(A(p) == B(p) + C(p));
}

static double seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void time_4k() {
  // the lambda form against the tagged functor on the same 4K int images (both resident in HBM after the first call)
  image2d<int> A(2160, 3840), B(A.domain()), C(A.domain());
  for (auto p : B.domain()) { B(p) = int(rng() >> 2); C(p) = int(rng() >> 2); }
  vpp_pixel_wise(A, B, C);
  pixel_wise(A, B, C) | ops::add();
  const int K = 200;
  double t0 = seconds();
  for (int k = 0; k < K; k++) vpp_pixel_wise(A, B, C);
  vpp_sync(nullptr);
  const double lam = (seconds() - t0) / K;
  t0 = seconds();
  for (int k = 0; k < K; k++) pixel_wise(A, B, C) | ops::add();
  vpp_sync(nullptr);
  const double tag = (seconds() - t0) / K;
  std::printf("4K int add, back-to-back calls (at most 2 queued): lambda %.2f us, ops::add %.2f us (ratio %.2f)\n", lam * 1e6, tag * 1e6, lam / tag);
  for (int r = 0; r < 2160; r += 97) for (int c = 0; c < 3840; c += 89) CHECK(A(r, c) == B(r, c) + C(r, c));
  // the 5x5 box of benchmarks/box_5x5_filter2.cc:71-81 as the opaque lambda against ops::box_mean<5, 5> (the hand-written K2i kernel), 4K int.  The calls rotate
  // over NS frame sets (NS x 66 MB: nothing survives in the 256 MiB Infinity Cache between two uses), like the bench's one-frame-per-call legs.
  const int NS = 10;
  {
    std::vector<image2d<int>> S, D, T, W;
    for (int q = 0; q < NS; q++) { S.emplace_back(2160, 3840, _border = 2); D.emplace_back(S[q].domain()); }
    T.emplace_back(S[0].domain()); W.emplace_back(S[0].domain());
    for (auto p : S[0].domain_with_border()) S[0](p) = int(rng() % 1000);
    for (int q = 1; q < NS; q++) for (auto p : S[0].domain_with_border()) S[q](p) = S[0](p);   // (the same pixels in distinct buffers: one result to check, no cache reuse)
    auto body = [] (int& b, auto a) {
      int sum = 0;
      for (int i = -2; i <= 2; i++)
      for (int j = -2; j <= 2; j++)
        sum += a(i, j);
      b = sum / 25;
    };
    for (int q = 0; q < NS; q++) vpp_pixel_wise(D[q], S[q]);
    pixel_wise(T[0], relative_access(S[0])) | ops::box_mean<5, 5>();
    pixel_wise(W[0], relative_access(S[0]))(_nbh_read_only) | body;
    CHECK(same_pixels(D[0], T[0]) && same_pixels(D[NS - 1], T[0]) && same_pixels(W[0], T[0]));
    vpp_sync(nullptr);
    t0 = seconds();
    for (int k = 0; k < K; k++) vpp_pixel_wise(D[k % NS], S[k % NS]);
    vpp_sync(nullptr);
    const double blam = (seconds() - t0) / K;
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_immediate) | ops::box_mean<5, 5>();
    vpp_sync(nullptr);
    const double btag = (seconds() - t0) / K;
    // the same body under `_nbh_read_only`: 4-byte pixels take the register window (pixel_wise_device.hh: pixel_wise_window_kernel)
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_nbh_read_only) | body;
    vpp_sync(nullptr);
    const double bwin = (seconds() - t0) / K;
    CHECK(same_pixels(D[1], T[0]));
    std::printf("4K int box 5x5, one launch per call over %d rotating frame sets: lambda %.2f us, lambda with _nbh_read_only (register window) %.2f us, ops::box_mean<5,5> %.2f us (ratio %.2f)\n", NS, blam * 1e6, bwin * 1e6, btag * 1e6, blam / btag);
    std::printf("LAMBDA_JSON {\"int_5x5\": {\"literal_us\": %.2f, \"nbh_read_only_us\": %.2f, \"ops_box_mean_us\": %.2f}}\n", blam * 1e6, bwin * 1e6, btag * 1e6);
  }
  // the same on vuchar3 (examples/box_filter.cc:23-32 body; BASELINE configs[1]'s pixel type), NS rotating frame sets of 50 MB
  {
    std::vector<image2d<vuchar3>> S, D, T;
    for (int q = 0; q < NS; q++) { S.emplace_back(2160, 3840, _border = 2); D.emplace_back(S[q].domain()); }
    T.emplace_back(S[0].domain());
    for (auto p : S[0].domain_with_border()) S[0](p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    for (int q = 1; q < NS; q++) for (auto p : S[0].domain_with_border()) S[q](p) = S[0](p);
    auto k3 = [] (vuchar3& out, auto nbh) {
      vint3 sum = vint3::Zero();
      for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) sum += nbh(i, j).template cast<int>();
      out = (sum / 25).template cast<unsigned char>();
    };
    for (int q = 0; q < NS; q++) pixel_wise(D[q], relative_access(S[q])) | k3;
    pixel_wise(T[0], relative_access(S[0])) | ops::box_mean<5, 5>();
    CHECK(same_pixels(D[0], T[0]) && same_pixels(D[NS - 1], T[0]));
    vpp_sync(nullptr);
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS])) | k3;
    vpp_sync(nullptr);
    const double blam = (seconds() - t0) / K;
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_immediate) | ops::box_mean<5, 5>();
    vpp_sync(nullptr);
    const double btag = (seconds() - t0) / K;
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_nbh_read_only) | k3;
    vpp_sync(nullptr);
    const double bro = (seconds() - t0) / K;
    CHECK(same_pixels(D[1], T[0]));
    std::printf("4K vuchar3 box 5x5, one launch per call over %d rotating frame sets: lambda %.2f us, lambda with _nbh_read_only (LDS tile) %.2f us, ops::box_mean<5,5> %.2f us (ratio %.2f)\n", NS, blam * 1e6, bro * 1e6, btag * 1e6, blam / btag);
    std::printf("LAMBDA_JSON {\"vuchar3_5x5\": {\"literal_us\": %.2f, \"nbh_read_only_us\": %.2f, \"ops_box_mean_us\": %.2f}}\n", blam * 1e6, bro * 1e6, btag * 1e6);
  }
  // block_wise on the device: 16 x 16 block sums (one wave per block) and 4 x 4 (one lane per block)
  for (int bs : {16, 4}) {
    image2d<int> A2(2160, 3840), S2(2160, 3840);
    for (auto p : A2.domain()) A2(p) = int(rng() % 100);
    fill(S2, 0);
    auto kb = [] (auto a, auto s) {
      int sum = 0;
      for (int r = 0; r < a.nrows(); r++) for (int c = 0; c < a.ncols(); c++) sum += a(r, c);
      s(0, 0) = sum;
    };
    block_wise(vint2(bs, bs), A2, S2) | kb;
    t0 = seconds();
    for (int k = 0; k < 50; k++) block_wise(vint2(bs, bs), A2, S2) | kb;
    vpp_sync(nullptr);
    std::printf("4K int block_wise %d x %d sums, back-to-back calls (at most 2 queued): %.2f us\n", bs, bs, (seconds() - t0) / 50 * 1e6);
  }
}

int main(int argc, char** argv) {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);
#define STEP(f) do { std::fprintf(stderr, "[device_lambda_test] " #f "\n"); f(); } while (0)
  if (argc > 1 && !std::strcmp(argv[1], "timeonly")) { STEP(time_4k); std::printf("device_lambda_test ok\n"); return 0; }   // bench.py's lambda_call leg
  STEP(test_reference_bodies);
  STEP(test_device_equals_host);
  STEP(test_block_wise_device);
  STEP(test_neighbourhood_tiles);
  STEP(test_register_window);
  STEP(test_writes_through_a_neighbourhood);
  STEP(test_box_lambdas_against_the_oracle);
  STEP(test_held_back_calls_and_lambdas_keep_their_order);
  if (argc > 1 && !std::strcmp(argv[1], "time")) STEP(time_4k);
  std::printf("device_lambda_test ok\n");
  return 0;
}
