// Host-side contract of the vpp-shaped C++ surface, restating what the reference's own unit tests pin
// (tests/pixel_wise.cc, tests/block_wise.cc, tests/border.cc, tests/imageNd.cc, tests/image2d.cc, tests/fill.cc,
// tests/sum.cc, tests/cast.cc, tests/window.cc, tests/boxNd_iterator.cc, tests/box_nbh2d.cc).  No GPU involved:
// opaque lambdas are evaluated on the host by design (BASELINE configs[0]).
#include <cstdio>
#include <cstdlib>
#include <vpp/vpp.hh>
#include <vpp/algorithms/optical_flow/gradient_descent.hh>

using namespace vpp;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static void test_layout_and_access() {
  // offset formula and row table (tests/imageNd.cc:23-27, tests/image2d.cc:8-33)
  image2d<int> img(6, 9, _border = 2);
  CHECK(img.coords_to_offset(vint2(3, 4)) == img.pitch() * 3 + 4 * int(sizeof(int)));
  for (int r = -2; r < 8; r++) for (int c = -2; c < 11; c++) CHECK(&img(r, c) == &img[r][c]);
  // alignment 256 of first pixel and pitch (tests/imageNd.cc:45-50)
  image2d<char> al(7, 13, _aligned = 256, _border = 1);
  CHECK((reinterpret_cast<unsigned long>(&al(0, 0)) % 256) == 0 && al.pitch() % 256 == 0 && al.alignment() == 256);
  // exact layouts (from imageNd.hpp:151-196)
  image2d<vuchar3> box_src(2160, 3840, _border = 2, _aligned = 16);
  CHECK(box_src.pitch() == 11552 && (char*)&box_src(0, 0) - (char*)box_src.data() == 23120);
  image2d<unsigned char> fast_src(2160, 3840, _border = 3, _aligned = 32);
  CHECK(fast_src.pitch() == 3904 && (char*)&fast_src(0, 0) - (char*)fast_src.data() == 11744);
  CHECK(sizeof(vuchar3) == 3 && sizeof(vfloat2) == 8 && sizeof(vint2) == 8);
  // shallow copies share pixels; clone does not (tests/imageNd.cc:58-71)
  image2d<int> a(4, 4); fill(a, 1);
  image2d<int> b = a; b(1, 1) = 7; CHECK(a(1, 1) == 7);
  image2d<int> c = clone(a, _border = 3); c(1, 1) = 9; CHECK(a(1, 1) == 7 && c.border() == 3 && c(0, 0) == 1);
  // sub-image (tests/imageNd.cc:74-83)
  image2d<int> s = a | box2d(vint2(1, 1), vint2(2, 3));
  CHECK(s.nrows() == 2 && s.ncols() == 3 && &s(0, 0) == &a(1, 1) && &s(1, 2) == &a(2, 3));
  // external data is borrowed (README.md:102-116)
  int raw[12] = {0};
  image2d<int> ext(3, 4, _data = raw, _pitch = int(4 * sizeof(int)));
  ext(2, 3) = 5; CHECK(raw[11] == 5);
  // bilinear interpolation truncates back to the pixel type (tests/imageNd.cc:87-107)
  image2d<vuchar1> t(2, 2, _border = 1);
  t(0, 0)[0] = 0; t(0, 1)[0] = 10; t(1, 0)[0] = 20; t(1, 1)[0] = 30;
  CHECK(t.linear_interpolate(vfloat2(0.5, 0.5))[0] == int((10 + 20 + 30) / 4.f));
  // casts (tests/cast.cc:16-21)
  vuchar1 m; m[0] = 2; CHECK(cast<vint1>(m)[0] == 2); CHECK(cast<int>(m) == 2); CHECK(cast<vfloat2>(vint2(1, 2))[1] == 2.f);
  // box iteration in raster order (tests/boxNd_iterator.cc)
  int k = 0; for (auto p : make_box2d(3, 4)) { CHECK(p[0] == k / 4 && p[1] == k % 4); k++; } CHECK(k == 12);
  image3d<int> v3(2, 3, 4); v3(1, 2, 3) = 4; CHECK(v3(1, 2, 3) == 4 && v3.nslices() == 2);
}

static void test_pixel_wise() {
  image2d<int> img(10, 12);
  pixel_wise(img) | [](int& i) { i = 42; };                                  // tests/pixel_wise.cc:13-20
  for (auto p : img.domain()) CHECK(img(p) == 42);
  int cnt = 0;                                                               // raster order over a box with _no_threads (:23-28)
  pixel_wise(img.domain())(_no_threads) | [&](vint2 p) { CHECK(p[0] * 12 + p[1] == cnt); cnt++; };
  // the four traversal directions through in-place prefix sums with relative_access (:34-60)
  auto run = [&](auto order, auto expect) {
    image2d<int> a(10, 12, _border = 1);
    fill_with_border(a, 0);
    order(a);
    for (auto p : a.domain()) CHECK(a(p) == expect(p));
  };
  run([](auto& a) { pixel_wise(relative_access(a))(_no_threads, _left_to_right) | [](auto n) { n(0, 0) = n(0, -1) + 1; }; }, [](vint2 p) { return p[1] + 1; });
  run([](auto& a) { pixel_wise(relative_access(a))(_no_threads, _right_to_left) | [](auto n) { n(0, 0) = n(0, 1) + 1; }; }, [](vint2 p) { return 12 - p[1]; });
  run([](auto& a) { pixel_wise(relative_access(a))(_no_threads, _top_to_bottom) | [](auto n) { n(0, 0) = n(-1, 0) + 1; }; }, [](vint2 p) { return p[0] + 1; });
  run([](auto& a) { pixel_wise(relative_access(a))(_no_threads, _bottom_to_top) | [](auto n) { n(0, 0) = n(1, 0) + 1; }; }, [](vint2 p) { return 10 - p[0]; });
  // a kernel that returns a value builds an image (:63-64)
  image2d<int> A(5, 6), B(5, 6); fill(A, 3); fill(B, 4);
  auto C = pixel_wise(A, B) | [](int& a, int& b) { return a * b; };
  for (auto p : C.domain()) CHECK(C(p) == 12);
  // the benchmark kernels as opaque lambdas and as tagged functors give the same pixels on the host
  image2d<int> S(5, 6), S2(5, 6);
  pixel_wise(S, A, B) | [](int& s, int& a, int& b) { s = a + b; };           // benchmarks/image_add.cc:53-56
  pixel_wise(S2, A, B) | [](int& s, const int& a, const int& b) { ops::add()(s, a, b); };
  for (auto p : S.domain()) CHECK(S(p) == 7 && S2(p) == 7);
  // 5x5 mean through relative_access and through the legacy box_nbh2d spelling (benchmarks/box_5x5_filter.cc:163-172)
  image2d<int> I(12, 14, _border = 2), O1(12, 14), O2(12, 14);
  int seed = 1; for (int r = -2; r < 14; r++) for (int c = -2; c < 16; c++) { seed = seed * 1103515245 + 12345; I(r, c) = (seed >> 8) % 1000; }
  pixel_wise(O1, relative_access(I)) | [](int& o, auto a) { int s = 0; for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) s += a(i, j); o = s / 25; };
  auto nb = box_nbh2d<int, 5, 5>(I);
  pixel_wise(O2, nb) | [](int& o, auto& n) { int s = 0; n.for_all([&](int x) { s += x; }); o = s / 25; };
  for (auto p : O1.domain()) CHECK(O1(p) == O2(p));
  auto at = box_nbh2d<int, 3, 3>(I, vint2(4, 4));                            // tests/box_nbh2d.cc:9-18
  CHECK(at.north() == I(3, 4) && at.south() == I(5, 4) && at.east() == I(4, 5) && at.west() == I(4, 3));
}

static void test_block_wise_and_fill() {
  image2d<int> img(10, 10, _border = 1);                                     // tests/block_wise.cc:100-114
  fill_border_with_value(img, 2);
  fill(img, 0);
  block_wise(vint2(3, 3), img) | [](auto si) { fill(si, 1); };
  for (auto p : img.domain_with_border()) CHECK(img(p) == (img.has(p) ? 1 : 2));
  std::vector<vint2> order;                                                   // block traversal order (:51-96)
  block_wise(vint2(4, 4), img.domain())(_no_threads) | [&](box2d b) { order.push_back(b.p1()); };
  CHECK(order.size() == 9 && order[0] == vint2(0, 0) && order[1] == vint2(0, 4) && order[3] == vint2(4, 0) && order[8] == vint2(8, 8));
  int rows = 0; row_wise(img.domain())(_no_threads) | [&](box2d b) { CHECK(b.nrows() == 1 && b.ncols() == 10); rows++; }; CHECK(rows == 10);
  // border fills (tests/border.cc:11-59; mirror semantics from fill.hh:48-83)
  image2d<int> b(3, 4, _border = 2);
  for (auto p : b.domain()) b(p) = p[0] * 10 + p[1];
  fill_border_closest(b); CHECK(b(-2, -2) == 0 && b(-1, 2) == 2 && b(4, 5) == 23 && b(1, -1) == 10);
  fill_border_mirror(b); CHECK(b(-1, 0) == 0 && b(-2, 0) == 10 && b(3, 1) == 21 && b(4, 1) == 11 && b(0, -2) == 1 && b(0, 5) == 2 && b(-1, -1) == 0 && b(-2, -2) == 11);
  CHECK(sum(img) == 100);                                                     // tests/sum.cc
  int n4 = 0; foreach(c4, [&](vint2 n) { n4 += std::abs(n[0]) + std::abs(n[1]); }); CHECK(n4 == 4);  // tests/window.cc
  keypoint_container<keypoint<int>, int> kc(make_box2d(20, 20));
  kc.add(keypoint<int>(vint2(3, 4))); kc.add(keypoint<int>(vint2(5, 6)));
  CHECK(kc.has(vint2(3, 4)) && kc.index_of(*new vint2(5, 6)) == 1);
  kc.move(0, vint2(4, 4)); CHECK(kc[0].velocity == vint2(1, 0) && kc[0].age == 2);
  kc.remove(1); kc.compact(); CHECK(kc.size() == 1 && kc.has(vint2(4, 4)) && !kc.has(vint2(5, 6)));
  // sync_attributes (keypoint_container.hpp:64-112): survivors keep their attribute at the compacted index, dead ones go
  // through die_fun, keypoints added since the last sync get new_value — as video_extruder.hpp:114-118 uses it
  keypoint_container<keypoint<int>, int> k2(make_box2d(40, 40));
  std::vector<keypoint_trajectory> traj;
  for (int i = 0; i < 6; i++) k2.add(keypoint<int>(vint2(2 * i + 1, 3 * i + 1)));
  k2.sync_attributes(traj, keypoint_trajectory(7));                          // no compact yet: plain resize
  CHECK(traj.size() == 6 && traj[5].start_frame() == 7);
  for (int i = 0; i < 6; i++) traj[i].move_to(vfloat2(float(i), 0.f));
  k2.prepare_matching();
  k2.remove(1); k2.remove(4);
  k2.add(keypoint<int>(vint2(30, 30))); k2.add(keypoint<int>(vint2(31, 35)));
  k2.compact();
  std::vector<keypoint_trajectory> dead;
  k2.sync_attributes(traj, keypoint_trajectory(9), dead);
  CHECK(k2.size() == 6 && traj.size() == 6 && dead.size() == 2);
  CHECK(traj[0].position()[0] == 0.f && traj[1].position()[0] == 2.f && traj[2].position()[0] == 3.f && traj[3].position()[0] == 5.f);
  CHECK(traj[4].size() == 0 && traj[4].start_frame() == 9 && traj[5].start_frame() == 9);
  CHECK(dead[0].position()[0] == 1.f && dead[1].position()[0] == 4.f);
}

static void test_colorspace_conversions() {                                  // tests/colorspace_conversions.cc
  image2d<vuchar3> i1(100, 100);
  unsigned char i = 0;
  for (vint2 p : i1.domain()) { i1(p) = vuchar3(i, i, i); i++; }
  image2d<vuchar1> i2 = rgb_to_graylevel<vuchar1>(i1);
  i = 0;
  for (vint2 p : i1.domain()) { CHECK(i2(p)[0] == i); i++; }
  image2d<vuchar4> i4(7, 9, _border = 2);
  for (int r = -2; r < 9; r++) for (int c = -2; c < 11; c++) i4(r, c) = vuchar4(r + 10, 3 * c + 20, 200, 255);
  image2d<unsigned char> g = rgb_to_graylevel<unsigned char>(i4);                // 4th component ignored, border mapped too
  CHECK(g.border() == 2);
  for (int r = -2; r < 9; r++) for (int c = -2; c < 11; c++) CHECK(g(r, c) == (r + 10 + 3 * c + 20 + 200) / 3);
  image2d<vuchar3> back = graylevel_to_rgb<vuchar3>(g);
  CHECK(back(3, 4) == vuchar3(g(3, 4), g(3, 4), g(3, 4)));
}

static void test_image3d_and_iterators() {
  image3d<int> img1(make_box3d(10, 20, 30));                                  // tests/image3d.cc
  image3d<int> img2(10, 20, 30);
  CHECK(img1.domain() == img2.domain());
  CHECK(img1.nslices() == 10 && img1.nrows() == 20 && img1.ncols() == 30);
  for (int s = 0; s < img1.nslices(); s++)
    for (int r = 0; r < img1.nrows(); r++)
      for (int c = 0; c < img1.ncols(); c++) { img1(s, r, c) = s * r * c; CHECK(img1(s, r, c) == s * r * c); }
  {
    auto s1 = img1 | box3d(vint3(2, 3, 4), vint3(5, 6, 7));
    CHECK(&s1(0, 0, 0) == &img1(vint3(2, 3, 4)));
    CHECK(&s1(0, 1, 1) == &img1(vint3(2, 3, 4) + vint3(0, 1, 1)));
    CHECK(&s1(1, 1, 1) == &img1(vint3(2, 3, 4) + vint3(1, 1, 1)));
    CHECK(&s1(2, 2, 2) == &img1(vint3(2, 3, 4) + vint3(2, 2, 2)));
  }
  image2d<int> img(3, 3, _border = 1);                                        // tests/imageNd_iterator.cc
  const vint2 ref[] = {vint2(0, 0), vint2(0, 1), vint2(0, 2), vint2(1, 0), vint2(1, 1), vint2(1, 2), vint2(2, 0), vint2(2, 1), vint2(2, 2)};
  int i = 0;
  for (auto& p : img) { CHECK(&p == &img(ref[i])); i++; }
  CHECK(i == 9);
}

// The container writes its 2-D index lazily (a log replayed on the first look-up): every interleaving of add / move / remove /
// compact / prepare_matching / look-ups must read exactly what the reference's eager stores produce (keypoint_container.hpp:22-167).
static void test_keypoint_index_is_the_eager_one() {
  const int NR = 24, NC = 31;
  unsigned rng = 12345u;
  auto next = [&](unsigned m) { rng = rng * 1664525u + 1013904223u; return (rng >> 8) % m; };
  for (int round = 0; round < 20; round++) {
    keypoint_container<keypoint<int>, int> kc(make_box2d(NR, NC));
    std::vector<int> model(NR * NC, -1);                                      // the eager index
    std::vector<vint2> pos; std::vector<int> age;
    auto cell = [&](vint2 p) -> int& { return model[p[0] * NC + p[1]]; };
    for (int step = 0; step < 400; step++) {
      const unsigned op = next(100);
      if (op < 30 || pos.empty()) {                                           // add
        vint2 p(next(NR), next(NC)); kc.add(keypoint<int>(p)); cell(p) = int(pos.size()); pos.push_back(p); age.push_back(1);
      } else if (op < 60) {                                                   // move
        int i = next(unsigned(pos.size())); vint2 p(next(NR), next(NC)); kc.move(i, p); pos[i] = p; age[i]++; cell(p) = i;
      } else if (op < 75) {                                                   // remove(int)
        int i = next(unsigned(pos.size())); kc.remove(i); age[i] = 0; if (cell(pos[i]) == i) cell(pos[i]) = -1;
      } else if (op < 80) {                                                   // prepare_matching
        kc.prepare_matching(); std::fill(model.begin(), model.end(), -1);
      } else if (op < 85) {                                                   // compact
        kc.compact();
        size_t w = 0;
        for (size_t i = 0; i < pos.size(); i++) if (age[i] > 0) { pos[w] = pos[i]; age[w] = age[i]; cell(pos[w]) = int(w); w++; }
        pos.resize(w); age.resize(w);
      } else if (op < 90) {                                                   // remove(position) when the cell is occupied
        vint2 p(next(NR), next(NC));
        if (cell(p) >= 0) { CHECK(kc.has(p)); int i = cell(p); kc.remove(p); age[i] = 0; if (cell(pos[i]) == i) cell(pos[i]) = -1; }
        else CHECK(!kc.has(p));
      } else if (op < 97) {                                                   // point look-ups
        vint2 p(next(NR), next(NC)); CHECK(kc.has(p) == (cell(p) >= 0)); CHECK(kc.index_of(p) == cell(p));
      } else {                                                                // whole image
        const keypoint_container<keypoint<int>, int>& ck = kc;
        for (int r = 0; r < NR; r++) for (int c = 0; c < NC; c++) CHECK(ck.index2d()(r, c) == model[r * NC + c]);
      }
      CHECK(kc.size() == int(pos.size()));
    }
    for (int r = 0; r < NR; r++) for (int c = 0; c < NC; c++) CHECK(kc.index2d()(r, c) == model[r * NC + c]);
    for (size_t i = 0; i < pos.size(); i++) CHECK(kc[i].position == pos[i] && kc[i].age == age[i]);
    // the mutable image may be written behind the container's back: the next prepare_matching must still clear it
    kc.index2d()(3, 3) = 77; kc.prepare_matching(); CHECK(!kc.has(vint2(3, 3)));
  }
}

// gradient_descent_match with an opaque distance (gradient_descent.hh:10-89) and the pyramid's sampling helpers (pyramid.hh:62-103)
static void test_gradient_descent_and_samplers() {
  const vint2 target(7, -3);
  int calls = 0;
  auto dist = [&](vint2, vint2 q, int) { calls++; const vint2 d = q - target; return std::abs(d[0]) + std::abs(d[1]); };
  auto m = gradient_descent_match(vint2(0, 0), vint2(5, -1), dist, 10);
  CHECK(m.flow == target && m.distance == 0);                                  // two diagonal moves, then a round without improvement
  auto far = gradient_descent_match(vint2(0, 0), vint2(0, 0), dist, 3);
  CHECK(far.distance == 4 && far.flow == vint2(3, -3));                       // limited to three rounds: one diagonal step per round
  auto flat = gradient_descent_match(vint2(2, 2), vint2(9, 9), [](vint2, vint2, int) { return 5; }, 10);
  CHECK(flat.flow == vint2(7, 7) && flat.distance == 5);                      // no strict improvement anywhere: stays on the prediction
  image2d<int> big(6, 8, _border = 1), half(3, 4), third(2, 3);
  for (auto p : big.domain()) big(p) = p[0] * 10 + p[1];
  subsample2(big, half);
  CHECK(half(0, 0) == 0 && half(1, 3) == 26 && half(2, 1) == 42);
  subsample(big, third, 2.5f);
  CHECK(third(0, 0) == 0 && third(1, 2) == 25 && third(1, 1) == 22);          // int(1 * 2.5) = 2, int(2 * 2.5) = 5
}

int main() {
  test_gradient_descent_and_samplers();
  test_keypoint_index_is_the_eager_one();
  test_image3d_and_iterators();
  test_colorspace_conversions();
  test_layout_and_access();
  test_pixel_wise();
  test_block_wise_and_fill();
  std::puts("host_api_test ok");
  return 0;
}
