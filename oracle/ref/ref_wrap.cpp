// ref_wrap.cpp — the REAL reference (matt-42/vpp headers, compiled unmodified from /root/reference) behind the same
// descriptor-based C surface as the oracle restatement.  TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libvpp_ref.so
// where the reference tree exists.  Third-party Eigen / iod are replaced by the stand-ins in shims/ (SURVEY.md §8c route A).
// Compiled without -fopenmp (like the reference's tests, tests/CMakeLists.txt:16): serial, deterministic order.
#include <vpp/vpp.hh>
#include <vpp/algorithms/filters/scharr.hh>
#include <vpp/algorithms/fast_detector/fast.hh>
#include <vpp/algorithms/pyrlk/pyrlk_match.hh>
#include <vpp/algorithms/lucas_kanade.hh>
#include <vpp/algorithms/lbp/lbp_transform.hh>
#include <iod/array_view.hh>
#include <vpp/algorithms/optical_flow/semi_dense_optical_flow.hpp>

#include <cstring>
#include <vector>

#include "../../include/vpp_amd.h"

using namespace vpp;

namespace {
template <class V> image2d<V> wrap(const vpp_image_desc* d) {
  return image2d<V>(make_box2d(d->nrows, d->ncols), _data = (V*)d->first_pixel, _pitch = (int)d->pitch, _border = (int)d->border);
}
// copy a reference-owned image (domain + border) into a caller image of the same domain / border
template <class V> void export_image(const image2d<V>& src, const vpp_image_desc* d) {
  auto dst = wrap<V>(d);
  const int b = d->border < src.border() ? d->border : src.border();
  for (int r = -b; r < src.nrows() + b; r++)
    for (int c = -b; c < src.ncols() + b; c++) dst(r, c) = src(r, c);
}
bool is(const vpp_image_desc* d, int dtype, int ch) { return d->dtype == dtype && d->channels == ch; }
}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int ref_pixelwise_add(const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b) {
  if (is(dst, VPP_I32, 1)) {  // benchmarks/image_add.cc:51-57
    auto A = wrap<int>(dst); auto B = wrap<int>(a); auto C = wrap<int>(b);
    pixel_wise(A, B, C) | [](int& x, int& y, int& z) { x = y + z; };
    return 0;
  }
  if (is(dst, VPP_U8, 3)) {
    auto A = wrap<vuchar3>(dst); auto B = wrap<vuchar3>(a); auto C = wrap<vuchar3>(b);
    pixel_wise(A, B, C) | [](vuchar3& x, vuchar3& y, vuchar3& z) { x = y + z; };
    return 0;
  }
  if (is(dst, VPP_F32, 1)) {
    auto A = wrap<float>(dst); auto B = wrap<float>(a); auto C = wrap<float>(b);
    pixel_wise(A, B, C) | [](float& x, float& y, float& z) { x = y + z; };
    return 0;
  }
  return VPP_ERR_UNSUPPORTED;
}

int ref_box_filter5x5(const vpp_image_desc* dst, const vpp_image_desc* src) {
  if (is(dst, VPP_I32, 1)) {  // benchmarks/box_5x5_filter2.cc:71-81
    auto B = wrap<int>(dst); auto A = wrap<int>(src);
    pixel_wise(B, relative_access(A)) | [&](int& b, auto a) {
      int sum = 0;
      for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) sum += a(i, j);
      b = sum / 25;
    };
    return 0;
  }
  if (is(dst, VPP_U8, 3)) {  // examples/box_filter.cc:23-32 widened to 5x5
    auto B = wrap<vuchar3>(dst); auto A = wrap<vuchar3>(src);
    pixel_wise(relative_access(A), B) | [](auto n, auto& b) {
      vint3 sum = vint3::Zero();
      for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) sum += n(i, j).template cast<int>();
      b = (sum / 25).cast<unsigned char>();
    };
    return 0;
  }
  if (is(dst, VPP_U8, 1)) {
    auto B = wrap<unsigned char>(dst); auto A = wrap<unsigned char>(src);
    pixel_wise(B, relative_access(A)) | [&](unsigned char& b, auto a) {
      int sum = 0;
      for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) sum += a(i, j);
      b = sum / 25;
    };
    return 0;
  }
  return VPP_ERR_UNSUPPORTED;
}

}  // extern "C"
#pragma GCC visibility pop
template <class V> static int fill_border_t(const vpp_image_desc* d, int mode, const void* value) {
  auto I = wrap<V>(d);
  if (mode == VPP_BORDER_MIRROR) fill_border_mirror(I);
  else if (mode == VPP_BORDER_CLOSEST) fill_border_closest(I);
  else { V v; memcpy(&v, value, sizeof(V)); fill_border_with_value(I, v); }
  return 0;
}
#pragma GCC visibility push(default)
extern "C" {
int ref_fill_border(const vpp_image_desc* d, int mode, const void* value) {
  if (is(d, VPP_U8, 1)) return fill_border_t<unsigned char>(d, mode, value);
  if (is(d, VPP_U8, 3)) return fill_border_t<vuchar3>(d, mode, value);
  if (is(d, VPP_I32, 1)) return fill_border_t<int>(d, mode, value);
  if (is(d, VPP_I32, 2)) return fill_border_t<vint2>(d, mode, value);
  if (is(d, VPP_F32, 2)) return fill_border_t<vfloat2>(d, mode, value);
  return VPP_ERR_UNSUPPORTED;
}

// pyramid2d<V>(img, nlevels, 2, _border = border) (pyramid.hh:146-158): levels exported into caller images.
}  // extern "C"
#pragma GCC visibility pop
template <class V> static int pyramid_t(const vpp_image_desc* img, int nlevels, int border, const vpp_image_desc* out) {
  auto I = wrap<V>(img);
  pyramid2d<V> pyr(I, nlevels, 2, _border = border);
  for (int l = 0; l < nlevels; l++) {
    if (out[l].nrows != pyr[l].nrows() || out[l].ncols != pyr[l].ncols()) return VPP_ERR_INVALID_ARG;
    export_image(pyr[l], &out[l]);
  }
  return 0;
}
#pragma GCC visibility push(default)
extern "C" {
int ref_pyramid(const vpp_image_desc* img, int nlevels, int border, const vpp_image_desc* out) {
  if (is(img, VPP_U8, 1)) return pyramid_t<unsigned char>(img, nlevels, border, out);
  if (is(img, VPP_I32, 2)) return pyramid_t<vint2>(img, nlevels, border, out);
  if (is(img, VPP_F32, 2)) return pyramid_t<vfloat2>(img, nlevels, border, out);
  return VPP_ERR_UNSUPPORTED;
}

int ref_scharr(const vpp_image_desc* out, const vpp_image_desc* in) {
  auto I = wrap<unsigned char>(in);
  if (is(out, VPP_F32, 2)) { auto O = wrap<vfloat2>(out); scharr(I, O); return 0; }
  if (is(out, VPP_I32, 2)) { auto O = wrap<vint2>(out); scharr(I, O); return 0; }
  return VPP_ERR_UNSUPPORTED;
}

// fast9(A, th, [_local_maxima | _blockwise, _block_size], _mask, _scores) (fast.hpp:931-955)
int ref_fast9(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size, int32_t* out_rc,
              int32_t* out_scores, int capacity, int* count) {
  auto A = wrap<unsigned char>(src);
  image2d<unsigned char> M;
  if (mask) M = wrap<unsigned char>(mask);
  std::vector<int> scores;
  std::vector<vint2> kps;
  try {
    if (mode == VPP_FAST9_LOCAL_MAXIMA) kps = fast9(A, th, _local_maxima, _mask = M, _scores = &scores);
    else if (mode == VPP_FAST9_BLOCKWISE) kps = fast9(A, th, _blockwise, _block_size = block_size, _mask = M, _scores = &scores);
    else kps = fast9(A, th, _mask = M, _scores = &scores);
  } catch (const std::runtime_error&) { return VPP_ERR_BORDER_TOO_SMALL; }
  *count = (int)kps.size();
  for (int i = 0; i < (int)kps.size() && i < capacity; i++) {
    out_rc[2 * i] = kps[i][0]; out_rc[2 * i + 1] = kps[i][1];
    if (out_scores) out_scores[i] = scores[i];
  }
  return (int)kps.size() > capacity ? VPP_ERR_CAPACITY : 0;
}

int ref_is_fast9_keypoint(const vpp_image_desc* src, int r, int c, int th) {
  auto A = wrap<unsigned char>(src);
  image2d<unsigned char> B(A.domain());
  FAST_internals::fast_detector9(A, B, th);  // dense scalar detector on the true ring (fast.hpp:512-551)
  return B(r, c);
}

// lucas_kanade(i1, i2, opts...) end to end (lucas_kanade.hpp:135-184): builds its own pyramids.
int ref_lucas_kanade(const vpp_image_desc* i1, const vpp_image_desc* i2, const float* pts, int n, int winsize, int nscales, int niterations,
                     double min_ev, double delta, float* out_flow, float* out_dist) {
  auto I1 = wrap<unsigned char>(i1); auto I2 = wrap<unsigned char>(i2);
  std::vector<vfloat2> keypoints;
  for (int i = 0; i < n; i++) keypoints.push_back(vfloat2(pts[2 * i], pts[2 * i + 1]));
  int k = 0;
  lucas_kanade(I1, I2, _keypoints = keypoints, _niterations = niterations, _winsize = winsize, _min_ev = min_ev, _delta = delta, _nscales = nscales,
               _flow = [&](vfloat2 p, vfloat2 f, float d) { out_flow[2 * k] = f[0]; out_flow[2 * k + 1] = f[1]; if (out_dist) out_dist[k] = d; k++; });
  return 0;
}

// pyrlk_match with lk_match_point_square_win<WS> over pyramids built as benchmarks/pyrlk_opencv_comparison.cc:49-60 does.
}  // extern "C"
#pragma GCC visibility pop
template <int WS>
static int pyrlk_t(const vpp_image_desc* i1, const vpp_image_desc* i2, int nlevels, int border, vpp_keypoint_f32* kps, int n, float min_ev,
                   float max_err, int max_it, float delta, int min_scale) {
  auto I1 = wrap<unsigned char>(i1); auto I2 = wrap<unsigned char>(i2);
  pyramid2d<unsigned char> pyr1(I1, nlevels, 2, _border = border);
  pyramid2d<unsigned char> pyr2(I2, nlevels, 2, _border = border);
  pyramid2d<vfloat2> grad(I1.domain(), nlevels, 2, _border = border);
  scharr(pyr1[0], grad[0]);
  grad.propagate_level0();
  pyrlk_keypoint_container kc(I1.domain());
  for (int i = 0; i < n; i++) {
    keypoint<float> kp(vfloat2(kps[i].pos_r, kps[i].pos_c));
    kp.velocity = vfloat2(kps[i].vel_r, kps[i].vel_c);
    kp.age = kps[i].age;
    kc.add(kp);
  }
  pyrlk_match(pyr1, grad, pyr2, kc, lk_match_point_square_win<WS>(), min_ev, max_err, max_it, delta, min_scale);
  for (int i = 0; i < n; i++) {
    kps[i].pos_r = kc[i].position[0]; kps[i].pos_c = kc[i].position[1];
    kps[i].vel_r = kc[i].velocity[0]; kps[i].vel_c = kc[i].velocity[1];
    kps[i].age = kc[i].age;
  }
  return 0;
}
#pragma GCC visibility push(default)
extern "C" {
int ref_pyrlk_match(const vpp_image_desc* i1, const vpp_image_desc* i2, int nlevels, int border, vpp_keypoint_f32* kps, int n, int winsize,
                    float min_ev, float max_err, int max_it, float delta, int min_scale) {
  switch (winsize) {
    case 5: return pyrlk_t<5>(i1, i2, nlevels, border, kps, n, min_ev, max_err, max_it, delta, min_scale);
    case 7: return pyrlk_t<7>(i1, i2, nlevels, border, kps, n, min_ev, max_err, max_it, delta, min_scale);
    case 9: return pyrlk_t<9>(i1, i2, nlevels, border, kps, n, min_ev, max_err, max_it, delta, min_scale);
  }
  return VPP_ERR_UNSUPPORTED;
}

int ref_semi_dense_optical_flow(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n, int winsize, int nscales,
                                int min_scale, int propagation, int patchsize, int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid) {
  auto I1 = wrap<unsigned char>(i1); auto I2 = wrap<unsigned char>(i2);
  for (int i = 0; i < n; i++) { out_pos[2 * i] = kps[2 * i]; out_pos[2 * i + 1] = kps[2 * i + 1]; out_dist[i] = 0; out_valid[i] = 0; }
  semi_dense_optical_flow(
      iod::array_view(n, [&](int i) { return vint2(kps[2 * i], kps[2 * i + 1]); }),
      [&](int i, vint2 pos, int distance) { out_pos[2 * i] = pos[0]; out_pos[2 * i + 1] = pos[1]; out_dist[i] = distance; out_valid[i] = 1; }, I1, I2,
      _winsize = winsize, _patchsize = patchsize, _propagation = propagation, _nscales = nscales, _min_scale = min_scale);
  return 0;
}

// rgb_to_graylevel<unsigned char>(image2d<vuchar3 | vuchar4>) (colorspace_conversions.hh:22-48); mirror != 0: the ingest chain of
// examples/video_extruder.cc:46-48 — clone(frame, _border = b); fill_border_mirror(frame); rgb_to_graylevel<unsigned char>(frame).
int ref_rgb_to_graylevel(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror) {
  if (!is(dst, VPP_U8, 1)) return VPP_ERR_UNSUPPORTED;
  if (is(src, VPP_U8, 3)) {
    auto S = wrap<vuchar3>(src);
    if (mirror) { auto F = clone(S, _border = (int)dst->border); fill_border_mirror(F); export_image(rgb_to_graylevel<unsigned char>(F), dst); }
    else export_image(rgb_to_graylevel<unsigned char>(S), dst);
    return 0;
  }
  if (is(src, VPP_U8, 4)) {
    auto S = wrap<vuchar4>(src);
    if (mirror) { auto F = clone(S, _border = (int)dst->border); fill_border_mirror(F); export_image(rgb_to_graylevel<unsigned char>(F), dst); }
    else export_image(rgb_to_graylevel<unsigned char>(S), dst);
    return 0;
  }
  return VPP_ERR_UNSUPPORTED;
}

// lbp_transform(A, B) (vpp/algorithms/lbp/lbp_transform.hh:6-38)
int ref_lbp_transform(const vpp_image_desc* out, const vpp_image_desc* in) {
  if (!is(out, VPP_U8, 1) || !is(in, VPP_U8, 1)) return VPP_ERR_UNSUPPORTED;
  auto A = wrap<unsigned char>(in); auto B = wrap<unsigned char>(out);
  lbp_transform(A, B);
  return 0;
}

// FAST_internals::fast_detector9(A, B, th) (vpp/algorithms/fast_detector/fast.hpp:511-551)
// (blockwise_maxima_filter, fast.hpp:577-614, cannot be wrapped: instantiating it fails to compile — it takes `const image2d<V>&`
// and stores `&A(r + i, 0)` into `V* rows[]`, fast.hpp:590 — so the reference never ran it; the oracle restatement of it is unpinned.)
int ref_fast9_dense(const vpp_image_desc* out, const vpp_image_desc* in, int th) {
  if (!is(in, VPP_U8, 1)) return VPP_ERR_UNSUPPORTED;
  auto A = wrap<unsigned char>(in);
  if (is(out, VPP_U8, 1)) { auto B = wrap<unsigned char>(out); FAST_internals::fast_detector9(A, B, th); return 0; }
  if (is(out, VPP_I32, 1)) { auto B = wrap<int>(out); FAST_internals::fast_detector9(A, B, th); return 0; }
  return VPP_ERR_UNSUPPORTED;
}


// local_maxima_filter(A, nbh_size) (fast.hpp:555-575), in place; the library is the serial build (tests/CMakeLists.txt:16
// has no OpenMP), where pixel_wise walks the rows top to bottom and each row left to right, so pixels above / left of the current one
// have already been filtered when they are compared.
int ref_local_maxima_filter(const vpp_image_desc* img) {
  if (is(img, VPP_U8, 1)) { auto A = wrap<unsigned char>(img); vpp::local_maxima_filter(A, 3); return 0; }
  if (is(img, VPP_I32, 1)) { auto A = wrap<int>(img); vpp::local_maxima_filter(A, 3); return 0; }
  if (is(img, VPP_F32, 1)) { auto A = wrap<float>(img); vpp::local_maxima_filter(A, 3); return 0; }
  return VPP_ERR_UNSUPPORTED;
}

}  // extern "C"
#pragma GCC visibility pop
