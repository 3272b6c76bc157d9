// oracle.cpp — CPU restatement of the matt-42/vpp hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Two builds (oracle/Makefile):
//   liboracle.so      : parity build, serial, -O2 -ffp-contract=off (numerics of the reference's tests/CMakeLists.txt:16:
//                       no FMA contraction, no fast-math).
//   liboracle_omp.so  : timing build, -O3 -march=native -fopenmp -DNDEBUG -DORC_OMP (mirrors
//                       benchmarks/CMakeLists.txt:10,18); OpenMP pragmas sit where the reference has them.
// Scratch that the reference leaves uninitialised is zero-filled here: that is the canonical value.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef ORC_OMP
#include <omp.h>
#define ORC_PRAGMA(x) _Pragma(#x)
#else
#define ORC_PRAGMA(x)
#endif

namespace {

struct Img {
  uint8_t* p0; int nr, nc, pitch, border, dtype, ch;
  Img() : p0(nullptr), nr(0), nc(0), pitch(0), border(0), dtype(0), ch(1) {}
  explicit Img(const vpp_image_desc* d)
      : p0((uint8_t*)d->first_pixel), nr(d->nrows), nc(d->ncols), pitch(d->pitch), border(d->border), dtype(d->dtype), ch(d->channels) {}
  template <class T> T* row(int r) const { return (T*)(p0 + (ptrdiff_t)r * pitch); }
  bool has(int r, int c) const { return r >= 0 && c >= 0 && r < nr && c < nc; }  // boxNd::has, boxNd.hh
};

int dtype_size(int dt) {
  switch (dt) { case VPP_U8: case VPP_I8: return 1; case VPP_U16: case VPP_I16: return 2; default: return 4; }
}

// Owning host image with the imageNd::allocate layout (vpp/core/imageNd.hpp:151-196), zero-filled.
struct OwnedImg {
  std::vector<uint8_t> buf; Img v;
  OwnedImg() {}
  OwnedImg(int nr, int nc, int dtype, int ch, int border, int align = 32) { alloc(nr, nc, dtype, ch, border, align); }
  void alloc(int nr, int nc, int dtype, int ch, int border, int align = 32) {
    int es = dtype_size(dtype) * ch;
    int border_size = border * es, border_padding = 0;
    if (border_size % align) { border_padding = align - (border_size % align); border_size += border_padding; }
    int pitch = nc * es + border_size * 2;
    if (pitch % align) pitch += align - (pitch % align);
    size_t size = (size_t)(nr + 2 * border) * pitch;
    buf.assign(size + align, 0);
    uint8_t* d = buf.data();
    if ((uintptr_t)d % align) d += align - ((uintptr_t)d % align);
    v.p0 = d + border_padding + (size_t)border * pitch + (size_t)border * es;
    v.nr = nr; v.nc = nc; v.pitch = pitch; v.border = border; v.dtype = dtype; v.ch = ch;
  }
};

// ---------------------------------------------------------------------------------------------
// pixel_wise (vpp/core/pixel_wise.hpp:84-105 row loop, :68-81 process_row) with an arithmetic lambda.
// Arithmetic in the promoted type (int for u8/i16/u16, wrap-around for 32-bit ints), converted back to T.
template <class T, class S> inline T apply_op(int op, T a, T b) {
  S x = (S)a, y = (S)b;
  switch (op) {
    case VPP_OP_ADD: return (T)(x + y);
    case VPP_OP_SUB: return (T)(x - y);
    case VPP_OP_MUL: return (T)(x * y);
    case VPP_OP_MIN: return a < b ? a : b;
    case VPP_OP_MAX: return a > b ? a : b;
    default: return (T)(a > b ? x - y : y - x);
  }
}
template <class T, class S> void pixelwise_binary_t(int op, const Img& d, const Img& a, const Img& b) {
  const int n = d.nc * d.ch;
  ORC_PRAGMA(omp parallel for)
  for (int r = 0; r < d.nr; r++) {  // pixel_wise.hpp:90-92
    T* o = d.row<T>(r); const T* x = a.row<T>(r); const T* y = b.row<T>(r);
    for (int c = 0; c < n; c++) o[c] = apply_op<T, S>(op, x[c], y[c]);  // pixel_wise.hpp:71-72
  }
}

// relative_access R x C mean (benchmarks/box_5x5_filter2.cc:73-80; examples/box_filter.cc:23-32).
template <class T, class S> void box_filter_t(const Img& d, const Img& s, int R, int C) {
  const int ch = d.ch, hr = R / 2, hc = C / 2;
  const int div = R * C;
  ORC_PRAGMA(omp parallel for)
  for (int r = 0; r < d.nr; r++) {
    T* o = d.row<T>(r);
    for (int c = 0; c < d.nc; c++)
      for (int k = 0; k < ch; k++) {
        S sum = 0;
        for (int dr = -hr; dr <= hr; dr++) {  // nbh(i,j) = line[dr][col+dc], relative_accessor.hh:28
          const T* l = s.row<T>(r + dr);
          for (int dc = -hc; dc <= hc; dc++) sum += (S)l[(c + dc) * ch + k];
        }
        o[c * ch + k] = (T)(sum / div);
      }
  }
}

// fill_border_* (vpp/core/fill.hh:31-122). Pixel = es bytes, copied bytewise.
void fill_border_generic(const Img& im, int es, int mode, const void* value) {
  const int b = im.border, nr = im.nr, nc = im.nc;
  auto px = [&](int r, int c) { return im.p0 + (ptrdiff_t)r * im.pitch + (ptrdiff_t)c * es; };
  // The reference fills eight regions one after the other (fill.hh:56-82: corners 1 3 6 8, then edges 2 7 4 5), each a serial
  // row-major pixel_wise.  The order only shows when border > nrows or border > ncols: a mirrored source position then lies in
  // another region of the border, which holds that region's new pixels if it was filled earlier and its old bytes otherwise.
  struct Region { int r0, r1, c0, c1; };
  const Region regions[8] = {{-b, -1, -b, -1}, {-b, -1, nc, nc + b - 1}, {nr, nr + b - 1, -b, -1}, {nr, nr + b - 1, nc, nc + b - 1},
                             {-b, -1, 0, nc - 1}, {nr, nr + b - 1, 0, nc - 1}, {0, nr - 1, -b, -1}, {0, nr - 1, nc, nc + b - 1}};
  for (const Region& g : regions)
    for (int r = g.r0; r <= g.r1; r++)
      for (int c = g.c0; c <= g.c1; c++) {
        const uint8_t* src;
        if (mode == VPP_BORDER_VALUE) src = (const uint8_t*)value;
        else if (mode == VPP_BORDER_MIRROR) {  // fill.hh:60-83: (-k) <- (k-1), (n-1+k) <- (n-k)
          int sr = r < 0 ? -r - 1 : (r >= nr ? 2 * nr - r - 1 : r);
          int sc = c < 0 ? -c - 1 : (c >= nc ? 2 * nc - c - 1 : c);
          src = px(sr, sc);
        } else {  // closest, fill.hh:86-122
          int sr = r < 0 ? 0 : (r >= nr ? nr - 1 : r);
          int sc = c < 0 ? 0 : (c >= nc ? nc - 1 : c);
          src = px(sr, sc);
        }
        memcpy(px(r, c), src, es);
      }
}

// antialiasing_lowpass_filter (vpp/core/pyramid.hh:12-59).  T component type, S = plus_promotion.
// `in` needs border >= 2 (columns) filled.  Result written to out rows 0..nr-1.
template <class T, class S> void lowpass5_t(const Img& in, const Img& out) {
  const int nr = in.nr, nc = in.nc, ch = in.ch;
  OwnedImg tmp(nr, nc, in.dtype, ch, 2);  // pyramid.hh:15
  const int n = nc * ch;
  ORC_PRAGMA(omp parallel for)
  for (int r = 0; r < nr; r++) {  // pyramid.hh:20-34
    const T* i = in.row<T>(r); T* o = tmp.v.row<T>(r);
    for (int c = 0; c < n; c++)
      o[c] = (T)((1 * (S)i[c - 2 * ch] + 4 * (S)i[c - ch] + 6 * (S)i[c] + 4 * (S)i[c + ch] + 1 * (S)i[c + 2 * ch]) / 16);
  }
  fill_border_generic(tmp.v, dtype_size(in.dtype) * ch, VPP_BORDER_MIRROR, nullptr);  // pyramid.hh:36
  ORC_PRAGMA(omp parallel for)
  for (int r = 0; r < nr; r++) {  // pyramid.hh:39-57
    const T *r1 = tmp.v.row<T>(r - 2), *r2 = tmp.v.row<T>(r - 1), *r3 = tmp.v.row<T>(r), *r4 = tmp.v.row<T>(r + 1), *r5 = tmp.v.row<T>(r + 2);
    T* o = out.row<T>(r);
    for (int c = 0; c < n; c++)
      o[c] = (T)((1 * (S)r1[c] + 4 * (S)r2[c] + 6 * (S)r3[c] + 4 * (S)r4[c] + 1 * (S)r5[c]) / 16);
  }
}

// One step of pyramid::propagate_level0, factor 2 (vpp/core/pyramid.hh:175-182) + subsample2 (:62-81).
template <class T, class S> void pyr_down_t(const Img& next, const Img& prev) {
  OwnedImg tmp(prev.nr, prev.nc, prev.dtype, prev.ch, 3);  // pyramid.hh:179 (zero = canonical for its border, Q4)
  lowpass5_t<T, S>(prev, tmp.v);                           // :180
  const int ch = prev.ch;
  ORC_PRAGMA(omp parallel for)
  for (int r = 0; r < next.nr; r++) {                      // :181 -> :67-79
    T* o = next.row<T>(r); const T* i = tmp.v.row<T>(2 * r);
    for (int c = 0; c < next.nc; c++)
      for (int k = 0; k < ch; k++) o[c * ch + k] = i[2 * c * ch + k];
  }
  fill_border_generic(next, dtype_size(next.dtype) * ch, VPP_BORDER_MIRROR, nullptr);  // :182
}

// ---------------------------------------------------------------------------------------------
// imageNd::linear_interpolate (vpp/core/imageNd.hpp:280-300), per component, float then cast<V>.
template <class T, int CH> inline void interp(const Img& I, float p0, float p1, T* out) {
  int x0 = (int)p0, x1 = (int)p1;
  float a0 = p0 - x0, a1 = p1 - x1;
  const T* l1 = I.row<T>(x0) + x1 * CH;
  const T* l2 = (const T*)((const uint8_t*)l1 + I.pitch);
  float w00 = (1 - a0) * (1 - a1), w10 = a0 * (1 - a1), w01 = (1 - a0) * a1, w11 = a0 * a1;
  for (int k = 0; k < CH; k++) {
    float v = w00 * (float)l1[k] + w10 * (float)l2[k] + w01 * (float)l1[CH + k] + w11 * (float)l2[CH + k];
    out[k] = (T)v;  // vpp::cast<V>: truncation for integer V
  }
}

struct Match { float f0, f1, err; };

// lk_match_point_square_win<WS>::operator() (vpp/algorithms/pyrlk/lk.hh:43-175) when PYRLK, GT=float;
// lk_internals::match (vpp/algorithms/lucas_kanade/lucas_kanade.hpp:12-131) when !PYRLK, GT=int.
template <class GT, bool PYRLK>
Match lk_match(float p0, float p1, float tr0, float tr1, const Img& A, const Img& B, const Img& Ag, int ws,
               float min_ev_th, int max_it, float delta) {
  const int hws = ws / 2;
  float G00 = 0, G01 = 0, G10 = 0, G11 = 0;
  int cpt = 0;
  for (int r = -hws; r <= hws; r++)       // lk.hh:56-72
    for (int c = -hws; c <= hws; c++) {
      float n0 = p0 + (float)r, n1 = p1 + (float)c;
      if (A.has((int)n0, (int)n1)) {
        GT g[2]; interp<GT, 2>(Ag, n0, n1, g);
        float gx = (float)g[0], gy = (float)g[1];
        G00 += gx * gx; G01 += gx * gy; G10 += gx * gy; G11 += gy * gy;
        cpt++;
      }
    }
  // lk.hh:75-81: min |eigenvalue| of G/cpt.  Eigen's general EigenSolver is replaced by the closed form for a
  // symmetric 2x2 (third-party arithmetic); it only feeds this threshold compare.
  {
    float fc = (float)cpt;
    float a = G00 / fc, b = G01 / fc, d = G11 / fc;
    float hm = (a + d) * 0.5f, hd = (a - d) * 0.5f;
    float s = std::sqrt(hd * hd + b * b);
    float e1 = std::fabs(hm + s), e2 = std::fabs(hm - s);
    float min_ev = 99999.f;
    if (e1 < min_ev) min_ev = e1;
    if (e2 < min_ev) min_ev = e2;
    if (min_ev < min_ev_th) return Match{-1.f, -1.f, FLT_MAX};
  }
  // lk.hh:83 G.inverse(): Eigen compute_inverse_size2_helper: invdet = 1/det; [d,-b;-c,a]*invdet
  float det = G00 * G11 - G10 * G01;
  float invdet = 1.f / det;
  float I00 = G11 * invdet, I10 = -G10 * invdet, I01 = -G01 * invdet, I11 = G00 * invdet;

  float v0 = p0 + tr0, v1 = p1 + tr1;  // lk.hh:86-87
  float nk0 = 1.f, nk1 = 1.f;
  std::vector<float> gs((size_t)ws * ws * 2, 0.f);  // zero = canonical for unset entries (Q5)
  std::vector<int> as((size_t)ws * ws, 0);
  {
    int i = 0;
    for (int r = -hws; r <= hws; r++)      // lk.hh:99-112
      for (int c = -hws; c <= hws; c++) {
        float n0 = p0 + (float)r, n1 = p1 + (float)c;
        if (Ag.has((int)n0, (int)n1)) {
          GT g[2]; interp<GT, 2>(Ag, n0, n1, g);
          gs[2 * i] = (float)g[0]; gs[2 * i + 1] = (float)g[1];
          uint8_t a; interp<uint8_t, 1>(A, n0, n1, &a);
          as[i] = (int)a;
        }
        i++;
      }
  }
  for (int k = 0; k <= max_it && std::sqrt(nk0 * nk0 + nk1 * nk1) >= delta; k++) {  // lk.hh:116
    float bk0 = 0.f, bk1 = 0.f;
    int i = 0;
    for (int r = -hws; r <= hws; r++)
      for (int c = -hws; c <= hws; c++) {
        float n0 = p0 + (float)r, n1 = p1 + (float)c;
        if (Ag.has((int)n0, (int)n1)) {
          float m0 = v0 + (float)r, m1 = v1 + (float)c;
          uint8_t b; interp<uint8_t, 1>(B, m0, m1, &b);
          float dt = (float)as[i] - (float)b;  // lk.hh:130
          bk0 += gs[2 * i] * dt; bk1 += gs[2 * i + 1] * dt;
        }
        i++;
      }
    nk0 = I00 * bk0 + I01 * bk1;  // lk.hh:137
    nk1 = I10 * bk0 + I11 * bk1;
    v0 += nk0; v1 += nk1;
    if (!B.has((int)v0, (int)v1)) return Match{0.f, 0.f, FLT_MAX};  // lk.hh:145-146
  }
  float err = 0.f;
  if (PYRLK) {
    const int n = ws * ws;
    float avg = 0.f, stddev = 0.f;         // lk.hh:151-159
    for (int i = 0; i < n; i++) avg += (float)as[i];
    avg /= n;
    for (int i = 0; i < n; i++) stddev += std::abs(avg - (float)as[i]);
    stddev /= n;
    for (int r = -hws; r <= hws; r++)      // lk.hh:161-171
      for (int c = -hws; c <= hws; c++) {
        float m0 = v0 + (float)r, m1 = v1 + (float)c;
        int i = (r + hws) * ws + (c + hws);
        uint8_t b; interp<uint8_t, 1>(B, m0, m1, &b);
        err += std::fabs((float)(as[i] - (int)b));
        cpt++;
      }
    return Match{v0 - p0, v1 - p1, err / (cpt * stddev)};  // lk.hh:173
  } else {
    for (int r = -hws; r <= hws; r++)      // lucas_kanade.hpp:116-126
      for (int c = -hws; c <= hws; c++) {
        float m0 = v0 + (float)r, m1 = v1 + (float)c;
        int i = (r + hws) * ws + (c + hws);
        uint8_t b; interp<uint8_t, 1>(B, m0, m1, &b);
        err += std::fabs((float)(as[i] - (int)b));
        cpt++;
      }
    return Match{v0 - p0, v1 - p1, err / (cpt)};  // lucas_kanade.hpp:128
  }
}

// ---------------------------------------------------------------------------------------------
// FAST-9.
inline bool fast9_check_code(unsigned code32) {  // fast.hpp:25-34
  uint64_t code48 = code32;
  code48 |= code48 << 32;
  code48 &= code48 << 8;
  code48 &= code48 << 4;
  code48 &= code48 << 2;
  return (code48 & (code48 << 2)) != 0;
}
// Ring offsets (dr,dc) for a0..a15.  Reference = as sampled by fast_detector9_simd (fast.hpp:327-465): a4 and a12
// come from row r-3 (fast.hpp:367-368).  Corrected = the ring of is_fast9_keypoint / fast9_score (fast.hpp:52-74,88-109).
const int kRingRef[16][2] = {{-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}, {-3, 3}, {1, 3}, {2, 2}, {3, 1},
                             {3, 0}, {3, -1}, {2, -2}, {1, -3}, {-3, -3}, {-1, -3}, {-2, -2}, {-3, -1}};
const int kRingTrue[16][2] = {{-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}, {0, 3}, {1, 3}, {2, 2}, {3, 1},
                              {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}};

inline bool nine_contiguous(unsigned m16) {  // >= 9 circularly contiguous set bits among 16
  unsigned x = m16 | (m16 << 16);
  // run of 9: x & x>>1 & ... & x>>8
  unsigned y = x & (x >> 1); y &= y >> 2; y &= y >> 4;  // runs of 8
  y &= x >> 8;                                          // runs of 9
  return (y & 0xFFFFu) != 0;
}
// plane byte of fast_detector9_simd: 0x10 if >=9 contiguous "x > sat(v+th)", 0x01 if >=9 contiguous "x < sat(v-th)".
inline int fast9_planes(const Img& A, int r, int c, int th, const int ring[16][2]) {
  int v = A.row<uint8_t>(r)[c];
  int hi = std::min(255, v + th), lo = std::max(0, v - th);  // u_adds / u_subs, fast.hpp:322-324
  unsigned mb = 0, md = 0;
  for (int i = 0; i < 16; i++) {
    int x = A.row<uint8_t>(r + ring[i][0])[c + ring[i][1]];
    if (x > hi) mb |= 1u << i;  // check(), fast.hpp:120-126
    if (x < lo) md |= 1u << i;
  }
  return (nine_contiguous(mb) ? 0x10 : 0) | (nine_contiguous(md) ? 0x01 : 0);
}
inline int fast9_score_at(const Img& A, int r, int c, int th) {  // fast.hpp:38-77
  int v = A.row<uint8_t>(r)[c];
  int sum_inf = 0, sum_sup = 0;
  for (int i = 0; i < 16; i++) {
    int a = A.row<uint8_t>(r + kRingTrue[i][0])[c + kRingTrue[i][1]];
    int diff = v - a;
    if (diff < -th) sum_inf -= diff;
    else if (diff > th) sum_sup += diff;
  }
  return std::max(sum_sup, sum_inf);
}

// ---------------------------------------------------------------------------------------------
// semi-dense optical flow helpers.
inline int sad_distance(const Img& i1, const Img& i2, int ar, int ac, int br, int bc, int winsize, int th) {
  // semi_dense_optical_flow.hpp:18-42
  int err = 0;
  const uint8_t* row1 = i1.row<uint8_t>(ar - winsize / 2) + (ac - winsize / 2);
  const uint8_t* row2 = i2.row<uint8_t>(br - winsize / 2) + (bc - winsize / 2);
  for (int r = 0; r < winsize && err <= th; r++) {
    int err2 = 0;
    for (int c = 0; c < winsize; c++) err2 += std::abs((int)row1[c] - (int)row2[c]);
    err += err2;
    row1 += i1.pitch; row2 += i2.pitch;
  }
  return err;
}
struct GdMatch { int f0, f1, distance; };
template <class D>
GdMatch gradient_descent_match(int p0, int p1, int pr0, int pr1, D distance, int max_iteration) {
  // gradient_descent.hh:10-89, tables verbatim
  int m0 = pr0, m1 = pr1;
  int match_distance = distance(p0, p1, pr0, pr1, INT_MAX);
  unsigned match_i = 8;
  static const int c8_it[9][2] = {{6, 3}, {0, 3}, {0, 5}, {2, 5}, {2, 7}, {4, 7}, {4, 1}, {6, 1}, {0, 0}};
  static const int c8[8][2] = {{-1, 1}, {0, 1}, {1, 1}, {-1, 0}, {1, 0}, {-1, -1}, {0, -1}, {1, -1}};
  for (int search = 0; search < max_iteration; search++) {
    int i = c8_it[match_i][0];
    int end = c8_it[match_i][1];
    {
      int n0 = pr0 + c8[i][0], n1 = pr1 + c8[i][1];
      int d = distance(p0, p1, n0, n1, match_distance);
      if (d < match_distance) { m0 = n0; m1 = n1; match_i = i; match_distance = d; }
      i = (i + 1) & 7;
    }
    for (; i != end; i = (i + 1) & 7) {
      int n0 = pr0 + c8[i][0], n1 = pr1 + c8[i][1];
      int d = distance(p0, p1, n0, n1, match_distance);
      if (d < match_distance) { m0 = n0; m1 = n1; match_i = i; match_distance = d; }
    }
    if (pr0 == m0 && pr1 == m1) break;
    pr0 = m0; pr1 = m1;
  }
  return GdMatch{m0 - p0, m1 - p1, match_distance};
}

template <class F> int dispatch_scalar(int dtype, F f) {
  switch (dtype) {
    case VPP_U8: f((uint8_t)0, (int)0); return VPP_OK;
    case VPP_I8: f((int8_t)0, (int)0); return VPP_OK;
    case VPP_U16: f((uint16_t)0, (int)0); return VPP_OK;
    case VPP_I16: f((int16_t)0, (int)0); return VPP_OK;
    case VPP_I32: f((int32_t)0, (int32_t)0); return VPP_OK;
    case VPP_U32: f((uint32_t)0, (uint32_t)0); return VPP_OK;
    case VPP_F32: f((float)0, (float)0); return VPP_OK;
  }
  return VPP_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

int orc_num_threads(void) {
#ifdef ORC_OMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int orc_pixelwise_binary(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b) {
  Img d(dst), x(a), y(b);
  if (d.dtype == VPP_I32) {  // signed wrap-around done in unsigned arithmetic (what the compiled reference does)
    if (op == VPP_OP_MIN || op == VPP_OP_MAX || op == VPP_OP_ABSDIFF) { pixelwise_binary_t<int32_t, int64_t>(op, d, x, y); return VPP_OK; }
    pixelwise_binary_t<uint32_t, uint32_t>(op, d, x, y); return VPP_OK;
  }
  return dispatch_scalar(d.dtype, [&](auto t, auto s) { pixelwise_binary_t<decltype(t), decltype(s)>(op, d, x, y); });
}

int orc_copy(const vpp_image_desc* dst, const vpp_image_desc* src, int with_border) {
  Img d(dst), s(src);
  int es = dtype_size(d.dtype) * d.ch;
  int b = with_border ? s.border : 0;  // copy.hh:22-27 iterates src.domain_with_border()
  for (int r = -b; r < d.nr + b; r++)
    memcpy(d.p0 + (ptrdiff_t)r * d.pitch - (ptrdiff_t)b * es, s.p0 + (ptrdiff_t)r * s.pitch - (ptrdiff_t)b * es, (size_t)(d.nc + 2 * b) * es);
  return VPP_OK;
}

int orc_fill(const vpp_image_desc* img, const void* value, int with_border) {
  Img d(img);
  int es = dtype_size(d.dtype) * d.ch;
  int b = with_border ? d.border : 0;
  for (int r = -b; r < d.nr + b; r++)
    for (int c = -b; c < d.nc + b; c++) memcpy(d.p0 + (ptrdiff_t)r * d.pitch + (ptrdiff_t)c * es, value, es);
  return VPP_OK;
}

int orc_box_filter(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C) {
  Img d(dst), s(src);
  if (s.border < std::max(R, C) / 2) return VPP_ERR_BORDER_TOO_SMALL;
  if (d.dtype == VPP_I32) { box_filter_t<int32_t, int32_t>(d, s, R, C); return VPP_OK; }
  return dispatch_scalar(d.dtype, [&](auto t, auto sx) { box_filter_t<decltype(t), decltype(sx)>(d, s, R, C); });
}

int orc_fill_border(const vpp_image_desc* img, int mode, const void* value) {
  Img d(img);
  fill_border_generic(d, dtype_size(d.dtype) * d.ch, mode, value);
  return VPP_OK;
}

int orc_lowpass5(const vpp_image_desc* out, const vpp_image_desc* in) {
  Img o(out), i(in);
  if (i.border < 2) return VPP_ERR_BORDER_TOO_SMALL;
  return dispatch_scalar(i.dtype, [&](auto t, auto s) { lowpass5_t<decltype(t), decltype(s)>(i, o); });
}

int orc_pyr_down(const vpp_image_desc* next, const vpp_image_desc* prev) {
  Img n(next), p(prev);
  if (p.border < 2) return VPP_ERR_BORDER_TOO_SMALL;
  if (n.nr != 1 + p.nr / 2 || n.nc != 1 + p.nc / 2) return VPP_ERR_INVALID_ARG;
  return dispatch_scalar(p.dtype, [&](auto t, auto s) { pyr_down_t<decltype(t), decltype(s)>(n, p); });
}

int orc_scharr(const vpp_image_desc* out, const vpp_image_desc* in) {  // scharr.hh:46-87
  Img o(out), i(in);
  if (i.border < 1) return VPP_ERR_BORDER_TOO_SMALL;
  if (i.dtype != VPP_U8 || i.ch != 1 || o.ch != 2) return VPP_ERR_UNSUPPORTED;
  const int nr = o.nr, nc = o.nc;
  if (o.dtype == VPP_F32) {
    ORC_PRAGMA(omp parallel for)
    for (int r = 0; r < nr; r++) {
      float* orow = o.row<float>(r);
      const uint8_t *row1 = i.row<uint8_t>(r - 1), *row2 = i.row<uint8_t>(r), *row3 = i.row<uint8_t>(r + 1);
      for (int c = 0; c < nc; c++) {
        typedef float V;
        orow[2 * c] = (3 * V(row3[c - 1]) + 10 * V(row3[c]) + 3 * V(row3[c + 1]) - 3 * V(row1[c - 1]) - 10 * V(row1[c]) - 3 * V(row1[c + 1])) / 32.f;
        orow[2 * c + 1] = (3 * V(row1[c + 1]) + 10 * V(row2[c + 1]) + 3 * V(row3[c + 1]) - 3 * V(row1[c - 1]) - 10 * V(row2[c - 1]) - 3 * V(row3[c - 1])) / 32.f;
      }
    }
    return VPP_OK;
  }
  if (o.dtype == VPP_I32) {  // V = int: integer arithmetic, then /32.f, then conversion to int (lucas_kanade.hpp:151-155)
    ORC_PRAGMA(omp parallel for)
    for (int r = 0; r < nr; r++) {
      int32_t* orow = o.row<int32_t>(r);
      const uint8_t *row1 = i.row<uint8_t>(r - 1), *row2 = i.row<uint8_t>(r), *row3 = i.row<uint8_t>(r + 1);
      for (int c = 0; c < nc; c++) {
        typedef int V;
        orow[2 * c] = (int32_t)((3 * V(row3[c - 1]) + 10 * V(row3[c]) + 3 * V(row3[c + 1]) - 3 * V(row1[c - 1]) - 10 * V(row1[c]) - 3 * V(row1[c + 1])) / 32.f);
        orow[2 * c + 1] = (int32_t)((3 * V(row1[c + 1]) + 10 * V(row2[c + 1]) + 3 * V(row3[c + 1]) - 3 * V(row1[c - 1]) - 10 * V(row2[c - 1]) - 3 * V(row3[c - 1])) / 32.f);
      }
    }
    return VPP_OK;
  }
  return VPP_ERR_UNSUPPORTED;
}

int orc_is_fast9_keypoint(const vpp_image_desc* src, int r, int c, int th) {  // fast.hpp:80-112
  Img A(src);
  int v = A.row<uint8_t>(r)[c];
  auto f = [&](int a) -> int { return ((a > v + th) << 1) ^ (a < v - th); };
  auto n = [&](int dr, int dc) { return (int)A.row<uint8_t>(r + dr)[c + dc]; };
  unsigned x = (f(n(3, -1)) << 20) + (f(n(3, 0)) << 18) + (f(n(3, +1)) << 16) + (f(n(2, -2)) << 22) + (f(n(2, 2)) << 14) +
               (f(n(1, -3)) << 24) + (f(n(1, 3)) << 12) + (f(n(0, -3)) << 26) + (f(n(0, 3)) << 10) + (f(n(-1, -3)) << 28) +
               (f(n(-1, 3)) << 8) + (f(n(-2, -2)) << 30) + (f(n(-2, 2)) << 6) + f(n(-3, -1)) + (f(n(-3, 0)) << 2) + (f(n(-3, 1)) << 4);
  return fast9_check_code(x) ? 1 : 0;
}

int orc_fast9_scores(const vpp_image_desc* src, int th, const int32_t* rc, int n, int32_t* out_scores) {
  Img A(src);
  ORC_PRAGMA(omp parallel for)
  for (int i = 0; i < n; i++) out_scores[i] = fast9_score_at(A, rc[2 * i], rc[2 * i + 1], th);  // fast.hpp:643-652
  return VPP_OK;
}

int orc_fast9_detect(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size, int compat,
                     int32_t* out_rc, int32_t* out_scores, int capacity, int* count) {
  Img A(src);
  if (A.border < 3) return VPP_ERR_BORDER_TOO_SMALL;  // fast.hpp:937-938
  Img M; if (mask) M = Img(mask);
  const int nr = A.nr, nc = A.nc;
  const int(*ring)[2] = compat == VPP_FAST9_REFERENCE ? kRingRef : kRingTrue;
  // fast_detector9_simd (fast.hpp:254-508), serial row-major order.
  std::vector<std::vector<int32_t>> rows(nr);
  ORC_PRAGMA(omp parallel for schedule(dynamic, 16))
  for (int r = 0; r < nr; r++)
    for (int c = 0; c < nc; c++) {
      int possible = mask ? M.row<uint8_t>(r)[c] : 255;  // fast.hpp:310-317
      if (!possible) continue;
      possible &= fast9_planes(A, r, c, th, ring);
      if (possible) rows[r].push_back(c);
    }
  std::vector<int32_t> kps;
  for (int r = 0; r < nr; r++) for (int c : rows[r]) { kps.push_back(r); kps.push_back(c); }
  int nk = (int)kps.size() / 2;
  std::vector<int32_t> res, res_scores;
  if (mode == VPP_FAST9_RAW) {  // fast.hpp:662-674
    res = kps; res_scores.resize(nk);
    orc_fast9_scores(src, th, kps.data(), nk, res_scores.data());
  } else {
    OwnedImg S(nr, nc, VPP_U8, 1, 1);  // fast.hpp:685-686 (zero-filled with border)
    ORC_PRAGMA(omp parallel for)
    for (int i = 0; i < nk; i++) {  // fast.hpp:688-694
      int r = kps[2 * i], c = kps[2 * i + 1];
      S.v.row<uint8_t>(r)[c] = (uint8_t)(fast9_score_at(A, r, c, th) / 16);
    }
    if (mode == VPP_FAST9_BLOCKWISE) {  // fast.hpp:763-789
      if (block_size <= 0) return VPP_ERR_INVALID_ARG;
      for (int r = 0; r < nr; r += block_size)
        for (int c = 0; c < nc; c += block_size) {
          int pm0 = 0, pm1 = 0; unsigned vmax = 0;
          for (int br = 0; br < block_size; br++)
            for (int bc = c; bc < c + block_size; bc++)
              if (r + br < nr && bc < nc) {
                unsigned v = S.v.row<uint8_t>(r + br)[bc];
                if (v > vmax) { vmax = v; pm0 = br; pm1 = bc; }
              }
          if (vmax > 0) { res.push_back(r + pm0); res.push_back(pm1); }
        }
    } else {  // local maxima, fast.hpp:896-927
      for (int i = 0; i < nk; i++) {
        int r = kps[2 * i], c = kps[2 * i + 1];
        auto nn = [&](int dr, int dc) { return (unsigned)S.v.row<uint8_t>(r + dr)[c + dc]; };
        unsigned a = nn(0, 0);
        int is_max = 1;
        is_max &= a > nn(-1, -1); is_max &= a > nn(-1, 0); is_max &= a > nn(-1, 1); is_max &= a > nn(0, -1);
        is_max &= a > nn(0, 1); is_max &= a > nn(1, -1); is_max &= a > nn(1, 0); is_max &= a > nn(1, 1);
        if (is_max) { res.push_back(r); res.push_back(c); }
      }
    }
    res_scores.resize(res.size() / 2);
    for (size_t i = 0; i < res_scores.size(); i++) res_scores[i] = S.v.row<uint8_t>(res[2 * i])[res[2 * i + 1]];  // fast.hpp:698-704
  }
  int n = (int)res.size() / 2;
  *count = n;
  int m = std::min(n, capacity);
  if (out_rc) memcpy(out_rc, res.data(), (size_t)m * 2 * sizeof(int32_t));
  if (out_scores) memcpy(out_scores, res_scores.data(), (size_t)m * sizeof(int32_t));
  return n > capacity ? VPP_ERR_CAPACITY : VPP_OK;
}

int orc_linear_interpolate(const vpp_image_desc* img, float pr, float pc, float* out) {
  Img I(img);
  if (I.dtype == VPP_U8 && I.ch == 1) { uint8_t v; interp<uint8_t, 1>(I, pr, pc, &v); out[0] = v; return VPP_OK; }
  if (I.dtype == VPP_F32 && I.ch == 2) { float v[2]; interp<float, 2>(I, pr, pc, v); out[0] = v[0]; out[1] = v[1]; return VPP_OK; }
  if (I.dtype == VPP_I32 && I.ch == 2) { int32_t v[2]; interp<int32_t, 2>(I, pr, pc, v); out[0] = (float)v[0]; out[1] = (float)v[1]; return VPP_OK; }
  if (I.dtype == VPP_F32 && I.ch == 1) { float v; interp<float, 1>(I, pr, pc, &v); out[0] = v; return VPP_OK; }
  return VPP_ERR_UNSUPPORTED;
}

int orc_pyrlk_match(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                    vpp_keypoint_f32* kps, int n, int winsize, float min_ev, float max_err, int max_iterations,
                    float convergence_delta, int min_scale, float* out_dist) {
  std::vector<Img> P, G, N;
  for (int l = 0; l < nlevels; l++) { P.emplace_back(prev + l); G.emplace_back(grad + l); N.emplace_back(next + l); }
  if (G[0].dtype != VPP_F32 || G[0].ch != 2 || P[0].dtype != VPP_U8) return VPP_ERR_UNSUPPORTED;
  const float factor = 2.f;
  ORC_PRAGMA(omp parallel for schedule(dynamic, 16))
  for (int i = 0; i < n; i++) {  // pyrlk_match.hh:24-51
    vpp_keypoint_f32& kp = kps[i];
    if (out_dist) out_dist[i] = 0.f;
    if (!(kp.age > 0)) continue;
    float tr0 = 0.f, tr1 = 0.f, dist = 0.f;
    for (int S = nlevels - 1; S >= min_scale; S--) {
      tr0 *= factor; tr1 *= factor;
      float sc = (float)std::pow(2, S);  // Eigen promotes the double scalar to float
      Match m = lk_match<float, true>(kp.pos_r / sc, kp.pos_c / sc, tr0, tr1, P[S], N[S], G[S], winsize, min_ev, max_iterations, convergence_delta);
      if (m.err < max_err) { tr0 = m.f0; tr1 = m.f1; }
      dist = m.err;
    }
    float nr0 = kp.pos_r + tr0, nr1 = kp.pos_c + tr1;
    if (out_dist) out_dist[i] = dist;
    if (dist > max_err || !P[0].has((int)nr0, (int)nr1)) kp.age = 0;  // keypoints.remove(i) -> die()
    else { kp.vel_r = nr0 - kp.pos_r; kp.vel_c = nr1 - kp.pos_c; kp.pos_r = nr0; kp.pos_c = nr1; kp.age++; }  // keypoint_container.hpp:156-167
  }
  return VPP_OK;
}

int orc_lucas_kanade(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                     const float* pts, const float* prediction, int n, int winsize, int min_ev, int niterations,
                     int delta, float* out_flow, float* out_dist) {
  std::vector<Img> P, G, N;
  for (int l = 0; l < nlevels; l++) { P.emplace_back(prev + l); G.emplace_back(grad + l); N.emplace_back(next + l); }
  if (G[0].dtype != VPP_I32 || G[0].ch != 2 || P[0].dtype != VPP_U8) return VPP_ERR_UNSUPPORTED;
  for (int i = 0; i < n; i++) {  // lucas_kanade.hpp:159-183 (serial in the reference)
    float k0 = pts[2 * i], k1 = pts[2 * i + 1];
    float d = float(std::pow(2, nlevels));
    float tr0 = (prediction ? prediction[2 * i] : 0.f) / d, tr1 = (prediction ? prediction[2 * i + 1] : 0.f) / d;
    float dist = 0.f;
    for (int S = nlevels - 1; S >= 0; S--) {
      tr0 *= 2.f; tr1 *= 2.f;
      int sc = int(std::pow(2, S));
      Match m = lk_match<int32_t, false>(k0 / sc, k1 / sc, tr0, tr1, P[S], N[S], G[S], winsize, (float)min_ev, niterations, (float)delta);
      tr0 = m.f0; tr1 = m.f1; dist = m.err;
    }
    out_flow[2 * i] = tr0; out_flow[2 * i + 1] = tr1;
    if (out_dist) out_dist[i] = dist;
  }
  return VPP_OK;
}

int orc_semi_dense_optical_flow(const vpp_image_desc* i1d, const vpp_image_desc* i2d, const int32_t* kps, int n, int winsize,
                                int nscales, int min_scale, int propagation_niters, int patchsize, int32_t* out_pos,
                                int32_t* out_dist, uint8_t* out_valid) {
  Img in1(i1d), in2(i2d);
  // semi_dense_optical_flow.hpp:68-74
  std::vector<OwnedImg> flow(nscales), mark(nscales), dmap(nscales), P1(nscales), P2(nscales);
  {
    int fr = in1.nr / patchsize, fc = in1.nc / patchsize, ir = in1.nr, ic = in1.nc;
    for (int s = 0; s < nscales; s++) {
      flow[s].alloc(fr, fc, VPP_I32, 2, nscales); mark[s].alloc(fr, fc, VPP_U8, 1, nscales); dmap[s].alloc(fr, fc, VPP_I32, 1, nscales);
      P1[s].alloc(ir, ic, VPP_U8, 1, 2 * winsize); P2[s].alloc(ir, ic, VPP_U8, 1, 2 * winsize);
      fr = 1 + fr / 2; fc = 1 + fc / 2; ir = 1 + ir / 2; ic = 1 + ic / 2;  // pyramid.hh:154
    }
  }
  auto build = [&](std::vector<OwnedImg>& P, const Img& in) {  // pyramid::update, pyramid.hh:194-198
    for (int r = 0; r < in.nr; r++) memcpy(P[0].v.row<uint8_t>(r), in.row<uint8_t>(r), in.nc);
    fill_border_generic(P[0].v, 1, VPP_BORDER_MIRROR, nullptr);
    for (int s = 1; s < nscales; s++) pyr_down_t<uint8_t, int>(P[s].v, P[s - 1].v);
  };
  build(P1, in1); build(P2, in2);

  for (int scale = nscales - 1; scale >= min_scale; scale--) {  // :92
    int scale_div = (int)std::pow(2, scale);
    const Img i1 = P1[scale].v, i2 = P2[scale].v;
    auto distance = [&](int a0, int a1, int b0, int b1, int max_distance) {  // :102-108
      if (i1.has(a0, a1) && i2.has(b0, b1)) return sad_distance(i1, i2, a0, a1, b0, b1, winsize, max_distance);
      return INT_MAX;
    };
    Img fm = flow[scale].v, mk = mark[scale].v, dm = dmap[scale].v;
    {  // fill_with_border(flow_map_mark, 0), :111
      uint8_t z = 0; vpp_image_desc d{mk.p0, mk.nr, mk.nc, mk.pitch, mk.border, VPP_U8, 1}; orc_fill(&d, &z, 1);
    }
    for (int i = 0; i < n; i++) {  // :114-143 (serial order = canonical)
      int p0 = kps[2 * i] / scale_div, p1 = kps[2 * i + 1] / scale_div;
      int pf0 = p0 / patchsize, pf1 = p1 / patchsize;
      if (!mk.row<uint8_t>(pf0)[pf1]) {
        int pfm0 = p0 / (2 * patchsize), pfm1 = p1 / (2 * patchsize);
        int pr0 = p0, pr1 = p1;
        if (scale < nscales - 1 && mark[scale + 1].v.row<uint8_t>(pfm0)[pfm1]) {
          const int32_t* f = flow[scale + 1].v.row<int32_t>(pfm0) + 2 * pfm1;
          pr0 = p0 + f[0] * 2; pr1 = p1 + f[1] * 2;
        }
        mk.row<uint8_t>(pf0)[pf1] = 1;
        GdMatch m = gradient_descent_match(p0, p1, pr0, pr1, distance, 5);
        int32_t* f = fm.row<int32_t>(pf0) + 2 * pf1;
        f[0] = m.f0; f[1] = m.f1;
        dm.row<int32_t>(pf0)[pf1] = m.distance;
        mk.row<uint8_t>(pf0)[pf1] = 2;
      }
    }
    for (int Ki = 0; Ki < propagation_niters; Ki++) {  // :146-201
      auto loop_body = [&](int r, int c) {
        int pf0 = r / patchsize, pf1 = c / patchsize;
        if (!mk.row<uint8_t>(pf0)[pf1]) return;
        int32_t* fpf = fm.row<int32_t>(pf0) + 2 * pf1;
        int prev0 = fpf[0], prev1 = fpf[1];
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            if (!dr && !dc) continue;
            int q0 = pf0 + dr, q1 = pf1 + dc;
            if (!(fm.has(q0, q1) && mk.row<uint8_t>(q0)[q1])) continue;
            const int32_t* fq = fm.row<int32_t>(q0) + 2 * q1;
            auto inorm = [](int a, int b) { return (int)std::sqrt((double)(a * a + b * b)); };  // Eigen int norm()
            if (inorm(fpf[0] - fq[0], fpf[1] - fq[1]) > 2 && inorm(prev0 - fq[0], prev1 - fq[1]) > 2) {
              int d1 = dm.row<int32_t>(pf0)[pf1];
              int d2 = distance(r, c, r + fq[0], c + fq[1], INT_MAX);
              if (d2 < d1) {
                GdMatch m = gradient_descent_match(r, c, r + fq[0], c + fq[1], distance, 5);
                if (m.distance < d1) {
                  mk.row<uint8_t>(pf0)[pf1] = 1;
                  fpf[0] = m.f0; fpf[1] = m.f1;
                  dm.row<int32_t>(pf0)[pf1] = m.distance;
                }
              }
            }
          }
      };
      if (Ki % 2) {
        for (int r = 0; r < i1.nr; r += patchsize) for (int c = 0; c < i1.nc; c += patchsize) loop_body(r, c);
      } else {
        for (int r = i1.nr - 1; r >= 0; r -= patchsize) for (int c = i1.nc - 1; c >= 0; c -= patchsize) loop_body(r, c);
      }
    }
  }
  int ms = (int)std::pow(2, min_scale);
  for (int pi = 0; pi < n; pi++) {  // :205-212
    int q0 = kps[2 * pi] / int(patchsize * std::pow(2, min_scale)), q1 = kps[2 * pi + 1] / int(patchsize * std::pow(2, min_scale));
    const Img& mk = mark[min_scale].v;
    out_valid[pi] = 0; out_pos[2 * pi] = kps[2 * pi]; out_pos[2 * pi + 1] = kps[2 * pi + 1]; out_dist[pi] = 0;
    if (mk.has(q0, q1) && mk.row<uint8_t>(q0)[q1]) {
      const int32_t* f = flow[min_scale].v.row<int32_t>(q0) + 2 * q1;
      out_pos[2 * pi] = kps[2 * pi] + f[0] * ms; out_pos[2 * pi + 1] = kps[2 * pi + 1] + f[1] * ms;
      out_dist[pi] = dmap[min_scale].v.row<int32_t>(q0)[q1];
      out_valid[pi] = 1;
    }
  }
  return VPP_OK;
}

// rgb_to_graylevel: `o = (i[0] + i[1] + i[2]) / 3` (colorspace_conversions.hh:10-20) mapped over domain_with_border
// (:27-31; the 4-channel overload :41-45 takes segment<3>(0)).  uchar components promote to int, the quotient is stored
// back into the uchar.  mirror != 0 restates the ingest chain literally (examples/video_extruder.cc:46-48):
// clone(frame, _border = b) -> fill_border_mirror -> rgb_to_graylevel.
int orc_rgb_to_graylevel(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror) {
  Img d(dst), s(src);
  if (d.dtype != VPP_U8 || d.ch != 1 || s.dtype != VPP_U8 || (s.ch != 3 && s.ch != 4)) return VPP_ERR_UNSUPPORTED;
  if (d.nr != s.nr || d.nc != s.nc) return VPP_ERR_INVALID_ARG;
  OwnedImg tmp;
  int ext = d.border < s.border ? d.border : s.border;
  if (mirror) {
    ext = d.border;
    if (ext > d.nr || ext > d.nc) return VPP_ERR_INVALID_ARG;
    tmp.alloc(s.nr, s.nc, VPP_U8, s.ch, ext);
    for (int r = 0; r < s.nr; r++) std::memcpy(tmp.v.row<uint8_t>(r), s.row<uint8_t>(r), (size_t)s.nc * s.ch);  // clone: copy.hh:10-17
    vpp_image_desc td = {tmp.v.p0, tmp.v.nr, tmp.v.nc, tmp.v.pitch, tmp.v.border, tmp.v.dtype, tmp.v.ch};
    int st = orc_fill_border(&td, VPP_BORDER_MIRROR, nullptr);
    if (st != VPP_OK) return st;
    s = tmp.v;
  }
  for (int r = -ext; r < d.nr + ext; r++) {
    const uint8_t* i = s.row<uint8_t>(r);
    uint8_t* o = d.row<uint8_t>(r);
    for (int c = -ext; c < d.nc + ext; c++) o[c] = (uint8_t)((i[c * s.ch] + i[c * s.ch + 1] + i[c * s.ch + 2]) / 3);
  }
  return VPP_OK;
}

// video_extruder.hpp:101-110: fill_with_border(mask, 1), then a 2s x 2s square of zeros per keypoint (every container
// entry, dead ones included).  The reference does not clip (keypoints lie inside the domain and border == spacing);
// writes outside the mask's border are dropped here.
int orc_keypoint_mask(const vpp_image_desc* mask, const int32_t* rc, int n, int spacing) {
  Img m(mask);
  if (m.dtype != VPP_U8 || m.ch != 1) return VPP_ERR_UNSUPPORTED;
  if (spacing <= 0 || n < 0) return VPP_ERR_INVALID_ARG;
  for (int r = -m.border; r < m.nr + m.border; r++) std::memset(m.row<uint8_t>(r) - m.border, 1, (size_t)m.nc + 2 * m.border);
  for (int i = 0; i < n; i++) {
    const int r = rc[2 * i], c = rc[2 * i + 1];
    for (int dr = -spacing; dr < spacing; dr++)
      for (int dc = -spacing; dc < spacing; dc++) {
        const int y = r + dr, x = c + dc;
        if (y >= -m.border && y < m.nr + m.border && x >= -m.border && x < m.nc + m.border) m.row<uint8_t>(y)[x] = 0;
      }
  }
  return VPP_OK;
}

// The merge step of video_extruder_update (video_extruder.hpp:60-84), restated literally: an index image of (nrows / spacing) x
// (ncols / spacing) cells with border 1 filled with -1, the keypoints visited in container order.  pos / age describe the container AFTER the
// flow callbacks (:48-53); removed[i] = 1 where the loop called keypoints.remove(i).  Pinned to the reference end to end by the 4K
// video_extruder run (tests/test_gpu_video_extruder.py), since the reference has no separate entry point for this loop.
int orc_keypoint_merge(const int32_t* pos_rc, const int32_t* age_in, int n, int nrows, int ncols, int spacing, uint8_t* removed) {
  if (n < 0 || nrows <= 0 || ncols <= 0 || spacing <= 0) return VPP_ERR_INVALID_ARG;
  const int gr = nrows / spacing, gc = ncols / spacing;
  OwnedImg idx(gr > 0 ? gr : 1, gc > 0 ? gc : 1, VPP_I32, 1, 1);
  for (int r = -1; r <= idx.v.nr; r++) for (int c = -1; c <= idx.v.nc; c++) idx.v.row<int32_t>(r)[c] = -1;   // fill_with_border(idx, -1)
  std::vector<int> age(age_in, age_in + n);
  for (int i = 0; i < n; i++) removed[i] = 0;
  for (int i = 0; i < n; i++) {
    const int pr = pos_rc[2 * i] / spacing, pc = pos_rc[2 * i + 1] / spacing;   // positions are inside the frame, so the cell is inside the border-1 image
    int32_t& cell = idx.v.row<int32_t>(pr)[pc];
    if (cell >= 0) {
      const int other_age = age[cell];                                            // `auto other = ctx.keypoints[idx(pos)]` is a COPY (:72)
      if (other_age < age[i]) { removed[cell] = 1; age[cell] = 0; cell = i; }
      if (other_age > age[i]) { removed[i] = 1; age[i] = 0; }
    } else cell = i;
  }
  return VPP_OK;
}

// lbp_transform (lbp_transform.hh:6-38): curB[i] = sum of ((neighbour > rows[1][i]) << k), k in row-major neighbour order.
int orc_lbp_transform(const vpp_image_desc* out, const vpp_image_desc* in) {
  Img o(out), a(in);
  if (o.dtype != VPP_U8 || o.ch != 1 || a.dtype != VPP_U8 || a.ch != 1) return VPP_ERR_UNSUPPORTED;
  if (o.nr != a.nr || o.nc != a.nc) return VPP_ERR_INVALID_ARG;
  if (a.border < 1) return VPP_ERR_BORDER_TOO_SMALL;
  for (int r = 0; r < a.nr; r++) {
    uint8_t* curB = o.row<uint8_t>(r);
    const uint8_t* rows[3];
    for (int i = -1; i <= 1; i++) rows[i + 1] = a.row<uint8_t>(r + i);
    for (int i = 0; i < a.nc; i++)
      curB[i] = (uint8_t)(((rows[0][i - 1] > rows[1][i]) << 0) + ((rows[0][i] > rows[1][i]) << 1) + ((rows[0][i + 1] > rows[1][i]) << 2) +
                          ((rows[1][i - 1] > rows[1][i]) << 3) + ((rows[1][i + 1] > rows[1][i]) << 4) +
                          ((rows[2][i - 1] > rows[1][i]) << 5) + ((rows[2][i] > rows[1][i]) << 6) + ((rows[2][i + 1] > rows[1][i]) << 7));
  }
  return VPP_OK;
}

// FAST_internals::fast_detector9(A, B, th) (fast.hpp:511-551) with fast9_check_code (fast.hpp:25-35), restated literally: the
// 2-bit codes of the 16 TRUE ring pixels at the reference's bit positions, doubled to 64 bits, and-shifted by 8, 4, 2 and 2.
int orc_fast9_dense(const vpp_image_desc* out, const vpp_image_desc* in, int th) {
  Img o(out), a(in);
  if (a.dtype != VPP_U8 || a.ch != 1 || o.ch != 1 || (o.dtype != VPP_U8 && o.dtype != VPP_I32)) return VPP_ERR_UNSUPPORTED;
  if (o.nr != a.nr || o.nc != a.nc) return VPP_ERR_INVALID_ARG;
  if (a.border < 3) return VPP_ERR_BORDER_TOO_SMALL;
  static const int pos[16][3] = {{3, -1, 20}, {3, 0, 18}, {3, 1, 16}, {2, -2, 22}, {2, 2, 14}, {1, -3, 24}, {1, 3, 12}, {0, -3, 26},
                                 {0, 3, 10}, {-1, -3, 28}, {-1, 3, 8}, {-2, -2, 30}, {-2, 2, 6}, {-3, -1, 0}, {-3, 0, 2}, {-3, 1, 4}};
  for (int r = 0; r < a.nr; r++)
    for (int c = 0; c < a.nc; c++) {
      const int v = a.row<uint8_t>(r)[c];
      unsigned x = 0;
      for (int k = 0; k < 16; k++) {
        const int n = a.row<uint8_t>(r + pos[k][0])[c + pos[k][1]];
        const unsigned f = (unsigned)(((n > v + th) << 1) ^ (n < v - th));  // fast.hpp:518
        x += f << pos[k][2];
      }
      uint64_t code48 = x;
      code48 |= code48 << 32;
      code48 &= code48 << 8;
      code48 &= code48 << 4;
      code48 &= code48 << 2;
      const bool corner = (code48 & (code48 << 2)) != 0;
      if (o.dtype == VPP_U8) o.row<uint8_t>(r)[c] = corner; else o.row<int32_t>(r)[c] = corner;
    }
  return VPP_OK;
}

// local_maxima_filter (fast.hpp:555-575), in place, the SERIAL form (the reference's test build has no OpenMP, tests/CMakeLists.txt:16):
// rows top to bottom, columns left to right; the four neighbours above / left of a pixel have already been filtered when it is
// compared with them, the other four still hold the input.  The border (>= 1) is read, never written.  Pinned to the reference
// by tests/test_ref_pins_oracle.py.
extern "C++" {
template <class V> static void local_maxima_t(Img a) {
  for (int r = 0; r < a.nr; r++) {
    V *up = a.row<V>(r - 1), *row = a.row<V>(r), *dn = a.row<V>(r + 1);
    for (int c = 0; c < a.nc; c++) {
      const V v = row[c];
      int is_max = 1;
      is_max &= v > up[c - 1]; is_max &= v > up[c]; is_max &= v > up[c + 1];
      is_max &= v > row[c - 1]; is_max &= v > row[c + 1];
      is_max &= v > dn[c - 1]; is_max &= v > dn[c]; is_max &= v > dn[c + 1];
      if (!is_max) row[c] = V(0);
    }
  }
}
}  // extern "C++"
int orc_local_maxima_filter(const vpp_image_desc* img) {
  Img a(img);
  if (a.ch != 1) return VPP_ERR_UNSUPPORTED;
  if (a.border < 1) return VPP_ERR_BORDER_TOO_SMALL;
  switch (a.dtype) {
    case VPP_U8: local_maxima_t<uint8_t>(a); break;
    case VPP_I8: local_maxima_t<int8_t>(a); break;
    case VPP_U16: local_maxima_t<uint16_t>(a); break;
    case VPP_I16: local_maxima_t<int16_t>(a); break;
    case VPP_I32: local_maxima_t<int32_t>(a); break;
    case VPP_U32: local_maxima_t<uint32_t>(a); break;
    case VPP_F32: local_maxima_t<float>(a); break;
    default: return VPP_ERR_UNSUPPORTED;
  }
  return VPP_OK;
}

// blockwise_maxima_filter (fast.hpp:577-614), in place.  PARITY UNPINNED: the reference template does not compile when
// instantiated (it stores &A(r + i, 0) of a const image into V* rows[], fast.hpp:590), so no reference output exists for it.
extern "C++" {
template <class V> static void blockwise_maxima_t(Img a, int bs) {
  for (int r = 0; r < a.nr; r += bs)
    for (int c = 0; c < a.nc; c += bs) {
      int pr = 0, pc = 0; V vmax = 0;
      for (int br = 0; br < bs; br++)
        for (int bc = c; bc < c + bs; bc++)
          if (r + br < a.nr && bc < a.nc) {
            const V v = a.row<V>(r + br)[bc];
            a.row<V>(r + br)[bc] = 0;
            if (v > vmax) { vmax = v; pr = br; pc = bc; }
          }
      if (vmax > 0) a.row<V>(r + pr)[pc] = vmax;
    }
}
}  // extern "C++"
int orc_blockwise_maxima_filter(const vpp_image_desc* img, int bs) {
  Img a(img);
  if (a.ch != 1) return VPP_ERR_UNSUPPORTED;
  if (bs <= 0) return VPP_ERR_INVALID_ARG;
  switch (a.dtype) {
    case VPP_U8: blockwise_maxima_t<uint8_t>(a, bs); break;
    case VPP_I8: blockwise_maxima_t<int8_t>(a, bs); break;
    case VPP_U16: blockwise_maxima_t<uint16_t>(a, bs); break;
    case VPP_I16: blockwise_maxima_t<int16_t>(a, bs); break;
    case VPP_I32: blockwise_maxima_t<int32_t>(a, bs); break;
    case VPP_U32: blockwise_maxima_t<uint32_t>(a, bs); break;
    case VPP_F32: blockwise_maxima_t<float>(a, bs); break;
    default: return VPP_ERR_UNSUPPORTED;
  }
  return VPP_OK;
}

}  // extern "C"
