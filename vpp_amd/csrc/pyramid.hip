// pyramid.hip — K4/K5: one step of pyramid::propagate_level0 for factor 2
// (reference: vpp/core/pyramid.hh:12-59 antialiasing_lowpass_filter, :62-81 subsample2, :175-182 the per-level loop).
//
// Fused: only the even rows / columns that subsample2 keeps are ever computed.  Semantics restated per output
// component (S = plus_promotion<V>, V() = conversion back to the element type, which truncates for integer V):
//   H(rr, cc) = V((1*S(in(rr,cc-2)) + 4*S(in(rr,cc-1)) + 6*S(in(rr,cc)) + 4*S(in(rr,cc+1)) + 1*S(in(rr,cc+2))) / 16)   rr in [0,nr)
//   H(-1)=H(0), H(-2)=H(1), H(nr)=H(nr-1), H(nr+1)=H(nr-2)                      (fill_border_mirror(tmp), pyramid.hh:36)
//   L(r, cc)  = V((1*S(H(r-2,cc)) + 4*S(H(r-1,cc)) + 6*S(H(r,cc)) + 4*S(H(r+1,cc)) + 1*S(H(r+2,cc))) / 16)
//   next(r,c) = (2r < nr && 2c < nc) ? L(2r, 2c) : 0        (the reference reads an unfilled temp border there)
// then fill_border_mirror(next) (pyramid.hh:182) as a second launch of the K3 kernel.
// L1-bound on its 5-tap gathers; each thread marches down 4 or 8 output rows so that the H pass of an input row is evaluated
// once per output column instead of up to three times.
#include "common.hpp"
using namespace vpp_amd;

namespace vpp_amd { int launch_fill_border(const vpp_image_desc* img, int mode, const void* value, hipStream_t st); }

namespace {

template <class T> struct Promo { typedef int type; };
template <> struct Promo<float> { typedef float type; };
template <> struct Promo<uint32_t> { typedef uint32_t type; };

template <class T, class S> __device__ __forceinline__ T tap5(S a, S b, S c, S d, S e) {
  return (T)((1 * a + 4 * b + 6 * c + 4 * d + 1 * e) / 16);
}
template <class T, class S> __device__ __forceinline__ T hpass(const DImg& in, int rr, int comp, int ch) {
  const T* i = in.row<T>(rr) + comp;
  return tap5<T, S>((S)i[-2 * ch], (S)i[-ch], (S)i[0], (S)i[ch], (S)i[2 * ch]);
}
__device__ __forceinline__ int mirror_row(int r, int nr) {
  const int m = r < 0 ? -r - 1 : (r >= nr ? 2 * nr - r - 1 : r);
  return min(max(m, 0), nr - 1);  // only differs for images of fewer than 2 rows, where the reference reads outside its buffer
}

template <class T, class S> __device__ __forceinline__ T lowpass_at(const DImg& in, int r, int comp, int ch) {
  S h[5];
#pragma unroll
  for (int k = 0; k < 5; k++) h[k] = (S)hpass<T, S>(in, mirror_row(r - 2 + k, in.nr), comp, ch);
  return tap5<T, S>(h[0], h[1], h[2], h[3], h[4]);
}

// One thread = one output component marching down TH output rows: consecutive output rows share three of their five
// H-pass rows (input rows 2r-2 .. 2r+2), so only two new H values are evaluated per row after the first.
// FUSE: the thread also writes the mirrored copies of its pixel into next's border (fill_border_mirror(next), pyramid.hh:182),
// which saves the separate border launch; used when the border is not wider than the level itself.
template <class T, class S, int TH, bool FUSE>
__global__ __launch_bounds__(256) void pyr_down_kernel(DImg next, DImg prev) {
  const int ch = prev.ch;
  const int comp = blockIdx.x * 256 + threadIdx.x;  // component index within the output row
  if (comp >= next.nc * ch) return;
  const int c = comp / ch, k = comp - c * ch;
  const int r0 = blockIdx.y * TH;
  const bool col_ok = 2 * c < prev.nc;
  const int icomp = 2 * c * ch + k;
  S h[5];
  if (col_ok && 2 * r0 < prev.nr) {
#pragma unroll
    for (int j = 0; j < 5; j++) h[j] = (S)hpass<T, S>(prev, mirror_row(2 * r0 - 2 + j, prev.nr), icomp, ch);
  }
#pragma unroll
  for (int j = 0; j < TH; j++) {
    const int r = r0 + j;
    if (r >= next.nr) break;
    T v = 0;
    if (col_ok && 2 * r < prev.nr) {
      if (j > 0) {
        h[0] = h[2]; h[1] = h[3]; h[2] = h[4];
        h[3] = (S)hpass<T, S>(prev, mirror_row(2 * r + 1, prev.nr), icomp, ch);
        h[4] = (S)hpass<T, S>(prev, mirror_row(2 * r + 2, prev.nr), icomp, ch);
      }
      v = tap5<T, S>(h[0], h[1], h[2], h[3], h[4]);
    }
    next.row<T>(r)[comp] = v;
    if (FUSE) {  // border pixel (-k, c) <- (k-1, c), (nr-1+k, c) <- (nr-k, c), same for columns, corners in both axes (fill.hh:60-83)
      const int b = next.border;
      const int mr = r < b ? -r - 1 : (r >= next.nr - b ? 2 * next.nr - r - 1 : r);
      const int mr2 = (r < b && r >= next.nr - b) ? 2 * next.nr - r - 1 : mr;          // a row within b of both ends mirrors both ways
      const int mc = c < b ? -c - 1 : (c >= next.nc - b ? 2 * next.nc - c - 1 : c);
      const int mc2 = (c < b && c >= next.nc - b) ? 2 * next.nc - c - 1 : mc;
      const int mcomp = mc * ch + k, mcomp2 = mc2 * ch + k;
      if (mc != c) next.row<T>(r)[mcomp] = v;
      if (mc2 != mc) next.row<T>(r)[mcomp2] = v;
      if (mr != r) { next.row<T>(mr)[comp] = v; if (mc != c) next.row<T>(mr)[mcomp] = v; if (mc2 != mc) next.row<T>(mr)[mcomp2] = v; }
      if (mr2 != mr) { next.row<T>(mr2)[comp] = v; if (mc != c) next.row<T>(mr2)[mcomp] = v; if (mc2 != mc) next.row<T>(mr2)[mcomp2] = v; }
    }
  }
}

// ---- float x 2 pixels (the gradient pyramid of pyrLK: vfloat2 Scharr responses) ------------------------------------------------
// The generic kernel above spends five 4-byte loads per H value at a 16-byte lane stride.  Here a lane owns one output pixel and the
// input pair (2c, 2c + 1) under it — ONE 16-byte load per input row through a buffer descriptor over prev's addressable bytes — and
// takes pixels 2c - 2, 2c - 1 from lane - 1 and pixel 2c + 2 from lane + 1 over DPP (lanes 0 and 63 of a wave are halo lanes: 62
// outputs per wave), so every input pixel is loaded once per row block.  Arithmetic is tap5<float, float> on the same values in the
// same order as the generic kernel (bit-identical); the march down TH rows and the fused border copies are the same as there.
constexpr int kF2Out = 62;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dpp_left(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }   // lane i <- lane i - 1
__device__ __forceinline__ float dpp_right(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }  // lane i <- lane i + 1

template <int TH, bool FUSE>
__global__ __launch_bounds__(256) void pyr_down_f32x2_kernel(DImg next, DImg prev, uint32_t pbytes, int nstrips) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (int)(blockDim.x >> 6) + wv);
  if (s >= nstrips) return;
  const int c = s * kF2Out - 1 + lane;   // output column of this lane (lanes 0 / 63: halo only)
  const int r0 = blockIdx.y * TH;
  const bool writer = lane >= 1 && lane <= kF2Out && c < next.nc;
  const bool col_ok = 2 * c < prev.nc;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(prev.p0 - (ptrdiff_t)prev.border * prev.pitch - (ptrdiff_t)prev.border * 8), 0, pbytes, 0x00020000);
  // byte offset of pixel 2c inside a row of the addressable area; out-of-range lanes (past the last addressable pixel) read 0, their
  // values reach only outputs that are 0 by definition (col_ok false) or not stored
  const uint32_t vo = (uint32_t)((2 * c + prev.border) * 8);
  // all 2 TH + 3 input rows of the block are requested before anything is computed or stored (the stores may alias the loads as
  // far as the compiler knows, so a load-per-iteration march waits one memory round trip per output row: 5.6 us for a 5 MB level)
  constexpr int NR = 2 * TH + 3;
  f32x4 raw[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int rr = mirror_row(2 * r0 - 2 + k, prev.nr);
    raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (rr + prev.border) * prev.pitch, 0));
  }
  float hx[NR], hy[NR];   // H pass of every loaded row at this lane's column, both components
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const f32x4 v = raw[k];
    const float ax = dpp_left(v.x), ay = dpp_left(v.y), bx = dpp_left(v.z), by = dpp_left(v.w), ex = dpp_right(v.x), ey = dpp_right(v.y);
    hx[k] = tap5<float, float>(ax, bx, v.x, v.z, ex);
    hy[k] = tap5<float, float>(ay, by, v.y, v.w, ey);
  }
#pragma unroll
  for (int j = 0; j < TH; j++) {
    const int r = r0 + j;
    if (r >= next.nr) break;
    float vx = 0.f, vy = 0.f;
    if (col_ok && 2 * r < prev.nr) {
      vx = tap5<float, float>(hx[2 * j], hx[2 * j + 1], hx[2 * j + 2], hx[2 * j + 3], hx[2 * j + 4]);
      vy = tap5<float, float>(hy[2 * j], hy[2 * j + 1], hy[2 * j + 2], hy[2 * j + 3], hy[2 * j + 4]);
    }
    if (!writer) continue;
    const float2 v = make_float2(vx, vy);
    next.row<float2>(r)[c] = v;
    if (FUSE) {  // the mirrored copies of this pixel in next's border, as in pyr_down_kernel
      const int b = next.border;
      const int mr = r < b ? -r - 1 : (r >= next.nr - b ? 2 * next.nr - r - 1 : r);
      const int mr2 = (r < b && r >= next.nr - b) ? 2 * next.nr - r - 1 : mr;
      const int mc = c < b ? -c - 1 : (c >= next.nc - b ? 2 * next.nc - c - 1 : c);
      const int mc2 = (c < b && c >= next.nc - b) ? 2 * next.nc - c - 1 : mc;
      if (mc != c) next.row<float2>(r)[mc] = v;
      if (mc2 != mc) next.row<float2>(r)[mc2] = v;
      if (mr != r) { next.row<float2>(mr)[c] = v; if (mc != c) next.row<float2>(mr)[mc] = v; if (mc2 != mc) next.row<float2>(mr)[mc2] = v; }
      if (mr2 != mr) { next.row<float2>(mr2)[c] = v; if (mc != c) next.row<float2>(mr2)[mc] = v; if (mc2 != mc) next.row<float2>(mr2)[mc2] = v; }
    }
  }
}

template <class T, class S>
__global__ __launch_bounds__(256) void lowpass_kernel(DImg out, DImg in) {
  const int ch = in.ch;
  const int comp = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (comp >= in.nc * ch) return;
  out.row<T>(r)[comp] = lowpass_at<T, S>(in, r, comp, ch);
}

template <class F> int by_dtype(int dtype, F f) {
  switch (dtype) {
    case VPP_U8: return f((uint8_t)0);
    case VPP_I8: return f((int8_t)0);
    case VPP_U16: return f((uint16_t)0);
    case VPP_I16: return f((int16_t)0);
    case VPP_I32: return f((int32_t)0);
    case VPP_U32: return f((uint32_t)0);
    case VPP_F32: return f((float)0);
  }
  return VPP_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

int vpp_pyr_down(const vpp_image_desc* next, const vpp_image_desc* prev, void* stream) {
  VPP_REQUIRE(valid_desc(next) && valid_desc(prev), VPP_ERR_INVALID_ARG, "vpp_pyr_down: invalid descriptor");
  VPP_REQUIRE(same_type(next, prev), VPP_ERR_INVALID_ARG, "vpp_pyr_down: element types differ");
  VPP_REQUIRE(prev->border >= 2, VPP_ERR_BORDER_TOO_SMALL, "vpp_pyr_down: prev needs border >= 2 (has %d)", prev->border);
  VPP_REQUIRE(next->nrows == 1 + prev->nrows / 2 && next->ncols == 1 + prev->ncols / 2, VPP_ERR_INVALID_ARG,
              "vpp_pyr_down: next must be (1+nr/2, 1+nc/2) = (%d,%d), got (%d,%d) (pyramid.hh:140)", 1 + prev->nrows / 2, 1 + prev->ncols / 2,
              next->nrows, next->ncols);
  hipStream_t st = as_stream(stream);
  const int gx = (next->ncols * next->channels + 255) / 256;
  const bool fuse = next->border <= next->nrows && next->border <= next->ncols && tuning("pyr.fuse_border", 1);
  const bool tall = tuning("pyr.rows", (long long)next->nrows * next->ncols >= (1 << 20) ? 8 : 4) == 8;  // enough waves either way
  DImg n = dimg(next), p = dimg(prev);
  const size_t paddr = (size_t)(prev->nrows + 2 * prev->border) * prev->pitch;   // rows -border .. nrows + border - 1 of the pitch
  if (prev->dtype == VPP_F32 && prev->channels == 2 && paddr < ((size_t)1 << 31) && prev->pitch % 8 == 0 && ((uintptr_t)prev->first_pixel & 7) == 0 &&
      ((uintptr_t)next->first_pixel & 7) == 0 && next->pitch % 8 == 0 && tuning("pyr.f2", 1)) {
    const uint32_t pbytes = (uint32_t)((size_t)(prev->nrows + 2 * prev->border - 1) * prev->pitch + (size_t)(prev->ncols + 2 * prev->border) * 8);
    const int nstrips = (next->ncols + kF2Out - 1) / kF2Out;
    const int wpb = tuning("pyr.f2_waves", 4) == 1 ? 1 : (tuning("pyr.f2_waves", 4) == 2 ? 2 : 4);
    const dim3 grid((nstrips + wpb - 1) / wpb, (next->nrows + 3) / 4);   // 4 rows per wave at every size (measured 4K level 1: 8 rows 20.1 us / 88 VGPRs, 4 rows 15 us / 48)
    if (fuse) pyr_down_f32x2_kernel<4, true><<<grid, 64 * wpb, 0, st>>>(n, p, pbytes, nstrips);
    else pyr_down_f32x2_kernel<4, false><<<grid, 64 * wpb, 0, st>>>(n, p, pbytes, nstrips);
    VPP_LAUNCH_CHECK();
    return fuse ? (int)VPP_OK : launch_fill_border(next, VPP_BORDER_MIRROR, nullptr, st);
  }
  int rc = by_dtype(prev->dtype, [&](auto t) {
    typedef decltype(t) T; typedef typename Promo<T>::type S;
    if (fuse) {
      if (tall) pyr_down_kernel<T, S, 8, true><<<dim3(gx, (next->nrows + 7) / 8), 256, 0, st>>>(n, p);
      else pyr_down_kernel<T, S, 4, true><<<dim3(gx, (next->nrows + 3) / 4), 256, 0, st>>>(n, p);
    } else {
      if (tall) pyr_down_kernel<T, S, 8, false><<<dim3(gx, (next->nrows + 7) / 8), 256, 0, st>>>(n, p);
      else pyr_down_kernel<T, S, 4, false><<<dim3(gx, (next->nrows + 3) / 4), 256, 0, st>>>(n, p);
    }
    return (int)VPP_OK;
  });
  if (rc != VPP_OK) return rc;
  VPP_LAUNCH_CHECK();
  return fuse ? (int)VPP_OK : launch_fill_border(next, VPP_BORDER_MIRROR, nullptr, st);
}

int vpp_lowpass5(const vpp_image_desc* out, const vpp_image_desc* in, void* stream) {
  VPP_REQUIRE(valid_desc(out) && valid_desc(in), VPP_ERR_INVALID_ARG, "vpp_lowpass5: invalid descriptor");
  VPP_REQUIRE(same_type(out, in) && same_domain(out, in), VPP_ERR_INVALID_ARG, "vpp_lowpass5: domain/type mismatch");
  VPP_REQUIRE(in->border >= 2, VPP_ERR_BORDER_TOO_SMALL, "vpp_lowpass5: in needs border >= 2 (has %d)", in->border);
  VPP_REQUIRE(out->first_pixel != in->first_pixel, VPP_ERR_INVALID_ARG, "vpp_lowpass5: in-place not supported");
  hipStream_t st = as_stream(stream);
  dim3 grid((in->ncols * in->channels + 255) / 256, in->nrows);
  DImg o = dimg(out), i = dimg(in);
  int rc = by_dtype(in->dtype, [&](auto t) {
    typedef decltype(t) T; typedef typename Promo<T>::type S;
    lowpass_kernel<T, S><<<grid, 256, 0, st>>>(o, i);
    return (int)VPP_OK;
  });
  if (rc != VPP_OK) return rc;
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

}  // extern "C"
