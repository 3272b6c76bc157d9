// scharr.hip — K6: 3x3 Scharr gradient (reference: vpp/algorithms/filters/scharr.hh:46-87).
// in: u8 x1 with border >= 1.  out: 2 components of V = float (pyrlk, benchmarks/pyrlk_opencv_comparison.cc:56-58) or
// V = int (lucas_kanade, lucas_kanade.hpp:151-155).  Arithmetic in V, then `/ 32.f`, then conversion to V
// (all products/sums are small integers, exact in float, so operation order cannot change the result).
// Output-bound: 8 B written per pixel (16.6 MB at 1080p); one lane = one pixel, 8 L1-served byte loads.
#include "common.hpp"
using namespace vpp_amd;

namespace {
template <class V>
__global__ __launch_bounds__(256) void scharr_kernel(DImg out, DImg in) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= out.nc) return;
  const uint8_t *row1 = in.row<uint8_t>(r - 1), *row2 = in.row<uint8_t>(r), *row3 = in.row<uint8_t>(r + 1);
  const V a1 = (V)row1[c - 1], b1 = (V)row1[c], c1 = (V)row1[c + 1];
  const V a2 = (V)row2[c - 1], c2 = (V)row2[c + 1];
  const V a3 = (V)row3[c - 1], b3 = (V)row3[c], c3 = (V)row3[c + 1];
  const float g0 = (3 * a3 + 10 * b3 + 3 * c3 - 3 * a1 - 10 * b1 - 3 * c1) / 32.f;
  const float g1 = (3 * c1 + 10 * c2 + 3 * c3 - 3 * a1 - 10 * a2 - 3 * a3) / 32.f;
  V* o = out.row<V>(r) + 2 * c;
  o[0] = (V)g0; o[1] = (V)g1;
}
// Four pixels per lane: three (possibly unaligned) 8-byte row loads cover columns c-1 .. c+6, the four results leave as two
// 16-byte stores; the 3x3 sums are the same small exact integers in float or int as in the per-pixel form.
// the mirrored copies of domain pixel (r, c) of a 2-component image in its border (fill_border_mirror, fill.hh:60-83; border <= nrows, ncols)
template <class V> __device__ __forceinline__ void border_copies2(const DImg& img, int r, int c, V v0, V v1) {
  const int b = img.border, nr = img.nr, nc = img.nc;
  if (b == 0 || (r >= b && r < nr - b && c >= b && c < nc - b)) return;
  auto put = [&](int rr, int cc) { V* p = img.row<V>(rr) + 2 * cc; p[0] = v0; p[1] = v1; };
  const int mr = r < b ? -r - 1 : (r >= nr - b ? 2 * nr - r - 1 : r);
  const int mr2 = (r < b && r >= nr - b) ? 2 * nr - r - 1 : mr;
  const int mc = c < b ? -c - 1 : (c >= nc - b ? 2 * nc - c - 1 : c);
  const int mc2 = (c < b && c >= nc - b) ? 2 * nc - c - 1 : mc;
  if (mc != c) put(r, mc);
  if (mc2 != mc) put(r, mc2);
  if (mr != r) { put(mr, c); if (mc != c) put(mr, mc); if (mc2 != mc) put(mr, mc2); }
  if (mr2 != mr) { put(mr2, c); if (mc != c) put(mr2, mc); if (mc2 != mc) put(mr2, mc2); }
}

template <class V, bool BORDER = false>
__global__ __launch_bounds__(256) void scharr4_kernel(DImg out, DImg in) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4, r = blockIdx.y;
  if (c >= out.nc) return;
  if (c + 4 > out.nc) {  // ragged row end: per pixel
    for (int x = c; x < out.nc; x++) {
      const uint8_t *row1 = in.row<uint8_t>(r - 1), *row2 = in.row<uint8_t>(r), *row3 = in.row<uint8_t>(r + 1);
      const V a1 = (V)row1[x - 1], b1 = (V)row1[x], c1 = (V)row1[x + 1], a2 = (V)row2[x - 1], c2 = (V)row2[x + 1];
      const V a3 = (V)row3[x - 1], b3 = (V)row3[x], c3 = (V)row3[x + 1];
      V* o = out.row<V>(r) + 2 * x;
      o[0] = (V)((3 * a3 + 10 * b3 + 3 * c3 - 3 * a1 - 10 * b1 - 3 * c1) / 32.f);
      o[1] = (V)((3 * c1 + 10 * c2 + 3 * c3 - 3 * a1 - 10 * a2 - 3 * a3) / 32.f);
      if (BORDER) border_copies2<V>(out, r, x, o[0], o[1]);
    }
    return;
  }
  uint64_t w1, w2, w3;  // bytes c-1 .. c+6 of the three rows (c+5, c+6 unused: <= 2 bytes past the last needed pixel, inside the row pitch or the border)
  __builtin_memcpy(&w1, in.row<uint8_t>(r - 1) + c - 1, 8);
  __builtin_memcpy(&w2, in.row<uint8_t>(r) + c - 1, 8);
  __builtin_memcpy(&w3, in.row<uint8_t>(r + 1) + c - 1, 8);
  V res[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const V a1 = (V)(uint8_t)(w1 >> (8 * k)), b1 = (V)(uint8_t)(w1 >> (8 * k + 8)), c1 = (V)(uint8_t)(w1 >> (8 * k + 16));
    const V a2 = (V)(uint8_t)(w2 >> (8 * k)), c2 = (V)(uint8_t)(w2 >> (8 * k + 16));
    const V a3 = (V)(uint8_t)(w3 >> (8 * k)), b3 = (V)(uint8_t)(w3 >> (8 * k + 8)), c3 = (V)(uint8_t)(w3 >> (8 * k + 16));
    res[2 * k] = (V)((3 * a3 + 10 * b3 + 3 * c3 - 3 * a1 - 10 * b1 - 3 * c1) / 32.f);
    res[2 * k + 1] = (V)((3 * c1 + 10 * c2 + 3 * c3 - 3 * a1 - 10 * a2 - 3 * a3) / 32.f);
  }
  V* o = out.row<V>(r) + 2 * c;
  if ((((uintptr_t)o) & 15) == 0) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v0, v1;
    __builtin_memcpy(&v0, res, 16); __builtin_memcpy(&v1, res + 4, 16);
    __builtin_nontemporal_store(v0, (u32x4*)o); __builtin_nontemporal_store(v1, (u32x4*)o + 1);
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = res[k];
  }
  if (BORDER) {
    const int b = out.border;
    if (r < b || r >= out.nr - b || c < b || c + 4 > out.nc - b) {
#pragma unroll
      for (int k = 0; k < 4; k++) border_copies2<V>(out, r, c + k, res[2 * k], res[2 * k + 1]);
    }
  }
}
}  // namespace

// scharr + fill_border_mirror(out) in one launch (the gradient pyramid's level 0); needs the wide kernel's preconditions
namespace vpp_amd {
int vpp_scharr_bordered(const vpp_image_desc* out, const vpp_image_desc* in, void* stream) {
  const bool fusable = in->border >= 3 && out->border <= out->nrows && out->border <= out->ncols && same_domain(out, in);
  if (!fusable) {
    const int rc = vpp_scharr(out, in, stream);
    return rc != VPP_OK ? rc : vpp_fill_border(out, VPP_BORDER_MIRROR, nullptr, stream);
  }
  // 128 lanes per workgroup (measured scharr + gradient pyramid at 1080p: 256 lanes 23.1 us, 128: 19.4-20.1, 64: 23.7; 4K: 41.4 -> 40.3)
  const int sbt = tuning("scharr.block", 128), sb = sbt == 64 ? 64 : (sbt == 256 ? 256 : 128);
  dim3 grid(((out->ncols + 3) / 4 + sb - 1) / sb, out->nrows);
  if (out->dtype == VPP_F32) scharr4_kernel<float, true><<<grid, sb, 0, as_stream(stream)>>>(dimg(out), dimg(in));
  else scharr4_kernel<int, true><<<grid, sb, 0, as_stream(stream)>>>(dimg(out), dimg(in));
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
}  // namespace vpp_amd

extern "C" int vpp_scharr(const vpp_image_desc* out, const vpp_image_desc* in, void* stream) {
  VPP_REQUIRE(valid_desc(out) && valid_desc(in), VPP_ERR_INVALID_ARG, "vpp_scharr: invalid descriptor");
  VPP_REQUIRE(in->dtype == VPP_U8 && in->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_scharr: input must be u8 x1");
  VPP_REQUIRE(out->channels == 2 && (out->dtype == VPP_F32 || out->dtype == VPP_I32), VPP_ERR_UNSUPPORTED, "vpp_scharr: output must be f32 x2 or i32 x2");
  VPP_REQUIRE(in->border >= 1, VPP_ERR_BORDER_TOO_SMALL, "vpp_scharr: input needs border >= 1 (scharr.hh:48)");
  VPP_REQUIRE(out->nrows <= in->nrows && out->ncols <= in->ncols, VPP_ERR_INVALID_ARG, "vpp_scharr: output larger than input");
  // the 8-byte row loads reach column nc + 2 of the input: only when the border holds it
  const bool wide = in->border >= 3 && tuning("scharr.wide", 1);
  if (wide) {
    dim3 grid(((out->ncols + 3) / 4 + 255) / 256, out->nrows);
    if (out->dtype == VPP_F32) scharr4_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in));
    else scharr4_kernel<int><<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in));
  } else {
    dim3 grid((out->ncols + 255) / 256, out->nrows);
    if (out->dtype == VPP_F32) scharr_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in));
    else scharr_kernel<int><<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in));
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
