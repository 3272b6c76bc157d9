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
}  // namespace

extern "C" int vpp_scharr(const vpp_image_desc* out, const vpp_image_desc* in, void* stream) {
  VPP_REQUIRE(valid_desc(out) && valid_desc(in), VPP_ERR_INVALID_ARG, "vpp_scharr: invalid descriptor");
  VPP_REQUIRE(in->dtype == VPP_U8 && in->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_scharr: input must be u8 x1");
  VPP_REQUIRE(out->channels == 2 && (out->dtype == VPP_F32 || out->dtype == VPP_I32), VPP_ERR_UNSUPPORTED, "vpp_scharr: output must be f32 x2 or i32 x2");
  VPP_REQUIRE(in->border >= 1, VPP_ERR_BORDER_TOO_SMALL, "vpp_scharr: input needs border >= 1 (scharr.hh:48)");
  VPP_REQUIRE(out->nrows <= in->nrows && out->ncols <= in->ncols, VPP_ERR_INVALID_ARG, "vpp_scharr: output larger than input");
  dim3 grid((out->ncols + 255) / 256, out->nrows);
  if (out->dtype == VPP_F32) scharr_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in));
  else scharr_kernel<int><<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in));
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
