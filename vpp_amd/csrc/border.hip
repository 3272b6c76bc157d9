// border.hip — K3: fill_border_mirror / fill_border_closest / fill_border_with_value
// (reference: vpp/core/fill.hh:31-122).  One launch over the frame perimeter: one lane per border pixel.
// The reference fills eight regions one after the other (fill.hh:56-82: corners, then top / bottom / left / right edge).  As
// long as border <= nrows and border <= ncols every mirrored source pixel is an interior pixel and the order is invisible; a
// wider mirror border reads other border regions, so that case runs the eight regions as eight stream-ordered launches (no
// region reads its own pixels), which reproduces the serial result.
#include "common.hpp"
#include <cstring>
using namespace vpp_amd;

namespace {
struct PixVal { uint8_t b[16]; };

template <int ES>
__global__ __launch_bounds__(256) void fill_border_kernel(DImg im, int mode, PixVal val, int region) {
  const int b = im.border, nr = im.nr, nc = im.nc;
  const int wfull = nc + 2 * b;
  const int ntop = b * wfull;                 // rows [-b, -1]
  const int nside = nr * 2 * b;               // rows [0, nr-1], cols [-b,-1] U [nc, nc+b-1]
  const int total = 2 * ntop + nside;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int r, c;
  if (i < ntop) { r = -b + i / wfull; c = -b + i % wfull; }
  else if (i < ntop + nside) { int k = i - ntop; r = k / (2 * b); int j = k % (2 * b); c = j < b ? j - b : nc + (j - b); }
  else { int k = i - ntop - nside; r = nr + k / wfull; c = -b + k % wfull; }
  if (region >= 0) {  // 0..7 = fill.hh's order: corners 1 3 6 8, edges 2 7 4 5
    const int vr = r < 0 ? 0 : (r >= nr ? 2 : 1), hc = c < 0 ? 0 : (c >= nc ? 2 : 1);
    const int of_cell[3][3] = {{0, 4, 1}, {6, -1, 7}, {2, 5, 3}};
    if (of_cell[vr][hc] != region) return;
  }
  uint8_t* dst = im.p0 + (ptrdiff_t)r * im.pitch + (ptrdiff_t)c * ES;
  if (mode == VPP_BORDER_VALUE) {
#pragma unroll
    for (int k = 0; k < ES; k++) dst[k] = val.b[k];
    return;
  }
  int sr, sc;
  if (mode == VPP_BORDER_MIRROR) {  // fill.hh:60-83
    sr = r < 0 ? -r - 1 : (r >= nr ? 2 * nr - r - 1 : r);
    sc = c < 0 ? -c - 1 : (c >= nc ? 2 * nc - c - 1 : c);
  } else {                          // fill.hh:93-121
    sr = r < 0 ? 0 : (r >= nr ? nr - 1 : r);
    sc = c < 0 ? 0 : (c >= nc ? nc - 1 : c);
  }
  const uint8_t* src = im.p0 + (ptrdiff_t)sr * im.pitch + (ptrdiff_t)sc * ES;
#pragma unroll
  for (int k = 0; k < ES; k++) dst[k] = src[k];
}
}  // namespace

namespace vpp_amd {
int launch_fill_border(const vpp_image_desc* img, int mode, const void* value, hipStream_t st) {
  const int es = elem_bytes(img);
  const int b = img->border;
  if (b == 0) return VPP_OK;
  PixVal v; memset(v.b, 0, sizeof v.b);
  if (mode == VPP_BORDER_VALUE) {
    VPP_REQUIRE(value, VPP_ERR_INVALID_ARG, "vpp_fill_border: VALUE mode needs a value");
    memcpy(v.b, value, es);
  }
  const long total = 2L * b * (img->ncols + 2 * b) + 2L * b * img->nrows;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  DImg d = dimg(img);
  const bool ordered = mode == VPP_BORDER_MIRROR && (b > img->nrows || b > img->ncols);
  for (int region = ordered ? 0 : -1; region < (ordered ? 8 : 0); region++) {
#define VPP_FB_CASE(ES) case ES: fill_border_kernel<ES><<<blocks, 256, 0, st>>>(d, mode, v, region); break;
    switch (es) {
      VPP_FB_CASE(1) VPP_FB_CASE(2) VPP_FB_CASE(3) VPP_FB_CASE(4) VPP_FB_CASE(6) VPP_FB_CASE(8) VPP_FB_CASE(12) VPP_FB_CASE(16)
      default: set_error("vpp_fill_border: unsupported element size %d", es); return VPP_ERR_UNSUPPORTED;
    }
#undef VPP_FB_CASE
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
}  // namespace vpp_amd

extern "C" int vpp_fill_border(const vpp_image_desc* img, int mode, const void* value, void* stream) {
  VPP_REQUIRE(valid_desc(img), VPP_ERR_INVALID_ARG, "vpp_fill_border: invalid descriptor");
  VPP_REQUIRE(mode >= VPP_BORDER_MIRROR && mode <= VPP_BORDER_VALUE, VPP_ERR_INVALID_ARG, "vpp_fill_border: bad mode %d", mode);
  return launch_fill_border(img, mode, value, as_stream(stream));
}
