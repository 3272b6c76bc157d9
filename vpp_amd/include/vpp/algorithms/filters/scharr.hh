// scharr.hh — 3x3 Scharr gradient (reference: vpp/algorithms/filters/scharr.hh:11-97).
#pragma once
#include <vpp/algorithms/device_only.hh>
#include <vpp/core/image2d.hh>

namespace vpp {
// gradient at one point (scharr.hh:11-43); a handful of host reads, not a hot path
template <class U> vfloat2 scharr(const image2d<U>& in, vint2 p) {
  assert(in.border() >= 1);
  const int r = p[0], c = p[1];
  const U *row1 = &in(r - 1, 0), *row2 = &in(r, 0), *row3 = &in(r + 1, 0);
  return vfloat2((3 * int(row3[c - 1]) + 10 * int(row3[c]) + 3 * int(row3[c + 1]) - 3 * int(row1[c - 1]) - 10 * int(row1[c]) - 3 * int(row1[c + 1])) / 32.f,
                 (3 * int(row1[c + 1]) + 10 * int(row2[c + 1]) + 3 * int(row3[c + 1]) - 3 * int(row1[c - 1]) - 10 * int(row2[c - 1]) - 3 * int(row3[c - 1])) / 32.f);
}
// dense gradient image (scharr.hh:46-87): V = float or int
template <class U, class V> void scharr(const image2d<U>& in, image2d<vector<V, 2>>& out) {
  static_assert(sizeof(U) == 1, "scharr: 8-bit single-channel input");
  const vpp_image_desc di = in.device_desc(false), dout = out.device_desc(true);
  device::check(vpp_scharr(&dout, &di, device::stream()), "vpp_scharr");
  device::call_done();   // queued, not drained: vpp/core/device.hh
}
template <class U, class V> void scharr(const image2d<vector<U, 1>>& in, image2d<vector<V, 2>>& out) { scharr(*(const image2d<U>*)&in, out); }
}  // namespace vpp
