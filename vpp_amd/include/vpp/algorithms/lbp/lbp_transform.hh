// lbp_transform.hh — local binary patterns (reference: vpp/algorithms/lbp/lbp_transform.hh:6-38).
#pragma once
#include <vpp/algorithms/device_only.hh>
#include <vpp/core/image2d.hh>

namespace vpp {
template <class V, class U> void lbp_transform(image2d<V>& A, image2d<U>& B) {
  static_assert(sizeof(V) == 1 && sizeof(U) == 1, "lbp_transform: 8-bit images");
  const vpp_image_desc da = A.device_desc(false), db = B.device_desc(true);
  device::check(vpp_lbp_transform(&db, &da, device::stream()), "vpp_lbp_transform");
  device::call_done();   // queued, not drained: vpp/core/device.hh
}
}  // namespace vpp
