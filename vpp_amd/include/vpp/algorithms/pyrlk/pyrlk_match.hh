// pyrlk_match.hh — coarse-to-fine Lucas-Kanade over prebuilt pyramids (reference: vpp/algorithms/pyrlk/pyrlk_match.hh:15-55).
#pragma once
#include <type_traits>
#include <vector>
#include <vpp/algorithms/device_only.hh>
#include <vpp/algorithms/pyrlk/lk.hh>
#include <vpp/core/keypoint_container.hh>
#include <vpp/core/pyramid.hh>

namespace vpp {
typedef keypoint_container<keypoint<float>, int> pyrlk_keypoint_container;
namespace detail {
template <class M, class = void> struct is_unsupported_matcher : std::false_type {};
template <class M> struct is_unsupported_matcher<M, decltype(void(M::vpp_amd_unsupported))> : std::true_type {};
}  // namespace detail

template <class M, class V, class U, class C>
void pyrlk_match(const pyramid2d<V>& pyramid_prev, const pyramid2d<vector<U, 2>>& pyramid_prev_grad, const pyramid2d<V>& pyramid_next, C& keypoints,
                 M /*matcher*/, float min_ev, float max_err, float max_iteration, float convergence_delta, int min_scale = 0) {
  static_assert(sizeof(typename C::keypoint_type) == sizeof(vpp_keypoint_f32), "keypoint<float> must match vpp_keypoint_f32 (20 bytes)");
  static_assert(!detail::is_unsupported_matcher<M>::value, "pyrlk_match: oriented_lk_match_point_square_win (reference lk.hh:181-317) has no device kernel; use lk_match_point_square_win<WS>");
  static_assert(M::window_size == 3 || M::window_size == 5 || M::window_size == 7 || M::window_size == 9 || M::window_size == 11 || M::window_size == 15 || M::window_size == 21,
                "pyrlk_match: the device kernels are instantiated for window sizes 3, 5, 7, 9, 11, 15 and 21");
  keypoints.prepare_matching();
  const int n = keypoints.size(), L = pyramid_prev.size();
  if (!n) return;
  std::vector<vpp_image_desc> P(L), G(L), N(L);
  for (int l = 0; l < L; l++) { P[l] = pyramid_prev[l].device_desc(false); G[l] = pyramid_prev_grad[l].device_desc(false); N[l] = pyramid_next[l].device_desc(false); }
  std::vector<typename C::keypoint_type> res(keypoints.keypoints());
  device::dbuf dk(size_t(n) * sizeof(vpp_keypoint_f32));
  dk.upload(res.data(), dk.bytes);
  device::check(vpp_pyrlk_match(P.data(), G.data(), N.data(), L, (vpp_keypoint_f32*)dk.p, n, int(M::window_size), min_ev, max_err, int(max_iteration),
                                convergence_delta, min_scale, nullptr, device::stream()), "vpp_pyrlk_match");
  dk.download(res.data(), dk.bytes);
  for (int i = 0; i < n; i++) {  // replay move / remove on the host container (index image, ages): pyrlk_match.hh:44-50
    if (!keypoints[i].alive()) continue;
    if (res[i].age == 0) keypoints.remove(i);
    else keypoints.move(i, res[i].position);
  }
}
}  // namespace vpp
