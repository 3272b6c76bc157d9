// pyrlk_match.hh — coarse-to-fine Lucas-Kanade over prebuilt pyramids (reference: vpp/algorithms/pyrlk/pyrlk_match.hh:15-55).
#pragma once
#include <algorithm>
#include <stdexcept>
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
// Several frame pairs in ONE launch (not in the reference, whose pyrlk_match takes one pair: pyrlk_match.hh:15-55): pair f is
// (pyramids_prev[f], pyramids_prev_grad[f], pyramids_next[f], keypoints[f]); every container ends exactly as pyrlk_match on its pair leaves it.  A launch of a
// few thousand keypoints costs the latency of one keypoint's chain whatever its width, so a caller holding several pairs (a group of decoded frames, a rank's
// slice of a keypoint-sharded job) gets near-linear throughput from handing them over together (include/vpp_amd.h: vpp_pyrlk_match_batch).
template <class M, class V, class U, class C>
void pyrlk_match(const std::vector<pyramid2d<V>>& pyramids_prev, const std::vector<pyramid2d<vector<U, 2>>>& pyramids_prev_grad, const std::vector<pyramid2d<V>>& pyramids_next,
                 std::vector<C>& keypoints, M /*matcher*/, float min_ev, float max_err, float max_iteration, float convergence_delta, int min_scale = 0) {
  static_assert(sizeof(typename C::keypoint_type) == sizeof(vpp_keypoint_f32), "keypoint<float> must match vpp_keypoint_f32 (20 bytes)");
  static_assert(!detail::is_unsupported_matcher<M>::value, "pyrlk_match: oriented_lk_match_point_square_win (reference lk.hh:181-317) has no device kernel; use lk_match_point_square_win<WS>");
  static_assert(M::window_size == 3 || M::window_size == 5 || M::window_size == 7 || M::window_size == 9 || M::window_size == 11 || M::window_size == 15 || M::window_size == 21,
                "pyrlk_match: the device kernels are instantiated for window sizes 3, 5, 7, 9, 11, 15 and 21");
  const int F = int(keypoints.size());
  if (int(pyramids_prev.size()) != F || int(pyramids_prev_grad.size()) != F || int(pyramids_next.size()) != F) throw std::runtime_error("pyrlk_match: one pyramid triple and one container per frame pair");
  if (!F) return;
  const int L = pyramids_prev[0].size();
  std::vector<vpp_image_desc> P(size_t(F) * L), G(size_t(F) * L), N(size_t(F) * L);
  std::vector<int> n(F);
  size_t total = 0;
  for (int f = 0; f < F; f++) {
    if (int(pyramids_prev[f].size()) != L || int(pyramids_prev_grad[f].size()) != L || int(pyramids_next[f].size()) != L) throw std::runtime_error("pyrlk_match: pyramids of different depths");
    keypoints[f].prepare_matching();
    n[f] = keypoints[f].size(); total += size_t(n[f]);
    for (int l = 0; l < L; l++) {
      P[size_t(f) * L + l] = pyramids_prev[f][l].device_desc(false); G[size_t(f) * L + l] = pyramids_prev_grad[f][l].device_desc(false); N[size_t(f) * L + l] = pyramids_next[f][l].device_desc(false);
    }
  }
  if (!total) return;
  std::vector<typename C::keypoint_type> res(total);   // all frames' records in one block: one upload, one download
  std::vector<vpp_keypoint_f32*> dptr(F);
  device::dbuf dk(total * sizeof(vpp_keypoint_f32));
  size_t at = 0;
  for (int f = 0; f < F; f++) {
    std::copy(keypoints[f].keypoints().begin(), keypoints[f].keypoints().end(), res.begin() + at);
    dptr[f] = (vpp_keypoint_f32*)dk.p + at;
    at += size_t(n[f]);
  }
  dk.upload(res.data(), dk.bytes);
  device::check(vpp_pyrlk_match_batch(P.data(), G.data(), N.data(), F, L, dptr.data(), n.data(), int(M::window_size), min_ev, max_err, int(max_iteration), convergence_delta, min_scale,
                                      nullptr, device::stream()), "vpp_pyrlk_match_batch");
  dk.download(res.data(), dk.bytes);
  at = 0;
  for (int f = 0; f < F; f++) {
    for (int i = 0; i < n[f]; i++) {  // replay move / remove on the host containers (index image, ages): pyrlk_match.hh:44-50
      if (!keypoints[f][i].alive()) continue;
      if (res[at + i].age == 0) keypoints[f].remove(i);
      else keypoints[f].move(i, res[at + i].position);
    }
    at += size_t(n[f]);
  }
}
}  // namespace vpp
