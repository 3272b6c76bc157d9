// optical_flow.hh — semi-dense optical flow (reference: vpp/algorithms/optical_flow.hh:6-35,
// optical_flow/semi_dense_optical_flow.hpp:48-214).  The epipolar options are not supported.
#pragma once
#include <vector>
#include <vpp/algorithms/device_only.hh>
#include <vpp/algorithms/symbols.hh>
#include <vpp/core/image2d.hh>

namespace vpp {
template <class K, class MC, class... OPTS>
void semi_dense_optical_flow(const K& keypoints, MC match_callback, const image2d<unsigned char>& i1, const image2d<unsigned char>& i2, OPTS... options) {
  auto opts = opt::make(options...);
  const int winsize = opts.get(_winsize, 7), nscales = opts.get(_nscales, 4), min_scale = opts.get(_min_scale, 0);
  const int propagation_niters = opts.get(_propagation, 2), patchsize = opts.get(_patchsize, 5);
  const int n = int(keypoints.size());
  if (!n) return;
  std::vector<vint2> kps(n), pos(n);
  std::vector<int> dist(n);
  std::vector<unsigned char> valid(n);
  for (int i = 0; i < n; i++) { auto k = keypoints[i]; kps[i] = vint2(k[0], k[1]); }
  device::dbuf dk(size_t(n) * 8), dp(size_t(n) * 8), dd(size_t(n) * 4), dv{size_t(n)};
  dk.upload(kps.data(), dk.bytes);
  const vpp_image_desc d1 = i1.device_desc(false), d2 = i2.device_desc(false);
  device::check(vpp_semi_dense_optical_flow(&d1, &d2, (const int32_t*)dk.p, n, winsize, nscales, min_scale, propagation_niters, patchsize, (int32_t*)dp.p,
                                            (int32_t*)dd.p, (uint8_t*)dv.p, device::stream()), "vpp_semi_dense_optical_flow");
  dp.download(pos.data(), dp.bytes); dd.download(dist.data(), dd.bytes); dv.download(valid.data(), dv.bytes);
  for (int i = 0; i < n; i++) if (valid[i]) match_callback(i, pos[i], dist[i]);  // semi_dense_optical_flow.hpp:205-212
}
}  // namespace vpp
