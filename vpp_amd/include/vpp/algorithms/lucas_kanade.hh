// lucas_kanade.hh — option-style pyramidal LK (reference: vpp/algorithms/lucas_kanade.hh:8-24, lucas_kanade/lucas_kanade.hpp:135-184).
#pragma once
#include <cmath>
#include <vector>
#include <vpp/algorithms/device_only.hh>
#include <vpp/algorithms/filters/scharr.hh>
#include <vpp/algorithms/symbols.hh>
#include <vpp/core/pyramid.hh>

namespace vpp {
template <class V, class... OPTS> void lucas_kanade(const image2d<V>& i1, const image2d<V>& i2, OPTS... opts) {
  static_assert(sizeof(V) == 1, "lucas_kanade: 8-bit single-channel frames");
  auto options = opt::make(opts...);
  const int niterations = options.get(_niterations, 21);
  const int winsize = options.get(_winsize, 11);
  const int nscales = options.get(_nscales, 3);
  const int min_ev = options.get(_min_ev, 0.0001);  // truncated to int exactly as lucas_kanade.hpp:143-144 does
  const int delta = options.get(_delta, 0.1);
  auto prediction = options.get(_prediction, [](auto) { return vfloat2(0.f, 0.f); });
  auto flow = options.get(_flow, 0);
  auto keypoints = options.get(_keypoints, 0);

  typedef typename std::decay<decltype(V() - V())>::type Gr;  // int for uchar
  pyramid2d<V> pyramid_prev(i1, nscales, 2, _border = winsize / 2);
  pyramid2d<vector<Gr, 2>> pyramid_prev_grad(i1.domain(), nscales, 2, _border = winsize / 2);
  pyramid2d<V> pyramid_next(i2, nscales, 2, _border = winsize / 2);
  scharr(pyramid_prev[0], pyramid_prev_grad[0]);
  pyramid_prev_grad.propagate_level0();

  const int n = int(keypoints.size());
  if (!n) return;
  std::vector<vfloat2> pts(n), pred(n), out(n);
  std::vector<float> dist(n);
  for (int i = 0; i < n; i++) { pts[i] = keypoints[i].template cast<float>(); pred[i] = prediction(keypoints[i]).template cast<float>(); }
  std::vector<vpp_image_desc> P(nscales), G(nscales), N(nscales);
  for (int l = 0; l < nscales; l++) { P[l] = pyramid_prev[l].device_desc(false); G[l] = pyramid_prev_grad[l].device_desc(false); N[l] = pyramid_next[l].device_desc(false); }
  device::dbuf dp(size_t(n) * 8), dq(size_t(n) * 8), df(size_t(n) * 8), dd(size_t(n) * 4);
  dp.upload(pts.data(), dp.bytes); dq.upload(pred.data(), dq.bytes);
  device::check(vpp_lucas_kanade(P.data(), G.data(), N.data(), nscales, (const float*)dp.p, (const float*)dq.p, n, winsize, min_ev, niterations, delta,
                                 (float*)df.p, (float*)dd.p, device::stream()), "vpp_lucas_kanade");
  df.download(out.data(), df.bytes); dd.download(dist.data(), dd.bytes);
  for (int i = 0; i < n; i++) flow(keypoints[i], out[i], dist[i]);  // lucas_kanade.hpp:181
}
}  // namespace vpp
