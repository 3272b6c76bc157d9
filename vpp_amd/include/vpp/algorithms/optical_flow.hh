// optical_flow.hh — semi-dense optical flow (reference: vpp/algorithms/optical_flow.hh:6-35,
// optical_flow/semi_dense_optical_flow.hpp:48-214).  The epipolar options are not supported.
#pragma once
#include <chrono>
#include <vector>
#include <vpp/algorithms/device_only.hh>
#include <vpp/algorithms/symbols.hh>
#include <vpp/core/image2d.hh>

namespace vpp {
namespace of_internals {
// optional wall-clock breakdown (benchmarks define VPP_AMD_TIMING): [0] keypoint gather + upload, [1] device call incl. frame
// upload, [2] result download, [3] match callbacks
inline double* timing() { static double t[4] = {0, 0, 0, 0}; return t; }
#ifdef VPP_AMD_TIMING
struct stopwatch { double& acc; std::chrono::steady_clock::time_point t0; explicit stopwatch(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                   ~stopwatch() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
#else
struct stopwatch { explicit stopwatch(double&) {} };
#endif
// pinned staging of one flow call: the keypoints as uploaded and the per-keypoint results the read-back kernel writes
struct flow_buffers {
  device::hbuf<vint2> kps, pos; device::hbuf<int> dist; device::hbuf<unsigned char> valid; device::dbuf dk;
  explicit flow_buffers(int n) : kps(n), pos(n), dist(n), valid(n), dk(size_t(n) * 8) {}
};
struct flow_params { int winsize, nscales, min_scale, propagation_niters, patchsize; };
// Uploads the keypoints, queues the flow, runs `queue_more()` (further device work that consumes the results in stream order,
// e.g. video_extruder's score cull) and waits once.  The results are written by the read-back kernel straight into pinned host
// memory (device-visible: vpp_malloc_host), so no device-to-host copy follows the flow.  (Measured: the first SDMA copy of more
// than a few 100 KB after the 2.5 ms of flow kernels takes ~0.8 ms — the copy engine waking up — while 1 MB of kernel stores
// over the link cost ~20 us.)
template <class K, class Q>
void run(const K& keypoints, const image2d<unsigned char>& i1, const image2d<unsigned char>& i2, const flow_params& fp, flow_buffers& b, Q queue_more) {
  const int n = int(keypoints.size());
  {
    stopwatch sw(timing()[0]);
    for (int i = 0; i < n; i++) { auto k = keypoints[i]; b.kps[i] = vint2(k[0], k[1]); }
    b.dk.upload(b.kps.data(), b.dk.bytes);
  }
  stopwatch sw(timing()[1]);
  const vpp_image_desc d1 = i1.device_desc(false), d2 = i2.device_desc(false);
  device::check(vpp_semi_dense_optical_flow(&d1, &d2, (const int32_t*)b.dk.p, n, fp.winsize, fp.nscales, fp.min_scale, fp.propagation_niters, fp.patchsize,
                                            (int32_t*)b.pos.data(), b.dist.data(), b.valid.data(), device::stream()), "vpp_semi_dense_optical_flow");
  queue_more();
  device::check(vpp_sync(device::stream()), "vpp_sync");
}
}
template <class K, class MC, class... OPTS>
void semi_dense_optical_flow(const K& keypoints, MC match_callback, const image2d<unsigned char>& i1, const image2d<unsigned char>& i2, OPTS... options) {
  auto opts = opt::make(options...);
  const of_internals::flow_params fp{opts.get(_winsize, 7), opts.get(_nscales, 4), opts.get(_min_scale, 0), opts.get(_propagation, 2), opts.get(_patchsize, 5)};
  const int n = int(keypoints.size());
  if (!n) return;
  of_internals::flow_buffers b(n);
  of_internals::run(keypoints, i1, i2, fp, b, [] {});
  of_internals::stopwatch sw(of_internals::timing()[3]);
  for (int i = 0; i < n; i++) if (b.valid[i]) match_callback(i, b.pos[i], b.dist[i]);  // semi_dense_optical_flow.hpp:205-212
}
}  // namespace vpp
