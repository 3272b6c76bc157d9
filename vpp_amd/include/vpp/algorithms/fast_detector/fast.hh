// fast.hh — FAST-9 front-ends (reference: vpp/algorithms/fast_detector/fast.hh:26-68, fast.hpp:643-707,745-799,889-955).
// Same names, options and exception; keypoints come back in the serial reference order (row-major pixels, or row-major
// blocks for _blockwise).  Extension options: _fast9_corrected_ring selects the true ring instead of the one
// fast_detector9_simd samples; the default reproduces the reference.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <stdexcept>
#include <type_traits>
#include <vector>
#include <vpp/algorithms/device_only.hh>
#include <vpp/algorithms/symbols.hh>
#include <vpp/core/image2d.hh>

namespace vpp {
namespace fast_internals {
inline std::vector<vint2> detect(const image2d<unsigned char>& A, int th, const image2d<unsigned char>& mask, int mode, int block_size,
                                 int compat, std::vector<int>* scores) {
  const vpp_image_desc da = A.device_desc(false);
  vpp_image_desc dm; const vpp_image_desc* pm = nullptr;
  if (mask.has_data()) { dm = mask.device_desc(false); pm = &dm; }
  int capacity = std::max(1024, (A.nrows() * A.ncols()) / 32), count = 0;
  std::vector<vint2> kps;
  for (int attempt = 0; attempt < 2; attempt++) {
    device::dbuf rc(size_t(capacity) * 8), sc(size_t(capacity) * 4);
    const int st = vpp_fast9_detect(&da, th, pm, mode, block_size, compat, (int32_t*)rc.p, (int32_t*)sc.p, capacity, &count, device::stream());
    if (st == VPP_ERR_CAPACITY) {
      if (attempt == 1) throw std::runtime_error("fast9: the detector reported more keypoints than the capacity it had just asked for");
      capacity = count;
      continue;
    }
    device::check(st, "vpp_fast9_detect");  // border < 3 -> std::runtime_error("Image need a border of 3px ...") like fast.hpp:937-938
    kps.resize(count);
    static_assert(sizeof(vint2) == 8, "vint2 must be two packed ints");
    rc.download(kps.data(), size_t(count) * 8);
    if (scores) { scores->resize(count); sc.download(scores->data(), size_t(count) * 4); }
    break;
  }
  return kps;
}
}  // namespace fast_internals

template <class V, class... OPTS> std::vector<vint2> fast9(const image2d<V>& A, int th, OPTS... opts_) {
  static_assert(sizeof(V) == 1, "fast9: 8-bit single-channel images");
  auto opts = opt::make(opts_...);
  if (A.border() < 3) throw std::runtime_error("Image need a border of 3px at least for the FAST detector");
  image2d<unsigned char> mask = opts.get(_mask, image2d<unsigned char>());
  std::vector<int>* scores = opts.get(_scores, (std::vector<int>*)nullptr);
  const int block_size = opts.get(_block_size, 10);
  const int compat = opts.has(_fast9_corrected_ring) ? VPP_FAST9_CORRECTED : VPP_FAST9_REFERENCE;
  const int mode = opts.has(_local_maxima) ? VPP_FAST9_LOCAL_MAXIMA : (opts.has(_blockwise) ? VPP_FAST9_BLOCKWISE : VPP_FAST9_RAW);
  return fast_internals::detect(*(const image2d<unsigned char>*)&A, th, mask, mode, block_size, compat, scores);
}

template <class V, class KPS> void fast9_scores(const image2d<V>& A, int th, const KPS& keypoints, std::vector<int>& scores) {
  const int n = int(keypoints.size());
  scores.resize(n);
  if (!n) return;
  std::vector<vint2> pts(n);
  for (int i = 0; i < n; i++) pts[i] = vint2(keypoints[i][0], keypoints[i][1]);
  device::dbuf rc(size_t(n) * 8), sc(size_t(n) * 4);
  rc.upload(pts.data(), size_t(n) * 8);
  const vpp_image_desc da = A.device_desc(false);
  device::check(vpp_fast9_scores(&da, th, (const int32_t*)rc.p, n, (int32_t*)sc.p, device::stream()), "vpp_fast9_scores");
  sc.download(scores.data(), size_t(n) * 4);
}
template <class V> int fast9_score(const image2d<V>& A, int th, vint2 p) {
  std::vector<vint2> k(1, p); std::vector<int> s;
  fast9_scores(A, th, k, s);
  return s[0];
}

// Dense detector on the true ring (fast.hpp:511-551): B(p) = 1 where 9 contiguous ring pixels are all brighter than A(p) + th or
// all darker than A(p) - th, else 0.  A needs a border of 3.
namespace FAST_internals {
template <class V, class U> void fast_detector9(const image2d<V>& A, image2d<U>& B, int th) {
  static_assert(sizeof(V) == 1, "fast_detector9: 8-bit single-channel source");
  static_assert(std::is_same<U, unsigned char>::value || std::is_same<U, int>::value, "fast_detector9: unsigned char or int flags");
  const vpp_image_desc da = A.device_desc(false), db = B.device_desc(true, B.border() == 0);
  device::check(vpp_fast9_dense(&db, &da, th, device::stream()), "vpp_fast9_dense");
}
}  // namespace FAST_internals

// blockwise_maxima_filter (fast.hpp:577-614), in place: per block only the first strict maximum > 0 survives.  (The reference
// local_maxima_filter(A, nbh_size) (fast.hpp:555-575; nbh_size is ignored there too), in place: the reference's serial raster-order result
template <class V> void local_maxima_filter(const image2d<V>& A, int /*nbh_size*/) {
  const vpp_image_desc da = A.device_desc(true);
  device::check(vpp_local_maxima_filter(&da, device::stream()), "vpp_local_maxima_filter");
}

// declares it on a const image and does not compile when instantiated; it is callable here.)
template <class V> void blockwise_maxima_filter(const image2d<V>& A, int block_size) {
  const vpp_image_desc da = A.device_desc(true);
  device::check(vpp_blockwise_maxima_filter(&da, block_size, device::stream()), "vpp_blockwise_maxima_filter");
}

// "old API" spellings (fast.hh:41-68)
template <class V> std::vector<vint2> fast_detector9(const image2d<V>& A, int th, const image2d<unsigned char>& mask = image2d<unsigned char>(), std::vector<int>* scores = nullptr) {
  return fast9(A, th, _mask = mask, _scores = scores);
}
template <class V> std::vector<vint2> fast_detector9_local_maxima(const image2d<V>& A, int th, const image2d<unsigned char>& mask = image2d<unsigned char>(), std::vector<int>* scores = nullptr) {
  return fast9(A, th, _local_maxima, _mask = mask, _scores = scores);
}
template <class V> std::vector<vint2> fast_detector9_blockwise_maxima(const image2d<V>& A, int th, int block_size, const image2d<unsigned char>& mask = image2d<unsigned char>(), std::vector<int>* scores = nullptr) {
  return fast9(A, th, _blockwise, _block_size = block_size, _mask = mask, _scores = scores);
}
}  // namespace vpp
