// block_wise.hh — map a kernel over a grid of blocks of the ranges (reference: vpp/core/block_wise.hh:26-78).
// The kernel receives, per range, the sub-image / sub-box of the block (clipped to the domain).  The traversal / threading options
// are those of pixel_wise and apply to the grid of blocks.  Evaluation routes:
//  * host: any callable, sub-images are real image2d views (nested pixel_wise, fill, ... work);
//  * -DVPP_AMD_DEVICE: the tagged functor ops::block_maxima runs vpp_blockwise_maxima_filter (fast.hpp:745-799's per-block step);
//  * single-source hipcc build (-DVPP_AMD_HIPCC): a stateless callable with default options is the body of a gfx950 kernel, one
//    lane per block; it sees pwdev::block_view<V> / pwdev::box_view instead of image2d / box2d (view(r, c), view[r], nrows(), ncols()),
//    so it must be written against that subset (generic `auto` parameters); `_host` keeps a call on the host.
#pragma once
#include <stdexcept>
#include <tuple>
#include <vpp/core/pixel_wise.hh>

namespace vpp {
namespace bw {
template <class V> image2d<V> cut(image2d<V>& img, const box2d& b) { return img | b; }
inline box2d cut(const box2d&, const box2d& b) { return b; }  // `box | b` is b (boxNd.hh:150)
}  // namespace bw

template <class OPTS, class... R> class block_wise_runner {
 public:
  block_wise_runner(vint2 bs, std::tuple<R...> t, OPTS o = OPTS()) : block_size_(bs), ranges_(t), options_(o) {}
  template <class... A> auto operator()(A... o) const { auto n = opt::make(o...); return block_wise_runner<decltype(n), R...>(block_size_, ranges_, n); }
  template <class F> void operator|(F fun) {
#if defined(VPP_AMD_DEVICE) && defined(__HIPCC__) && defined(VPP_AMD_HIPCC)
    if constexpr (device_eligible<F>()) { run_device(fun, std::index_sequence_for<R...>()); return; }
#endif
    run(fun, std::index_sequence_for<R...>());
  }
#ifdef VPP_AMD_DEVICE
  // keep, per block, only the first strict maximum > 0 in raster order, zero the rest (fast.hpp:773-789 as an image filter)
  void operator|(ops::block_maxima) {
    static_assert(sizeof...(R) == 1, "ops::block_maxima needs block_wise(block_size, image)");
    auto& img = std::get<0>(ranges_);
    if (block_size_[0] != block_size_[1]) throw std::runtime_error("block_wise | ops::block_maxima: square blocks only");
    const vpp_image_desc d = img.device_desc(true);
    device::check(vpp_blockwise_maxima_filter(&d, block_size_[0], device::stream()), "vpp_blockwise_maxima_filter");
    device::call_done();   // queued, not drained: vpp/core/device.hh
  }
#endif
 private:
#if defined(VPP_AMD_DEVICE) && defined(__HIPCC__) && defined(VPP_AMD_HIPCC)
  template <class T> struct block_range : std::false_type {};
  template <class V> struct block_range<imageNd<V, 2>> : std::is_trivially_copyable<V> {};
  template <class F> static constexpr bool device_eligible() {
    return std::is_trivially_copyable<F>::value && (std::is_empty<F>::value || OPTS::has(_device)) && !OPTS::has(_host) && !OPTS::has(_no_threads) &&
           !OPTS::has(_right_to_left) && !OPTS::has(_bottom_to_top) && !OPTS::has(_left_to_right) && !OPTS::has(_top_to_bottom) &&
           !OPTS::has(_mem_forward) && !OPTS::has(_mem_backward) && ((block_range<R>::value || std::is_same<R, box2d>::value) && ...);
  }
  template <class F, std::size_t... I> void run_device(F& fun, std::index_sequence<I...>) {
    const auto p1 = std::get<0>(ranges_).first_point_coordinates();
    const auto p2 = std::get<0>(ranges_).last_point_coordinates();
    pwdev::launch_blocks(fun, p1[0], p1[1], p2[0], p2[1], block_size_[0], block_size_[1], pw::device_accessor(std::get<I>(ranges_))...);
  }
#endif
  template <class F, std::size_t... I> void run(F& fun, std::index_sequence<I...>) {
    const auto p1 = std::get<0>(ranges_).first_point_coordinates();
    const auto p2 = std::get<0>(ranges_).last_point_coordinates();
    const int rstart = p1[0], rend = p2[0], cstart = p1[1], cend = p2[1];
    const int nr = (rend - rstart) / block_size_[0] + 1, nc = (cend - cstart) / block_size_[1] + 1;  // ceil(extent / block), block_wise.hh:37-38
    box2d grid(vint2(0, 0), vint2(nr - 1, nc - 1));
    auto body = [&](vint2 b) {
      const vint2 q1(rstart + b[0] * block_size_[0], cstart + b[1] * block_size_[1]);
      const vint2 q2(std::min(q1[0] + block_size_[0] - 1, rend), std::min(q1[1] + block_size_[1] - 1, cend));
      const box2d blk(q1, q2);  // in the coordinates of the first range, like the reference (block_wise.hh:45-54)
      fun(bw::cut(std::get<I>(ranges_), blk)...);
    };
    pixel_wise(grid)(options_) | body;
  }
  vint2 block_size_;
  std::tuple<R...> ranges_;
  OPTS options_;
};
template <class... R> auto block_wise(vint2 block_size, R&&... r) {
  return block_wise_runner<opt::set<>, typename std::decay<R>::type...>(block_size, std::tuple<typename std::decay<R>::type...>(r...));
}
template <class P0, class... R> auto row_wise(P0&& a, R&&... r) {  // one block per row (block_wise.hh:71-78)
  const auto p1 = a.first_point_coordinates(); const auto p2 = a.last_point_coordinates();
  return block_wise(vint2(1, p2[1] - p1[1] + 1), a, r...);
}
}  // namespace vpp
