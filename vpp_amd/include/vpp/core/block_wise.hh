// block_wise.hh — map a kernel over a grid of blocks of the ranges (reference: vpp/core/block_wise.hh:26-78).
// The kernel receives, per range, the sub-image / sub-box of the block (clipped to the domain).  Host evaluation; the
// traversal / threading options are those of pixel_wise and apply to the grid of blocks.
#pragma once
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
  template <class F> void operator|(F fun) { run(fun, std::index_sequence_for<R...>()); }
 private:
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
