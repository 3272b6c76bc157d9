// lk.hh — matcher tag types of pyrlk_match (reference: vpp/algorithms/pyrlk/lk.hh:10-38).  The matcher's body
// (lk.hh:43-175) is the gfx950 kernel pyrlk_match_kernel<WS>; here the type only carries the window size.
#pragma once
namespace vpp {
template <unsigned WS> struct lk_match_point_square_win { enum { window_size = WS }; };
// the direction-constrained variant of the reference (lk.hh:24-41,181-): no algorithm on the replaced path uses it; a tag only
template <unsigned WS> struct oriented_lk_match_point_square_win { enum { window_size = WS }; };
}  // namespace vpp
