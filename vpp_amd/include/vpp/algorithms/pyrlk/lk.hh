// lk.hh — matcher tag types of pyrlk_match (reference: vpp/algorithms/pyrlk/lk.hh:10-38).  The matcher's body
// (lk.hh:43-175) is the gfx950 kernel pyrlk_match_kernel<WS>; here the type only carries the window size.
#pragma once
namespace vpp {
template <unsigned WS> struct lk_match_point_square_win { enum { window_size = WS }; };
// The direction-constrained variant of the reference (lk.hh:24-41,181-317) computes something else (1-D search along a
// direction, `k < max_it`): no algorithm on the replaced path uses it and there is no kernel for it.  The name exists so that
// code mentioning it gets a readable error instead of silently running the square-window kernel (pyrlk_match.hh checks it).
template <unsigned WS> struct oriented_lk_match_point_square_win { enum { window_size = WS, vpp_amd_unsupported = 1 }; };
}  // namespace vpp
