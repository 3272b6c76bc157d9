// make_array.hh — std::array from a list of values of one type (reference: vpp/core/make_array.hh:8-16; the element type is the
// first argument's, as written there).
#pragma once
#include <array>
#include <utility>

namespace vpp {
template <class First, class... Rest> decltype(auto) make_array(First&& first, Rest&&... rest) {
  return std::array<First, 1 + sizeof...(Rest)>{{std::forward<First>(first), std::forward<Rest>(rest)...}};
}
}  // namespace vpp
