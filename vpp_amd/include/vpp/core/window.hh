// window.hh — static neighbourhood windows c4 / c5 / c8 / c9 and foreach (reference: vpp/core/window.hh:11-75).  A window wraps a
// callable that returns the array of offsets, so that `foreach` over a window defined by a lambda inlines to straight-line code.
#pragma once
#include <vpp/core/make_array.hh>
#include <vpp/core/vector.hh>

namespace vpp {
template <class N> struct window {
  explicit window(N n) : offsets(n) {}
  decltype(auto) operator()() { return offsets(); }
  N offsets;
};
template <class F> window<F> make_window(F f) { return window<F>(f); }
template <class N, class F> void foreach(window<N> w, F f) {
  auto o = w();
  for (std::size_t i = 0; i < o.size(); i++) f(vint2(o[i][0], o[i][1]));
}
static auto c9 = make_window([] { return make_array(vint2{-1, -1}, vint2{-1, 0}, vint2{-1, 1}, vint2{0, -1}, vint2{0, 0}, vint2{0, 1}, vint2{1, -1}, vint2{1, 0}, vint2{1, 1}); });
static auto c8 = make_window([] { return make_array(vint2{-1, -1}, vint2{-1, 0}, vint2{-1, 1}, vint2{0, -1}, vint2{0, 1}, vint2{1, -1}, vint2{1, 0}, vint2{1, 1}); });
static auto c5 = make_window([] { return make_array(vint2{-1, 0}, vint2{0, -1}, vint2{0, 0}, vint2{0, 1}, vint2{1, 0}); });
static auto c4 = make_window([] { return make_array(vint2{-1, 0}, vint2{0, -1}, vint2{0, 1}, vint2{1, 0}); });
}  // namespace vpp
