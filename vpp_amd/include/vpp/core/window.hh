// window.hh — static neighbourhood windows c4 / c5 / c8 / c9 and foreach (reference: vpp/core/window.hh:11-61).
#pragma once
#include <array>
#include <vpp/core/vector.hh>

namespace vpp {
template <unsigned K> struct window { std::array<vint2, K> offsets; const std::array<vint2, K>& operator()() const { return offsets; } };
template <unsigned K, class F> void foreach(const window<K>& w, F f) { for (unsigned i = 0; i < K; i++) f(w.offsets[i]); }
template <class... P> window<sizeof...(P)> make_window(P... p) { return window<sizeof...(P)>{{{p...}}}; }
static const window<9> c9 = make_window(vint2(-1, -1), vint2(-1, 0), vint2(-1, 1), vint2(0, -1), vint2(0, 0), vint2(0, 1), vint2(1, -1), vint2(1, 0), vint2(1, 1));
static const window<8> c8 = make_window(vint2(-1, -1), vint2(-1, 0), vint2(-1, 1), vint2(0, -1), vint2(0, 1), vint2(1, -1), vint2(1, 0), vint2(1, 1));
static const window<5> c5 = make_window(vint2(-1, 0), vint2(0, -1), vint2(0, 0), vint2(0, 1), vint2(1, 0));
static const window<4> c4 = make_window(vint2(-1, 0), vint2(0, -1), vint2(0, 1), vint2(1, 0));
}  // namespace vpp
