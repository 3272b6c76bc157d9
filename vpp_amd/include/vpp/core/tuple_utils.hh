// tuple_utils.hh — calling functions with the elements of a tuple (reference: vpp/core/tuple_utils.hh; the names pixel_wise and
// block_wise are written against: internals::apply_args / apply_args_star / apply_args_transform / tuple_map / tuple_transform).
#pragma once
#include <cstddef>
#include <tuple>
#include <utility>

namespace vpp {
namespace internals {
namespace tu {
template <class T, class F, std::size_t... I> void call(T& t, F&& f, std::index_sequence<I...>) { f(std::get<I>(t)...); }
template <class T, class F, std::size_t... I> void call_star(T& t, F&& f, std::index_sequence<I...>) { f(*std::get<I>(t)...); }
template <class T, class F, class G, std::size_t... I> void call_through(T& t, F& f, G& g, std::index_sequence<I...>) { f(g(std::get<I>(t))...); }
template <class T, class F, std::size_t... I> void each(T& t, F& f, std::index_sequence<I...>) { (void)std::initializer_list<int>{((void)f(std::get<I>(t)), 0)...}; }
template <class T, class F, std::size_t... I> auto mapped(T& t, F& f, std::index_sequence<I...>) { return std::make_tuple(f(std::get<I>(t))...); }
}  // namespace tu

// f(t0, t1, ...)
template <class... A, class F> void apply_args(std::tuple<A...>& t, F&& f) { tu::call(t, f, std::index_sequence_for<A...>()); }
// f(*t0, *t1, ...)
template <class... A, class F> void apply_args_star(std::tuple<A...>& t, F&& f) { tu::call_star(t, f, std::index_sequence_for<A...>()); }
// f(g(t0), g(t1), ...)
template <class... A, class F, class G> void apply_args_transform(std::tuple<A...>& t, F f, G g) { tu::call_through(t, f, g, std::index_sequence_for<A...>()); }
// f(t0); f(t1); ...
template <class F, class... A> void tuple_map(std::tuple<A...>& t, F f) { tu::each(t, f, std::index_sequence_for<A...>()); }
// make_tuple(f(t0), f(t1), ...)
template <class F, class... A> auto tuple_transform(std::tuple<A...>& t, F f) { return tu::mapped(t, f, std::index_sequence_for<A...>()); }
}  // namespace internals
}  // namespace vpp
