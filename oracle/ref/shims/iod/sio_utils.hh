#pragma once
#include "sio.hh"

namespace iod {

namespace internal {
// one option -> one sio member: `_x = v` keeps v, a bare flag symbol becomes a bool member set to true
template <class A, class = void> struct member_of;
template <class S, class V> struct member_of<assign_exp<S, V>, void> {
  typedef typename S::template variable_type<V> type;
  static type make(const assign_exp<S, V>& a) { return type(a.value); }
};
template <class S> struct member_of<S, typename std::enable_if<std::is_base_of<symbol<S>, S>::value>::type> {
  typedef typename S::template variable_type<bool> type;
  static type make(const S&) { return type(true); }
};
}  // namespace internal

inline sio<> D() { return sio<>(); }
template <class... A> auto D(const A&... a) {
  return sio<typename internal::member_of<A>::type...>(internal::member_of<A>::make(a)...);
}

template <bool C, class F, class G, class... A, class = typename std::enable_if<C>::type>
decltype(auto) static_if(F&& f, G&&, A&&... a) { return f(std::forward<A>(a)...); }
template <bool C, class F, class G, class... A, class = typename std::enable_if<!C>::type, class = void>
decltype(auto) static_if(F&&, G&& g, A&&... a) { return g(std::forward<A>(a)...); }

}  // namespace iod
