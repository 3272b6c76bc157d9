#pragma once
#include "symbol.hh"

namespace iod {

namespace internal {
template <class S, class... M> struct find_member { typedef void type; };
template <class S, class M0, class... M> struct find_member<S, M0, M...> {
  typedef typename std::conditional<std::is_same<typename M0::symbol_type, S>::value, M0, typename find_member<S, M...>::type>::type type;
};
}  // namespace internal

template <class... M> struct sio : M... {
  sio() {}
  template <class... A, class = typename std::enable_if<sizeof...(A) == sizeof...(M) && (sizeof...(A) > 0)>::type>
  explicit sio(const A&... a) : M(a)... {}

  template <class S> static constexpr bool has(const S&) { return !std::is_void<typename internal::find_member<S, M...>::type>::value; }

  template <class S, class D> auto get(const S&, const D& dflt) const {
    typedef typename internal::find_member<S, M...>::type Mem;
    if constexpr (std::is_void<Mem>::value) return dflt;
    else return static_cast<const Mem&>(*this).iod_member();
  }
};

template <class O, class S> struct has_symbol { static constexpr bool value = false; };
template <class... M, class S> struct has_symbol<sio<M...>, S> {
  static constexpr bool value = !std::is_void<typename internal::find_member<S, M...>::type>::value;
};

}  // namespace iod
