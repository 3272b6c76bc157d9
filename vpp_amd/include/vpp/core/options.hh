// options.hh — named options (`_border = 3`, bare flags such as `_no_threads`) for the vpp-shaped API.
// Replaces the third-party iod library the reference uses for the same syntax (reference: vpp/core/symbols.hh,
// vpp/core/imageNd.hpp:99-141, README.md:85-116,177-191).  Only the call-site syntax is kept compatible.
#pragma once
#include <type_traits>
#include <utility>

namespace vpp {
namespace opt {

template <class S, class V> struct bound { typedef S symbol_type; V value; };

template <class S> struct symbol {
  template <class V> constexpr bound<S, typename std::decay<V>::type> operator=(V&& v) const {
    return bound<S, typename std::decay<V>::type>{std::forward<V>(v)};
  }
};

namespace detail {
template <class A> struct as_bound { typedef A type; static const A& get(const A& a) { return a; } };  // already `_x = v`
template <class S, class... B> struct find { typedef void type; };
template <class S, class B0, class... B> struct find<S, B0, B...> {
  typedef typename std::conditional<std::is_same<typename B0::symbol_type, S>::value, B0, typename find<S, B...>::type>::type type;
};
template <class A, class = void> struct normalize { typedef A type; static A make(const A& a) { return a; } };
template <class S> struct normalize<S, typename std::enable_if<std::is_base_of<symbol<S>, S>::value>::type> {
  typedef bound<S, bool> type;  // a bare flag symbol
  static type make(const S&) { return type{true}; }
};
}  // namespace detail

// The set of options of one call.
template <class... B> struct set : B... {
  set() {}
  template <class... A, class = typename std::enable_if<(sizeof...(A) > 0) && sizeof...(A) == sizeof...(B)>::type>
  explicit set(const A&... b) : B(b)... {}
  template <class S> static constexpr bool has(const S&) { return !std::is_void<typename detail::find<S, B...>::type>::value; }
  template <class S, class D> auto get(const S&, const D& dflt) const {
    typedef typename detail::find<S, B...>::type M;
    if constexpr (std::is_void<M>::value) return dflt;
    else return static_cast<const M&>(*this).value;
  }
};
template <class O, class S> struct has_symbol : std::false_type {};
template <class... B, class S> struct has_symbol<set<B...>, S> : std::integral_constant<bool, !std::is_void<typename detail::find<S, B...>::type>::value> {};

inline set<> make() { return set<>(); }
template <class... A> auto make(const A&... a) { return set<typename detail::normalize<A>::type...>(detail::normalize<A>::make(a)...); }

}  // namespace opt
}  // namespace vpp

#define VPP_DEFINE_SYMBOL(NAME)                                   \
  namespace vpp { namespace s {                                   \
  struct _##NAME##_t : ::vpp::opt::symbol<_##NAME##_t> {          \
    using ::vpp::opt::symbol<_##NAME##_t>::operator=;             \
  };                                                              \
  static constexpr _##NAME##_t _##NAME{};                         \
  } }
