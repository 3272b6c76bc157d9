// Minimal stand-in for matt-42/iod (named-parameter library), covering what vpp's hot path uses:
// iod_define_symbol, `_x = value` options, bare flag symbols, iod::D -> iod::sio with has / get / named members,
// iod::static_if, iod::has_symbol, iod::array_view.  TEST INFRASTRUCTURE for oracle/ref only.  No arithmetic lives here.
#pragma once
#include <type_traits>
#include <utility>

namespace iod {

template <class S, class V> struct assign_exp { typedef S symbol_type; typedef V value_type; V value; };

template <class S> struct symbol {
  template <class V> constexpr assign_exp<S, typename std::decay<V>::type> operator=(V&& v) const {
    return assign_exp<S, typename std::decay<V>::type>{std::forward<V>(v)};
  }
};

}  // namespace iod

#define iod_define_symbol(NAME)                                                              \
  namespace s {                                                                              \
  struct _##NAME##_t : iod::symbol<_##NAME##_t> {                                            \
    using iod::symbol<_##NAME##_t>::operator=;                                               \
    template <class T> struct variable_type {                                                \
      typedef T value_type;                                                                  \
      typedef _##NAME##_t symbol_type;                                                       \
      T NAME;                                                                                \
      variable_type() : NAME() {}                                                            \
      variable_type(const T& v) : NAME(v) {}                                                 \
      const T& iod_member() const { return NAME; }                                           \
    };                                                                                       \
  };                                                                                         \
  static constexpr _##NAME##_t _##NAME{};                                                    \
  }

#define iod_define_number_symbol(NUM)                                                        \
  namespace s {                                                                              \
  struct _##NUM##_t : iod::symbol<_##NUM##_t> {                                              \
    using iod::symbol<_##NUM##_t>::operator=;                                                \
    template <class T> struct variable_type {                                                \
      typedef T value_type;                                                                  \
      typedef _##NUM##_t symbol_type;                                                        \
      T n##NUM;                                                                              \
      variable_type() : n##NUM() {}                                                          \
      variable_type(const T& v) : n##NUM(v) {}                                               \
      const T& iod_member() const { return n##NUM; }                                         \
    };                                                                                       \
  };                                                                                         \
  static constexpr _##NUM##_t _##NUM{};                                                      \
  }
