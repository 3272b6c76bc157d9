// const.hh — constify / unconstify (reference: vpp/core/const.hh:6-12).
#pragma once
#include <type_traits>

namespace vpp {
template <class T> using unconstify = typename std::remove_const<T>::type;
template <class T> using constify = typename std::add_const<T>::type;
}  // namespace vpp
