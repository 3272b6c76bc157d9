#pragma once
namespace iod {
template <class F> struct array_view_t {
  int n; F f;
  int size() const { return n; }
  auto operator[](int i) const { return f(i); }
};
template <class F> array_view_t<F> array_view(int n, F f) { return array_view_t<F>{n, f}; }
}  // namespace iod
