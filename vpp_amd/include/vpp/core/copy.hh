// copy.hh / clone.hh / sum.hh (reference: vpp/core/copy.hh:10-27, clone.hh:10-19, sum.hh:12-20)
#pragma once
#include <cassert>
#include <vpp/core/pixel_wise.hh>

namespace vpp {
template <class I, class J> void copy(const I& src, J&& dst) { pixel_wise(src, dst) | [](const auto& in, auto& out) { out = in; }; }
template <class I, class J> void copy_with_border(const I& src, J&& dst) {
  assert(src.domain() == dst.domain());
  assert(src.border() <= dst.border());
  pixel_wise(src.domain_with_border(), src, dst) | [](vint2, const auto& in, auto& out) { out = in; };
}
template <class I, class... O> I clone(I img, const O&... options) {
  auto o = opt::make(options...);
  const int border = o.has(_border) ? o.get(_border, 0) : img.border();
  const int aligned = o.has(_aligned) ? o.get(_aligned, 0) : img.alignment();
  assert(aligned != 0);
  I n(img.domain(), _border = border, _aligned = aligned);
  if (img.border() <= border) copy_with_border(img, n); else copy(img, n);
  return n;
}
template <class V, unsigned N> plus_promotion<V> sum(const imageNd<V, N>& img) {
  plus_promotion<V> s = zero<plus_promotion<V>>();
  for (auto p : img.domain()) s = s + vpp::cast<plus_promotion<V>>(img(p));
  return s;
}
}  // namespace vpp
