// relative_accessor.hh — access to the neighbourhood of one pixel (reference: vpp/core/relative_accessor.hh:16-33):
//   auto ra = relative_accessor(img, vint2(10, 10));  ra(-1, -1) is the north-west neighbour, ra(vint2(dr, dc)) the same by offset.
// Built on the image's row-pointer table (imageNd.hpp:176-186), like the reference's.
#pragma once
#include <vpp/core/image2d.hh>

namespace vpp {
template <class V> struct relative_access_kernel {
  V* const* line; int col;
  V& operator()(int dr, int dc) const { return line[dr][col + dc]; }
  V& operator()(vint2 p) const { return line[p[0]][col + p[1]]; }
};
template <class V> relative_access_kernel<V> relative_accessor(const image2d<V>& img, vint2 p) {
  return relative_access_kernel<V>{&img[p[0]], p[1]};
}
}  // namespace vpp
