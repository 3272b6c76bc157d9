// fill.hh — fill / border fills (reference: vpp/core/fill.hh:12-122).  Host implementations of the value fills;
// the three border fills run on the device when the image's HBM mirror is the current copy (vpp_fill_border).
#pragma once
#include <vpp/core/pixel_wise.hh>

namespace vpp {

template <class V, class U, unsigned N> void fill(imageNd<V, N>& img, U&& value) { pixel_wise(img) | [=](V& pix) { pix = value; }; }
template <class V, unsigned N> void fill(imageNd<V, N>& img, V value, const boxNd<N>& box) { pixel_wise(box, img) | [=](vint2, V& pix) { pix = value; }; }
template <class V, class U, unsigned N> void fill_with_border(imageNd<V, N>& img, U&& value) {
  auto box = img.domain_with_border();
  pixel_wise(box, img) | [=](vint2, V& pix) { pix = value; };
}

namespace detail {
// mode 0 mirror (edge pixel repeated: (-k) <- (k-1)), 1 closest, 2 value  — fill.hh:48-122
template <class V> void fill_border_host(image2d<V>& img, int mode, const V* value) {
  const int b = img.border(), nr = img.nrows(), nc = img.ncols();
  img.host_write();
  for (int r = -b; r < nr + b; r++) {
    V* row = img[r];
    const int sr = mode == 0 ? (r < 0 ? -r - 1 : (r >= nr ? 2 * nr - r - 1 : r)) : (r < 0 ? 0 : (r >= nr ? nr - 1 : r));
    const V* srow = img[sr];
    for (int c = -b; c < nc + b; c++) {
      if (r >= 0 && r < nr && c >= 0 && c < nc) { c = nc - 1; continue; }
      if (mode == 2) { row[c] = *value; continue; }
      const int sc = mode == 0 ? (c < 0 ? -c - 1 : (c >= nc ? 2 * nc - c - 1 : c)) : (c < 0 ? 0 : (c >= nc ? nc - 1 : c));
      row[c] = srow[sc];
    }
  }
}
template <class V> void fill_border_any(image2d<V>& img, int mode, const V* value) {
#ifdef VPP_AMD_DEVICE
  const vpp_image_desc d = img.device_desc(true);
  device::check(vpp_fill_border(&d, mode, value, device::stream()), "vpp_fill_border");
  device::call_done();   // queued, not drained: vpp/core/device.hh
#else
  fill_border_host(img, mode, value);
#endif
}
}  // namespace detail

template <class V, class U> void fill_border_with_value(image2d<V>& img, U&& value) { V v = value; detail::fill_border_any(img, 2, &v); }
template <class V> void fill_border_mirror(image2d<V>& img) { detail::fill_border_any<V>(img, 0, nullptr); }
template <class V> void fill_border_closest(image2d<V>& img) { detail::fill_border_any<V>(img, 1, nullptr); }

}  // namespace vpp
