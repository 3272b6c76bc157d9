// colorspace_conversions.hh — rgb_to_graylevel / graylevel_to_rgb (reference: vpp/core/colorspace_conversions.hh:10-63).
// 8-bit rgb / rgba -> 8-bit gray runs on the device (vpp_rgb_to_graylevel); every other instantiation is the reference's
// pixel_wise lambda, evaluated by the host expression engine like any user kernel.
#pragma once
#include <type_traits>
#include <vpp/core/pixel_wise.hh>

namespace vpp {

template <class T, class U> VPP_HD void rgb_to_graylevel(const vector<U, 3>& i, vector<T, 1>& o) { o[0] = (i[0] + i[1] + i[2]) / 3; }
template <class T, class U> VPP_HD void rgb_to_graylevel(const vector<U, 3>& i, T& o) { o = (i[0] + i[1] + i[2]) / 3; }

namespace colorspace_internals {
template <class T> struct is_u8_gray : std::integral_constant<bool, sizeof(T) == 1 && (std::is_same<T, unsigned char>::value || std::is_same<T, vector<unsigned char, 1>>::value)> {};
}

// immediate: launch at once instead of holding the frame back for a batched launch (vpp/core/device.hh; the `_immediate` option of the overloads below)
template <class T, class U, unsigned N, unsigned C> imageNd<T, N> rgb_to_graylevel_impl(const imageNd<vector<U, C>, N>& in, bool immediate = false) {
  imageNd<T, N> out(in.domain(), _border = in.border(), _aligned = in.alignment());
#ifdef VPP_AMD_DEVICE
  if constexpr (N == 2 && std::is_same<U, unsigned char>::value && colorspace_internals::is_u8_gray<T>::value) {
    const vpp_image_desc di = in.device_desc(false), dout = out.device_desc(true, true);
    if (immediate) { device::check(vpp_rgb_to_graylevel(&dout, &di, 0, device::stream()), "vpp_rgb_to_graylevel"); device::call_done(); return out; }
    device::check(vpp_rgb_to_graylevel_deferred(&dout, &di, 0, device::stream()), "vpp_rgb_to_graylevel");
    device::deferred_call_done();   // held back and launched in batches by the library: vpp/core/device.hh
    return out;
  }
#endif
  pixel_wise(in.domain_with_border(), in, out) | [](vint<N>, const vector<U, C>& i, T& o) {
    vector<U, 3> tmp(i[0], i[1], i[2]);
    rgb_to_graylevel(tmp, o);
  };
  return out;
}
template <class T, class U, unsigned N> imageNd<T, N> rgb_to_graylevel(const imageNd<vector<U, 3>, N>& in) { return rgb_to_graylevel_impl<T, U, N, 3>(in); }
template <class T, class U, unsigned N> imageNd<T, N> rgb_to_graylevel(const imageNd<vector<U, 4>, N>& in) { return rgb_to_graylevel_impl<T, U, N, 4>(in); }
template <class T, class U, unsigned N> imageNd<T, N> rgb_to_graylevel(const imageNd<vector<U, 3>, N>& in, s::_immediate_t) { return rgb_to_graylevel_impl<T, U, N, 3>(in, true); }
template <class T, class U, unsigned N> imageNd<T, N> rgb_to_graylevel(const imageNd<vector<U, 4>, N>& in, s::_immediate_t) { return rgb_to_graylevel_impl<T, U, N, 4>(in, true); }

#ifdef VPP_AMD_DEVICE
// Frame ingest, one pass on the device: the result of `auto f = clone(frame, _border = border); fill_border_mirror(f);
// return rgb_to_graylevel<unsigned char>(f);` (examples/video_extruder.cc:46-48) without the intermediate rgb image.
template <unsigned C> image2d<unsigned char> rgb_to_graylevel_mirror(const image2d<vector<unsigned char, C>>& frame, int border, int aligned = 32) {
  static_assert(C == 3 || C == 4, "rgb or rgba");
  image2d<unsigned char> out(frame.domain(), _border = border, _aligned = aligned);
  const vpp_image_desc di = frame.device_desc(false), dout = out.device_desc(true, true);
  device::check(vpp_rgb_to_graylevel_deferred(&dout, &di, 1, device::stream()), "vpp_rgb_to_graylevel");
  device::deferred_call_done();   // held back and launched in batches by the library: vpp/core/device.hh
  return out;
}
#endif

template <class T, class U, unsigned N> imageNd<T, N> graylevel_to_rgb(const imageNd<U, N>& in) {
  imageNd<T, N> out(in.domain(), _border = in.border(), _aligned = in.alignment());
  pixel_wise(in.domain_with_border(), in, out) | [](vint<N>, const U& i, T& o) { o = T(i, i, i); };
  return out;
}

}  // namespace vpp
