// pyramid.hh — image pyramids (reference: vpp/core/pyramid.hh:126-221).  Level sizes, the border option and the
// update / propagate_level0 protocol are the reference's; the per-level work (5-tap binomial low-pass + subsample2 +
// mirror border, pyramid.hh:12-81) is the gfx950 kernel behind vpp_pyr_down.  Factor 2 only (what the hot path uses).
#pragma once
#include <algorithm>
#include <vector>
#include <vpp/core/clone.hh>
#include <vpp/core/copy.hh>
#include <vpp/core/fill.hh>

namespace vpp {

// The free functions the pyramid is made of (pyramid.hh:12-123).  The low-pass runs on the device (vpp_lowpass5: both passes and
// the mirror of the temporary in one kernel); the two samplers are plain index maps evaluated by the host engine like any other
// opaque pixel_wise kernel (the pyramid itself never calls them: vpp_pyr_down computes only the samples subsample2 keeps).
template <class V> void antialiasing_lowpass_filter(const image2d<V>& in, image2d<V>& out) {  // :12-59, `in` needs a border of 2
#ifdef VPP_AMD_DEVICE
  const vpp_image_desc di = in.device_desc(false), dout = out.device_desc(true);
  device::check(vpp_lowpass5(&dout, &di, device::stream()), "vpp_lowpass5");
#else
  static_assert(sizeof(V) == 0, "antialiasing_lowpass_filter runs on the device: build with -DVPP_AMD_DEVICE and link libvpp_amd");
#endif
}
template <class V> void subsample2(const image2d<V>& in, image2d<V>& out) {  // :62-81: out(r, c) = in(2r, 2c), no bounds check in the reference either
  for (int r = 0; r < out.nrows(); r++) for (int c = 0; c < out.ncols(); c++) out(r, c) = in(r * 2, c * 2);
}
template <class V> void subsample(const image2d<V>& in, image2d<V>& out, float factor) {  // :84-103
  for (int r = 0; r < out.nrows(); r++) for (int c = 0; c < out.ncols(); c++) out(r, c) = in(int(r * factor), int(c * factor));
}
template <class V> image2d<V> antialias_subsample2(const image2d<V>& in) {  // :105-123
  auto tmp = clone(in, _border = std::max(in.border(), 1));
  fill_border_mirror(tmp);
  image2d<V> in2 = in;
  if (in2.border() < 2) { in2 = clone(in2, _border = 2); fill_border_mirror(in2); }
  antialiasing_lowpass_filter(in2, tmp);
  image2d<V> tmp2(1 + (in.nrows() / 2), 1 + (in.ncols() / 2), _border = std::max(in.border(), 0));
  subsample2(tmp, tmp2);
  fill_border_mirror(tmp2);
  return tmp2;
}

template <class V, unsigned N> struct pyramid {
  typedef imageNd<V, N> image_type;
  template <class... O> pyramid(boxNd<N> d, int nlevels, float factor, const O&... image_options) : levels_(nlevels), factor_(factor) {
    for (int i = 0; i < nlevels; i++) {
      levels_[i] = image_type(d, image_options...);
      d = make_box2d(1 + (d.nrows() / factor), 1 + (d.ncols() / factor));
    }
  }
  template <class... O> pyramid(const image_type& img, int nlevels, float factor, const O&... image_options) : pyramid(img.domain(), nlevels, factor, image_options...) { update(img); }

  image_type& operator[](unsigned i) { return levels_[i]; }
  const image_type& operator[](unsigned i) const { return levels_[i]; }

  void propagate_level0() {
#ifdef VPP_AMD_DEVICE
    if (factor_ != 2.f) throw std::runtime_error("pyramid: only factor 2 is implemented on the device");
    fill_border_mirror(levels_[0]);
    for (size_t i = 1; i < levels_.size(); i++) {
      // the low-pass reads 2 pixels past each edge: the reference filters from a border-2 copy whatever the level's own border
      // is (pyramid.hh:15-36), so a pyramid built with _border < 2 (or none) gets a mirror-filled border-2 temporary here
      image_type src = levels_[i - 1];
      if (src.border() < 2) { src = clone(src, _border = 2); fill_border_mirror(src); }
      const vpp_image_desc prev = src.device_desc(false), next = levels_[i].device_desc(true);
      device::check(vpp_pyr_down(&next, &prev, device::stream()), "vpp_pyr_down");
    }
    device::call_done();   // queued, not drained: vpp/core/device.hh
#else
    static_assert(sizeof(V) == 0, "pyramid::propagate_level0 runs on the device: build with -DVPP_AMD_DEVICE and link libvpp_amd");
#endif
  }
  void update(const image_type& in) {  // pyramid.hh:194-198
#ifdef VPP_AMD_DEVICE
    bool direct = factor_ == 2.f && in.domain() == levels_[0].domain();
    for (auto& l : levels_) direct = direct && l.border() >= 2;   // the per-level kernels behind vpp_pyramid_build read 2 border pixels
    if (direct) {  // copy + mirror border + every level in one submission (one launch for u8 pyramids of 2-3 levels)
      std::vector<vpp_image_desc> d(levels_.size());
      for (size_t i = 0; i < levels_.size(); i++) d[i] = levels_[i].device_desc(true, true);
      const vpp_image_desc s = in.device_desc(false);
      device::check(vpp_pyramid_build(d.data(), int(d.size()), &s, device::stream()), "vpp_pyramid_build");
      device::call_done();   // queued, not drained: vpp/core/device.hh
      return;
    }
#endif
    copy(in, levels_[0]); propagate_level0();
  }

  float factor() const { return factor_; }
  int size() const { return int(levels_.size()); }
  void swap(pyramid& o) { levels_.swap(o.levels_); std::swap(factor_, o.factor_); }
  std::vector<image_type>& levels() { return levels_; }
  const std::vector<image_type>& levels() const { return levels_; }
 private:
  std::vector<image_type> levels_;
  float factor_;
};
template <class V> using pyramid2d = pyramid<V, 2>;
template <class V> using pyramid3d = pyramid<V, 3>;

}  // namespace vpp
