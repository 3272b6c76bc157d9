// pyramid.hh — image pyramids (reference: vpp/core/pyramid.hh:126-221).  Level sizes, the border option and the
// update / propagate_level0 protocol are the reference's; the per-level work (5-tap binomial low-pass + subsample2 +
// mirror border, pyramid.hh:12-81) is the gfx950 kernel behind vpp_pyr_down.  Factor 2 only (what the hot path uses).
#pragma once
#include <vector>
#include <vpp/core/copy.hh>
#include <vpp/core/fill.hh>

namespace vpp {

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
      const vpp_image_desc prev = levels_[i - 1].device_desc(false), next = levels_[i].device_desc(true);
      device::check(vpp_pyr_down(&next, &prev, device::stream()), "vpp_pyr_down");
    }
    device::check(vpp_sync(device::stream()), "vpp_sync");
#else
    static_assert(sizeof(V) == 0, "pyramid::propagate_level0 runs on the device: build with -DVPP_AMD_DEVICE and link libvpp_amd");
#endif
  }
  void update(const image_type& in) { copy(in, levels_[0]); propagate_level0(); }

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
