// pixel_wise.hh — the expression engine: pixel_wise(ranges...)(options...) | kernel.
// Reference: vpp/core/pixel_wise.hh:41-50, vpp/core/pixel_wise.hpp:14-217 (ranges = images, boxes, relative_access),
// vpp/core/relative_accessor.hh:18-33; legacy box_nbh2d<V,R,C> (tests/box_nbh2d.cc:9-18, benchmarks/box_5x5_filter.cc:165-171).
//
// Three evaluation routes:
//  * an opaque callable (any lambda) is applied on the host, row by row, exactly like the reference
//    (OpenMP over rows unless _no_threads; the four traversal-order options are honoured);
//  * the same callable in a translation unit compiled by hipcc with -DVPP_AMD_DEVICE -DVPP_AMD_HIPCC is the body of a generic
//    gfx950 kernel (vpp/core/pixel_wise_device.hh) when it carries no state (a by-reference capture would hold host addresses;
//    `_device` vouches for a by-value one), the traversal options are the defaults and the ranges do not alias; `_host` keeps a
//    call on the host (a callable that uses host-only functions such as rand() must say so, since it cannot be compiled for the GPU);
//  * STACKS OF FRAMES (extension; the reference's pixel_wise is 2-d only, pixel_wise.hpp:146-165): when the ranges are image3d<V> (slices = frames) or
//    std::vector<image2d<V>>, the kernel is applied to every frame as the 2-d expression it would be on that frame — and a tagged functor evaluates
//    the WHOLE stack in one device launch (vpp_box_filter_batch / vpp_pixelwise_binary_batch: the form the 4K roofline numbers are measured on);
//  * a tagged functor from vpp::ops (add, sub, mul, min, max, absdiff, box_mean<R,C>) is, in a -DVPP_AMD_DEVICE build,
//    dispatched through the C ABI to the hand-written gfx950 kernels (vpp_pixelwise_binary, vpp_box_filter).  A failing
//    device call throws; it never silently re-runs on the host.
#pragma once
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include <vpp/core/image2d.hh>
#include <vpp/core/image3d.hh>
#include <vpp/core/relative_accessor.hh>
#include <vpp/core/pixel_wise_device.hh>

namespace vpp {

// ---- neighbourhood access ---------------------------------------------------------------------------------
// relative_access_kernel / relative_accessor(img, p): vpp/core/relative_accessor.hh
template <class I> struct relative_access_ {
  I img;  // the (shallow, shared) handle itself — the reference keeps a dangling reference here
  auto first_point_coordinates() const { return img.domain().p1(); }
  auto last_point_coordinates() const { return img.domain().p2(); }
};
template <class I> relative_access_<I> relative_access(I i) { return relative_access_<I>{i}; }

// legacy spelling: R x C neighbourhood of every pixel, usable as a pixel_wise range or built at a point
template <class V, int R, int C> struct box_nbh2d {
  image2d<V> img; V* const* line = nullptr; int col = 0;
  explicit box_nbh2d(const image2d<V>& i) : img(i) {}
  box_nbh2d(const image2d<V>& i, vint2 p) : img(i), line(&img[p[0]]), col(p[1]) {}
  auto first_point_coordinates() const { return img.domain().p1(); }
  auto last_point_coordinates() const { return img.domain().p2(); }
  V& operator()(int dr, int dc) const { return line[dr][col + dc]; }
  V& north() const { return (*this)(-1, 0); }
  V& south() const { return (*this)(1, 0); }
  V& east() const { return (*this)(0, 1); }
  V& west() const { return (*this)(0, -1); }
  template <class F> void for_all(F f) const {
    for (int dr = -(R / 2); dr <= R / 2; dr++)
      for (int dc = -(C / 2); dc <= C / 2; dc++) f((*this)(dr, dc));
  }
};

// ---- tagged functors (device-dispatchable kernels; also plain callables on the host) ----------------------------
namespace ops {
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_MIN = 3, OP_MAX = 4, OP_ABSDIFF = 5 };  // = vpp_binary_op
template <int OP> struct binary {
  template <class V> void operator()(V& a, const V& b, const V& c) const {
    if (OP == OP_ADD) a = b + c; else if (OP == OP_SUB) a = b - c; else if (OP == OP_MUL) a = V(b * c);
    else if (OP == OP_MIN) a = b < c ? b : c; else if (OP == OP_MAX) a = b > c ? b : c; else a = b > c ? V(b - c) : V(c - b);
  }
};
typedef binary<OP_ADD> add; typedef binary<OP_SUB> sub; typedef binary<OP_MUL> mul;
typedef binary<OP_MIN> min; typedef binary<OP_MAX> max; typedef binary<OP_ABSDIFF> absdiff;
// out = mean of the R x C neighbourhood, per component, in the promoted type, truncating division
// (benchmarks/box_5x5_filter2.cc:73-80, examples/box_filter.cc:23-32)
template <int R, int C> struct box_mean {
  template <class V, class NB> void operator()(V& out, NB nbh) const {
    plus_promotion<V> sum = zero<plus_promotion<V>>();
    for (int i = -(R / 2); i <= R / 2; i++)
      for (int j = -(C / 2); j <= C / 2; j++) sum = sum + vpp::cast<plus_promotion<V>>(nbh(i, j));
    out = vpp::cast<V>(sum / (R * C));
  }
};
struct block_maxima {};  // block_wise(vint2(b, b), img) | ops::block_maxima(): see block_wise.hh
}  // namespace ops

namespace pw {
template <class V> struct image_row { V* row; V& operator()(int c) const { return row[c]; } };
struct box_row { int r; vint2 operator()(int c) const { return vint2(r, c); } };
template <class V> struct nbh_row { V* const* line; relative_access_kernel<V> operator()(int c) const { return relative_access_kernel<V>{line, c}; } };
template <class V, int R, int C> struct box_nbh_row {
  image2d<V> img; V* const* line;
  box_nbh2d<V, R, C> operator()(int c) const { box_nbh2d<V, R, C> n(img); n.line = line; n.col = c; return n; }
};
template <class V> image_row<V> row_access(image2d<V>& img, int r) { return image_row<V>{img[r]}; }
template <class V> image_row<const V> row_access(const image2d<V>& img, int r) { return image_row<const V>{img[r]}; }
inline box_row row_access(const box2d&, int r) { return box_row{r}; }
template <class V> nbh_row<V> row_access(const relative_access_<image2d<V>>& ra, int r) { return nbh_row<V>{&ra.img[r]}; }
template <class V, int R, int C> box_nbh_row<V, R, C> row_access(const box_nbh2d<V, R, C>& n, int r) { return box_nbh_row<V, R, C>{n.img, &n.img[r]}; }

// make the host copy current (and the mirror stale where the kernel may write) once, before raw row pointers are taken
template <class V> void touch(image2d<V>& i) { i.host_write(); }
template <class V> void touch(const image2d<V>& i) { i.host_read(); }
inline void touch(const box2d&) {}
template <class I> void touch(const relative_access_<I>& r) { r.img.host_write(); }
template <class V, int R, int C> void touch(const box_nbh2d<V, R, C>& n) { n.img.host_write(); }

template <class F, class... A> inline void call_lvalues(F& f, A&&... a) { f(a...); }  // kernels may take `auto&` accessors
template <class F, class... ROWS> inline void process_row(bool right_to_left, F& f, int c0, int c1, ROWS... rows) {
  if (!right_to_left) for (int c = c0; c <= c1; c++) call_lvalues(f, rows(c)...);
  else for (int c = c1; c >= c0; c--) call_lvalues(f, rows(c)...);
}
#if defined(VPP_AMD_DEVICE) && defined(__HIPCC__)
// ranges the generic device kernel can address, their accessors (mirror pointers; the mirror becomes the newer copy: the kernel
// may write through any reference it is handed) and the storage they live in (for the aliasing test)
template <class T> struct device_range : std::false_type {};
template <class V> struct device_range<imageNd<V, 2>> : std::is_trivially_copyable<V> {};
template <> struct device_range<box2d> : std::true_type {};
template <class V> struct device_range<relative_access_<imageNd<V, 2>>> : std::is_trivially_copyable<V> {};
template <class V, int R, int C> struct device_range<box_nbh2d<V, R, C>> : std::is_trivially_copyable<V> {};
template <class V> pwdev::image_acc<V> device_accessor(const image2d<V>& i) { const vpp_image_desc d = i.device_desc(true); return pwdev::image_acc<V>{(V*)d.first_pixel, d.pitch}; }
inline pwdev::box_acc device_accessor(const box2d&) { return pwdev::box_acc{}; }
template <class V> pwdev::nbh_acc<V> device_accessor(const relative_access_<image2d<V>>& r) {
  const vpp_image_desc d = r.img.device_desc(true);
  pwdev::nbh_acc<V> a{(V*)d.first_pixel, d.pitch, nullptr, nullptr};
  r.img.device_allocation(&a.lo, &a.hi);
  return a;
}
template <class V, int R, int C> pwdev::boxnbh_acc<V, R, C> device_accessor(const box_nbh2d<V, R, C>& n) {
  const vpp_image_desc d = n.img.device_desc(true);
  pwdev::boxnbh_acc<V, R, C> a{(V*)d.first_pixel, d.pitch, nullptr, nullptr};
  n.img.device_allocation(&a.lo, &a.hi);
  return a;
}
template <class V> const void* storage_of(const image2d<V>& i) { return i.storage_id(); }
inline const void* storage_of(const box2d&) { return nullptr; }
template <class V> const void* storage_of(const relative_access_<image2d<V>>& r) { return r.img.storage_id(); }
template <class V, int R, int C> const void* storage_of(const box_nbh2d<V, R, C>& n) { return n.img.storage_id(); }
#endif
// stacks of frames: image3d<V> (frame k = slice k, vpp/core/imageNd.hh: slice) and std::vector<image2d<V>>; relative_access / of a stack = of every frame
template <class T> struct stack : std::false_type {};
template <class V> struct stack<imageNd<V, 3>> : std::true_type {
  static int size(const imageNd<V, 3>& s) { return s.nslices(); }
  static imageNd<V, 2> frame(const imageNd<V, 3>& s, int k) { return s.slice(k); }
};
template <class V> struct stack<std::vector<imageNd<V, 2>>> : std::true_type {
  static int size(const std::vector<imageNd<V, 2>>& s) { return (int)s.size(); }
  static imageNd<V, 2> frame(const std::vector<imageNd<V, 2>>& s, int k) { return s[k]; }
};
template <class I> struct stack<relative_access_<I>> : stack<I> {
  static int size(const relative_access_<I>& r) { return stack<I>::size(r.img); }
  static auto frame(const relative_access_<I>& r, int k) { return relative_access(stack<I>::frame(r.img, k)); }
};
template <class T> struct is_image2d : std::false_type {};
template <class V> struct is_image2d<imageNd<V, 2>> : std::true_type { typedef V value_type; };
}  // namespace pw

template <class OPTS, class... R> class pixel_wise_impl {
 public:
  pixel_wise_impl(std::tuple<R...> t, OPTS o) : ranges_(t), options_(o) {}
  template <class... A> auto operator()(A... o) const { auto n = opt::make(o...); return pixel_wise_impl<decltype(n), R...>(ranges_, n); }
  template <class... B> auto operator()(opt::set<B...> n) const { return pixel_wise_impl<opt::set<B...>, R...>(ranges_, n); }

  // (the ranges enter through a type that depends on F: for stacks of frames there is no row_access, and the expression must not be looked at before a kernel is applied)
  template <class F, class X> struct later { typedef X type; };
  template <class F> using kernel_return_type = decltype(std::declval<F&>()(std::declval<decltype(pw::row_access(std::declval<typename later<F, R>::type&>(), 0)(0))&>()...));

  static constexpr bool stacked = (pw::stack<R>::value || ...);   // ranges are stacks of frames: the expression is the 2-d one on every frame

  // opaque callable (pixel_wise.hpp:146-165,188-213): host evaluation, or — single-source hipcc build — the generic device kernel
  template <class F> auto operator|(F fun) {
    if constexpr (stacked) { for_each_frame(fun, std::index_sequence_for<R...>()); return; }
    else return apply(fun);
  }
  template <class F> auto apply(F fun) {
#if defined(VPP_AMD_DEVICE) && defined(__HIPCC__) && defined(VPP_AMD_HIPCC)
    if constexpr (device_eligible<F>()) {
      if (!ranges_alias(std::index_sequence_for<R...>())) return eval_device(fun, std::is_void<kernel_return_type<F>>());
    }
#endif
    return eval(fun, std::is_void<kernel_return_type<F>>());
  }

#ifdef VPP_AMD_DEVICE
  // tagged functors: gfx950 kernels through the C ABI (a stack of frames: ONE launch for all of them)
  template <int OP> void operator|(ops::binary<OP>) {
    if constexpr (stacked) device_binary_stack(OP); else device_binary(OP, std::index_sequence_for<R...>());
  }
  template <int RR, int CC> void operator|(ops::box_mean<RR, CC>) {
    if constexpr (stacked) device_box_stack(RR, CC); else device_box(RR, CC);
  }
#endif

#if defined(VPP_AMD_DEVICE) && defined(__HIPCC__) && defined(VPP_AMD_HIPCC)
  template <class F> static constexpr bool device_eligible() {
    return std::is_trivially_copyable<F>::value && (std::is_empty<F>::value || OPTS::has(_device)) && !OPTS::has(_host) && !OPTS::has(_no_threads) &&
           !OPTS::has(_right_to_left) && !OPTS::has(_bottom_to_top) && !OPTS::has(_left_to_right) && !OPTS::has(_top_to_bottom) &&
           !OPTS::has(_mem_forward) && !OPTS::has(_mem_backward) && (pw::device_range<R>::value && ...);
  }
  template <std::size_t... I> bool ranges_alias(std::index_sequence<I...>) const {
    const void* st[] = {pw::storage_of(std::get<I>(ranges_))...};
    for (std::size_t a = 0; a < sizeof...(I); a++)
      for (std::size_t b = a + 1; b < sizeof...(I); b++)
        if (st[a] && st[a] == st[b]) return true;
    return false;
  }
  template <class F, std::size_t... I> void run_device(F& fun, std::index_sequence<I...>) {
    const auto p1 = std::get<0>(ranges_).first_point_coordinates();
    const auto p2 = std::get<0>(ranges_).last_point_coordinates();
    pwdev::launch<OPTS::has(_nbh_read_only)>(fun, p1[0], p1[1], p2[0] - p1[0] + 1, p2[1] - p1[1] + 1, pw::device_accessor(std::get<I>(ranges_))...);
  }
  template <class F> void eval_device(F& fun, std::true_type) { run_device(fun, std::index_sequence_for<R...>()); }
  template <class F> auto eval_device(F& fun, std::false_type) {
    typedef typename std::decay<kernel_return_type<F>>::type value_type;
    const auto p1 = std::get<0>(ranges_).first_point_coordinates();
    const auto p2 = std::get<0>(ranges_).last_point_coordinates();
    image2d<value_type> out(box2d(p1, p2));
    auto all = std::tuple_cat(std::make_tuple(out), ranges_);
    auto wrapper = [fun](value_type& o, auto&... ps) mutable { o = fun(ps...); };
    pixel_wise_impl<OPTS, image2d<value_type>, R...> sub(all, options_);
    sub.run_device(wrapper, std::make_index_sequence<sizeof...(R) + 1>());
    return out;
  }
#endif

 private:
  template <class F> void eval(F& fun, std::true_type) { run(fun, std::index_sequence_for<R...>()); }
  template <class F> auto eval(F& fun, std::false_type) {  // the kernel returns a pixel value: build an image (pixel_wise.hpp:196-211)
    typedef typename std::decay<kernel_return_type<F>>::type value_type;
    const auto p1 = std::get<0>(ranges_).first_point_coordinates();
    const auto p2 = std::get<0>(ranges_).last_point_coordinates();
    image2d<value_type> out(box2d(p1, p2));
    auto all = std::tuple_cat(std::make_tuple(out), ranges_);
    auto wrapper = [&fun](value_type& o, auto&... ps) { o = fun(ps...); };
    pixel_wise_impl<OPTS, image2d<value_type>, R...> sub(all, options_);
    sub.run(wrapper, std::make_index_sequence<sizeof...(R) + 1>());
    return out;
  }

 public:
  template <class F, std::size_t... I> void run(F& fun, std::index_sequence<I...>) {
    const auto p1 = std::get<0>(ranges_).first_point_coordinates();
    const auto p2 = std::get<0>(ranges_).last_point_coordinates();
    const int r0 = p1[0], r1 = p2[0], c0 = p1[1], c1 = p2[1];
    const bool rtl = OPTS::has(_right_to_left), btt = OPTS::has(_bottom_to_top);
    (void)std::initializer_list<int>{(pw::touch(std::get<I>(ranges_)), 0)...};
    if (OPTS::has(_no_threads)) {
      if (!btt) for (int r = r0; r <= r1; r++) pw::process_row(rtl, fun, c0, c1, pw::row_access(std::get<I>(ranges_), r)...);
      else for (int r = r1; r >= r0; r--) pw::process_row(rtl, fun, c0, c1, pw::row_access(std::get<I>(ranges_), r)...);
    } else {
      if (!btt) {
#pragma omp parallel for
        for (int r = r0; r <= r1; r++) pw::process_row(rtl, fun, c0, c1, pw::row_access(std::get<I>(ranges_), r)...);
      } else {
#pragma omp parallel for
        for (int r = r1; r >= r0; r--) pw::process_row(rtl, fun, c0, c1, pw::row_access(std::get<I>(ranges_), r)...);
      }
    }
  }

 private:
#ifdef VPP_AMD_DEVICE
  template <std::size_t... I> void device_binary(int op, std::index_sequence<I...>) {
    static_assert(sizeof...(R) == 3, "ops::binary needs pixel_wise(dst, a, b)");
    auto& d = std::get<0>(ranges_); auto& a = std::get<1>(ranges_); auto& b = std::get<2>(ranges_);
    const vpp_image_desc da = a.device_desc(false), db = b.device_desc(false), dd = d.device_desc(true);
    if (OPTS::has(_immediate)) { device::check(vpp_pixelwise_binary(op, &dd, &da, &db, device::stream()), "vpp_pixelwise_binary"); device::call_done(); return; }
    device::check(vpp_pixelwise_binary_deferred(op, &dd, &da, &db, device::stream()), "vpp_pixelwise_binary");
    device::deferred_call_done();   // held back and launched in batches by the library, queued not drained: vpp/core/device.hh
  }
  void device_box(int rr, int cc) {
    static_assert(sizeof...(R) == 2, "ops::box_mean needs pixel_wise(dst, relative_access(src)) or pixel_wise(dst, box_nbh2d(src))");
    auto& d = std::get<0>(ranges_); auto& n = std::get<1>(ranges_);
    const vpp_image_desc ds = n.img.device_desc(false), dd = d.device_desc(true);
    if (OPTS::has(_immediate)) { device::check(vpp_box_filter(&dd, &ds, rr, cc, device::stream()), "vpp_box_filter"); device::call_done(); return; }
    device::check(vpp_box_filter_deferred(&dd, &ds, rr, cc, device::stream()), "vpp_box_filter");
    device::deferred_call_done();   // held back and launched in batches by the library, queued not drained: vpp/core/device.hh
  }
#endif
  // a stack: the same expression on every frame (a kernel that returns a pixel value would have to build a stack: not provided)
  template <class F, std::size_t... I> void for_each_frame(F& fun, std::index_sequence<I...>) {
    static_assert((pw::stack<R>::value && ...), "pixel_wise over stacks of frames: every range must be an image3d, a std::vector<image2d> or relative_access of one");
    const int n = pw::stack<typename std::tuple_element<0, std::tuple<R...>>::type>::size(std::get<0>(ranges_));
    for (int k = 0; k < n; k++) {
      auto frames = std::make_tuple(pw::stack<R>::frame(std::get<I>(ranges_), k)...);
      pixel_wise_impl<OPTS, typename std::decay<decltype(pw::stack<R>::frame(std::get<I>(ranges_), k))>::type...> sub(frames, options_);
      static_assert(std::is_void<decltype(sub | fun)>::value, "pixel_wise over stacks of frames: the kernel must return void");
      sub | fun;
    }
  }
#ifdef VPP_AMD_DEVICE
  template <class S> static int stack_size(const S& s) { return pw::stack<S>::size(s); }
  void device_binary_stack(int op) {
    static_assert(sizeof...(R) == 3 && (pw::stack<R>::value && ...), "ops::binary on stacks needs pixel_wise(dst, a, b) with three stacks of frames");
    auto& d = std::get<0>(ranges_); auto& a = std::get<1>(ranges_); auto& b = std::get<2>(ranges_);
    const int n = stack_size(d);
    if (stack_size(a) != n || stack_size(b) != n) throw std::runtime_error("pixel_wise: stacks of different sizes");
    std::vector<vpp_image_desc> dd(n), da(n), db(n);
    for (int k = 0; k < n; k++) {   // sources first: a destination frame that shares its storage with a source must not mark the mirror newer before the upload
      da[k] = pw::stack<typename std::decay<decltype(a)>::type>::frame(a, k).device_desc(false);
      db[k] = pw::stack<typename std::decay<decltype(b)>::type>::frame(b, k).device_desc(false);
    }
    for (int k = 0; k < n; k++) dd[k] = pw::stack<typename std::decay<decltype(d)>::type>::frame(d, k).device_desc(true);
    device::check(vpp_pixelwise_binary_batch(op, dd.data(), da.data(), db.data(), n, device::stream()), "vpp_pixelwise_binary_batch");
    device::call_done();   // queued, not drained: vpp/core/device.hh
  }
  void device_box_stack(int rr, int cc) {
    static_assert(sizeof...(R) == 2 && (pw::stack<R>::value && ...), "ops::box_mean on stacks needs pixel_wise(dst, relative_access(src)) with two stacks of frames");
    auto& d = std::get<0>(ranges_); auto& nb = std::get<1>(ranges_);
    const int n = stack_size(d);
    if (stack_size(nb) != n) throw std::runtime_error("pixel_wise: stacks of different sizes");
    std::vector<vpp_image_desc> dd(n), ds(n);
    for (int k = 0; k < n; k++) ds[k] = pw::stack<typename std::decay<decltype(nb)>::type>::frame(nb, k).img.device_desc(false);
    for (int k = 0; k < n; k++) dd[k] = pw::stack<typename std::decay<decltype(d)>::type>::frame(d, k).device_desc(true);
    device::check(vpp_box_filter_batch(dd.data(), ds.data(), n, rr, cc, device::stream()), "vpp_box_filter_batch");
    device::call_done();   // queued, not drained: vpp/core/device.hh
  }
#endif
  std::tuple<R...> ranges_;
  OPTS options_;
  template <class O2, class... R2> friend class pixel_wise_impl;
};

struct pixel_wise_caller {
  template <class... T> auto operator()(T&&... t) const {
    return pixel_wise_impl<opt::set<>, typename std::decay<T>::type...>(std::tuple<typename std::decay<T>::type...>(t...), opt::set<>());
  }
};
static const pixel_wise_caller pixel_wise;

}  // namespace vpp
