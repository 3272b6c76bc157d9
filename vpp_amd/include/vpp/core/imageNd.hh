// imageNd.hh — N-d pitched image with border, alignment and shared ownership: the container of the vpp API.
// Reference: vpp/core/imageNd.hh:43-177, vpp/core/imageNd.hpp:99-341 (layout arithmetic :151-196 is reproduced exactly:
// pitches and first-pixel offsets are part of the contract), README.md:85-116.
// New: every buffer can carry an HBM mirror (vpp/core/device.hh); host accessors keep the two coherent.
#pragma once
#include <cassert>
#include <cstdlib>
#include <initializer_list>
#include <memory>
#include <new>
#include <vector>
#include <sys/mman.h>

#include <vpp/core/boxNd.hh>
#include <vpp/core/device.hh>
#include <vpp/core/symbols.hh>

#ifndef VPP_DEFAULT_IMAGE_ALIGNMENT
#ifdef __AVX2__
#define VPP_DEFAULT_IMAGE_ALIGNMENT 32
#else
#define VPP_DEFAULT_IMAGE_ALIGNMENT 16
#endif
#endif

namespace vpp {

template <class V, unsigned N> struct imageNd_data {
  V* data_ = nullptr;        // aligned start of the buffer
  V* data_end_ = nullptr;
  V* begin_ = nullptr;       // pixel (0,..,0)
  std::vector<V*> rows_;     // N == 2: one pointer per row of domain_with_border
  V** rows_array_start_ = nullptr;
  std::shared_ptr<void> data_sptr_;                 // owner of the host bytes (malloc'ed, external holder, or nothing)
  std::shared_ptr<device::storage> store_;          // host range + HBM mirror, shared by sub-images
  boxNd<N> domain_, buffer_domain_;
  int border_ = 0, pitch_ = 0, alignment_ = 0;
};

template <class V, unsigned N> class imageNd {
 public:
  typedef imageNd<V, N> self;
  typedef V value_type;
  typedef vint<N> coord_type;
  typedef boxNd<N> domain_type;
  enum { dimension = N };

  imageNd() {}
  template <class... O> imageNd(const std::initializer_list<int>& dims, const O&... o) { init(std::vector<int>(dims), opt::make(o...)); }
  template <class... O> imageNd(const std::vector<int>& dims, const O&... o) { init(dims, opt::make(o...)); }
  template <class... O> imageNd(int ncols, const O&... o) { static_assert(N == 1, "imageNd constructor: bad dimension."); init({ncols}, opt::make(o...)); }
  template <class... O> imageNd(int nrows, int ncols, const O&... o) { static_assert(N == 2, "imageNd constructor: bad dimension."); init({nrows, ncols}, opt::make(o...)); }
  template <class... O> imageNd(int nslices, int nrows, int ncols, const O&... o) { static_assert(N == 3, "imageNd constructor: bad dimension."); init({nslices, nrows, ncols}, opt::make(o...)); }
  template <class... O> imageNd(const boxNd<N>& d, const O&... o) {
    std::vector<int> dims(N);
    for (unsigned i = 0; i < N; i++) dims[i] = d.size(i);
    init(dims, opt::make(o...));
  }
  // copies share the pixels (imageNd.hpp:77-87)
  imageNd(const imageNd&) = default;
  imageNd(imageNd&&) = default;
  imageNd& operator=(const imageNd&) = default;
  imageNd& operator=(imageNd&&) = default;

  const int& nslices() const { static_assert(N >= 3, "nslices require dimension >= 3."); return domain().size(N - 3); }
  const int& nrows() const { static_assert(N >= 2, "nrows require dimension >= 2."); return domain().size(N - 2); }
  const int& ncols() const { return domain().size(N - 1); }

  V& operator()(const vint<N>& p) { host_write(); return *address_of_(p); }
  const V& operator()(const vint<N>& p) const { host_read(); return *address_of_(p); }
  template <class... T> V& operator()(int c, T... cs) { static_assert(1 + sizeof...(cs) == N, "Wrong dimension of coordinates passed to imageNd::operator()."); return (*this)(coord_type(c, cs...)); }
  template <class... T> const V& operator()(int c, T... cs) const { static_assert(1 + sizeof...(cs) == N, "Wrong dimension of coordinates passed to imageNd::operator()."); return (*this)(coord_type(c, cs...)); }
  V*& operator[](int r) { host_write(); return ptr_->rows_array_start_[r]; }
  V* const& operator[](int r) const { host_read(); return ptr_->rows_array_start_[r]; }

  // bilinear interpolation, result converted back to V (imageNd.hpp:280-300)
  V linear_interpolate(const vfloat<N>& p) const {
    static_assert(N == 2, "linear_interpolate only supports 2d images.");
    host_read();
    const vint2 x = p.template cast<int>();
    const float a0 = p[0] - x[0], a1 = p[1] - x[1];
    const V* l1 = address_of_(x);
    const V* l2 = (const V*)((const char*)l1 + ptr_->pitch_);
    typedef cast_to_float<V> S;
    return vpp::cast<V>((1 - a0) * (1 - a1) * vpp::cast<S>(l1[0]) + a0 * (1 - a1) * vpp::cast<S>(l2[0]) + (1 - a0) * a1 * vpp::cast<S>(l1[1]) + a0 * a1 * vpp::cast<S>(l2[1]));
  }

  V* address_of(const vint<N>& p) { host_write(); return address_of_(p); }
  const V* address_of(const vint<N>& p) const { host_read(); return address_of_(p); }
  int offset_of(const vint<N>& p) const { return coords_to_offset(p); }
  int coords_to_offset(const vint<N>& p) const {  // byte offset of p from pixel 0 (imageNd.hpp:224-234)
    int row_idx = N >= 2 ? p[N - 2] : 0, ds = 1;
    for (int i = int(N) - 3; i >= 0; i--) { ds *= ptr_->buffer_domain_.size(i + 1) + 2 * ptr_->border_; row_idx += ds * p[i]; }
    return row_idx * ptr_->pitch_ + p[N - 1] * int(sizeof(V));
  }

  bool has(const coord_type& p) const { return ptr_->domain_.has(p); }
  bool has(const V* p) const { return p >= ptr_->data_ && p < ptr_->data_end_; }
  bool has_data() const { return !!ptr_; }
  V* data() { host_write(); return ptr_->data_; }
  const V* data() const { host_read(); return ptr_->data_; }
  const V* data_end() const { return ptr_->data_end_; }
  int pitch() const { return ptr_->pitch_; }
  int border() const { return ptr_->border_; }
  int alignment() const { return ptr_->alignment_; }
  template <class U> imageNd<U, N>& cast() { return *(imageNd<U, N>*)this; }
  template <class U> const imageNd<U, N>& cast() const { return *(const imageNd<U, N>*)this; }

  const boxNd<N>& domain() const { return ptr_->domain_; }
  boxNd<N> domain_with_border() const { return ptr_->domain_ + vpp::border(ptr_->border_); }
  const vint<N>& first_point_coordinates() const { return ptr_->domain_.p1(); }
  const vint<N>& last_point_coordinates() const { return ptr_->domain_.p2(); }

  // iteration over the domain pixels in raster order (vpp/core/imageNd_iterator.hh)
  struct iterator {
    boxNd_iterator<N> it; self* img;
    V& operator*() { return *img->address_of_(*it); }
    iterator& operator++() { ++it; return *this; }
    bool operator!=(const iterator& o) const { return it != o.it; }
    bool operator==(const iterator& o) const { return it == o.it; }
  };
  iterator begin() { host_write(); return iterator{domain().begin(), this}; }
  iterator end() { return iterator{domain().end(), this}; }

  // sub-image sharing the buffer (imageNd.hpp:325-341); coordinates restart at 0
  self subimage(const boxNd<N>& d) const {
    self res;
    res.ptr_ = std::make_shared<imageNd_data<V, N>>(*ptr_);
    res.ptr_->begin_ = const_cast<V*>(address_of_(d.p1()));
    res.ptr_->domain_ = boxNd<N>(d.p1() - d.p1(), d.p2() - d.p1());
    res.index_rows();
    return res;
  }
  const self const_subimage(const boxNd<N>& d) const { return subimage(d); }

  // Frame k of a stack of frames (image3d: slices x rows x cols) as an image2d over the same pixels: the slice's rows, its border rows included (with
  // border b every slice carries its own b rows above and below, imageNd.hpp:151-196), share the buffer, its owner and its HBM mirror.  This is what
  // lets `pixel_wise` treat an image3d as a batch of frames (vpp/core/pixel_wise.hh) — one device launch for the whole stack.
  imageNd<V, 2> slice(int k) const {
    static_assert(N == 3, "slice(k): the frames of an image3d");
    imageNd<V, 2> res;
    res.ptr_ = std::make_shared<imageNd_data<V, 2>>();
    auto& d = *res.ptr_;
    const auto& s = *ptr_;
    vint<N> p = vint<N>::Zero(); p[0] = k;
    d.data_ = s.data_; d.data_end_ = s.data_end_; d.begin_ = const_cast<V*>(address_of_(p));
    d.data_sptr_ = s.data_sptr_; d.store_ = s.store_;
    d.domain_ = boxNd<2>(vint2(0, 0), vint2(s.domain_.size(1) - 1, s.domain_.size(2) - 1));
    d.buffer_domain_ = d.domain_;
    d.border_ = s.border_; d.pitch_ = s.pitch_; d.alignment_ = s.alignment_;
    res.index_rows();
    return res;
  }

  const void* storage_id() const { return ptr_ ? (const void*)ptr_->store_.get() : nullptr; }  // identity of the pixel buffer (shared by sub-images)
  void set_external_data_holder(void* data, void (*deleter)(void*)) { ptr_->data_sptr_ = std::shared_ptr<void>(data, deleter); }
  void swap(imageNd& o) { o.ptr_.swap(ptr_); }

  // ---- coherence with the HBM mirror -----------------------------------------------------------------
  void host_read() const {
#ifdef VPP_AMD_DEVICE
    if (ptr_ && __builtin_expect(ptr_->store_->state == 2, 0)) ptr_->store_->to_host(false);
#endif
  }
  void host_write() const {
#ifdef VPP_AMD_DEVICE
    if (ptr_ && __builtin_expect(ptr_->store_->state != 0, 0)) ptr_->store_->to_host(true);
#endif
  }
#ifdef VPP_AMD_DEVICE
  // true when the HBM mirror exists and is as new as the host pixels (device_desc would not upload)
  bool device_current() const { return ptr_ && ptr_->store_->dev && ptr_->store_->state != 0; }
  // C-ABI descriptor of the HOST pixels (for entry points that take host frames and stage them themselves); a newer mirror is downloaded first
  vpp_image_desc host_desc() const {
    static_assert(N == 2, "device evaluation handles image2d");
    typedef pixel_traits<V> PT;
    host_read();
    vpp_image_desc d;
    d.first_pixel = (void*)ptr_->begin_;
    d.nrows = nrows(); d.ncols = ncols(); d.pitch = pitch(); d.border = border();
    d.dtype = device::dtype_of<typename PT::component>::value; d.channels = PT::channels;
    return d;
  }
  // C-ABI descriptor of this image in HBM (vpp_image_desc); the mirror is uploaded if stale.  will_write marks the
  // mirror as the newer copy: the next host access downloads it; discard (with will_write) skips the upload of a stale
  // mirror when the callee overwrites the whole image, border included.
  vpp_image_desc device_desc(bool will_write, bool discard = false) const {
    static_assert(N == 2, "device evaluation handles image2d");
    typedef pixel_traits<V> PT;
    vpp_image_desc d;
    d.first_pixel = ptr_->store_->to_device(ptr_->begin_, will_write, discard);
    d.nrows = nrows(); d.ncols = ncols(); d.pitch = pitch(); d.border = border();
    d.dtype = device::dtype_of<typename PT::component>::value; d.channels = PT::channels;
    return d;
  }
  // [first, one past the last) byte of the HBM mirror of the WHOLE buffer this image (or view) lives in — what device code may read around a view
  // (valid after a device_desc call: the mirror exists)
  void device_allocation(const char** lo, const char** hi) const { *lo = (const char*)ptr_->store_->dev; *hi = *lo + ptr_->store_->bytes; }
#endif

 protected:
  V* address_of_(const vint<N>& p) const { return (V*)((char*)ptr_->begin_ + coords_to_offset(p)); }

  template <class OPTS> void init(const std::vector<int>& dims, const OPTS& options) {
    static_assert(!OPTS::has(_data) || OPTS::has(_pitch), "You must provide the pitch (number of bytes between the beginning of two successive lines) when providing a data pointer to the image constructor.");
    assert(dims.size() == N);
    ptr_ = std::make_shared<imageNd_data<V, N>>();
    auto& d = *ptr_;
    vint<N> p2;
    for (unsigned i = 0; i < N; i++) p2[i] = dims[i] - 1;
    d.domain_ = boxNd<N>(vint<N>::Zero(), p2);
    d.buffer_domain_ = d.domain_;
    d.border_ = options.get(_border, 0);
    d.store_ = std::make_shared<device::storage>();
    size_t rows = 1;
    for (unsigned i = 0; i + 1 < N; i++) rows *= size_t(dims[i] + 2 * d.border_);
    if (OPTS::has(_data)) {  // borrowed pixels: never freed here (imageNd.hpp:112-136, README.md:102-116)
      d.data_ = (V*)options.get(_data, (V*)0);
      d.begin_ = d.data_;
      d.pitch_ = options.get(_pitch, 0);
      unsigned data_al = 1, pitch_al = 1;
      while ((reinterpret_cast<unsigned long>(d.data_) % (data_al * 2)) == 0 && data_al < (1u << 20)) data_al *= 2;
      while (((unsigned long)d.pitch_ % (pitch_al * 2)) == 0 && pitch_al < (1u << 20)) pitch_al *= 2;
      d.alignment_ = int(data_al < pitch_al ? data_al : pitch_al);
      size_t size = d.pitch_;
      for (unsigned n = 0; n + 1 < N; n++) size *= d.domain_.size(n);
      d.data_end_ = (V*)((char*)d.data_ + size);
      // the caller vouches for `border` pixels around the domain: that is the range the mirror covers
      vint<N> o = vint<N>::Ones() * (-d.border_);
      d.store_->host = (char*)d.begin_ + coords_to_offset(o);
      d.store_->bytes = rows * size_t(d.pitch_) - (size_t(d.pitch_) - size_t(dims[N - 1] + 2 * d.border_) * sizeof(V));  // up to the last bordered pixel
    } else {
      const int align_size = options.get(_aligned, VPP_DEFAULT_IMAGE_ALIGNMENT);
      assert(align_size != 0);
      d.alignment_ = align_size;
      int border_size = d.border_ * int(sizeof(V)), border_padding = 0;
      if (border_size % align_size) { border_padding = align_size - (border_size % align_size); border_size += border_padding; }
      d.pitch_ = dims[N - 1] * int(sizeof(V)) + border_size * 2;
      if (d.pitch_ % align_size) d.pitch_ += align_size - (d.pitch_ % align_size);
      const size_t size = rows * size_t(d.pitch_);
      // zero-filled: bytes the algorithms read before writing are 0.  Large buffers are mapped directly: glibc raises its mmap
      // threshold after the first large free, from then on calloc() memsets recycled heap (0.8 ms for a 4K uchar image) even for
      // an image that only ever lives in HBM; fresh anonymous pages are zero and cost nothing until the host touches them.
      char* raw = nullptr;
      const size_t total = size + size_t(align_size);
      if (total >= (size_t(1) << 20)) {
        void* m = ::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        raw = (char*)m;
        d.data_sptr_ = std::shared_ptr<void>(raw, [total](void* p) { ::munmap(p, total); });
      } else {
        raw = (char*)std::calloc(1, total);
        if (!raw) throw std::bad_alloc();
        d.data_sptr_ = std::shared_ptr<void>(raw, [](void* p) { std::free(p); });
      }
      const unsigned long mis = reinterpret_cast<unsigned long>(raw) % (unsigned long)align_size;
      d.data_ = (V*)(mis ? raw + (align_size - mis) : raw);
      d.data_end_ = (V*)((char*)d.data_ + size);
      vint<N> first = vint<N>::Ones() * d.border_;
      d.begin_ = (V*)((char*)d.data_ + border_padding + coords_to_offset(first));
      d.store_->host = (char*)d.data_;
      d.store_->bytes = size;
    }
    index_rows();
  }

  void index_rows() {
    if (N != 2) return;
    auto& d = *ptr_;
    d.rows_.clear();
    const int b = d.border_;
    for (int r = -b; r < d.domain_.size(0) + b; r++) { vint<N> p = vint<N>::Zero(); p[0] = r; d.rows_.push_back(address_of_(p)); }
    d.rows_array_start_ = &d.rows_[b];
  }

  std::shared_ptr<imageNd_data<V, N>> ptr_;
  template <class U, unsigned M> friend class imageNd;
};

template <class V, unsigned N> imageNd<V, N> operator|(imageNd<V, N>& img, const boxNd<N>& b) { return img.subimage(b); }
template <class V, unsigned N> const imageNd<V, N> operator|(const imageNd<V, N>& img, const boxNd<N>& b) { return img.const_subimage(b); }

}  // namespace vpp
