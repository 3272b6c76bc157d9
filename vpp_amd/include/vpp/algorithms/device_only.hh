// The algorithm front-ends exist only as device dispatchers: there is no host implementation to fall back to.
#pragma once
#include <vpp/core/device.hh>
#ifndef VPP_AMD_DEVICE
#error "vpp/algorithms/* run on the MI355X engine: compile with -DVPP_AMD_DEVICE -I<repo>/include and link libvpp_amd.so"
#endif
#include <vector>
namespace vpp { namespace device {
// RAII scratch in HBM for keypoint lists and results
struct dbuf {
  void* p = nullptr; size_t bytes = 0;
  explicit dbuf(size_t n) : bytes(n) { check(vpp_malloc(n ? n : 1, &p), "vpp_malloc"); }
  ~dbuf() { if (p) vpp_free(p); }
  dbuf(const dbuf&) = delete; dbuf& operator=(const dbuf&) = delete;
  void upload(const void* h, size_t n) { if (n) check(vpp_memcpy_h2d(p, h, n, stream()), "vpp_memcpy_h2d"); }
  void download(void* h, size_t n) const { if (n) check(vpp_memcpy_d2h(h, p, n, stream()), "vpp_memcpy_d2h"); }
};
// pinned host staging of n elements of T (keypoint lists / results moved to and from HBM every frame)
template <class T> struct hbuf {
  T* p = nullptr; size_t n = 0;
  explicit hbuf(size_t count) : n(count) { void* v = nullptr; check(vpp_malloc_host((count ? count : 1) * sizeof(T), &v), "vpp_malloc_host"); p = (T*)v; }
  ~hbuf() { if (p) vpp_free_host(p); }
  hbuf(const hbuf&) = delete; hbuf& operator=(const hbuf&) = delete;
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p; }
  size_t bytes() const { return n * sizeof(T); }
};
} }
