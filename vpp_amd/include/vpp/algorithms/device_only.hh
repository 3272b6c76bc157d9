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
} }
