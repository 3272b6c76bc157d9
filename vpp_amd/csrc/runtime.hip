// runtime.hip — device/runtime entry points of the C ABI (include/vpp_amd.h "runtime" block).
#include "common.hpp"
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace vpp_amd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
static std::mutex g_tune_mu;
static std::map<std::string, int> g_tune;
int tuning(const char* name, int dflt) {
  std::lock_guard<std::mutex> l(g_tune_mu);
  auto it = g_tune.find(name);
  return it == g_tune.end() ? dflt : it->second;
}
}  // namespace vpp_amd
using namespace vpp_amd;

extern "C" {

const char* vpp_last_error(void) { return g_err; }
const char* vpp_version(void) { return "vpp_amd 0.1 (gfx950)"; }

int vpp_set_tuning(const char* name, int value) {
  VPP_REQUIRE(name, VPP_ERR_INVALID_ARG, "vpp_set_tuning: null name");
  std::lock_guard<std::mutex> l(g_tune_mu);
  if (value < 0) g_tune.erase(name); else g_tune[name] = value;  // negative = back to the built-in default
  return VPP_OK;
}

int vpp_device_count(int* n) {
  VPP_REQUIRE(n, VPP_ERR_INVALID_ARG, "vpp_device_count: null");
  VPP_HIP_TRY(hipGetDeviceCount(n));
  return VPP_OK;
}
int vpp_init(int device) {
  VPP_HIP_TRY(hipSetDevice(device));
  VPP_HIP_TRY(hipFree(nullptr));
  return VPP_OK;
}
int vpp_malloc(size_t bytes, void** dptr) {
  VPP_REQUIRE(dptr, VPP_ERR_INVALID_ARG, "vpp_malloc: null out pointer");
  VPP_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return VPP_OK;
}
int vpp_free(void* dptr) { VPP_HIP_TRY(hipFree(dptr)); return VPP_OK; }
int vpp_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  VPP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  VPP_HIP_TRY(hipStreamSynchronize(as_stream(stream)));  // src is pageable host memory the caller may reuse
  return VPP_OK;
}
int vpp_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  VPP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  VPP_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return VPP_OK;
}
int vpp_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  VPP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return VPP_OK;
}
int vpp_memset(void* dst, int byte, size_t bytes, void* stream) {
  VPP_HIP_TRY(hipMemsetAsync(dst, byte, bytes, as_stream(stream)));
  return VPP_OK;
}
int vpp_sync(void* stream) { VPP_HIP_TRY(hipStreamSynchronize(as_stream(stream))); return VPP_OK; }

int vpp_image_layout(int nrows, int ncols, int elem_bytes, int border, int align, int32_t* pitch, size_t* alloc_bytes,
                     size_t* first_pixel_offset) {
  // imageNd::allocate, vpp/core/imageNd.hpp:151-196
  VPP_REQUIRE(nrows > 0 && ncols > 0 && elem_bytes > 0 && border >= 0 && align > 0, VPP_ERR_INVALID_ARG, "vpp_image_layout: bad argument");
  int border_size = border * elem_bytes, border_padding = 0;
  if (border_size % align) { border_padding = align - (border_size % align); border_size += border_padding; }
  int p = ncols * elem_bytes + border_size * 2;
  if (p % align) p += align - (p % align);
  if (pitch) *pitch = p;
  if (alloc_bytes) *alloc_bytes = (size_t)(nrows + 2 * border) * p;
  if (first_pixel_offset) *first_pixel_offset = (size_t)border_padding + (size_t)border * p + (size_t)border * elem_bytes;
  return VPP_OK;
}

}  // extern "C"
