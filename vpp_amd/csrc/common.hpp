// common.hpp — shared host-side plumbing of the gfx950 engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/vpp_amd.h"

namespace vpp_amd {

void set_error(const char* fmt, ...);
int tuning(const char* name, int dflt);  // runtime tuning knobs (vpp_set_tuning)

#define VPP_HIP_TRY(expr)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ::vpp_amd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VPP_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

#define VPP_REQUIRE(cond, status, ...)        \
  do {                                        \
    if (!(cond)) {                            \
      ::vpp_amd::set_error(__VA_ARGS__);      \
      return (status);                        \
    }                                         \
  } while (0)

#define VPP_LAUNCH_CHECK() VPP_HIP_TRY(hipGetLastError())

// RCCL plumbing shared by the sharded entry points (comm.hip): every rank holds `bytes_per_rank` bytes at base + rank * bytes_per_rank
// and ends up with all of them (in-place all-gather); several gathers between comm_group_begin / _end go out as one RCCL launch.
int comm_info(const vpp_comm* comm, int* rank, int* nranks);
int comm_allgather_inplace(vpp_comm* comm, void* base, size_t bytes_per_rank, hipStream_t st);
int comm_group_begin();
int comm_group_end();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int dtype_size(int dt) {
  switch (dt) { case VPP_U8: case VPP_I8: return 1; case VPP_U16: case VPP_I16: return 2; default: return 4; }
}

// Device-side image handle, passed to kernels by value.
struct DImg {
  uint8_t* p0; int nr, nc, pitch, border, dtype, ch;
  template <class T> __device__ __forceinline__ T* row(int r) const { return (T*)(p0 + (ptrdiff_t)r * pitch); }
  __device__ __forceinline__ bool has(int r, int c) const { return r >= 0 && c >= 0 && r < nr && c < nc; }
};
inline DImg dimg(const vpp_image_desc* d) {
  return DImg{(uint8_t*)d->first_pixel, d->nrows, d->ncols, d->pitch, d->border, d->dtype, d->channels};
}
inline bool valid_desc(const vpp_image_desc* d) {
  return d && d->first_pixel && d->nrows > 0 && d->ncols > 0 && d->pitch > 0 && d->border >= 0 && d->channels > 0 &&
         d->dtype >= VPP_U8 && d->dtype <= VPP_F32;
}
inline int elem_bytes(const vpp_image_desc* d) { return dtype_size(d->dtype) * d->channels; }
inline bool same_domain(const vpp_image_desc* a, const vpp_image_desc* b) { return a->nrows == b->nrows && a->ncols == b->ncols; }
inline bool same_type(const vpp_image_desc* a, const vpp_image_desc* b) { return a->dtype == b->dtype && a->channels == b->channels; }
inline bool aligned16(const vpp_image_desc* d) { return ((uintptr_t)d->first_pixel % 16) == 0 && (d->pitch % 16) == 0; }

// Grow-only device scratch, per host thread and per user (the C ABI is re-entrant across host threads), one buffer per
// (device, stream): work that uses it is stream-ordered, so calls on one stream reuse one buffer and calls on different streams
// (several frame pairs in flight) or devices never share one — no cross-stream synchronisation, nothing that is illegal under
// stream capture (a capture must not be the FIRST call on its stream at a given size: growing allocates).  At most kSlots
// buffers are kept; the least recently used one is released (after a device synchronise) when another stream shows up, which
// also retires the buffers of destroyed streams.
struct Scratch {
  static constexpr int kSlots = 16;   // one per (device, stream) that has called in: 8 was the ceiling of the frame-pairs-in-flight case (the 9th stream evicted — and synchronised — every call)
  // user[]: the owner's notes about what the buffer holds (e.g. "this region is zeroed"); cleared whenever the buffer is (re)allocated
  struct Slot { void* p = nullptr; size_t cap = 0; int dev = -1; hipStream_t st = nullptr; unsigned long long used = 0; unsigned long long user[4] = {0, 0, 0, 0}; };
  Slot* cur = nullptr;   // the slot of the last ensure()
  Slot slots[kSlots];
  unsigned long long tick = 0;
  void* p = nullptr;   // the buffer of the last ensure()
  int ensure(size_t bytes, hipStream_t st) {
    int dev = 0;
    VPP_HIP_TRY(hipGetDevice(&dev));
    Slot* s = nullptr;
    for (Slot& c : slots) if (c.p && c.dev == dev && c.st == st) { s = &c; break; }
    if (!s) {
      for (Slot& c : slots) if (!c.p) { s = &c; break; }
      if (!s) {  // evict the least recently used buffer; it may still be in use by queued work, or belong to another device
        s = &slots[0];
        for (Slot& c : slots) if (c.used < s->used) s = &c;
        if (release(*s) != VPP_OK) return VPP_ERR_HIP;
      }
      s->dev = dev; s->st = st;
    }
    s->used = ++tick;
    if (bytes > s->cap) {
      if (s->p) { VPP_HIP_TRY(hipStreamSynchronize(st)); VPP_HIP_TRY(hipFree(s->p)); s->p = nullptr; s->cap = 0; }
      VPP_HIP_TRY(hipMalloc(&s->p, bytes));
      s->cap = bytes;
      for (unsigned long long& u : s->user) u = 0;
    }
    p = s->p; cur = s;
    return VPP_OK;
  }
  static int release(Slot& c) {
    if (!c.p) return VPP_OK;
    int cur = 0;
    VPP_HIP_TRY(hipGetDevice(&cur));
    if (cur != c.dev) VPP_HIP_TRY(hipSetDevice(c.dev));
    (void)hipDeviceSynchronize();
    const hipError_t e = hipFree(c.p);
    if (cur != c.dev) VPP_HIP_TRY(hipSetDevice(cur));
    c = Slot();
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("scratch: hipFree failed: %s", hipGetErrorString(e)); return VPP_ERR_HIP; }
    return VPP_OK;
  }
  ~Scratch() { for (Slot& c : slots) if (c.p) (void)hipFree(c.p); }
};

// blockIdx remap so that consecutive logical blocks share an XCD (hardware places block b on XCD b % 8;
// MI355X_MICROARCH.md "Workgroup dispatch").  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblocks) {
  const unsigned per = nblocks >> 3;            // blocks per XCD (floor)
  const unsigned main = per << 3;
  if (b >= main) return b;                      // ragged tail keeps its index
  return (b & 7u) * per + (b >> 3);
}

}  // namespace vpp_amd
