// device.hh — glue between the vpp-shaped C++ surface and the C ABI (include/vpp_amd.h) of the gfx950 engine.
// With -DVPP_AMD_DEVICE every image owns (lazily) a mirror of its whole pitched buffer in HBM; the algorithm front-ends
// and the tagged pixel_wise functors run there.  Without it the containers and the host expression engine still work
// (BASELINE configs[0]: "plumbing, no GPU"), and every device-only entry point is a hard compile error — there is no CPU
// re-implementation of the hot-path algorithms in these headers to fall back to.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#ifdef VPP_AMD_DEVICE
#include <vpp_amd.h>
#endif

namespace vpp {
namespace device {

#ifdef VPP_AMD_DEVICE
inline void check(int status, const char* what) {
  if (status == VPP_OK) return;
  // the reference's only hot-path exception (fast.hpp:937-938) keeps its type and text
  throw std::runtime_error(status == VPP_ERR_BORDER_TOO_SMALL ? std::string(vpp_last_error()) : std::string(what) + ": " + vpp_last_error());
}
inline void* stream() { return nullptr; }
// What every device call of the drop-in surface ends with.  The reference's calls are synchronous (pixel_wise joins its OpenMP loop); what a caller can
// OBSERVE of that is kept without draining the GPU after every call (a drained 4K call costs ~6 us more than its kernel, and nothing overlaps):
//  * pixels: every host accessor waits for the image's mirror (storage::to_host copies on the same stream and waits for the copy);
//  * the wall clock of a loop of calls: a call returns once the call BEFORE THE PREVIOUS ONE has completed — at most kCallsInFlight calls are queued, so a
//    timed loop measures the device's sustained rate to within that many calls, and memory handed back by a dying image is reused in stream order;
//  * errors: a failing launch throws from the call itself, a fault in a running kernel from one of the next calls (or the next host access).
// VPP_AMD_SYNC_CALLS=1 in the environment: every call waits for its own completion (the behaviour up to round 3).
constexpr int kCallsInFlight = 2;
inline void call_done() {
  static const bool sync_all = [] { const char* e = std::getenv("VPP_AMD_SYNC_CALLS"); return e && e[0] == '1'; }();
  if (sync_all) { check(vpp_sync(stream()), "vpp_sync"); return; }
  struct ring { void* ev[kCallsInFlight] = {}; bool set[kCallsInFlight] = {}; int k = 0; };
  static thread_local ring r;   // (events live as long as the process: the HIP runtime may be gone before thread-exit destructors run)
  if (!r.ev[r.k]) check(vpp_event_create(&r.ev[r.k]), "vpp_event_create");
  check(vpp_event_record(r.ev[r.k], stream()), "vpp_event_record");
  r.set[r.k] = true;
  r.k = (r.k + 1) % kCallsInFlight;
  if (r.set[r.k]) check(vpp_event_synchronize(r.ev[r.k]), "vpp_event_synchronize");   // the oldest call still queued
}
// Deferred per-frame calls (include/vpp_amd.h: vpp_*_deferred): the tagged functors and rgb_to_graylevel do not launch per call — the library holds the frame back
// and launches whole batches (up to 64 frames), always before anything else this thread queues, before anything ANY thread queues on the same stream (a host
// accessor or sync() on another thread, once this thread has handed the image over) and when the thread ends — so every host accessor, every other call and sync()
// see the results exactly as with per-call launches.  `pixel_wise(...)(_immediate) | ops::...` launches that call at once instead (symbols.hh).
// The queue bound of call_done() then applies per BATCH: after a call that made the library launch a window.
inline void deferred_call_done() {
  static thread_local unsigned long long seen = 0;
  static thread_local int unthrottled = 0;
  const unsigned long long f = vpp_deferred_flushes();
  if (f == seen) return;
  seen = f;
  // (call_done records an event, which would launch a window that has just been opened: throttle when nothing is pending — the full-window case — or after
  // many small windows in a row, a chain of dependent calls)
  if (vpp_deferred_pending() == 0 || ++unthrottled >= 64) { unthrottled = 0; call_done(); }
}
// Kernels that the CALLER's translation unit launches itself (the opaque-lambda engines of pixel_wise_device.hh) do not pass through the ABI, so nothing would
// launch the frames this thread's tagged functors have held back before them: every such launch calls this first (stream order = call order).
inline void flush_held_back() { if (vpp_deferred_pending()) check(vpp_flush(stream()), "vpp_flush"); }
// Everything queued so far — deferred frames included — has completed when sync() returns.
inline void sync() { check(vpp_sync(stream()), "vpp_sync"); }
// RAII: the deferred frames of the enclosed calls are launched when the scope ends (they are launched earlier whenever order demands it; see above).
struct batch_scope {
  batch_scope() {}
  ~batch_scope() { (void)vpp_flush(stream()); }
  batch_scope(const batch_scope&) = delete;
  batch_scope& operator=(const batch_scope&) = delete;
};
template <class T> struct dtype_of { static_assert(sizeof(T) == 0, "pixel component type not supported on the device"); };
template <> struct dtype_of<unsigned char> { enum { value = VPP_U8 }; };
template <> struct dtype_of<signed char> { enum { value = VPP_I8 }; };
template <> struct dtype_of<char> { enum { value = VPP_I8 }; };
template <> struct dtype_of<unsigned short> { enum { value = VPP_U16 }; };
template <> struct dtype_of<short> { enum { value = VPP_I16 }; };
template <> struct dtype_of<int> { enum { value = VPP_I32 }; };
template <> struct dtype_of<unsigned int> { enum { value = VPP_U32 }; };
template <> struct dtype_of<float> { enum { value = VPP_F32 }; };  // the null stream: calls are synchronous from the caller's point of view, like the reference
#endif

// One pitched host buffer and its HBM mirror.  state: 0 = host is authoritative (mirror stale or absent),
// 1 = both valid, 2 = the mirror is newer.
struct storage {
  char* host = nullptr;
  size_t bytes = 0;
  int state = 0;
  void* dev = nullptr;
  ~storage() {
#ifdef VPP_AMD_DEVICE
    if (dev) vpp_free(dev);
#endif
  }
  void to_host(bool for_write) {
#ifdef VPP_AMD_DEVICE
    if (state == 2) { check(vpp_memcpy_d2h(host, dev, bytes, stream()), "vpp_memcpy_d2h"); state = 1; }
    if (for_write) state = 0;
#else
    (void)for_write;
#endif
  }
#ifdef VPP_AMD_DEVICE
  // pointer into the mirror corresponding to host address p; the mirror is brought up to date first
  // discard: the caller overwrites every pixel (domain and border), so a stale mirror is not refreshed from the host first
  void* to_device(const void* p, bool will_write, bool discard = false) {
    if (!dev) check(vpp_malloc(bytes, &dev), "vpp_malloc");
    if (state == 0 && !(discard && will_write)) { check(vpp_memcpy_h2d(dev, host, bytes, stream()), "vpp_memcpy_h2d"); state = 1; }
    if (will_write) state = 2;
    return (char*)dev + ((const char*)p - host);
  }
#endif
};

}  // namespace device
}  // namespace vpp
