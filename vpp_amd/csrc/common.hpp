// common.hpp — shared host-side plumbing of the gfx950 engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <vector>
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

// ---- held-back per-frame calls (vpp_*_deferred, and the plain per-frame entry points on a stream recorded through vpp_graph_begin) ------------------------
// The reference's call form is one frame per call (benchmarks/box_5x5_filter2.cc:43-81, benchmarks/image_add.cc:51-57); eagerly that is one launch per call, and
// a 50 MB launch reaches 46 % of the HBM peak where 64 frames in one launch reach 74 %.  A held-back call does not launch: it appends its frame to the calling
// thread's window while the frame is unrelated to the pending ones (no pending result overlaps its source or result, its result overlaps no pending source) and
// has their geometry; the window goes out as ONE batched launch (the vpp_*_batch kernels: "the results of the n calls one after the other") when it holds
// kDeferMax frames, when a call that cannot join arrives, at vpp_flush / vpp_graph_end — and before ANYTHING else is queued through this ABI by the same thread (on
// any stream) or by ANY thread on the window's stream: every entry point turns its stream argument into a hipStream_t with as_stream(), which launches those
// windows first.  Stream order is therefore exactly that of the calls, also when one host thread makes the calls and another one synchronises the stream.
// ONE mechanism serves the eager and the recorded form (round 6; until then a recorded call re-parameterised the previous call's node with
// hipGraphKernelNodeSetParams while the graph was under capture): on a stream this thread records through vpp_graph_begin the plain per-frame entry points hold
// their frames back the same way, and a window that closes records ONE kernel node.  No node of a graph under capture is ever edited.
constexpr int kDeferMax = 64;
enum { kDeferNone = 0, kDeferBox = 1, kDeferBinary = 2, kDeferGray = 3 };
struct DeferBatch {                // what one launch carries
  int n = 0;                       // pending calls
  int kind = kDeferNone, p0 = 0, p1 = 0;   // the entry point and its scalar parameters (box: R, C; binary: op; gray: mirror)
  int nsrc = 0;
  void* stream = nullptr;
  int dev = 0;                     // the device that was current when the window opened (its stream's device)
  vpp_image_desc dst[kDeferMax], src[2][kDeferMax];
};
struct DeferWindow;
DeferWindow& defer_window();       // the calling thread's (created and registered at first use, launched and unregistered when the thread ends)
extern std::atomic<int> g_defer_pending;   // frames held back by ALL threads: the one word as_stream() reads on the common path
extern thread_local int g_defer_bypass;    // > 0 while this thread launches a window (the batch entry points are ordinary entry points: they must not hold back or flush again)
int defer_flush();                 // launches the calling thread's window (if any) and empties it; the launch's status (kept as the window's sticky error when it failed)
int defer_flush_stream(void* stream);   // + every other thread's window that waits on `stream`
// the call joins the calling thread's window (which is launched first when the call cannot join it: another entry point / parameters / stream / geometry, or
// data flow between the call and a pending one) and launches it when it is full.  Always VPP_OK: a window that fails to launch is not THIS call's failure — it
// is kept as the window's sticky error (status, entry point, frame count) and reported by this thread's next vpp_flush / vpp_sync.
int defer_call(int kind, int p0, int p1, void* stream, const vpp_image_desc* dst, const vpp_image_desc* src0, const vpp_image_desc* src1);
// true: `stream` is being recorded by this thread between vpp_graph_begin and vpp_graph_end (then vpp_graph_end closes the last window, so frames may be held
// back; a capture begun by other means ends where this library cannot see it: every call records its own node there)
bool defer_recording(void* stream);

inline hipStream_t as_stream(void* s) {
  if (g_defer_pending.load(std::memory_order_acquire) && !g_defer_bypass) (void)defer_flush_stream(s);   // whatever this call queues comes after the calls held back before it
  return reinterpret_cast<hipStream_t>(s);
}

// memset by a kernel of this library (runtime.hip), for every fill that may be recorded into a launch graph: on ROCm 7.2 the runtime's memset NODES were
// not ordered with the kernel nodes around them (a replayed flow read a half-zeroed control block and faulted), and eagerly hipMemsetAsync is two dispatches.
int device_fill(void* dst, int byte, size_t bytes, hipStream_t st);

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
// A batch entry point promises the results of n calls made one after the other.  In ONE launch that only holds when no frame's result is another frame's
// input or result: true when dst[j] overlaps dst[k] or any source of frame k, k != j (a frame's own in-place / aliasing rules are the single call's).
// Byte extents are the whole bordered areas.  O(n log n): the extents sorted by start, and for each kind the two furthest-reaching open extents of
// different frames.
inline bool batch_frames_interfere(int n, const vpp_image_desc* dst, const vpp_image_desc* const* srcs, int nsrc) {
  struct Ext { uintptr_t lo, hi; int frame; bool is_dst; };
  std::vector<Ext> v;
  v.reserve((size_t)n * (1 + nsrc));
  auto ext = [](const vpp_image_desc& d, int frame, bool is_dst) {
    const uintptr_t p = (uintptr_t)d.first_pixel;
    const size_t es = (size_t)elem_bytes(&d);
    return Ext{p - (size_t)d.border * d.pitch - (size_t)d.border * es, p + (size_t)(d.nrows - 1 + d.border) * d.pitch + (size_t)(d.ncols + d.border) * es, frame, is_dst};
  };
  for (int k = 0; k < n; k++) {
    v.push_back(ext(dst[k], k, true));
    for (int q = 0; q < nsrc; q++) v.push_back(ext(srcs[q][k], k, false));
  }
  std::sort(v.begin(), v.end(), [](const Ext& a, const Ext& b) { return a.lo < b.lo; });
  struct Top { uintptr_t hi[2] = {0, 0}; int frame[2] = {-1, -1}; };   // the furthest end seen, and the furthest end of another frame than that one
  auto reaches = [](const Top& t, uintptr_t lo, int frame) { return (t.frame[0] >= 0 && t.frame[0] != frame && t.hi[0] > lo) || (t.frame[1] >= 0 && t.frame[1] != frame && t.hi[1] > lo); };
  auto add = [](Top& t, uintptr_t hi, int frame) {
    if (t.frame[0] == frame) { t.hi[0] = std::max(t.hi[0], hi); return; }
    if (t.frame[0] < 0 || hi > t.hi[0]) { t.hi[1] = t.hi[0]; t.frame[1] = t.frame[0]; t.hi[0] = hi; t.frame[0] = frame; return; }
    if (t.frame[1] < 0 || t.frame[1] == frame || hi > t.hi[1]) { if (t.frame[1] != frame || hi > t.hi[1]) { t.hi[1] = hi; t.frame[1] = frame; } }
  };
  Top dsts, srcx;
  for (const Ext& e : v) {
    if (e.is_dst ? (reaches(dsts, e.lo, e.frame) || reaches(srcx, e.lo, e.frame)) : reaches(dsts, e.lo, e.frame)) return true;
    add(e.is_dst ? dsts : srcx, e.hi, e.frame);
  }
  return false;
}

// ---- independent calls recorded side by side -------------------------------------------------------------------------------------------------------
// A stream orders every call behind the one before it, and a 4K streaming launch pays ~4.6 us of ramp and drain that the next launch cannot hide
// behind a kernel boundary (LABNOTES.md section 5: one 50 MB launch per call reaches 47 % of the HBM peak, 64 frames in one launch 70-75 %).  When a
// stream is being RECORDED into a launch graph, the order only has to hold where data flows: a call that brackets its launches with an
// IndependentCall — stating the byte extents it writes and reads — is recorded behind (a) whatever the stream depended on when the window opened
// (work of callers this library cannot see), (b) the last recorded call of its lane (at most `launch.capture_width` calls side by side) and (c) every
// recorded call whose extents its own overlap (write / write, write / read, read / write): data flow is kept exactly, independent calls become
// sibling nodes whose tails overlap when the graph runs.  After each call the stream's dependency set is the join of all lanes, so anything recorded
// by anyone else afterwards waits for all of them — for every other observer the stream's semantics are unchanged.  Outside a capture it does nothing.
struct Extent { uintptr_t lo, hi; };
inline Extent extent_of(const vpp_image_desc& d) {   // the whole bordered area
  const uintptr_t p = (uintptr_t)d.first_pixel;
  const size_t es = (size_t)elem_bytes(&d);
  return Extent{p - (size_t)d.border * d.pitch - (size_t)d.border * es, p + (size_t)(d.nrows - 1 + d.border) * d.pitch + (size_t)(d.ncols + d.border) * es};
}
class IndependentCall {
 public:
  IndependentCall(hipStream_t st, const Extent* writes, int nw, const Extent* reads, int nr);
  ~IndependentCall() { if (active_) (void)finish(); }
  IndependentCall(const IndependentCall&) = delete;
  IndependentCall& operator=(const IndependentCall&) = delete;
  // the call's launches are recorded: books them, rejoins the lanes; returns the call's (last) node — nullptr when there is no single one
  hipGraphNode_t finish();
 private:
  hipStream_t st_;
  bool active_ = false;
  int lane_ = 0;
  Extent w_[2], r_[3];
  int nw_ = 0, nr_ = 0;
};

// Bumped whenever something may have left ANY scratch buffer in an unknown state (a device-side protocol that gave up, check_device_error): every note
// taken before the bump stops being believed.
unsigned notes_epoch();
void invalidate_scratch_notes();
// Launch graphs hold the ADDRESS of the scratch buffer their calls were recorded on.  A later, larger eager call on the stream gets a NEW buffer and the recorded
// one is retired alive (Scratch::ensure), so those graphs keep replaying.  Only the eviction of a recorded buffer (a 17th stream on one host thread) frees it:
// this generation is then bumped; a vpp_graph remembers the generation it was recorded under and vpp_graph_launch refuses a stale one (VPP_ERR_INVALID_ARG,
// "record it again") instead of replaying into freed memory.  Conservative: graphs that never touched that buffer are refused as well.
unsigned recorded_scratch_generation();
void recorded_scratch_freed();

struct Scratch {
  static constexpr int kSlots = 16;   // one per (device, stream) that has called in: 8 was the ceiling of the frame-pairs-in-flight case (the 9th stream evicted — and synchronised — every call)
  // user[]: the owner's notes about what the buffer holds (e.g. "this region is zeroed"); cleared whenever the buffer is (re)allocated.
  // Notes describe the buffer as the LAST QUEUED call leaves it.  A call recorded into a launch graph runs later, any number of times, between any
  // other calls — so once a capture has recorded work on a buffer (`recorded`, sticky until the buffer is reallocated) no note about it can be
  // trusted any more and none is taken: note() then always reports "unknown", and every call (recorded or eager) carries its own resets.
  struct Slot {
    void* p = nullptr; size_t cap = 0; int dev = -1; hipStream_t st = nullptr; unsigned long long used = 0; unsigned long long user[4] = {0, 0, 0, 0};
    bool recorded = false;     // a stream capture has recorded work that uses this buffer
    bool capturing = false;    // the call that is being queued right now is being captured
    unsigned epoch = 0;        // notes_epoch() when the notes were last written
    bool note(int i, unsigned long long sig) const { return !recorded && epoch == notes_epoch() && user[i] == sig; }
    void set_note(int i, unsigned long long sig) {
      if (epoch != notes_epoch()) { for (unsigned long long& u : user) u = 0; epoch = notes_epoch(); }
      user[i] = recorded ? 0 : sig;
    }
  };
  Slot* cur = nullptr;   // the slot of the last ensure()
  Slot slots[kSlots];
  unsigned long long tick = 0;
  void* p = nullptr;   // the buffer of the last ensure()
  int ensure(size_t bytes, hipStream_t st) {
    int dev = 0;
    VPP_HIP_TRY(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    Slot* s = nullptr;
    for (Slot& c : slots) if (c.p && c.dev == dev && c.st == st) { s = &c; break; }
    // growing the buffer synchronises the stream and allocates — both illegal while the stream is captured, and the graph would keep the address of a
    // buffer that a later, larger call frees: refuse readably instead of failing inside the runtime
    VPP_REQUIRE(!capturing || (s && bytes <= s->cap), VPP_ERR_UNSUPPORTED,
                "the call needs %zu bytes of scratch on this stream and a stream capture cannot allocate them: run the same call once on this stream before recording it", bytes);
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
      if (s->p && s->recorded) { retired.push_back(s->p); s->p = nullptr; s->cap = 0; }   // launch graphs hold this buffer's address: it stays alive (and is never handed out again) until the thread's scratch dies
      if (s->p) { VPP_HIP_TRY(hipStreamSynchronize(st)); VPP_HIP_TRY(hipFree(s->p)); s->p = nullptr; s->cap = 0; }
      VPP_HIP_TRY(hipMalloc(&s->p, bytes));
      s->cap = bytes;
      for (unsigned long long& u : s->user) u = 0;
      s->recorded = false;   // (graphs recorded on the old buffer keep replaying on it: retired above, not freed)
    }
    s->capturing = capturing;
    if (capturing) { s->recorded = true; for (unsigned long long& u : s->user) u = 0; }
    p = s->p; cur = s;
    return VPP_OK;
  }
  static int release(Slot& c) {
    if (!c.p) return VPP_OK;
    int cur = 0;
    VPP_HIP_TRY(hipGetDevice(&cur));
    if (cur != c.dev) VPP_HIP_TRY(hipSetDevice(c.dev));
    (void)hipDeviceSynchronize();
    if (c.recorded) recorded_scratch_freed();
    const hipError_t e = hipFree(c.p);
    if (cur != c.dev) VPP_HIP_TRY(hipSetDevice(cur));
    c = Slot();
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("scratch: hipFree failed: %s", hipGetErrorString(e)); return VPP_ERR_HIP; }
    return VPP_OK;
  }
  std::vector<void*> retired;   // buffers that launch graphs were recorded on and that a larger eager call has since replaced
  ~Scratch() { for (Slot& c : slots) if (c.p) (void)hipFree(c.p); for (void* q : retired) (void)hipFree(q); }
};

// Sticky device-side error word (runtime.hip): one 32-bit word in pinned, device-visible host memory that kernels raise bits of when a device-side
// protocol gives up (a grid barrier's poll limit); vpp_sync and the tracker's count read-back check it after the stream has drained and return
// VPP_ERR_HIP once (the word is cleared by the report).  nullptr when the allocation failed (the kernels then skip the store).
// The word is PROCESS-GLOBAL: whichever stream's synchronisation through this ABI comes first reports (and clears) a fault raised by any stream's kernels
// (include/vpp_amd.h says so).  A caller that synchronises outside the ABI never has it reported; the flow entry points therefore also peek at it before they
// queue anything (peek_device_error: no clearing) and stop trusting their scratch notes while a bit is up.
unsigned* device_error_word();
int check_device_error(const char* where);   // VPP_OK, or VPP_ERR_HIP + vpp_last_error() when a bit is up
unsigned peek_device_error();                // the bits that are up, left as they are (0 when the word was never allocated)
enum { kDevErrSweepBarrier = 1u, kDevErrFastFuse = 2u };   // the flow's grid barrier; FAST-9's in-launch ordered write (fast9.hip)

// blockIdx remap so that consecutive logical blocks share an XCD (hardware places block b on XCD b % 8;
// MI355X_MICROARCH.md "Workgroup dispatch").  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblocks) {
  const unsigned per = nblocks >> 3;            // blocks per XCD (floor)
  const unsigned main = per << 3;
  if (b >= main) return b;                      // ragged tail keeps its index
  return (b & 7u) * per + (b >> 3);
}

}  // namespace vpp_amd
