// runtime.hip — device/runtime entry points of the C ABI (include/vpp_amd.h "runtime" block).
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace vpp_amd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
static std::mutex g_tune_mu;
static std::map<std::string, int> g_tune;
int tuning(const char* name, int dflt) {
  std::lock_guard<std::mutex> l(g_tune_mu);
  // VPP_TUNE="knob=value,knob=value" presets the knobs of vpp_set_tuning for programs that cannot call it (the C++ harnesses under tools/ A/B runs); read once
  static const bool env_read = [] {
    if (const char* e = getenv("VPP_TUNE")) {
      std::string t(e);
      for (size_t i = 0; i < t.size();) {
        const size_t c = t.find(',', i), end = c == std::string::npos ? t.size() : c, eq = t.find('=', i);
        if (eq != std::string::npos && eq < end) g_tune[t.substr(i, eq - i)] = atoi(t.substr(eq + 1, end - eq - 1).c_str());
        i = end + 1;
      }
    }
    return true;
  }();
  (void)env_read;
  auto it = g_tune.find(name);
  return it == g_tune.end() ? dflt : it->second;
}
// The sticky device-side error word (common.hpp): pinned host memory mapped into every device, written by kernels with plain stores (a PCIe write,
// only ever on a failure path), read by the host after a synchronisation at the cost of one load.
static std::once_flag g_deverr_once;
static unsigned* g_deverr = nullptr;
unsigned* device_error_word() {
  std::call_once(g_deverr_once, [] {
    void* q = nullptr;
    if (hipHostMalloc(&q, 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) { memset(q, 0, 64); g_deverr = (unsigned*)q; }
    else (void)hipGetLastError();
  });
  return g_deverr;
}
int check_device_error(const char* where) {
  unsigned* w = g_deverr;   // never allocated: no kernel could have raised it
  if (!w) return VPP_OK;
  const unsigned bits = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);
  if (!bits) return VPP_OK;
  invalidate_scratch_notes();   // whatever that kernel left in its scratch region is unknown: the next call resets it
  set_error("%s: a device-side protocol gave up waiting (bits 0x%x:%s%s) - the results of the calls queued before this point are not valid", where, bits,
            (bits & kDevErrSweepBarrier) ? " grid barrier of the semi-dense flow's propagation rounds" : "", (bits & kDevErrFastFuse) ? " FAST-9's in-launch ordered write" : "");
  return VPP_ERR_HIP;
}
unsigned peek_device_error() {
  unsigned* w = g_deverr;
  return w ? __atomic_load_n(w, __ATOMIC_ACQUIRE) : 0u;
}
static std::atomic<unsigned> g_recorded_scratch_gen{1};
unsigned recorded_scratch_generation() { return g_recorded_scratch_gen.load(std::memory_order_acquire); }
void recorded_scratch_freed() { g_recorded_scratch_gen.fetch_add(1, std::memory_order_acq_rel); }
static std::atomic<unsigned> g_notes_epoch{1};
unsigned notes_epoch() { return g_notes_epoch.load(std::memory_order_relaxed); }
void invalidate_scratch_notes() { g_notes_epoch.fetch_add(1, std::memory_order_relaxed); }

// ---- held-back per-frame calls (common.hpp) ------------------------------------------------------------------------------------------------------------
std::atomic<int> g_defer_pending{0};
thread_local int g_defer_bypass = 0;
extern "C" int vpp_box_filter_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, int R, int C, void* stream);
extern "C" int vpp_pixelwise_binary_batch(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, int n, void* stream);
extern "C" int vpp_rgb_to_graylevel_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, int mirror, void* stream);
// A window belongs to the thread that fills it.  Everything that touches it — an append, a launch — happens under `mu`: the owner takes it for every call
// (uncontended: tens of nanoseconds), another thread only when it queues something on the window's stream while frames are pending (it then launches them
// itself, and the owner's next call waits for that launch to be queued: the frames of one thread never overtake each other).
struct DeferWindow {
  std::mutex mu;
  DeferBatch b;
  std::atomic<unsigned long long> flushes{0};   // batches launched for this thread so far (vpp_deferred_flushes: the C++ surface throttles per batch, not per call)
  int last_rc = VPP_OK;                         // sticky: a window of this thread failed to launch; reported (once) by its next vpp_flush / vpp_sync
  char last_msg[200] = "";
};
namespace {
// lock order: registry, then a window.  Both objects are leaked on purpose: a thread that ends (or flushes) while the process is already running its static
// destructors must still find them alive.
std::mutex& g_defer_reg_mu = *new std::mutex;
std::vector<DeferWindow*>& g_defer_reg = *new std::vector<DeferWindow*>;   // every live thread's window
const char* defer_kind_name(int kind) { return kind == kDeferBox ? "vpp_box_filter" : kind == kDeferBinary ? "vpp_pixelwise_binary" : kind == kDeferGray ? "vpp_rgb_to_graylevel" : "?"; }
// launches the window and empties it (w.mu held by the caller)
int defer_launch_locked(DeferWindow& w) {
  DeferBatch& b = w.b;
  const int n = b.n;
  if (!n) return VPP_OK;
  b.n = 0;
  // (the process-wide count goes down only when the launch has been QUEUED, below: a thread that reads 0 in as_stream() and queues its own work at once must come
  // after these frames — while the count is up it comes here instead and waits for w.mu)
  int rc = VPP_OK;
  int cur = b.dev;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = b.dev; }
  if (cur != b.dev && hipSetDevice(b.dev) != hipSuccess) { (void)hipGetLastError(); set_error("cannot return to device %d", b.dev); rc = VPP_ERR_HIP; }
  if (rc == VPP_OK) {
    g_defer_bypass++;   // the batch entry points are ordinary entry points: they neither hold back nor flush while a window is being launched
    switch (b.kind) {
      case kDeferBox: rc = vpp_box_filter_batch(b.dst, b.src[0], n, b.p0, b.p1, b.stream); break;
      case kDeferBinary: rc = vpp_pixelwise_binary_batch(b.p0, b.dst, b.src[0], b.src[1], n, b.stream); break;
      case kDeferGray: rc = vpp_rgb_to_graylevel_batch(b.dst, b.src[0], n, b.p0, b.stream); break;
      default: break;
    }
    g_defer_bypass--;
    if (cur != b.dev) (void)hipSetDevice(cur);   // the caller has moved to another device since: back to it
  }
  g_defer_pending.fetch_sub(n, std::memory_order_acq_rel);
  w.flushes.fetch_add(1, std::memory_order_release);
  if (rc != VPP_OK) {   // the n frames were not queued: say so at the thread's next vpp_flush / vpp_sync
    w.last_rc = rc;
    snprintf(w.last_msg, sizeof w.last_msg, "a held-back batch of %d %s call%s failed to launch and was dropped (status %d): %.80s", n, defer_kind_name(b.kind), n == 1 ? "" : "s", rc, g_err);
  }
  return rc;
}
struct DeferOwner {
  DeferWindow* w;
  DeferOwner() : w(new DeferWindow()) { std::lock_guard<std::mutex> l(g_defer_reg_mu); g_defer_reg.push_back(w); }
  // a thread that ends with frames held back launches them (until round 6 it had to vpp_flush before it ended, and silently dropped its calls when it did not)
  ~DeferOwner() {
    { std::lock_guard<std::mutex> l(w->mu); (void)defer_launch_locked(*w); }
    { std::lock_guard<std::mutex> l(g_defer_reg_mu); g_defer_reg.erase(std::find(g_defer_reg.begin(), g_defer_reg.end(), w)); }
    delete w;
  }
};
}  // namespace
DeferWindow& defer_window() { static thread_local DeferOwner owner; return *owner.w; }
int defer_flush() {
  DeferWindow& w = defer_window();
  std::lock_guard<std::mutex> l(w.mu);
  return defer_launch_locked(w);
}
int defer_flush_stream(void* stream) {
  int rc = defer_flush();   // this thread's own window, whatever its stream
  if (!g_defer_pending.load(std::memory_order_acquire)) return rc;
  // other threads' windows that wait on this stream: this thread is about to queue behind calls they have already made (the threads ordered themselves — a
  // thread that hands a stream over has returned from its calls).  Under the registry lock: an owner cannot end, and free its window, meanwhile.
  DeferWindow* mine = &defer_window();
  std::lock_guard<std::mutex> lr(g_defer_reg_mu);
  for (DeferWindow* o : g_defer_reg) {
    if (o == mine) continue;
    std::lock_guard<std::mutex> l(o->mu);
    if (o->b.n && o->b.stream == stream) { const int r2 = defer_launch_locked(*o); if (rc == VPP_OK) rc = r2; }
  }
  return rc;
}
namespace {
inline bool same_frame_geometry(const vpp_image_desc& a, const vpp_image_desc& b) {
  return a.nrows == b.nrows && a.ncols == b.ncols && a.pitch == b.pitch && a.border == b.border && a.dtype == b.dtype && a.channels == b.channels &&
         (((uintptr_t)a.first_pixel ^ (uintptr_t)b.first_pixel) & 15) == 0;
}
inline bool extents_overlap(const Extent& a, const Extent& b) { return a.lo < b.hi && b.lo < a.hi; }
}  // namespace
int defer_call(int kind, int p0, int p1, void* stream, const vpp_image_desc* dst, const vpp_image_desc* src0, const vpp_image_desc* src1) {
  DeferWindow& w = defer_window();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  std::lock_guard<std::mutex> l(w.mu);
  DeferBatch& b = w.b;
  if (b.n) {
    bool join = b.dev == dev && b.kind == kind && b.p0 == p0 && b.p1 == p1 && b.stream == stream && same_frame_geometry(*dst, b.dst[0]) && same_frame_geometry(*src0, b.src[0][0]) &&
                (!src1 || same_frame_geometry(*src1, b.src[1][0]));
    if (join) {   // no data flow between this frame and a pending one (reads of one source by several frames are fine)
      const Extent d = extent_of(*dst), s0 = extent_of(*src0), s1 = src1 ? extent_of(*src1) : Extent{0, 0};
      for (int k = 0; join && k < b.n; k++) {
        const Extent pd = extent_of(b.dst[k]);
        join = !extents_overlap(d, pd) && !extents_overlap(s0, pd) && !(src1 && extents_overlap(s1, pd)) && !extents_overlap(d, extent_of(b.src[0][k])) &&
               !(b.nsrc > 1 && extents_overlap(d, extent_of(b.src[1][k])));
      }
    }
    if (!join) (void)defer_launch_locked(w);   // (its failure is the window's sticky error, not this call's: this call's frame is only held back)
  }
  if (!b.n) { b.kind = kind; b.p0 = p0; b.p1 = p1; b.stream = stream; b.dev = dev; b.nsrc = src1 ? 2 : 1; }
  b.dst[b.n] = *dst; b.src[0][b.n] = *src0; if (src1) b.src[1][b.n] = *src1;
  b.n++;
  g_defer_pending.fetch_add(1, std::memory_order_acq_rel);
  if (b.n == kDeferMax) (void)defer_launch_locked(w);
  return VPP_OK;
}
// streams this thread records through vpp_graph_begin, with the recorded-scratch generation at the start of the capture
static thread_local std::map<void*, unsigned> g_recording;
bool defer_recording(void* stream) { return !g_recording.empty() && g_recording.count(stream) != 0; }

// ---- device_fill (common.hpp): 16-byte units for the aligned body, the first workgroup also writes the unaligned head and tail bytes
namespace {
__global__ __launch_bounds__(256) void fill_bytes_kernel(uint8_t* __restrict__ p, size_t head, size_t units, size_t tail, uint32_t v32) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < units) ((uint4*)(p + head))[i] = make_uint4(v32, v32, v32, v32);
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) p[threadIdx.x] = (uint8_t)v32;
    if (threadIdx.x < tail) p[head + units * 16 + threadIdx.x] = (uint8_t)v32;
  }
}
}  // namespace
int device_fill(void* dst, int byte, size_t bytes, hipStream_t st) {
  if (!bytes) return VPP_OK;
  VPP_REQUIRE(dst, VPP_ERR_INVALID_ARG, "device_fill: null");
  const size_t head = std::min(bytes, (size_t)((16 - ((uintptr_t)dst & 15)) & 15)), units = (bytes - head) / 16, tail = bytes - head - units * 16;
  const uint32_t v32 = 0x01010101u * (uint32_t)(byte & 255);
  fill_bytes_kernel<<<(unsigned)std::max<size_t>(1, (units + 255) / 256), 256, 0, st>>>((uint8_t*)dst, head, units, tail, v32);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

// ---- IndependentCall (common.hpp) --------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxLanes = 16;
struct Lane { hipGraphNode_t last = nullptr; };
// what the recorded calls of this window touched: per distinct extent its last writer and, per lane, the last reader since that write
struct Touch { Extent e; hipGraphNode_t writer = nullptr; hipGraphNode_t reader[kMaxLanes] = {}; };
struct CaptureWindow {
  unsigned long long id = 0;
  std::vector<hipGraphNode_t> base, join;
  Lane lanes[kMaxLanes];
  int nlanes = 0;
  size_t count = 0;
  std::vector<Touch> touched;
  std::vector<hipGraphNode_t> deps;   // of the call being recorded
  unsigned long long serial = 0;      // of this window (a node recorded in an earlier window must not be extended any more)
};
thread_local std::map<hipStream_t, CaptureWindow> g_windows;
thread_local unsigned long long g_window_serial = 0;
inline bool overlap(const Extent& a, const Extent& b) { return a.lo < b.hi && b.lo < a.hi; }
inline bool same_set(const std::vector<hipGraphNode_t>& a, const hipGraphNode_t* b, size_t nb) {
  if (a.size() != nb) return false;
  for (size_t i = 0; i < nb; i++) if (std::find(a.begin(), a.end(), b[i]) == a.end()) return false;
  return true;
}
inline void add_unique(std::vector<hipGraphNode_t>& v, hipGraphNode_t n) { if (n && std::find(v.begin(), v.end(), n) == v.end()) v.push_back(n); }
}  // namespace

IndependentCall::IndependentCall(hipStream_t st, const Extent* writes, int nw, const Extent* reads, int nr) : st_(st) {
  const int W = std::min(tuning("launch.capture_width", 2), kMaxLanes);
  if (W < 1 || nw > 2 || nr > 3) return;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  if (hipStreamGetCaptureInfo_v2(st, &status, &id, &graph, &deps, &ndeps) != hipSuccess) { (void)hipGetLastError(); return; }
  if (status != hipStreamCaptureStatusActive) return;
  CaptureWindow& cw = g_windows[st];
  // a new capture, or somebody else has recorded on the stream since the last bracketed call: a new window opens behind whatever the stream depends on now
  if (cw.id != id || !same_set(cw.join, deps, ndeps) || cw.touched.size() > 4096) {
    cw = CaptureWindow();
    cw.id = id;
    cw.serial = ++g_window_serial;
    cw.base.assign(deps, deps + ndeps);
  }
  lane_ = cw.nlanes < W ? cw.nlanes : (int)(cw.count % (size_t)W);
  cw.deps = cw.base;
  if (lane_ < cw.nlanes) add_unique(cw.deps, cw.lanes[lane_].last);
  for (const Touch& t : cw.touched) {   // recorded calls whose extents this call's overlap: recorded behind them
    bool ww = false, rw = false;   // this call writes what was touched / reads what was written
    for (int i = 0; i < nw; i++) ww = ww || overlap(writes[i], t.e);
    for (int i = 0; i < nr; i++) rw = rw || overlap(reads[i], t.e);
    if (ww || rw) add_unique(cw.deps, t.writer);
    if (ww) for (hipGraphNode_t r : t.reader) add_unique(cw.deps, r);
  }
  if (hipStreamUpdateCaptureDependencies(st, cw.deps.data(), cw.deps.size(), hipStreamSetCaptureDependencies) != hipSuccess) { (void)hipGetLastError(); cw = CaptureWindow(); return; }
  nw_ = nw; nr_ = nr;
  for (int i = 0; i < nw; i++) w_[i] = writes[i];
  for (int i = 0; i < nr; i++) r_[i] = reads[i];
  active_ = true;
}

namespace {
void book(CaptureWindow& cw, hipGraphNode_t node, int lane, const Extent* w, int nw, const Extent* r, int nr) {
  auto touch = [&](const Extent& e) -> Touch& {
    for (Touch& t : cw.touched) if (t.e.lo == e.lo && t.e.hi == e.hi) return t;
    cw.touched.push_back(Touch()); cw.touched.back().e = e;
    return cw.touched.back();
  };
  for (int i = 0; i < nw; i++) { Touch& t = touch(w[i]); t.writer = node; for (hipGraphNode_t& x : t.reader) x = nullptr; }
  for (int i = 0; i < nr; i++) touch(r[i]).reader[lane] = node;
}
// after a call: the stream depends on every lane's last node (and on what the window opened behind), so whatever anyone records next waits for all of them
bool rejoin(CaptureWindow& cw, hipStream_t st) {
  cw.join = cw.base;
  for (int l = 0; l < cw.nlanes; l++) add_unique(cw.join, cw.lanes[l].last);
  if (hipStreamUpdateCaptureDependencies(st, cw.join.data(), cw.join.size(), hipStreamSetCaptureDependencies) == hipSuccess) return true;
  (void)hipGetLastError();
  return false;
}
}  // namespace

hipGraphNode_t IndependentCall::finish() {
  if (!active_) return nullptr;
  active_ = false;
  CaptureWindow& cw = g_windows[st_];
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  const bool ok = hipStreamGetCaptureInfo_v2(st_, &status, &id, &graph, &deps, &ndeps) == hipSuccess && status == hipStreamCaptureStatusActive && id == cw.id;
  if (!ok) { (void)hipGetLastError(); cw = CaptureWindow(); return nullptr; }   // the capture ended or failed inside the call: nothing to join
  // the call's launches were recorded one behind the other: its last node is now the stream's only dependency (nothing recorded: the set is unchanged)
  if (!(ndeps == 1 && !same_set(cw.deps, deps, ndeps))) {
    // no single new node to name (an empty call, or a call that forked internally): close the window — the stream continues behind everything recorded so far
    std::vector<hipGraphNode_t> all(deps, deps + ndeps);
    for (hipGraphNode_t b : cw.base) add_unique(all, b);
    for (int l = 0; l < cw.nlanes; l++) add_unique(all, cw.lanes[l].last);
    if (hipStreamUpdateCaptureDependencies(st_, all.data(), all.size(), hipStreamSetCaptureDependencies) != hipSuccess) (void)hipGetLastError();
    cw = CaptureWindow();
    return nullptr;
  }
  const hipGraphNode_t node = deps[0];
  if (lane_ == cw.nlanes) cw.nlanes++;
  cw.lanes[lane_].last = node;
  cw.count++;
  book(cw, node, lane_, w_, nw_, r_, nr_);
  if (!rejoin(cw, st_)) { cw = CaptureWindow(); return nullptr; }
  return node;
}
}  // namespace vpp_amd
using namespace vpp_amd;

namespace {
struct DevicePool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;   // rounded size -> block
  std::map<void*, size_t> live;               // every block handed out, with its rounded size
  size_t cached_bytes = 0;
};
std::mutex g_pools_mu;
std::map<int, DevicePool*> g_pools;
DevicePool& pool_of_current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> l(g_pools_mu);
  DevicePool*& p = g_pools[dev];
  if (!p) p = new DevicePool();   // intentionally never destroyed: the HIP runtime may already be gone at exit
  return *p;
}
std::vector<DevicePool*> all_pools() {
  std::lock_guard<std::mutex> l(g_pools_mu);
  std::vector<DevicePool*> v;
  for (auto& kv : g_pools) v.push_back(kv.second);
  return v;
}
// size classes: 256 B granules, 4 KiB above 64 KiB — per-frame buffers whose element count drifts by a few entries keep hitting the same class
inline size_t round_block(size_t bytes) { const size_t g = bytes >= (64u << 10) ? 4096 : 256; return ((bytes ? bytes : 1) + g - 1) / g * g; }
// Every device block starts kGuard bytes after its hipMalloc'd address.  The streaming stencil kernels load whole 16-byte chunks
// and let a chunk begin up to 16 bytes before an image's first addressable byte (the bytes are never used; the range check of the
// buffer descriptor covers the far end, not the near one): with the guard that read stays inside this allocation whatever lies before it
// in the address space.  include/vpp_amd.h states the same requirement for memory the caller allocates itself.
constexpr size_t kGuard = 256;
inline void* user_ptr(void* raw) { return (char*)raw + kGuard; }
inline void* raw_ptr(void* user) { return (char*)user - kGuard; }
}  // namespace

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
// vpp_malloc / vpp_free keep freed blocks in per-device, exact-size free lists instead of returning them to the driver:
// hipMalloc / hipFree cost 0.1-0.3 ms each and hipFree synchronises the device, which a per-frame caller (keypoint lists,
// result buffers, image mirrors of the same shapes every frame) would pay several times per frame.  A cached block is handed
// out again without synchronisation: callers order their work on streams, exactly as they would around a real allocation
// that returns recently freed memory.  Cap: tuning "runtime.pool_mb" (default 2048 MiB per device, 0 disables the cache).
int vpp_malloc(size_t bytes, void** dptr) {
  VPP_REQUIRE(dptr, VPP_ERR_INVALID_ARG, "vpp_malloc: null out pointer");
  const size_t rb = round_block(bytes + kGuard);
  DevicePool& P = pool_of_current_device();
  {
    std::lock_guard<std::mutex> l(P.mu);
    auto it = P.free_blocks.find(rb);
    if (it != P.free_blocks.end()) {
      *dptr = it->second;
      P.free_blocks.erase(it);
      P.cached_bytes -= rb;
      P.live[*dptr] = rb;
      return VPP_OK;
    }
  }
  void* raw = nullptr;
  hipError_t e = hipMalloc(&raw, rb);
  if (e != hipSuccess) {  // out of memory: give the cached blocks back to the driver and retry once
    (void)hipGetLastError();
    vpp_release_cached_memory();
    e = hipMalloc(&raw, rb);
  }
  if (e != hipSuccess) { set_error("vpp_malloc: hipMalloc(%zu) failed: %s", rb, hipGetErrorString(e)); return VPP_ERR_HIP; }
  *dptr = user_ptr(raw);   // the pool's maps hold user pointers throughout
  std::lock_guard<std::mutex> l(P.mu);
  P.live[*dptr] = rb;
  return VPP_OK;
}

int vpp_free(void* dptr) {
  if (!dptr) return VPP_OK;
  if (g_defer_pending.load(std::memory_order_acquire) && !g_defer_bypass) (void)defer_flush();   // a held-back call may use the block: launched before the block can be handed out again (callers order reuse on streams)
  const size_t cap = (size_t)tuning("runtime.pool_mb", 2048) << 20;
  // the block goes back to the pool of the device it was allocated on, whichever device is current now
  DevicePool* P = nullptr;
  size_t rb = 0;
  for (DevicePool* cand : all_pools()) {
    std::lock_guard<std::mutex> l(cand->mu);
    auto it = cand->live.find(dptr);
    if (it == cand->live.end()) continue;
    rb = it->second;
    cand->live.erase(it);
    if (cand->cached_bytes + rb <= cap) { cand->free_blocks.emplace(rb, dptr); cand->cached_bytes += rb; return VPP_OK; }
    P = cand;
    break;
  }
  // a block of the pool that does not fit the cache, or a pointer the pool never handed out (freed as it is); hipFree takes any device's pointer
  VPP_HIP_TRY(hipFree(P ? raw_ptr(dptr) : dptr));
  return VPP_OK;
}

// Pinned (page-locked) host staging memory, cached by size like the device blocks: a copy between HBM and pageable memory makes
// the runtime pin the host pages first, which costs ~1 ms for a range it has not seen recently; per-frame result buffers
// (keypoint lists, flow results) come from here instead.
namespace { struct HostPool { std::mutex mu; std::multimap<size_t, void*> free_blocks; std::map<void*, size_t> live; size_t cached_bytes = 0; }; HostPool g_host_pool; }

// A request is served by the smallest cached block that holds it and is at most twice its size (per-frame buffers sized by a
// drifting keypoint count keep landing in one block instead of adding a size class per frame); the cache is capped by the
// tuning "runtime.host_pool_mb" (default 512 MiB of page-locked memory), beyond which freed blocks go back to the driver.
int vpp_malloc_host(size_t bytes, void** hptr) {
  VPP_REQUIRE(hptr, VPP_ERR_INVALID_ARG, "vpp_malloc_host: null out pointer");
  const size_t rb = ((bytes ? bytes : 1) + 4095) / 4096 * 4096;
  {
    std::lock_guard<std::mutex> l(g_host_pool.mu);
    auto it = g_host_pool.free_blocks.lower_bound(rb);
    if (it != g_host_pool.free_blocks.end() && it->first <= 2 * rb) {
      *hptr = it->second;
      g_host_pool.live[*hptr] = it->first;
      g_host_pool.cached_bytes -= it->first;
      g_host_pool.free_blocks.erase(it);
      return VPP_OK;
    }
  }
  VPP_HIP_TRY(hipHostMalloc(hptr, rb, hipHostMallocDefault));
  std::lock_guard<std::mutex> l(g_host_pool.mu);
  g_host_pool.live[*hptr] = rb;
  return VPP_OK;
}

int vpp_free_host(void* hptr) {
  if (!hptr) return VPP_OK;
  const size_t cap = (size_t)tuning("runtime.host_pool_mb", 512) << 20;
  {
    std::lock_guard<std::mutex> l(g_host_pool.mu);
    auto it = g_host_pool.live.find(hptr);
    VPP_REQUIRE(it != g_host_pool.live.end(), VPP_ERR_INVALID_ARG, "vpp_free_host: not a vpp_malloc_host block");
    const size_t rb = it->second;
    g_host_pool.live.erase(it);
    if (g_host_pool.cached_bytes + rb <= cap) { g_host_pool.free_blocks.emplace(rb, hptr); g_host_pool.cached_bytes += rb; return VPP_OK; }
  }
  VPP_HIP_TRY(hipHostFree(hptr));
  return VPP_OK;
}

int vpp_release_cached_memory(void) {
  {
    std::multimap<size_t, void*> hb;
    { std::lock_guard<std::mutex> l(g_host_pool.mu); hb.swap(g_host_pool.free_blocks); g_host_pool.cached_bytes = 0; }
    for (auto& b : hb) VPP_HIP_TRY(hipHostFree(b.second));
  }
  DevicePool& P = pool_of_current_device();
  std::multimap<size_t, void*> blocks;
  {
    std::lock_guard<std::mutex> l(P.mu);
    blocks.swap(P.free_blocks);
    P.cached_bytes = 0;
  }
  for (auto& b : blocks) VPP_HIP_TRY(hipFree(raw_ptr(b.second)));
  return VPP_OK;
}
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
  return device_fill(dst, byte, bytes, as_stream(stream));   // a kernel of this library: recordable into launch graphs (see common.hpp)
}
namespace {
// this thread's sticky held-back failure, reported once
int defer_report(const char* where) {
  DeferWindow& w = defer_window();
  std::lock_guard<std::mutex> l(w.mu);
  if (w.last_rc == VPP_OK) return VPP_OK;
  const int rc = w.last_rc;
  w.last_rc = VPP_OK;
  set_error("%s: %s", where, w.last_msg);
  return rc;
}
}  // namespace
int vpp_sync(void* stream) {
  VPP_HIP_TRY(hipStreamSynchronize(as_stream(stream)));   // (as_stream launches the held-back calls first: this thread's, and any thread's on this stream)
  const int rc = defer_report("vpp_sync");
  if (rc != VPP_OK) return rc;
  return check_device_error("vpp_sync");
}
int vpp_flush(void* stream) {
  (void)defer_flush_stream(stream);   // this thread's window whatever its stream, and other threads' windows on `stream`
  return defer_report("vpp_flush");
}
unsigned long long vpp_deferred_flushes(void) { return defer_window().flushes.load(std::memory_order_acquire); }
int vpp_deferred_pending(void) { DeferWindow& w = defer_window(); std::lock_guard<std::mutex> l(w.mu); return w.b.n; }
int vpp_stream_create(void** stream) {
  VPP_REQUIRE(stream, VPP_ERR_INVALID_ARG, "vpp_stream_create: null");
  hipStream_t s;
  VPP_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return VPP_OK;
}
int vpp_stream_destroy(void* stream) {
  if (stream) VPP_HIP_TRY(hipStreamDestroy(as_stream(stream)));
  return VPP_OK;
}
int vpp_event_create(void** event) {
  VPP_REQUIRE(event, VPP_ERR_INVALID_ARG, "vpp_event_create: null");
  hipEvent_t e;
  VPP_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = (void*)e;
  return VPP_OK;
}
int vpp_event_record(void* event, void* stream) {
  VPP_REQUIRE(event, VPP_ERR_INVALID_ARG, "vpp_event_record: null");
  VPP_HIP_TRY(hipEventRecord((hipEvent_t)event, as_stream(stream)));
  return VPP_OK;
}
int vpp_event_synchronize(void* event) {
  VPP_REQUIRE(event, VPP_ERR_INVALID_ARG, "vpp_event_synchronize: null");
  VPP_HIP_TRY(hipEventSynchronize((hipEvent_t)event));
  return check_device_error("vpp_event_synchronize");
}
int vpp_event_destroy(void* event) {
  if (event) VPP_HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return VPP_OK;
}
int vpp_stream_wait_event(void* stream, void* event) {
  VPP_REQUIRE(event, VPP_ERR_INVALID_ARG, "vpp_stream_wait_event: null");
  VPP_HIP_TRY(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
  return VPP_OK;
}

// Launch graphs for C / C++ hosts: every entry point of this ABI is stream-ordered and allocation-free on its fast paths, so a
// frame loop (or K benchmark launches) can be recorded once and replayed with one submission.  With `timed`, the graph gets an
// event-record node in front of its root nodes and one behind its leaves: the pair brackets the recorded kernels on the
// device's own clock at every replay, excluding the host's submission latency.
struct vpp_graph { hipGraph_t g = nullptr; hipGraphExec_t exec = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; bool timed = false; unsigned scratch_gen = 0; };
int vpp_graph_begin(void* stream) {
  VPP_HIP_TRY(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  g_recording[stream] = recorded_scratch_generation();   // per-frame calls on this stream are held back until their window closes (common.hpp); vpp_graph_end closes the last one
  return VPP_OK;
}
int vpp_graph_end(void* stream, int timed, vpp_graph** out) {
  VPP_REQUIRE(out, VPP_ERR_INVALID_ARG, "vpp_graph_end: null");
  vpp_graph* gr = new vpp_graph();
  hipStream_t st = as_stream(stream);   // the last held-back window is recorded here
  // The generation the capture STARTED under: an eager call of this thread on another stream may evict (free) a recorded buffer while this capture is open — the
  // graph then holds a freed address and must be refused, which stamping the generation current at the end would hide.
  auto rec = g_recording.find(stream);
  gr->scratch_gen = rec != g_recording.end() ? rec->second : recorded_scratch_generation();
  if (rec != g_recording.end()) g_recording.erase(rec);
  hipError_t e = hipStreamEndCapture(st, &gr->g);
  if (e != hipSuccess || !gr->g) { delete gr; set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return VPP_ERR_HIP; }
  if (timed) {
    size_t nn = 0, ne = 0;
    bool ok = hipGraphGetNodes(gr->g, nullptr, &nn) == hipSuccess && nn > 0 && hipGraphGetEdges(gr->g, nullptr, nullptr, &ne) == hipSuccess;
    std::vector<hipGraphNode_t> nodes(nn), from(ne), to(ne);
    ok = ok && hipGraphGetNodes(gr->g, nodes.data(), &nn) == hipSuccess && (ne == 0 || hipGraphGetEdges(gr->g, from.data(), to.data(), &ne) == hipSuccess);
    // device-scope release: the default (system-scope) record writes the L2 back before it takes its timestamp, which lands inside the bracket
    ok = ok && hipEventCreateWithFlags(&gr->e0, hipEventReleaseToDevice) == hipSuccess && hipEventCreateWithFlags(&gr->e1, hipEventReleaseToDevice) == hipSuccess;
    if (ok) {
      std::vector<hipGraphNode_t> roots, leaves;
      for (hipGraphNode_t n : nodes) {
        bool has_in = false, has_out = false;
        for (size_t k = 0; k < ne; k++) { has_in |= to[k] == n; has_out |= from[k] == n; }
        if (!has_in) roots.push_back(n);
        if (!has_out) leaves.push_back(n);
      }
      hipGraphNode_t n0 = nullptr, n1 = nullptr;
      ok = hipGraphAddEventRecordNode(&n0, gr->g, nullptr, 0, gr->e0) == hipSuccess;
      for (hipGraphNode_t r : roots) ok = ok && hipGraphAddDependencies(gr->g, &n0, &r, 1) == hipSuccess;
      ok = ok && hipGraphAddEventRecordNode(&n1, gr->g, leaves.data(), leaves.size(), gr->e1) == hipSuccess;
    }
    (void)hipGetLastError();
    if (!ok) { set_error("vpp_graph_end: event-record nodes are not supported by this runtime"); (void)hipGraphDestroy(gr->g); if (gr->e0) (void)hipEventDestroy(gr->e0); if (gr->e1) (void)hipEventDestroy(gr->e1); delete gr; return VPP_ERR_UNSUPPORTED; }
    gr->timed = true;
  }
  e = hipGraphInstantiate(&gr->exec, gr->g, nullptr, nullptr, 0);
  if (e != hipSuccess) { set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); (void)hipGraphDestroy(gr->g); delete gr; return VPP_ERR_HIP; }
  *out = gr;
  return VPP_OK;
}
// diagnostics (not part of include/vpp_amd.h): raise bits of the sticky device error word from the host, as a kernel whose protocol gave up would (tests)
int vpp_debug_raise_device_error(unsigned bits) {
  unsigned* w = device_error_word();
  VPP_REQUIRE(w, VPP_ERR_HIP, "vpp_debug_raise_device_error: no error word");
  __atomic_fetch_or(w, bits, __ATOMIC_RELEASE);
  return VPP_OK;
}
// diagnostics (not part of include/vpp_amd.h): kernel nodes of a recorded graph — how many launches a replay makes (record-time batching, tests)
int vpp_debug_graph_kernel_nodes(vpp_graph* gr, int* count) {
  VPP_REQUIRE(gr && gr->g && count, VPP_ERR_INVALID_ARG, "vpp_debug_graph_kernel_nodes: null");
  size_t nn = 0;
  VPP_HIP_TRY(hipGraphGetNodes(gr->g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  if (nn) VPP_HIP_TRY(hipGraphGetNodes(gr->g, nodes.data(), &nn));
  int k = 0;
  for (hipGraphNode_t n : nodes) { hipGraphNodeType t; if (hipGraphNodeGetType(n, &t) == hipSuccess && t == hipGraphNodeTypeKernel) k++; }
  *count = k;
  return VPP_OK;
}
int vpp_graph_launch(vpp_graph* gr, void* stream) {
  VPP_REQUIRE(gr && gr->exec, VPP_ERR_INVALID_ARG, "vpp_graph_launch: null");
  VPP_REQUIRE(gr->scratch_gen == recorded_scratch_generation(), VPP_ERR_INVALID_ARG,
              "vpp_graph_launch: a scratch buffer that launch graphs were recorded on has been evicted since this graph was recorded (more than 16 streams in use on "
              "one host thread): the graph may hold a freed address - record it again");
  VPP_HIP_TRY(hipGraphLaunch(gr->exec, as_stream(stream)));
  return VPP_OK;
}
int vpp_graph_elapsed_ms(vpp_graph* gr, float* ms) {  // of the last completed replay; waits for it
  VPP_REQUIRE(gr && ms && gr->timed, VPP_ERR_INVALID_ARG, "vpp_graph_elapsed_ms: not a timed graph");
  VPP_HIP_TRY(hipEventSynchronize(gr->e1));
  VPP_HIP_TRY(hipEventElapsedTime(ms, gr->e0, gr->e1));
  return VPP_OK;
}
int vpp_graph_destroy(vpp_graph* gr) {
  if (!gr) return VPP_OK;
  if (gr->exec) (void)hipGraphExecDestroy(gr->exec);
  if (gr->g) (void)hipGraphDestroy(gr->g);
  if (gr->e0) (void)hipEventDestroy(gr->e0);
  if (gr->e1) (void)hipEventDestroy(gr->e1);
  delete gr;
  return VPP_OK;
}

// Stream gate: a one-thread kernel that holds `stream` until the host opens the gate, so that a batch of launches (or a
// graph replay and the events around it) can be queued completely before the first of them starts — the events then
// bracket the kernels themselves, not the host's submission latency.  The wait gives up after ~50 ms so that a gate that
// is never opened cannot hang the device.
struct vpp_gate { volatile uint32_t* flag; };
__global__ void gate_wait_kernel(volatile uint32_t* flag) {
  const unsigned long long t0 = wall_clock64();       // constant 100 MHz counter
  while (__hip_atomic_load((const uint32_t*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
    if (wall_clock64() - t0 > 5000000ull) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
int vpp_gate_create(vpp_gate** gate) {
  VPP_REQUIRE(gate, VPP_ERR_INVALID_ARG, "vpp_gate_create: null");
  void* p = nullptr;
  VPP_HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *(volatile uint32_t*)p = 0u;
  *gate = new vpp_gate{(volatile uint32_t*)p};
  return VPP_OK;
}
int vpp_gate_wait(vpp_gate* gate, void* stream) {   // closes the gate and queues the wait on `stream`
  VPP_REQUIRE(gate, VPP_ERR_INVALID_ARG, "vpp_gate_wait: null");
  *gate->flag = 0u;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  gate_wait_kernel<<<1, 1, 0, as_stream(stream)>>>(gate->flag);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
int vpp_gate_open(vpp_gate* gate) {
  VPP_REQUIRE(gate, VPP_ERR_INVALID_ARG, "vpp_gate_open: null");
  __atomic_store_n((uint32_t*)gate->flag, 1u, __ATOMIC_SEQ_CST);
  return VPP_OK;
}
int vpp_gate_destroy(vpp_gate* gate) {
  if (!gate) return VPP_OK;
  (void)hipHostFree((void*)gate->flag);
  delete gate;
  return VPP_OK;
}

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
