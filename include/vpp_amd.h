/* vpp_amd.h — C ABI of the MI355X (gfx950) evaluation engine behind the Video++ (matt-42/vpp) API.
 *
 * The reference is a header-only C++14 template library with NO plugin/FFI layer.
 * The drop-in boundary is therefore the set of C++ entry points listed next to each function below
 * (paths relative to the reference tree); the vpp-shaped C++ headers in vpp_amd/include/vpp/ call these
 * functions from exactly those entry points.  Signatures use plain pointers and sizes only.
 *
 * Conventions
 *  - coordinates are (row, col); boxes inclusive (vpp/core/boxNd.hh:58-62)
 *  - every image argument is a BORROWED descriptor of DEVICE memory laid out as imageNd::allocate does
 *    (vpp/core/imageNd.hpp:151-196): `first_pixel` is the address of pixel (0,0); `border` pixels of valid,
 *    addressable memory surround the domain on every side; rows are `pitch` bytes apart.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls enqueue work and return;
 *    functions that hand results back to HOST memory synchronise the stream themselves (documented per call).
 *  - every function returns a vpp_status; nothing throws across this boundary.  vpp_last_error() returns
 *    a thread-local description of the last non-OK status.
 */
#ifndef VPP_AMD_H_
#define VPP_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vpp_status {
  VPP_OK = 0,
  VPP_ERR_INVALID_ARG = 1,
  VPP_ERR_BORDER_TOO_SMALL = 2, /* e.g. fast9 needs border >= 3 (vpp/algorithms/fast_detector/fast.hpp:937-938) */
  VPP_ERR_HIP = 3,
  VPP_ERR_UNSUPPORTED = 4,
  VPP_ERR_CAPACITY = 5          /* caller-provided output buffer too small; required size reported */
} vpp_status;

typedef enum vpp_dtype { VPP_U8 = 0, VPP_I8 = 1, VPP_U16 = 2, VPP_I16 = 3, VPP_I32 = 4, VPP_U32 = 5, VPP_F32 = 6 } vpp_dtype;

/* image2d<V> as seen by the device (vpp/core/imageNd.hh:17-40).  V = vector<dtype, channels>.
 * Memory contract for images in memory the caller allocated itself: the addressable area is rows -border .. nrows + border - 1 of the
 * pitch, as imageNd::allocate lays it out (imageNd.hpp:151-196).  The streaming stencil kernels load whole ALIGNED 16-byte granules: a
 * granule that holds an addressable byte may extend up to 15 bytes in front of the area's first byte or past its last one.  Such bytes lie
 * in the same page as an addressable byte, so the load cannot fault, and they are never used; nothing else outside the area is touched. */
typedef struct vpp_image_desc {
  void*   first_pixel; /* imageNd_data::begin_ */
  int32_t nrows;       /* domain().nrows() */
  int32_t ncols;       /* domain().ncols() */
  int32_t pitch;       /* bytes between rows, imageNd_data::pitch_ */
  int32_t border;      /* pixels, imageNd_data::border_ */
  int32_t dtype;       /* vpp_dtype of one component */
  int32_t channels;    /* 1 for scalars, N for vector<T,N> */
} vpp_image_desc;

/* ---- runtime (no reference counterpart: device residency is new; host code keeps vpp's shared_ptr ownership,
 *      vpp/core/imageNd.hpp:177-180, and hangs the device mirror's deleter beside it) ---- */
int vpp_init(int device);                       /* hipSetDevice + warm the context */
int vpp_device_count(int* n);
int vpp_malloc(size_t bytes, void** dptr);        /* freed blocks are cached per device and reused by size (no hipFree sync per frame); 256 B of slack precede every block */
int vpp_free(void* dptr);
int vpp_malloc_host(size_t bytes, void** hptr);   /* pinned host staging memory (hipHostMalloc), cached by size on vpp_free_host */
int vpp_free_host(void* hptr);
int vpp_release_cached_memory(void);             /* give everything vpp_free / vpp_free_host are holding back to the driver */
int vpp_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int vpp_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int vpp_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int vpp_memset(void* dst, int byte, size_t bytes, void* stream);
int vpp_sync(void* stream);
/* A stream of the current device for hosts that have none of their own (every entry point takes `void* stream` = a hipStream_t, NULL =
 * the null stream): launch graphs are recorded on one, and independent frame pairs overlap on several. */
int vpp_stream_create(void** stream);
int vpp_stream_destroy(void* stream);
/* Completion events (no reference counterpart: the reference's calls are synchronous, vpp/core/pixel_wise.hpp:146-165 joins its OpenMP loop before it
 * returns).  The C++ drop-in surface uses them to bound how many of its calls are queued (vpp/core/device.hh: call_done) instead of draining the
 * GPU after every call; vpp_stream_wait_event orders one stream behind work recorded on another. */
int vpp_event_create(void** event);
int vpp_event_record(void* event, void* stream);
int vpp_event_synchronize(void* event);          /* returns once the work recorded before the event has completed (at once if never recorded) */
int vpp_event_destroy(void* event);
int vpp_stream_wait_event(void* stream, void* event);
/* Launch graphs (no reference counterpart: the reference's frame loops call the algorithms directly, e.g.
 * examples/video_extruder.cc:40-60; on a stream the per-launch host cost is what a graph removes).  Everything queued on
 * `stream` between vpp_graph_begin and vpp_graph_end is recorded instead of run; vpp_graph_launch replays it with one
 * submission.  timed != 0 adds an event-record node in front of the first and behind the last recorded node (the reference
 * times with clock_gettime around host calls, benchmarks/get_time.hh:2-7): vpp_graph_elapsed_ms then returns the device-clock
 * duration of the last replay.  VPP_ERR_UNSUPPORTED when the runtime has no event-record nodes (retry with timed = 0).
 * Scratch rule: entry points that need device scratch (FAST-9, the flow, the tracker, local maxima) keep one grow-only buffer per (host thread, stream), and a
 * recorded call bakes that buffer's ADDRESS into the graph.  So (a) run a call once eagerly on the stream before recording it (a capture cannot allocate:
 * VPP_ERR_UNSUPPORTED says so); (b) a later eager call on that stream that needs MORE scratch gets a new buffer and the recorded one stays alive for its graphs
 * (until the host thread ends); (c) only a 17th stream on one host thread, which evicts the least recently used buffer, frees a recorded buffer: every graph
 * recorded before that is then refused by vpp_graph_launch (VPP_ERR_INVALID_ARG, "record it again") instead of replaying into freed memory.
 * Device-side faults: a kernel whose in-kernel protocol gives up (the flow's grid barrier after ~4 s) raises a bit in ONE process-global sticky word; the first
 * vpp_sync / vpp_event_synchronize on ANY stream afterwards returns VPP_ERR_HIP once and clears it — the fault is reported to whoever synchronises first, not to
 * the stream that raised it, and never to a caller that only synchronises outside this ABI. */
typedef struct vpp_graph vpp_graph;
int vpp_graph_begin(void* stream);
int vpp_graph_end(void* stream, int timed, vpp_graph** graph);
int vpp_graph_launch(vpp_graph* graph, void* stream);
int vpp_graph_elapsed_ms(vpp_graph* graph, float* ms);
int vpp_graph_destroy(vpp_graph* graph);
/* Stream gate (no reference counterpart; batching / measurement aid): vpp_gate_wait queues a one-thread kernel that holds `stream`
 * until vpp_gate_open is called from the host (or ~50 ms pass), so that a batch of launches can be queued completely before the
 * first of them starts. */
typedef struct vpp_gate vpp_gate;
int vpp_gate_create(vpp_gate** gate);
int vpp_gate_wait(vpp_gate* gate, void* stream);
int vpp_gate_open(vpp_gate* gate);
int vpp_gate_destroy(vpp_gate* gate);
const char* vpp_last_error(void);
const char* vpp_version(void);
/* runtime tuning knob (launch geometry variants; used by bench/tuning scripts, never changes results) */
int vpp_set_tuning(const char* name, int value);

/* imageNd::allocate arithmetic (vpp/core/imageNd.hpp:151-196), host-only helper: pitch, allocation size
 * (excluding the `align` slack) and byte offset of pixel (0,0) from the aligned buffer start. */
int vpp_image_layout(int nrows, int ncols, int elem_bytes, int border, int align,
                     int32_t* pitch, size_t* alloc_bytes, size_t* first_pixel_offset);

/* ---- pixel_wise (vpp/core/pixel_wise.hpp:68-105,146-165 with an arithmetic kernel lambda, e.g.
 *      benchmarks/image_add.cc:53-56 `a = b + c`) ---- */
typedef enum vpp_binary_op { VPP_OP_ADD = 0, VPP_OP_SUB = 1, VPP_OP_MUL = 2, VPP_OP_MIN = 3, VPP_OP_MAX = 4, VPP_OP_ABSDIFF = 5 } vpp_binary_op;
/* dst(p) = V(a(p) op b(p)) over dst's domain; arithmetic in the C++ promoted type of V's component then
 * converted back (so u8 wraps modulo 256 and i32 wraps modulo 2^32, as the compiled reference does). */
int vpp_pixelwise_binary(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, void* stream);
/* dst[k] = a[k] op b[k] for n image triples of one size in ONE launch (see vpp_box_filter_batch); same results as n single calls. */
int vpp_pixelwise_binary_batch(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, int n, void* stream);
/* copy(src,dst) (vpp/core/copy.hh:10-20); with_border=1 is copy_with_border (copy.hh:22-27): copies src's border too. */
int vpp_copy(const vpp_image_desc* dst, const vpp_image_desc* src, int with_border, void* stream);
/* fill (vpp/core/fill.hh:12-16) / fill_with_border (fill.hh:24-29): `value` points to one pixel (host memory). */
int vpp_fill(const vpp_image_desc* img, const void* value, int with_border, void* stream);

/* ---- neighbourhood access: box_nbh2d<V,R,C> / relative_access (vpp/core/pixel_wise.hpp:14-25,57-63;
 *      vpp/core/relative_accessor.hh:26-33) with the R x C mean kernel of benchmarks/box_5x5_filter2.cc:73-80
 *      and examples/box_filter.cc:23-32: per component, sum over the window in the promoted type (taps in
 *      row-major order), C++ `/ (R*C)`, cast back.  Reads src's border (needs border >= max(R,C)/2). ---- */
int vpp_box_filter(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, void* stream);
/* The same filter over n frame pairs dst[k] <- src[k] of one geometry (a decoded group of frames, the levels of a frame ring) in ONE launch:
 * the chip does not drain between frames and the host submits once (no reference counterpart: the reference's frame loops call the
 * filter per frame, benchmarks/box_5x5_filter2.cc:43-69).  Results are those of n vpp_box_filter calls; shapes the batched kernel does not
 * serve go out as exactly those calls. */
int vpp_box_filter_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, int R, int C, void* stream);

/* ---- deferred per-frame calls: the reference's call form (ONE frame per call: benchmarks/box_5x5_filter2.cc:43-81, benchmarks/image_add.cc:51-57,
 *      examples/video_extruder.cc:44-48; vpp/core/pixel_wise.hpp:188-213 evaluates at operator|) at the batched kernels' rate, without a launch graph ----
 * A *_deferred entry point validates its arguments as its plain namesake does and returns VPP_OK; its launch is held back in a window of the CALLING HOST THREAD and
 * goes out together with later deferred calls of the same entry point, parameters, stream and geometry as ONE batched launch (vpp_*_batch: "the results of the n
 * calls one after the other").  A call joins the window only while no data flows between it and the pending calls (no pending result overlaps its operands or result,
 * its result overlaps no pending operand; bordered extents); otherwise the window is launched first.  The window is also launched when it holds 64 frames, by
 * vpp_flush, when its thread ends, and before ANYTHING else is queued through this ABI by the same thread (on any stream) or by ANY host thread on the window's
 * stream — every entry point that takes a stream (copies, events, vpp_sync, graphs, every kernel) and vpp_free — so stream order, results and what a vpp_sync waits
 * for are exactly those of the plain calls, also when one host thread makes the calls and another one (which the first has handed the stream to) synchronises.
 * What is NOT covered: work queued on the stream by other means than this ABI (raw HIP calls, another library) — call vpp_flush before such work, and before an
 * image's memory is released by any other means than vpp_free.  A window belongs to the device that was current when it opened; it is launched there whatever device
 * is current later.  Frames the batched kernels do not serve run as the plain call at once.
 * Recorded streams: on a stream that the calling thread records through vpp_graph_begin, the PLAIN per-frame entry points (vpp_box_filter, vpp_pixelwise_binary,
 * vpp_rgb_to_graylevel) hold their frames back in the same window, and a window that closes (as above, or at vpp_graph_end) records ONE node of the batched
 * kernel: a recorded frame loop replays as batched launches.  On a stream captured by other means (hipStreamBeginCapture by the caller) every call records its own
 * node at once, deferred or not — the library cannot see that capture end.
 * The C++ drop-in surface (vpp/core/pixel_wise.hh: ops::box_mean, ops::add/sub; colorspace_conversions.hh) calls these (its `_immediate` option selects the plain
 * call): a frame loop written like the reference's reaches the batch rate (4K vuchar3 box5x5: 8.3 us per frame instead of 13.5).
 * Errors: argument errors are reported at the call.  A window that fails to LAUNCH is never reported as the failure of the call that happened to close it: the
 * status, the entry point and the number of dropped frames are kept and returned (once) by the owning thread's next vpp_flush / vpp_sync. */
int vpp_box_filter_deferred(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, void* stream);
int vpp_pixelwise_binary_deferred(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, void* stream);
int vpp_rgb_to_graylevel_deferred(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror, void* stream);
int vpp_flush(void* stream);                       /* launches the calling thread's window (whatever its stream) and any thread's window that waits on `stream`; VPP_OK when there is none */
unsigned long long vpp_deferred_flushes(void);     /* batches launched for the calling thread so far (the C++ surface throttles per batch, not per call) */
int vpp_deferred_pending(void);                    /* calls held back in the calling thread's window right now */

/* ---- borders (vpp/core/fill.hh:31-122) ---- */
typedef enum vpp_border_mode { VPP_BORDER_MIRROR = 0, VPP_BORDER_CLOSEST = 1, VPP_BORDER_VALUE = 2 } vpp_border_mode;
int vpp_fill_border(const vpp_image_desc* img, int mode, const void* value /* VALUE mode: one pixel, host */, void* stream);

/* ---- pyramid (vpp/core/pyramid.hh:12-81,169-192): one level step of propagate_level0 for factor 2:
 *      next = subsample2(antialiasing_lowpass_filter(prev)); fill_border_mirror(next).
 *      prev must have border >= 2 already filled.  next dims must be (1+nr/2, 1+nc/2).
 *      The reference's temporaries are uninitialised heap; the canonical value is 0. ---- */
int vpp_pyr_down(const vpp_image_desc* next, const vpp_image_desc* prev, void* stream);
/* A whole pyramid: pyramid2d<V>(src, nlevels, 2, _border = levels[0].border) (pyramid.hh:146-198) = copy src's domain into
 * level 0, fill_border_mirror, propagate_level0 — for u8 x1 pyramids of 2 or 3 levels ONE launch (a workgroup owns a tile of the
 * coarsest level and computes everything underneath it in LDS), otherwise the per-level kernels.  Bit-identical either way. */
int vpp_pyramid_build(const vpp_image_desc* levels, int nlevels, const vpp_image_desc* src, void* stream);
/* The gradient pyramid of pyrlk_match / lucas_kanade in one launch: scharr(img, grad[0]) (scharr.hh:46-87), fill_border_mirror,
 * propagate_level0 (pyrlk_opencv_comparison.cc:56-60, lucas_kanade.hpp:151-157).  img: u8 x1, filled border >= 1. */
int vpp_scharr_pyramid_build(const vpp_image_desc* grad_levels, int nlevels, const vpp_image_desc* img, void* stream);
/* antialiasing_lowpass_filter alone (pyramid.hh:12-59); out may have any border (left untouched). */
int vpp_lowpass5(const vpp_image_desc* out, const vpp_image_desc* in, void* stream);

/* ---- scharr(in, out) (vpp/algorithms/filters/scharr.hh:46-87): in u8 x1 (border>=1), out 2 channels f32 or i32 ---- */
int vpp_scharr(const vpp_image_desc* out, const vpp_image_desc* in, void* stream);

/* ---- FAST-9 (vpp/algorithms/fast_detector/fast.hpp:254-508 detector, :38-77 score, :676-707 maxima,
 *      :745-799 blockwise, :889-928 local maxima, :931-955 front end `fast9`) ---- */
typedef enum vpp_fast9_mode { VPP_FAST9_RAW = 0, VPP_FAST9_LOCAL_MAXIMA = 1, VPP_FAST9_BLOCKWISE = 2 } vpp_fast9_mode;
typedef enum vpp_fast9_compat {
  VPP_FAST9_REFERENCE = 0, /* ring samples 4 and 12 taken from row r-3 as fast.hpp:367-368 does */
  VPP_FAST9_CORRECTED = 1  /* true Bresenham ring, = is_fast9_keypoint (fast.hpp:80-112) */
} vpp_fast9_compat;
/* Detect on `src` (u8 x1, border >= 3 else VPP_ERR_BORDER_TOO_SMALL).  `mask` may be NULL (fast.hpp:317) else u8 x1
 * same domain, AND-ed bitwise with the plane byte (0x10 brighter | 0x01 darker; fast.hpp:120-126,312,333).
 * Writes up to `capacity` keypoints as (row,col) int32 pairs in ROW-MAJOR order into device buffer out_rc and
 * (if non-NULL) scores into device buffer out_scores: RAW -> full fast9_score; maxima modes -> score/16 as
 * stored in the uchar score image (fast.hpp:693,698-704).  *count (HOST int) receives the number found;
 * synchronises the stream.  count > capacity => VPP_ERR_CAPACITY (first `capacity` entries valid). */
int vpp_fast9_detect(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size,
                     int compat, int32_t* out_rc, int32_t* out_scores, int capacity, int* count, void* stream);
/* The same detection without the host round trip: everything is queued on `stream` and the number of keypoints found (not clamped to
 * `capacity`; entries beyond it are not written) lands in *count_dev, a device-visible 32-bit word (HBM or vpp_malloc_host memory), for a
 * consumer on the same stream — or for the host after its own synchronisation.  Can be recorded into a launch graph. */
int vpp_fast9_detect_async(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size,
                           int compat, int32_t* out_rc, int32_t* out_scores, int capacity, uint32_t* count_dev, void* stream);
/* fast9_scores (fast.hpp:643-652): n (row,col) pairs in device memory -> n int32 full scores. */
int vpp_fast9_scores(const vpp_image_desc* src, int th, const int32_t* rc, int n, int32_t* out_scores, void* stream);
/* FAST_internals::fast_detector9(A, B, th) (fast.hpp:511-551): dst(r,c) = 1 where 9 contiguous pixels of the TRUE 16-pixel
 * ring are all > src + th or all < src - th (plain int compares, fast9_check_code fast.hpp:25-35), else 0.  src u8 x1 with
 * border >= 3, dst u8 or int32 x1, same domain. */
int vpp_fast9_dense(const vpp_image_desc* dst, const vpp_image_desc* src, int th, void* stream);
/* blockwise_maxima_filter(A, block_size) (fast.hpp:577-614), in place on a scalar image: per block_size x block_size block
 * (clipped to the domain) only the first strict maximum > 0 in row-major order survives, every other pixel becomes 0. */
int vpp_blockwise_maxima_filter(const vpp_image_desc* img, int block_size, void* stream);
/* local_maxima_filter(A, nbh_size) (fast.hpp:555-575; nbh_size is ignored there), in place on a scalar image with border >= 1: a
 * pixel that is not strictly greater than its 8 neighbours becomes 0, the neighbours above / left of it being compared AFTER their
 * own filtering — the result of the reference's serial (no-OpenMP) build; its OpenMP build races on exactly those neighbours.
 * Synchronises the stream (the number of order-dependent pixels is read back). */
int vpp_local_maxima_filter(const vpp_image_desc* img, void* stream);
/* The score cull of video_extruder_update (video_extruder/video_extruder.hpp:44-56,87-91) queued behind the flow instead of
 * after a host round trip: keypoint i is scored at rc_moved[i] when that lies inside src's domain (the match callback moved it
 * there), else at rc_prev[i] (the callback removed it, or never ran: its position is unchanged).  All three arrays are
 * device-visible (HBM or vpp_malloc_host memory); no synchronisation. */
int vpp_fast9_scores_moved(const vpp_image_desc* src, int th, const int32_t* rc_moved, const int32_t* rc_prev, int n,
                           int32_t* out_scores, void* stream);

/* ---- pyramidal Lucas-Kanade ----
 * vpp_pyrlk_match = pyrlk_match (vpp/algorithms/pyrlk/pyrlk_match.hh:15-55) with matcher
 *   lk_match_point_square_win<winsize> (vpp/algorithms/pyrlk/lk.hh:43-175).
 * prev/next: nlevels u8x1 images; grad: nlevels f32x2 images (scharr of level 0 propagated, as
 *   benchmarks/pyrlk_opencv_comparison.cc:56-60 builds them).  All levels need border >= 1 + (their reads).
 * kps: device array of n records {pos_r,pos_c,vel_r,vel_c (f32), age (i32)} = keypoint<float>
 *   (vpp/core/keypoint_container.hh:13-25); updated IN PLACE exactly as keypoints.move / remove do
 *   (keypoint_container.hpp:136-167): moved => velocity=new-old, position=new, age++ ; removed => age=0.
 *   Records with age<=0 are skipped (kp.alive()).  out_dist (optional, device, n floats) = last level error. */
typedef struct vpp_keypoint_f32 { float pos_r, pos_c, vel_r, vel_c; int32_t age; } vpp_keypoint_f32;
int vpp_pyrlk_match(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                    vpp_keypoint_f32* kps, int n, int winsize, float min_ev, float max_err, int max_iterations,
                    float convergence_delta, int min_scale, float* out_dist, void* stream);
/* F independent frame pairs in ONE launch (no reference counterpart: the reference matches one pair per call, pyrlk_match.hh:15-55; a launch of a few thousand
 * keypoints is bound by the latency of ONE keypoint's levels x iterations chain — 1 250 keypoints cost what 5 000 do — so a caller with several pairs in hand, e.g.
 * a rank's share of a keypoint-sharded job over a group of frames, gets near-linear throughput from batching them).  prev / grad / next: nframes * nlevels
 * descriptors, frame-major (frame f's level l at [f * nlevels + l]); kps[f]: frame f's n[f] records, updated in place; out_dist: NULL, or per frame NULL / n[f]
 * floats.  Results per frame are exactly those of vpp_pyrlk_match on that frame.  Frames of one geometry (per level: sizes, pitches, borders) share a launch,
 * up to 16 pairs (64 frame-levels) each; anything else — other geometries, windows > 11 — goes out as the nframes calls. */
int vpp_pyrlk_match_batch(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nframes, int nlevels,
                          vpp_keypoint_f32* const* kps, const int* n, int winsize, float min_ev, float max_err, int max_iterations,
                          float convergence_delta, int min_scale, float* const* out_dist, void* stream);
/* vpp_lucas_kanade = the per-keypoint loop of lucas_kanade (vpp/algorithms/lucas_kanade/lucas_kanade.hpp:159-183)
 * over pyramids the caller built (u8x1 images, i32x2 gradients; :150-157).  min_ev/delta are the already-truncated
 * ints of :143-144.  pts: n (row,col) f32 pairs; prediction: n (row,col) f32 pairs or NULL (= 0);
 * out_flow: n f32 pairs; out_dist: n f32. */
int vpp_lucas_kanade(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                     const float* pts, const float* prediction, int n, int winsize, int min_ev, int niterations,
                     int delta, float* out_flow, float* out_dist, void* stream);

/* ---- semi-dense optical flow (vpp/algorithms/optical_flow/semi_dense_optical_flow.hpp:48-214 with
 *      gradient_descent_match, gradient_descent.hh:10-89; epipolar options unsupported) ----
 * i1/i2: u8x1 frames (any border; pyramids with border 2*winsize are built internally, :70-73).
 * kps: n (row,col) int32 pairs (device).  Outputs (device): out_pos n int32 pairs, out_dist n int32, out_valid n
 * u8 (1 where match_callback would have fired, :205-212).  Serial-order semantics. */
int vpp_semi_dense_optical_flow(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n,
                                int winsize, int nscales, int min_scale, int propagation, int patchsize,
                                int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid, void* stream);
/* The same call over pyramids the caller already holds (a video loop builds each frame's pyramid once and uses it twice): pyr1 / pyr2 = nscales
 * u8 x1 levels each, level l of (1 + n / 2)-halved size (pyramid.hh:154), with a mirror-filled border of winsize / 2 pixels at least — what
 * vpp_pyramid_build / vpp_rgb_pyramid_build leave.  Results identical to vpp_semi_dense_optical_flow on the levels 0. */
int vpp_semi_dense_optical_flow_pyramids(const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, int nscales, const int32_t* kps, int n,
                                         int winsize, int min_scale, int propagation, int patchsize,
                                         int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid, void* stream);
/* The same call with its per-keypoint phases (claim, descent) sharded by row strips of the flow maps — the decomposition of SURVEY 8e
 * bullet 2: every strip has private maps and its own stream, the strips' rows are gathered into the owner's maps, the owner runs the
 * Jacobi pre-passes and the ordered sweeps, and the swept maps are broadcast back as the next scale's prediction.  One process, one GPU:
 * the exchanges are device copies; across GPUs they are RCCL gathers / broadcasts of the same rows.  Results identical to nstrips = 1. */
int vpp_semi_dense_optical_flow_strips(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n,
                                       int winsize, int nscales, int min_scale, int propagation, int patchsize, int nstrips,
                                       int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid, void* stream);

/* ---- the steps either side of the algorithms in the reference's video loop (examples/video_extruder.cc:44-58) ---- */
/* rgb_to_graylevel (vpp/core/colorspace_conversions.hh:10-33; 4-channel input :36-48 ignores the 4th component):
 * dst (u8 x1) = (c0 + c1 + c2) / 3, integer.  mirror == 0: the reference call — mapped over the domain extended by
 * min(dst.border, src.border) (the reference maps domain_with_border of equal-border images).  mirror != 0: frame
 * ingest — dst's WHOLE border is written with the gray value of the mirrored source pixel, i.e. the result of
 * `clone(frame, _border = b); fill_border_mirror(frame); rgb_to_graylevel<uchar>(frame)` in one pass; src's border
 * is not read (it may be 0). */
int vpp_rgb_to_graylevel(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror, void* stream);
/* n frames of one geometry in ONE launch (a 33 MB launch alone pays its ramp and drain: 54 % of the HBM peak); results are those of the n calls in sequence —
 * mixed geometries and frames that feed each other go out as the n calls.  Per-frame calls recorded into a launch graph fold into such launches by themselves. */
int vpp_rgb_to_graylevel_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, int mirror, void* stream);
/* The ingest fused with the image pyramid it feeds (examples/video_extruder.cc:46-48 followed by pyramid<uchar>::update, vpp/core/pyramid.hh:
 * 169-198): levels[0] (u8 x1, any border <= its extents) = rgb_to_graylevel<uchar>(rgb) with a mirror-filled border, levels[1..] = propagate_level0.
 * Bit-identical to vpp_rgb_to_graylevel(gray, rgb, 1) + vpp_pyramid_build(levels, nlevels, gray); one launch for 2 or 3 levels. */
int vpp_rgb_pyramid_build(const vpp_image_desc* levels, int nlevels, const vpp_image_desc* rgb, void* stream);
/* Re-detection mask of video_extruder (video_extruder/video_extruder.hpp:95-110): mask (u8 x1) = 1 over its domain
 * with border, then 0 over [r - spacing, r + spacing) x [c - spacing, c + spacing) for each of the n (row, col) int32
 * pairs in DEVICE memory (clipped to the mask's border). */
int vpp_keypoint_mask(const vpp_image_desc* mask, const int32_t* rc, int n, int spacing, void* stream);
/* The merge step of video_extruder_update (video_extruder/video_extruder.hpp:60-84), queued behind the flow like
 * vpp_fast9_scores_moved.  Keypoint i sits at rc_moved[i] with age age_prev[i] + 1 when matched[i] != 0 and rc_moved[i] lies
 * inside the nrows x ncols frame (move, :51); at rc_prev[i] with age 0 when matched but outside (remove, :52); at rc_prev[i]
 * with age_prev[i] when unmatched.  removed[i] = 1 iff the reference's serial loop over i (one champion per
 * spacing x spacing cell; a strictly older newcomer evicts the champion, a strictly younger one is removed, a tie keeps both
 * and the champion) calls remove(i).  All arrays device-visible; positions must be inside the frame; no synchronisation. */
int vpp_keypoint_merge(const int32_t* rc_moved, const int32_t* rc_prev, const uint8_t* matched, const int32_t* age_prev, int n,
                       int nrows, int ncols, int spacing, uint8_t* removed, void* stream);

/* lbp_transform (vpp/algorithms/lbp/lbp_transform.hh:6-38): out(r,c) bit k = (k-th neighbour > centre), neighbours in
 * row-major order without the centre; u8 x1 -> u8 x1, in needs border >= 1. */
int vpp_lbp_transform(const vpp_image_desc* out, const vpp_image_desc* in, void* stream);

/* ---- video_extruder with its state resident in HBM (vpp/algorithms/video_extruder/video_extruder.hpp:24-135; SURVEY 8b
 *      `video_extruder_step`, 8f row 2).  vpp_video_extruder_create = video_extruder_init(domain) (:14-20); one
 *      vpp_video_extruder_step = one video_extruder_update(ctx, frame1, frame2, options...) with the options of :35-41 in `params`
 *      (defaults 10, 10, 5, 15, 3, 9, 2).  Keypoints (position, velocity, age: keypoint_container.hh:13-25, dead entries kept in
 *      place until the next compaction exactly as keypoint_container does) and trajectories (keypoint_trajectory.hh:11-72, as
 *      rings of trajectory_capacity + 1 slots, newest first from `head`) never leave the device; the step never waits for the
 *      device (a re-detection's counts are read back when the next call needs them).  The accessors copy the state to HOST buffers and synchronise. ---- */
typedef struct vpp_video_extruder vpp_video_extruder;
typedef struct vpp_video_extruder_params {
  int32_t detector_th, keypoint_spacing, detector_period, max_trajectory_length, nscales, winsize, propagation;
} vpp_video_extruder_params;
int vpp_video_extruder_create(vpp_video_extruder** ve, int nrows, int ncols, int trajectory_capacity);
int vpp_video_extruder_destroy(vpp_video_extruder* ve);
int vpp_video_extruder_step(vpp_video_extruder* ve, const vpp_image_desc* frame1, const vpp_image_desc* frame2,
                            const vpp_video_extruder_params* params, void* stream);
/* The video loop's own shape (examples/video_extruder.cc:44-58: `prev` is kept, one video_extruder_update(ctx, prev, frame) per frame): one frame in,
 * one update out.  frame: u8 x1 (gray), or u8 x3 / x4 — then the rgb_to_graylevel of the loop (:48) happens inside, fused with the pyramid's
 * level 0.  The tracker keeps the previous frame's pyramid, so a frame's pyramid is built once (vpp_video_extruder_step builds both per
 * update); the frame's own border is not read (the levels are mirror-filled, as fill_border_mirror on the gray frame does).  The first frame
 * after create — or after a change of nscales / winsize — only becomes `prev`: no update, frame_id unchanged.  Results are those of
 * vpp_video_extruder_step on the mirror-bordered gray frames.  Do not interleave with vpp_video_extruder_step on one tracker. */
int vpp_video_extruder_push_frame(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* params, void* stream);
/* The same for a frame in HOST memory (a decoder's output; pinned — vpp_malloc_host — for the copy engine's full rate): the frame is copied into one
 * of two staging frames on a stream of the tracker's own, so the upload of frame t + 1 overlaps the update of frame t.  Returns once the host buffer
 * has been read (it may be refilled at once), with the update queued on `stream`.  frame->border is ignored.  Not recordable into a launch graph. */
int vpp_video_extruder_push_host_frame(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* params, void* stream);
/* For callers that rotate two host buffers (decoding frame t + 1 into one while frame t uploads from the other): the same without the wait.
 * vpp_video_extruder_wait_host_frame(ve, back) returns once the frame pushed last (back = 0) or the one before it (back = 1) has been read. */
int vpp_video_extruder_push_host_frame_nowait(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* params, void* stream);
int vpp_video_extruder_wait_host_frame(vpp_video_extruder* ve, int back);
/* container size (dead entries included), frame_id; either may be NULL.  Asking for the size after a re-detection frame waits for that frame's FAST count
 * (the update itself does not: everything behind the re-detection is queued against the two counts in HBM). */
int vpp_video_extruder_count(const vpp_video_extruder* ve, int* n, int* frame_id);
/* n (row, col) int32 pairs for position and velocity, n ages; any output may be NULL */
int vpp_video_extruder_keypoints(const vpp_video_extruder* ve, int32_t* pos_rc, int32_t* vel_rc, int32_t* age, int capacity, void* stream);
/* per trajectory: length, start frame, alive flag, ring head; ring_rc = n x slots x (row, col) f32, entry k of trajectory i (0 =
 * newest) at slot (head[i] + k) % slots; any output may be NULL */
int vpp_video_extruder_trajectories(const vpp_video_extruder* ve, int32_t* len, int32_t* start_frame, uint8_t* alive, int32_t* head,
                                    float* ring_rc, int capacity, void* stream);
int vpp_video_extruder_trajectory_slots(const vpp_video_extruder* ve, int* slots);
/* replace the state with host data laid out as the two accessors return it (a caller that edited the container between updates) */
int vpp_video_extruder_upload(vpp_video_extruder* ve, int n, int frame_id, const int32_t* pos_rc, const int32_t* vel_rc, const int32_t* age,
                              const int32_t* len, const int32_t* start_frame, const uint8_t* alive, const int32_t* head, const float* ring_rc,
                              void* stream);

/* ---- multi-GPU: the one exchange step of the keypoint-sharded path (no reference counterpart: the reference is a
 *      single OpenMP process; pyrlk_match.hh:24-51 iterates independent keypoints).  One process per GPU: rank 0 obtains
 *      the 128-byte id and ships it to the other ranks out of band (launcher, file, socket), every rank calls
 *      vpp_comm_init after vpp_init(local device).  vpp_allgather_tracks: each rank contributes n_per_rank records
 *      (pad the last shard), `all` receives nranks * n_per_rank records in rank order — an RCCL all-gather over xGMI on
 *      `stream`.  RCCL is loaded at first use; VPP_ERR_UNSUPPORTED when no librccl can be found. ---- */
typedef struct vpp_comm vpp_comm;
int vpp_comm_unique_id(void* id128);
int vpp_comm_init(vpp_comm** comm, int nranks, const void* id128, int rank);
int vpp_comm_destroy(vpp_comm* comm);
int vpp_allgather_tracks(vpp_comm* comm, const vpp_keypoint_f32* shard, int n_per_rank, vpp_keypoint_f32* all, void* stream);
/* Row-strip sharding of the image-space phases (SURVEY 8e bullet 2; no reference counterpart).  A strip is a bordered image holding
 * rows [r0, r1) of a frame; its border rows are the halo.  At a true frame edge they are an ordinary border (vpp_fill_border); at an
 * inner edge they must hold the neighbouring strip's rows — full pitch-wide rows, column borders included — so that a stencil kernel
 * (FAST-9: 3 rows, box R x C: R / 2, low-pass: 2) run on the strip computes what it would on the frame.  vpp_halo_exchange: one strip
 * per rank, ranks ordered top to bottom, grouped RCCL send / recv with the two neighbours.  vpp_halo_copy: the same transfer between
 * two strips of one process (two streams of one GPU, peer GPUs). */
int vpp_halo_exchange(vpp_comm* comm, const vpp_image_desc* strip, int halo_rows, void* stream);
int vpp_halo_copy(const vpp_image_desc* upper, const vpp_image_desc* lower, int halo_rows, void* stream);
/* The semi-dense flow (semi_dense_optical_flow.hpp:48-214) with its per-keypoint phase (:114-143) sharded over the ranks of `comm`, one
 * process per GPU (BASELINE configs[4]; no reference counterpart).  A match may land anywhere in the frame (prediction from the coarser
 * scale + descent + adopted neighbour flows), so a fixed image halo would not be exact: both frames are complete on every rank and the
 * pyramids replicated.  vpp_allgather_rows completes a frame of which rank g holds rows [g nrows / G, (g + 1) nrows / G) (what a sharded
 * capture / decode front-end leaves on each GPU) with one in-place RCCL all-gather of pitch-wide rows; nrows must divide by G.
 * vpp_semi_dense_optical_flow_sharded: rank g claims and descends only the keypoints whose flow-map cell lies in its row strip of each
 * scale, one grouped in-place all-gather per scale completes the three maps on every rank, the (deterministic, cheap) propagation sweeps
 * (:146-201) run on every rank, and every rank returns the full result — identical to vpp_semi_dense_optical_flow.  All ranks pass the
 * same frames, keypoints and parameters. */
int vpp_allgather_rows(vpp_comm* comm, const vpp_image_desc* frame, void* stream);
int vpp_semi_dense_optical_flow_sharded(vpp_comm* comm, const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps_rc, int n, int winsize,
                                        int nscales, int min_scale, int propagation, int patchsize, int32_t* out_pos_rc,
                                        int32_t* out_distance, uint8_t* out_valid, void* stream);

#ifdef __cplusplus
}
#endif
#endif
