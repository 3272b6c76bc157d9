// video.hip — K13/K14: the steps either side of the algorithms in the reference's video loop.
//  * rgb_to_graylevel (vpp/core/colorspace_conversions.hh:10-33 for 3 channels, :36-48 for 4): o = (i0 + i1 + i2) / 3 in int,
//    mapped over domain_with_border.  The ingest form (mirror = 1) fuses the reference's
//    `clone(frame, _border = b); fill_border_mirror(frame); rgb_to_graylevel<uchar>(frame)` (examples/video_extruder.cc:46-48)
//    into one pass: a border pixel of dst is the gray value of the mirrored source pixel, which is exactly what the three
//    separate steps produce (gray is per-pixel, so it commutes with the mirror copy).
//    HBM-bound: CH bytes read + 1 written per pixel; one lane = 16 output pixels (CH 16-B loads, one 16-B store).
//  * lbp_transform (vpp/algorithms/lbp/lbp_transform.hh:6-38), a 3x3 stencil on the same machinery.
//  * keypoint mask of video_extruder's re-detection (video_extruder/video_extruder.hpp:95-110): mask = 1 over the domain with
//    border, then the [r - s, r + s) x [c - s, c + s) square of every keypoint is zeroed.
#include "common.hpp"
#include "tracker_device.hpp"
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>
using namespace vpp_amd;

namespace {

__device__ __forceinline__ int mirror_index(int x, int n) { return x < 0 ? -x - 1 : (x >= n ? 2 * n - x - 1 : x); }  // fill.hh:60-83
__device__ __forceinline__ uint32_t div3(uint32_t s) { return (s * 43691u) >> 17; }  // exact for s <= 765 (43691 * 3 = 2^17 + 1)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kGrayChunk = 16;  // output pixels per lane: CH x 16 B loaded, one 16-B store

// byte k of a little-endian dword array
__device__ __forceinline__ uint32_t byte_at(const uint32_t* w, int k) { return (w[k >> 2] >> (8 * (k & 3))) & 255u; }

// 16 gray pixels from their 16 * CH source bytes (little-endian dwords).  The three channel bytes of a pixel are summed by
// v_dot4_u32_u8 against a 0 / 1 byte mask (two chained dots where a pixel straddles a dword), the quotient (s * 21846) >> 16 is
// exact for s <= 765 (3 * 21846 = 2^16 + 2: the excess 2 s / (3 * 2^16) < 1/3) and lands in byte 2 of a 24-bit product, so one
// v_perm_b32 packs two quotients: 13 VALU instructions per 4 pixels against 35 for the extract / add / multiply / shift / or form
// (the kernel has one resident round of waves whose loads all land together — the arithmetic that follows is not hidden).
template <int CH> __device__ __forceinline__ void gray_chunk(const uint32_t* w, uint32_t* o) {
  auto dot = [](uint32_t a, uint32_t mask, uint32_t acc) { return __builtin_amdgcn_udot4(a, mask, acc, false); };
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t s0, s1, s2, s3;
    if constexpr (CH == 3) {
      const uint32_t a = w[3 * q], b = w[3 * q + 1], c = w[3 * q + 2];
      s0 = dot(a, 0x00010101u, 0u);
      s1 = dot(a, 0x01000000u, dot(b, 0x00000101u, 0u));
      s2 = dot(b, 0x01010000u, dot(c, 0x00000001u, 0u));
      s3 = dot(c, 0x01010100u, 0u);
    } else {
      s0 = dot(w[4 * q], 0x00010101u, 0u); s1 = dot(w[4 * q + 1], 0x00010101u, 0u); s2 = dot(w[4 * q + 2], 0x00010101u, 0u); s3 = dot(w[4 * q + 3], 0x00010101u, 0u);
    }
    const uint32_t p0 = __umul24(s0, 21846u), p1 = __umul24(s1, 21846u), p2 = __umul24(s2, 21846u), p3 = __umul24(s3, 21846u);
    o[q] = __builtin_amdgcn_perm(p1, p0, 0x0c0c0602u) | __builtin_amdgcn_perm(p3, p2, 0x06020c0cu);
  }
}

// One lane = one 16-pixel chunk.  The chunks that lie entirely inside the source columns that map to themselves ("main": CH 16-byte
// loads, one 16-byte store) are numbered densely over (row, chunk) — no idle lanes at row ends, no divergence in those waves; the
// chunks that cross the left / right end of a row ("edge", 2 per row: mirrored / clipped columns, per-pixel form) are handled by
// the FIRST blocks of the same launch, so that their dependent byte loads run under the main stream instead of after it.
// One launch serves a BATCH of frames of one geometry (vpp_rgb_to_graylevel_batch; the single call = a batch of one): the block grid is the frames' grids back
// to back (`blocks_per_frame` each), so the chip does not drain between frames — a 33 MB launch alone reaches 54 % of the HBM peak (ramp + drain).
constexpr int kGrayBatchMax = 64;
struct GrayBatch { uint8_t* d[kGrayBatchMax]; const uint8_t* s[kGrayBatchMax]; };
template <int CH, bool MIRROR>
__global__ __launch_bounds__(256) void rgb_to_gray_kernel(DImg dst, DImg src, int ext, int c_start, int nchunks, int n_left, int n_main, int edge_blocks, int vec_ok,
                                                          const GrayBatch frames, unsigned blocks_per_frame) {
  const int nrows_out = dst.nr + 2 * ext;
  const unsigned frame = blockIdx.x / blocks_per_frame, fblock = blockIdx.x - frame * blocks_per_frame;
  dst.p0 = frames.d[frame]; src.p0 = const_cast<uint8_t*>(frames.s[frame]);
  if ((int)fblock < edge_blocks) {
    const int n_edge = nchunks - n_main, t = fblock * blockDim.x + threadIdx.x;
    const int row = t / n_edge, e = t - row * n_edge;
    if (row >= nrows_out) return;
    const int r = row - ext, c0 = c_start + kGrayChunk * (e < n_left ? e : e + n_main);
    const uint8_t* srow = src.row<uint8_t>(MIRROR ? mirror_index(r, src.nr) : r);
    uint8_t* drow = dst.row<uint8_t>(r);
    for (int k = 0; k < kGrayChunk; k++) {
      const int c = c0 + k;
      if (c < -ext || c >= dst.nc + ext) continue;
      const uint8_t* p = srow + (ptrdiff_t)(MIRROR ? mirror_index(c, src.nc) : c) * CH;
      drow[c] = (uint8_t)div3((uint32_t)p[0] + p[1] + p[2]);
    }
    return;
  }
  const long long t = (long long)(fblock - edge_blocks) * blockDim.x + threadIdx.x;
  const int row = (int)(t / n_main), chunk = n_left + (int)(t - (long long)row * n_main);
  if (row >= nrows_out) return;
  const int r = row - ext, c0 = c_start + kGrayChunk * chunk;
  const uint8_t* srow = src.row<uint8_t>(MIRROR ? mirror_index(r, src.nr) : r);
  uint8_t* drow = dst.row<uint8_t>(r);
  uint32_t w[4 * CH];
  // default cache policy: the CH loads of a lane (and of its neighbours) share cache lines; non-temporal loads measured 25 % slower
  __builtin_memcpy(w, srow + (ptrdiff_t)c0 * CH, 16 * CH);   // CH (possibly unaligned) 16-B loads
  uint32_t o[4];
  gray_chunk<CH>(w, o);
  if (vec_ok & 1) __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, (u32x4*)(drow + c0));
  else {
#pragma unroll
    for (int k = 0; k < kGrayChunk; k++) drow[c0 + k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
  }
}

// lbp_transform (vpp/algorithms/lbp/lbp_transform.hh:6-38): bit k of B(r, c) = neighbour k > centre, neighbours in row-major
// order without the centre ((-1,-1) = bit 0 ... (1,1) = bit 7).  Four pixels per lane: three 8-byte row loads (columns
// c-1 .. c+6), one dword store; ragged row ends per pixel.
__global__ __launch_bounds__(256) void lbp_kernel(DImg out, DImg in, int wide) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4, r = blockIdx.y;
  if (c >= out.nc) return;
  const uint8_t *r0 = in.row<uint8_t>(r - 1), *r1 = in.row<uint8_t>(r), *r2 = in.row<uint8_t>(r + 1);
  auto code = [](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t v, uint32_t b2, uint32_t c0, uint32_t c1, uint32_t c2) -> uint32_t {
    return (a0 > v ? 1u : 0u) | (a1 > v ? 2u : 0u) | (a2 > v ? 4u : 0u) | (b0 > v ? 8u : 0u) | (b2 > v ? 16u : 0u) | (c0 > v ? 32u : 0u) |
           (c1 > v ? 64u : 0u) | (c2 > v ? 128u : 0u);
  };
  uint8_t* o = out.row<uint8_t>(r) + c;
  if (!wide || c + 4 > out.nc) {
    for (int x = c; x < min(c + 4, out.nc); x++) o[x - c] = (uint8_t)code(r0[x - 1], r0[x], r0[x + 1], r1[x - 1], r1[x], r1[x + 1], r2[x - 1], r2[x], r2[x + 1]);
    return;
  }
  uint64_t w0, w1, w2;
  __builtin_memcpy(&w0, r0 + c - 1, 8); __builtin_memcpy(&w1, r1 + c - 1, 8); __builtin_memcpy(&w2, r2 + c - 1, 8);
  uint32_t res = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    auto b = [&](uint64_t w, int j) { return (uint32_t)(uint8_t)(w >> (8 * (k + j))); };
    res |= code(b(w0, 0), b(w0, 1), b(w0, 2), b(w1, 0), b(w1, 1), b(w1, 2), b(w2, 0), b(w2, 1), b(w2, 2)) << (8 * k);
  }
  if ((((uintptr_t)o) & 3) == 0) *(uint32_t*)o = res;
  else { o[0] = (uint8_t)res; o[1] = (uint8_t)(res >> 8); o[2] = (uint8_t)(res >> 16); o[3] = (uint8_t)(res >> 24); }
}

__global__ __launch_bounds__(256) void keypoint_mask_kernel(DImg mask, const int32_t* __restrict__ rc, int n, int s) {
  // one thread per (keypoint, square row): the row's 2s zero bytes go out as (unaligned) 16- and 4-byte stores — 2 stores for s = 10 where byte stores
  // were 20 per thread, 30 M on a 4K frame with 75 k keypoints (27 -> 8 us)
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;   // (n * 2 s < 2^32: checked by the launcher — a 64-bit division here was half of the kernel's instructions)
  const uint32_t rows = 2u * (uint32_t)s;
  const int k = (int)(t / rows), dr = (int)(t - (uint32_t)k * rows) - s;
  if (k >= n) return;
  const int r = rc[2 * k] + dr, c = rc[2 * k + 1];
  if (r < -mask.border || r >= mask.nr + mask.border) return;
  uint8_t* row = mask.row<uint8_t>(r);
  const int cb = max(c - s, -mask.border), ce = min(c + s, mask.nc + mask.border);
  int x = cb;
  const uint4 z16 = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t z4 = 0u;
  for (; x + 16 <= ce; x += 16) __builtin_memcpy(row + x, &z16, 16);
  for (; x + 4 <= ce; x += 4) __builtin_memcpy(row + x, &z4, 4);
  for (; x < ce; x++) row[x] = 0;
}

}  // namespace

namespace {
// geometry of one frame's launch (shared by every frame of a batch) and the kernel to launch / to re-parameterise a recorded node with
struct GrayGeom { DImg d, s; int ext, c_start, nchunks, n_left, n_main, edge_blocks, vec_ok, bsz, ch, mirror; unsigned blocks_per_frame; };
// bsz: threads per workgroup.  Measured at 4K (tools/ingest_ab.py, round 5): ONE frame per launch 64 threads 9.19 us, 256 threads 10.05 (more, shorter workgroups:
// the single resident round of waves ends more evenly); 64 frames per launch 64 threads 6.14 us = 0.676 of the HBM peak, 256 threads 5.85 us = 0.709 (a quarter
// of the workgroups to dispatch).  A wave-cooperative variant (coalesced 16-byte loads turned around through LDS, 256 threads) measured 5.89 / 9.79 us: the
// lane-per-chunk loads were never the limit, it was not kept.  So: 64 threads for single frames, 256 once a launch carries several (`frames`).
int gray_geometry(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror, GrayGeom* g, int frames = 1) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(src) && same_domain(dst, src), VPP_ERR_INVALID_ARG, "vpp_rgb_to_graylevel: invalid descriptors / domain mismatch");
  VPP_REQUIRE(dst->dtype == VPP_U8 && dst->channels == 1 && src->dtype == VPP_U8 && (src->channels == 3 || src->channels == 4), VPP_ERR_UNSUPPORTED,
              "vpp_rgb_to_graylevel: u8 x3 / x4 -> u8 x1 only");
  VPP_REQUIRE(dst->first_pixel != src->first_pixel, VPP_ERR_INVALID_ARG, "vpp_rgb_to_graylevel: in-place not supported");
  const int ext = mirror ? dst->border : (dst->border < src->border ? dst->border : src->border);
  VPP_REQUIRE(!mirror || (ext <= dst->nrows && ext <= dst->ncols), VPP_ERR_INVALID_ARG, "vpp_rgb_to_graylevel: border larger than the image");
  const int c_start = -((ext + kGrayChunk - 1) / kGrayChunk) * kGrayChunk;   // chunks are 16-B aligned relative to dst's first pixel
  const int nchunks = (dst->ncols + ext - c_start + kGrayChunk - 1) / kGrayChunk;
  // main chunks: c0 >= lo && c0 + 16 <= hi with [lo, hi) = the source columns that map to themselves
  const int lo = mirror ? 0 : -ext, hi = mirror ? src->ncols : src->ncols + ext;
  const int n_left = (lo - c_start + kGrayChunk - 1) / kGrayChunk;                                 // first chunk with c0 >= lo
  const int n_main = std::max(0, std::min(nchunks, (hi - c_start) / kGrayChunk) - n_left);            // chunks [n_left, n_left + n_main)
  const int nrows_out = dst->nrows + 2 * ext;
  int bsz = tuning("ingest.block", frames >= 4 ? 256 : 64);
  if (bsz != 128 && bsz != 256) bsz = 64;
  const int edge_blocks = (int)(((long long)nrows_out * (nchunks - n_main) + bsz - 1) / bsz);
  const long long main_blocks = ((long long)nrows_out * n_main + bsz - 1) / bsz;
  VPP_REQUIRE(edge_blocks + main_blocks < (1ll << 31) / kGrayBatchMax, VPP_ERR_UNSUPPORTED, "vpp_rgb_to_graylevel: image too large for one launch");
  *g = GrayGeom{dimg(dst), dimg(src), ext, c_start, nchunks, n_left, n_main, edge_blocks, aligned16(dst) ? 1 : 0, bsz, src->channels, mirror ? 1 : 0, (unsigned)(edge_blocks + main_blocks)};
  return VPP_OK;
}
void* gray_kernel(const GrayGeom& g) {
  if (g.ch == 3) return g.mirror ? (void*)rgb_to_gray_kernel<3, true> : (void*)rgb_to_gray_kernel<3, false>;
  return g.mirror ? (void*)rgb_to_gray_kernel<4, true> : (void*)rgb_to_gray_kernel<4, false>;
}
int gray_launch(const GrayGeom& g0, const GrayBatch& fr, int n, hipStream_t st) {
  GrayGeom g = g0; GrayBatch frames = fr;
  void* args[11] = {&g.d, &g.s, &g.ext, &g.c_start, &g.nchunks, &g.n_left, &g.n_main, &g.edge_blocks, &g.vec_ok, &frames, &g.blocks_per_frame};
  VPP_HIP_TRY(hipLaunchKernel(gray_kernel(g), dim3(g.blocks_per_frame * (unsigned)n), dim3((unsigned)g.bsz), args, 0, st));
  return VPP_OK;
}
inline bool same_gray_geometry(const vpp_image_desc& a, const vpp_image_desc& b) {
  return a.nrows == b.nrows && a.ncols == b.ncols && a.pitch == b.pitch && a.border == b.border && a.dtype == b.dtype && a.channels == b.channels && (((uintptr_t)a.first_pixel ^ (uintptr_t)b.first_pixel) & 15) == 0;
}
}  // namespace

extern "C" int vpp_rgb_to_graylevel(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror, void* stream) {
  GrayGeom g;
  int rc = gray_geometry(dst, src, mirror, &g);
  if (rc != VPP_OK) return rc;
  // on a stream this thread records through vpp_graph_begin: the frame is held back and recorded as part of ONE batched node (box.hip, common.hpp)
  if (!g_defer_bypass && defer_recording(stream) && tuning("ingest.coalesce", 1) && tuning("ingest.batch", 1)) return defer_call(kDeferGray, mirror ? 1 : 0, 0, stream, dst, src, nullptr);
  hipStream_t st = as_stream(stream);
  const Extent wr = extent_of(*dst), rd = extent_of(*src);
  IndependentCall side(st, &wr, 1, &rd, 1);   // recorded streams: calls on unrelated images become sibling nodes (common.hpp)
  GrayBatch fr{};
  fr.d[0] = (uint8_t*)dst->first_pixel; fr.s[0] = (const uint8_t*)src->first_pixel;
  return gray_launch(g, fr, 1, st);
}

// The per-frame call form without its per-frame launch (common.hpp, "held-back per-frame calls"); argument errors are reported at the call (gray_geometry is
// what vpp_rgb_to_graylevel checks with).
extern "C" int vpp_rgb_to_graylevel_deferred(const vpp_image_desc* dst, const vpp_image_desc* src, int mirror, void* stream) {
  GrayGeom g;
  const int rc = gray_geometry(dst, src, mirror, &g);
  if (rc != VPP_OK) return rc;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if ((cap != hipStreamCaptureStatusNone && !defer_recording(stream)) || g_defer_bypass || !tuning("defer", 1) || !tuning("ingest.batch", 1) || dst->first_pixel == src->first_pixel)
    return vpp_rgb_to_graylevel(dst, src, mirror, stream);
  return defer_call(kDeferGray, mirror ? 1 : 0, 0, stream, dst, src, nullptr);
}

// n frames of one geometry (same sizes, pitches, borders, channel count, 16-byte phase of the first pixels) in ONE launch; anything else, and frames that feed
// each other, go out as the n calls in sequence (whose results are the contract)
extern "C" int vpp_rgb_to_graylevel_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, int mirror, void* stream) {
  VPP_REQUIRE(n >= 0 && (n == 0 || (dst && src)), VPP_ERR_INVALID_ARG, "vpp_rgb_to_graylevel_batch: invalid argument");
  if (n == 0) return VPP_OK;
  bool same = n > 1 && tuning("ingest.batch", 1);
  for (int k = 0; same && k < n; k++) same = valid_desc(&dst[k]) && valid_desc(&src[k]) && same_gray_geometry(dst[k], dst[0]) && same_gray_geometry(src[k], src[0]);
  if (same) { const vpp_image_desc* srcs[1] = {src}; same = !batch_frames_interfere(n, dst, srcs, 1); }
  if (same) {
    GrayGeom g;
    int rc = gray_geometry(&dst[0], &src[0], mirror, &g, std::min(n, kGrayBatchMax));
    if (rc != VPP_OK) return rc;
    for (int k = 0; k < n; k++) VPP_REQUIRE(dst[k].first_pixel != src[k].first_pixel, VPP_ERR_INVALID_ARG, "vpp_rgb_to_graylevel_batch: in-place not supported (frame %d)", k);
    for (int b0 = 0; b0 < n; b0 += kGrayBatchMax) {
      const int nb = std::min(kGrayBatchMax, n - b0);
      GrayBatch fr{};
      for (int k = 0; k < nb; k++) { fr.d[k] = (uint8_t*)dst[b0 + k].first_pixel; fr.s[k] = (const uint8_t*)src[b0 + k].first_pixel; }
      rc = gray_launch(g, fr, nb, as_stream(stream));
      if (rc != VPP_OK) return rc;
    }
    return VPP_OK;
  }
  for (int k = 0; k < n; k++) { const int rc = vpp_rgb_to_graylevel(&dst[k], &src[k], mirror, stream); if (rc) return rc; }
  return VPP_OK;
}

// the squares of a mask that is already 1 everywhere (the tracker owns its mask block and fills it with one memset)
namespace vpp_amd {
int keypoint_mask_squares(const vpp_image_desc* mask, const int32_t* rc, int n, int spacing, hipStream_t st) {
  if (n == 0) return VPP_OK;
  const long long threads = (long long)n * 2 * spacing;
  VPP_REQUIRE(threads + 256 < (1ll << 32), VPP_ERR_INVALID_ARG, "vpp_keypoint_mask: %d keypoints x %d rows exceed one launch", n, 2 * spacing);
  keypoint_mask_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(dimg(mask), rc, n, spacing);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
}  // namespace vpp_amd

extern "C" int vpp_keypoint_mask(const vpp_image_desc* mask, const int32_t* rc, int n, int spacing, void* stream) {
  VPP_REQUIRE(valid_desc(mask) && n >= 0 && (rc || n == 0) && spacing > 0, VPP_ERR_INVALID_ARG, "vpp_keypoint_mask: invalid argument");
  VPP_REQUIRE(mask->dtype == VPP_U8 && mask->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_keypoint_mask: u8 x1 mask only");
  const uint8_t one = 1;
  int st = vpp_fill(mask, &one, 1, stream);
  if (st != VPP_OK) return st;
  return keypoint_mask_squares(mask, rc, n, spacing, as_stream(stream));
}

// ---- video_extruder merge (video_extruder.hpp:60-84) ------------------------------------------------------------
// The serial loop keeps one champion per cell: the first keypoint of the cell, replaced by any later one that is strictly
// older (which removes the champion); a later keypoint strictly younger than the champion is removed; a tie changes nothing.
// The champion's age is therefore the running maximum of the cell, and with E = max age of the earlier keypoints of the cell
// and L = max age of the later ones:   first of its cell or age > E  ->  it becomes champion, removed iff L > age
//                                       age == E -> kept (ties never become champion, so nothing evicts them);  age < E -> removed.
// Pass 1 threads every keypoint onto a per-cell list (atomic exchange of the list head); pass 2 walks the list of its cell —
// a handful of entries — and compares indices, so the unordered list gives the ordered answer.
__global__ __launch_bounds__(256) void merge_link_kernel(const int32_t* __restrict__ moved, const int32_t* __restrict__ prev, const uint8_t* __restrict__ matched,
                                                         const int32_t* __restrict__ age_prev, int n, int nr, int nc, int spacing, int gr, int gc,
                                                         int32_t* __restrict__ head, int32_t* __restrict__ next, int32_t* __restrict__ age_now, int32_t* __restrict__ cell_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  merge_link_one(MergeLinkArgs{age_prev, nr, nc, spacing, gr, gc, head, next, age_now, cell_of}, i, moved[2 * i], moved[2 * i + 1], prev[2 * i], prev[2 * i + 1], matched[i] != 0);
}
__global__ __launch_bounds__(256) void merge_fate_kernel(int n, MergeLists m, uint8_t* __restrict__ removed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) removed[i] = merge_removes(m, i) ? 1 : 0;
}

// head[] = -1 in one launch (hipMemsetAsync of this size is two dispatches of the runtime's fill kernel: head + aligned body, ~4.5 us each in a chain)
__global__ __launch_bounds__(256) void merge_heads_reset_kernel(int4* __restrict__ head4, int n4) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u < n4) head4[u] = make_int4(-1, -1, -1, -1);
}

namespace vpp_amd {
int keypoint_merge_prepare(const int32_t* age_prev, int n, int nrows, int ncols, int spacing, MergeLinkArgs* args, size_t* head_units16, hipStream_t st) {
  const int gr = nrows / spacing + 1, gc = ncols / spacing + 1;  // the reference's idx image is (nrows/s) x (ncols/s) with border 1 (:63-64)
  const size_t cells = ((size_t)gr * gc + 3) / 4 * 4;   // head[] padded to whole 16-byte units
  static thread_local Scratch scratch;  // per host thread, like the FAST / flow scratch
  size_t want = 1 << 20;  // grown in powers of two: the keypoint count creeps up at every re-detection
  while (want < (cells + 3 * (size_t)n) * sizeof(int32_t)) want <<= 1;
  const int rc = scratch.ensure(want, st);
  if (rc != VPP_OK) return rc;
  int32_t *head = (int32_t*)scratch.p, *next = head + cells, *age_now = next + n, *cell_of = age_now + n;
  *args = MergeLinkArgs{age_prev, nrows, ncols, spacing, gr, gc, head, next, age_now, cell_of};
  *head_units16 = cells / 4;
  return VPP_OK;
}
int keypoint_merge_link(const int32_t* rc_moved, const int32_t* rc_prev, const uint8_t* matched, const int32_t* age_prev, int n, int nrows, int ncols, int spacing,
                        MergeLists* lists, hipStream_t st) {
  MergeLinkArgs a; size_t units = 0;
  const int rc = keypoint_merge_prepare(age_prev, n, nrows, ncols, spacing, &a, &units, st);
  if (rc != VPP_OK) return rc;
  merge_heads_reset_kernel<<<(unsigned)((units + 255) / 256), 256, 0, st>>>((int4*)a.head, (int)units);   // (the scratch block is 256-byte aligned)
  merge_link_kernel<<<(n + 255) / 256, 256, 0, st>>>(rc_moved, rc_prev, matched, age_prev, n, nrows, ncols, spacing, a.gr, a.gc, a.head, a.next, a.age_now, a.cell_of);
  VPP_LAUNCH_CHECK();
  *lists = MergeLists{a.head, a.next, a.age_now, a.cell_of};
  return VPP_OK;
}
}  // namespace vpp_amd

extern "C" int vpp_keypoint_merge(const int32_t* rc_moved, const int32_t* rc_prev, const uint8_t* matched, const int32_t* age_prev, int n,
                                  int nrows, int ncols, int spacing, uint8_t* removed, void* stream) {
  VPP_REQUIRE(n >= 0 && nrows > 0 && ncols > 0 && spacing > 0 && (n == 0 || (rc_moved && rc_prev && matched && age_prev && removed)), VPP_ERR_INVALID_ARG,
              "vpp_keypoint_merge: invalid argument");
  if (n == 0) return VPP_OK;
  hipStream_t st = as_stream(stream);
  MergeLists m;
  const int rc = keypoint_merge_link(rc_moved, rc_prev, matched, age_prev, n, nrows, ncols, spacing, &m, st);
  if (rc != VPP_OK) return rc;
  merge_fate_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, m, removed);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

extern "C" int vpp_lbp_transform(const vpp_image_desc* out, const vpp_image_desc* in, void* stream) {
  VPP_REQUIRE(valid_desc(out) && valid_desc(in) && same_domain(out, in), VPP_ERR_INVALID_ARG, "vpp_lbp_transform: invalid descriptors / domain mismatch");
  VPP_REQUIRE(out->dtype == VPP_U8 && out->channels == 1 && in->dtype == VPP_U8 && in->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_lbp_transform: u8 x1 only");
  VPP_REQUIRE(in->border >= 1, VPP_ERR_BORDER_TOO_SMALL, "vpp_lbp_transform: input needs border >= 1 (lbp_transform.hh:17-19 reads rows r-1 .. r+1, columns -1 .. nc)");
  VPP_REQUIRE(out->first_pixel != in->first_pixel, VPP_ERR_INVALID_ARG, "vpp_lbp_transform: in-place not supported");
  dim3 grid(((out->ncols + 3) / 4 + 255) / 256, out->nrows);
  lbp_kernel<<<grid, 256, 0, as_stream(stream)>>>(dimg(out), dimg(in), in->border >= 3 ? 1 : 0);  // the 8-byte loads reach column nc + 2
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
