// fast9.hip — K7-K9: FAST-9 detector, score and non-maximum suppression.
// Reference: vpp/algorithms/fast_detector/fast.hpp:254-508 (fast_detector9_simd), :38-77 (fast9_score),
// :676-707 (fast_detector9_maxima), :745-799 (blockwise maxima), :889-928 (local maxima), :931-955 (front end fast9).
//
// Pipeline (all on one stream, no host round trip until the final count):
//   1. fast9_detect2_kernel  (default; fast9_detect_kernel = the single-phase form, tuning fast9.impl = 1)
//                            LDS tile (64x32 px + 3 px halo, dword loads), one lane per pixel column marching down 8 rows.
//                            Segment test per lane: saturated thresholds, 4 cardinal samples first (any 9-arc holds >= 2
//                            of ring indices {0,4,8,12}), then 16-bit brighter/darker masks and a shift-and "9 circularly
//                            contiguous" test.  corner = mask_byte & (0x10*B9 | 0x01*D9) != 0 (fast.hpp:120-126,312,333).
//                            Corner lanes then evaluate fast9_score on the TRUE ring (fast.hpp:52-74) from the same LDS tile
//                            and every lane writes the dense u16 map F(p) = score + 1 (0 = not a corner) — the reference's
//                            scores_img (:685-694) without the intermediate keypoint list (a single global append counter
//                            saturates at ~90 atomics/us, which measured 790 us on a 4K frame; the dense write costs 16.6 MB).
//                            Each wave also ballots its row of 64 flags into one u64 of a corner bitmap (1 bit / px) and the
//                            edge tiles zero F's 1-px border, so no memset precedes the launch.  The default kernel splits this in
//                            two phases per wave — pre-test for every pixel, then ring test + score on an LDS-compacted list of
//                            the survivors — so that the lanes of the expensive part are all busy (see fast9_detect2_kernel).
//   2. count / scan / write  units in the reference's serial output order, one per thread: 16-px bitmap segments (RAW,
//                            LOCAL_MAXIMA — the count pass rewrites a segment with the bits that survive the strict 8-neighbour
//                            test on F) or bs x bs blocks (BLOCKWISE — one lane per block row walks the bitmap words, rows meet
//                            in an LDS u64 max of score << 32 | ~position).  A flat exclusive scan of the unit counts (each write
//                            workgroup sums the per-workgroup counts before it: no scan launch) gives the output index, so the list comes out row-major (pixels / blocks) like the serial reference,
//                            which the OpenMP reference itself does not guarantee.  4K, 454k corners: RAW
//                            5 + 5 + 12 us, LOCAL 18 + 5 + 6 us, BLOCKWISE 13 + 5 + 6 us after the 30 us detect kernel.
// VALU-bound (the ring test is ~100 lane-ops per pixel against 1 B/px of HBM traffic); see LABNOTES.md section 3.
#include "common.hpp"
#include "tracker_device.hpp"
#include <mutex>
#include <type_traits>
using namespace vpp_amd;

namespace {

// ring offsets (dr, dc) for a0..a15.  REF: as sampled by fast_detector9_simd — a4/a12 come from row r-3 (fast.hpp:367-368);
// otherwise the true Bresenham ring of is_fast9_keypoint / fast9_score (fast.hpp:52-74, 88-109).
template <bool REF> __host__ __device__ constexpr int ring_dr(int i) {
  constexpr int ref[16] = {-3, -3, -2, -1, -3, 1, 2, 3, 3, 3, 2, 1, -3, -1, -2, -3};
  constexpr int tru[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  return REF ? ref[i] : tru[i];
}
__host__ __device__ constexpr int ring_dc(int i) {
  constexpr int dc[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  return dc[i];
}

constexpr int TW = 64, TH = 32, HALO = 3;   // 64 x 32 px per workgroup (8 rows per wave): measured 4K raw / blockwise 64 rows 53.8 / 46.0 us, 48: 53.4 / 43.8, 32: 52.1 / 42.3, 16: 54.5 / 43.4
constexpr int LP = 72;                       // LDS row pitch: cols [c0-4, c0+68)
constexpr int LROWS = TH + 2 * HALO;         // 38
constexpr int LDW = LP / 4;                  // 18 dwords per LDS row

__device__ __forceinline__ uint32_t ld_dword_guarded_px(const uint8_t* __restrict__ p, int off, int lo, int hi, bool aligned) {
  if (aligned && off >= lo && off + 4 <= hi) return *(const uint32_t*)(p + off);
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int q = off + k;
    const int qc = min(max(q, lo), hi - 1);
    const uint32_t b = p[qc];
    v |= (q >= lo && q < hi) ? (b << (8 * k)) : 0u;
  }
  return v;
}

__device__ __forceinline__ bool nine_contiguous(uint32_t m16) {
  const uint32_t x = m16 | (m16 << 16);
  uint32_t y = x & (x >> 1);
  y &= y >> 2;
  y &= y >> 4;   // runs of 8
  y &= x >> 8;   // runs of 9
  return (y & 0xFFFFu) != 0;
}

// LDS hand-off inside one wave: make this wave's LDS writes visible to its own later reads (no workgroup barrier needed)
__device__ __forceinline__ void wave_fence_lds() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// shift a comparison result into a ring mask without a compare: (m << 1) | (d < 0), one v_alignbit_b32
__device__ __forceinline__ uint32_t push_sign(uint32_t m, int d) { return __builtin_amdgcn_alignbit(m, (uint32_t)d, 31); }

template <bool REF>
__global__ __launch_bounds__(256) void fast9_detect_kernel(DImg A, DImg M, int has_mask, int th, DImg F, uint64_t* __restrict__ bitmap, int ntc) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[LROWS * LP];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH;
  const int lo = -A.border, hi = A.nc + A.border;
  const bool aligned = (((uintptr_t)A.p0 | (uintptr_t)A.pitch) & 3) == 0;
  for (int i = threadIdx.x; i < LROWS * LDW; i += 256) {
    const int lr = i / LDW, ld = i - lr * LDW;
    const int r = r0 - HALO + lr;
    uint32_t v = 0;
    if (r < A.nr + A.border) v = ld_dword_guarded_px(A.p0 + (ptrdiff_t)r * A.pitch, c0 - 4 + 4 * ld, lo, hi, aligned);
    ((uint32_t*)tile)[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = c0 + lane;
  if (c >= A.nc) return;
  const int thc = min(max(th, 0), 255);
  // F's 1-px border (read by the local-maxima pass) is zeroed here instead of by separate memsets
  if ((blockIdx.y == 0 && wv == 0) || (r0 + TH >= A.nr && wv == 1)) {
    uint16_t* fb = F.row<uint16_t>(wv == 0 ? -1 : A.nr);
    fb[c] = 0;
    if (c == 0) fb[-1] = 0;
    if (c == A.nc - 1) fb[A.nc] = 0;
  }
#pragma unroll 1
  for (int j = 0; j < TH / 4; j++) {
    const int lr = wv * (TH / 4) + j;
    const int r = r0 + lr;
    if (r >= A.nr) break;
    const uint8_t* p = tile + (lr + HALO) * LP + lane + 4;
    uint32_t f = 0;
    const int v = p[0];
    const int vhi = min(v + thc, 255), vlo = max(v - thc, 0);  // u_adds / u_subs (fast.hpp:322-324)
    int x[16];
#pragma unroll
    for (int i = 0; i < 16; i += 4) x[i] = p[ring_dr<REF>(i) * LP + ring_dc(i)];
    // ring masks from sign bits, no compares: x brighter <=> (vhi - x) < 0, darker <=> (x - vlo) < 0; one v_alignbit shifts the
    // sign into the mask.  First the four cardinal samples: any 9 contiguous ring positions contain two of {0, 4, 8, 12}.
    uint32_t qb = 0, qd = 0;
#pragma unroll
    for (int i = 0; i < 16; i += 4) { qb = push_sign(qb, vhi - x[i]); qd = push_sign(qd, x[i] - vlo); }
    if ((qb & (qb - 1)) | (qd & (qd - 1))) {
#pragma unroll
      for (int i = 0; i < 16; i++) if (i & 3) x[i] = p[ring_dr<REF>(i) * LP + ring_dc(i)];
      uint32_t mb = 0, md = 0;
#pragma unroll
      for (int i = 15; i >= 0; i--) { mb = push_sign(mb, vhi - x[i]); md = push_sign(md, x[i] - vlo); }
      int planes = (nine_contiguous(mb) ? 0x10 : 0) | (nine_contiguous(md) ? 0x01 : 0);
      if (has_mask && planes) planes &= M.row<uint8_t>(r)[c];
      if (planes) {
        // fast9_score on the TRUE ring (fast.hpp:38-77), branch-free: sum over {d > th} of d = th * count + sum max(d - th, 0);
        // the counts are popcounts of the true-ring masks (x < v - th <=> x < vlo, x > v + th <=> x > vhi for 0 <= th <= 255)
        if (REF) {  // only a4 / a12 differ from the samples the detector used
          x[4] = p[3]; x[12] = p[-3];
          mb = (mb & ~0x1010u) | ((uint32_t)(vhi - x[4]) >> 31 << 4) | ((uint32_t)(vhi - x[12]) >> 31 << 12);
          md = (md & ~0x1010u) | ((uint32_t)(x[4] - vlo) >> 31 << 4) | ((uint32_t)(x[12] - vlo) >> 31 << 12);
        }
        // sum max(dn - x, 0) = (sum |dn - x| + 16 dn - sum x) / 2 and sum max(x - up, 0) = (sum |x - up| + sum x - 16 up) / 2
        // with dn, up clamped to a byte: three v_sad_u8 chains over the ring packed 4 samples per dword
        const uint32_t dnc = (uint32_t)max(v - th, 0), upc = (uint32_t)min(v + th, 255);
        const uint32_t dn4 = dnc * 0x01010101u, up4 = upc * 0x01010101u;
        uint32_t sx = 0, sdn = 0, sup = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t X = (uint32_t)x[4 * q] | ((uint32_t)x[4 * q + 1] << 8) | ((uint32_t)x[4 * q + 2] << 16) | ((uint32_t)x[4 * q + 3] << 24);
          sx = __builtin_amdgcn_sad_u8(X, 0u, sx); sdn = __builtin_amdgcn_sad_u8(X, dn4, sdn); sup = __builtin_amdgcn_sad_u8(X, up4, sup);
        }
        const int over_sup = (int)(sdn + 16u * dnc - sx) >> 1, over_inf = (int)(sup + sx - 16u * upc) >> 1;
        f = (uint32_t)max(th * __popc(md) + over_sup, th * __popc(mb) + over_inf) + 1u;
      }
    }
    uint16_t* fr = F.row<uint16_t>(r);
    fr[c] = (uint16_t)f;
    if (c == 0) fr[-1] = 0;
    if (c == A.nc - 1) fr[A.nc] = 0;
    const uint64_t corners = __ballot(f != 0);  // one 64-px word of the corner bitmap per wave and row
    if (lane == 0) bitmap[(size_t)r * ntc + blockIdx.x] = corners;
  }
}

// Two-phase variant of the detect kernel.  In the kernel above a wave pays the full ring test whenever ANY of its 64 lanes
// passes the four-sample pre-test, and on natural images nearly every wave has such a lane.  Here phase 1 runs the pre-test for
// every pixel of the wave's TH / 4 rows without divergence, writes F = 0 rows (coalesced) and appends the survivors to a per-wave
// LDS list (ballot + mbcnt prefix); phase 2 runs the ring test and the score on the list, 64 candidates per pass, so its lanes
// are all busy; corners overwrite their F entry and set their bit in a per-wave LDS copy of the 16 bitmap words.
// MODE (what the selection that follows needs): VPP_FAST9_LOCAL_MAXIMA reads F around every corner, so F is written densely (zeros
// included); RAW reads F at the corners only (no zero fill: 16.6 MB of 2-byte stores less on a 4K frame); BLOCKWISE needs neither F
// nor the bitmap — a corner raises the key (score / 16) << 32 | ~position of its bs x bs block with one 64-bit atomic max in L2
// (keys all-zero before the launch: zeroed by the write pass of the previous call, or by a memset node), which is the reduction fast9_count_blocks_kernel did in a second pass over F.
// FUSED (RAW only, round 6): the ordered write happens in THIS launch — no F map, no bitmap, no second kernel.  A corner's output index is
//   (corners of the bands above) + (corners of the rows above it in its band) + (corners left of its tile in its row) + (its rank in the tile's row word).
// No tile ever waits for another one (a first version in which every tile waited for its band's prefix measured 98 us per 4K call instead of 42: thousands of
// resident workgroups polling, and a slow tile stalling every later one — which keeps its slot — into a convoy).  Instead every tile STAGES its corners — packed
// {score + 1, row in the band, column in the tile}, tile-major, in the tile's own 8 KB of a staging area — publishes its 32 row counts (write-through bytes,
// row-major: a row's counts of all tiles are contiguous), adds its total to its band's word, arrives on the band's counter and leaves.  The band's LAST arriver
// stays: it publishes {aggregate} for the band, looks back over the bands above (decoupled look-back: one lane per predecessor, an {inclusive prefix} ends the
// walk — the only wait in the launch, for bands whose tiles were all dispatched earlier), publishes the band's {inclusive prefix}, builds the band's offset tables
// from the 32 x ntc count matrix in LDS and moves the band's staged records (a few thousand) to their places in the output.  The gathers of all bands but the
// last few overlap other tiles' detection.  The last tile of the launch hands the control block back zeroed.
constexpr int kFuseMaxTiles = 64;             // tile columns the fused write serves (frames up to 4096 px wide): the band tables fit the detect phase's LDS
constexpr int kStagePerTile = TW * TH;        // staged records per tile (u32 each): every pixel a corner
struct RawFuse {
  uint8_t* rowcnt; int ntcp;                    // [nby * TH rows][ntcp] corner counts per (row, tile); ntcp = ntc rounded up to 8
  uint32_t* stage;                              // [nby * ntc tiles][kStagePerTile] packed records
  // per band, kBandStride words apart (returning atomics on one cache line retire one after the other, ~11 ns each: with the bands' counters side by side the
  // 4 080 arrivals of a 4K frame alone took longer than the detection): [0] corners << 32 | tiles arrived — ONE atomic per tile; [1] flag << 32 | value; flag 1:
  // value = the band's corners, flag 2: value = corners of this band and all above
  unsigned long long* band;
  uint32_t* finished;                           // bands whose gather is done
  uint32_t* total; int32_t* out_rc; int32_t* out_scores; int capacity;
  unsigned* err;                                // the sticky device error word (common.hpp)
};
constexpr unsigned kFuseSpinLimit = 1u << 22;
constexpr int kBandStride = 32;   // 256 bytes
__device__ __forceinline__ uint32_t ld_u32_sc1(const uint32_t* p) { return __hip_atomic_load((uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64_sc1(const unsigned long long* p) { return __hip_atomic_load((unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE kDenseU8 / kDenseI32 (round 6, vpp_fast9_dense): no selection at all — the kernel's result is the dense 0 / 1 flag image (F = the destination, u8 or int32):
// phase 1 writes the zeros, four pixels per item, a corner of phase 2 writes its 1 (no score).
constexpr int kDenseU8 = 3, kDenseI32 = 4;
template <bool REF, int MODE, bool FUSED = false>
__global__ __launch_bounds__(256) void fast9_detect2_kernel(DImg A, DImg M, int has_mask, int th, DImg F, uint64_t* __restrict__ bitmap, int ntc,
                                                            unsigned long long* __restrict__ blkkey, uint32_t bs_magic, int nbc,
                                                            uint8_t* __restrict__ rowcnt, uint16_t* __restrict__ tiletot, RawFuse fz) {
  static_assert(!FUSED || MODE == VPP_FAST9_RAW, "the fused write serves RAW");
  __shared__ __attribute__((aligned(16))) uint8_t tile[LROWS * LP];
  __shared__ uint16_t cand[4][TH / 4 * TW];          // per wave: (row in the wave's band) * 64 + column
  __shared__ unsigned long long words[4][TH / 4];    // per wave: corner bitmap words of its TH / 4 rows
  __shared__ uint16_t fsc[FUSED ? 4 : 1][FUSED ? TH / 4 * TW : 1];   // FUSED: F(p) = score + 1 of the wave's corners, at (row in the wave's band) * 64 + column
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH;
  const int lo = -A.border, hi = A.nc + A.border;
  const bool aligned = (((uintptr_t)A.p0 | (uintptr_t)A.pitch) & 3) == 0;
  // (round 5) A tile whose 38 x 72-byte patch lies inside the bordered area — every tile but the frame's last column / row of tiles — loads it without the per-dword
  // guards and without a division per dword: thread t owns dword t % 18 of the rows t / 18, + 14, + 28 (252 threads, three loads in flight each).  ~20 instead of
  // ~54 instructions per thread for the staging, of the kernel's ~580 per thread.
  if (aligned && c0 - 4 >= lo && c0 + TW + 4 <= hi && r0 + TH + HALO <= A.nr + A.border) {   // (r0 - HALO >= -border always: border >= 3; wave-uniform)
    const int t = threadIdx.x;
    if (t < 14 * LDW) {
      const int lr = t / LDW, ld = t - lr * LDW;
      const uint8_t* src = A.p0 + (ptrdiff_t)(r0 - HALO + lr) * A.pitch + (c0 - 4 + 4 * ld);
      const uint32_t v0 = *(const uint32_t*)src, v1 = *(const uint32_t*)(src + (ptrdiff_t)14 * A.pitch);
      uint32_t v2 = 0;
      if (lr + 28 < LROWS) v2 = *(const uint32_t*)(src + (ptrdiff_t)28 * A.pitch);
      ((uint32_t*)tile)[t] = v0; ((uint32_t*)tile)[t + 14 * LDW] = v1;
      if (lr + 28 < LROWS) ((uint32_t*)tile)[t + 28 * LDW] = v2;
    }
  } else
  for (int i = threadIdx.x; i < LROWS * LDW; i += 256) {
    const int lr = i / LDW, ld = i - lr * LDW;
    const int r = r0 - HALO + lr;
    uint32_t v = 0;
    if (r < A.nr + A.border) v = ld_dword_guarded_px(A.p0 + (ptrdiff_t)r * A.pitch, c0 - 4 + 4 * ld, lo, hi, aligned);
    ((uint32_t*)tile)[i] = v;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane < TH / 4) words[wv][lane] = 0;
  __syncthreads();
  const int c = c0 + lane;
  const bool col_ok = c < A.nc;
  const int thc = min(max(th, 0), 255);
  if (MODE == VPP_FAST9_LOCAL_MAXIMA && col_ok && ((blockIdx.y == 0 && wv == 0) || (r0 + TH >= A.nr && wv == 1))) {
    uint16_t* fb = F.row<uint16_t>(wv == 0 ? -1 : A.nr);
    fb[c] = 0;
    if (c == 0) fb[-1] = 0;
    if (c == A.nc - 1) fb[A.nc] = 0;
  }
  // ---- phase 1: pre-test on the four cardinal samples, FOUR PIXELS PER LANE on packed 16-bit fields (round 3) ----
  // Any 9 consecutive ring positions contain two cardinal samples that are NEIGHBOURS among {0, 4, 8, 12} (8 consecutive positions already
  // hold exactly two multiples of 4, four apart), both on the same side: pass = (b0 | b8) & (b4 | b12) for "brighter", the same for "darker"
  // (stricter than "any two of the four", which lets every line through the centre pass, and still necessary: the result is unchanged).
  // A lane takes the dword of its 4 pixels and the dwords of the 4 samples (a4 / a12 sit 3 columns away: two dwords + v_alignbyte), splits
  // each into its even and odd bytes (two u16 fields per register) and evaluates x > v + th as bit 15 of x + (0x8000 - v - th - 1) and
  // x < v - th as bit 15 of (0x8000 + v - th - 1) - x — plain 32-bit adds, no field ever borrows from its neighbour (|values| < 512):
  // ~12 instead of ~30 VALU instructions per pixel and 7 dword LDS reads per 4 pixels instead of 20 byte reads.
  int ncand = 0;
  {
    const uint32_t* tile32 = (const uint32_t*)tile;
    const uint32_t th1 = (uint32_t)(thc + 1) * 0x00010001u;
#pragma unroll
    for (int it = 0; it < (TH / 4) * (TW / 4) / 64; it++) {
      const int item = it * 64 + lane, j = item / (TW / 4), g = item - j * (TW / 4);
      const int lr = wv * (TH / 4) + j, r = r0 + lr, cg = c0 + 4 * g;
      uint32_t pass4 = 0;   // bit k: pixel cg + k passes
      if (r < A.nr && cg < A.nc) {
        const int rowc = (lr + HALO) * LDW + g + 1, rowm = rowc - 3 * LDW, rowp = rowc + 3 * LDW;
        const int rows = REF ? rowm : rowc;   // REF: a4 / a12 come from row r - 3 (fast.hpp:367-368), else from the pixel's own row
        const uint32_t V = tile32[rowc], X0 = tile32[rowm], X8 = tile32[rowp];
        const uint32_t sl = tile32[rows - 1], sm = REF ? X0 : V, sr = tile32[rows + 1];
        const uint32_t X4 = __builtin_amdgcn_alignbyte(sr, sm, 3), X12 = __builtin_amdgcn_alignbyte(sm, sl, 1);   // columns + 3 / - 3
        auto ev = [](uint32_t d) { return __builtin_amdgcn_perm(0u, d, 0x0c020c00u); };
        auto od = [](uint32_t d) { return __builtin_amdgcn_perm(0u, d, 0x0c030c01u); };
        uint32_t pe, po;
        {
          const uint32_t v = ev(V), kb = 0x80008000u - v - th1, kd = 0x80008000u + v - th1;
          const uint32_t x0 = ev(X0), x4 = ev(X4), x8 = ev(X8), x12 = ev(X12);
          pe = (((x0 + kb) | (x8 + kb)) & ((x4 + kb) | (x12 + kb))) | (((kd - x0) | (kd - x8)) & ((kd - x4) | (kd - x12)));
        }
        {
          const uint32_t v = od(V), kb = 0x80008000u - v - th1, kd = 0x80008000u + v - th1;
          const uint32_t x0 = od(X0), x4 = od(X4), x8 = od(X8), x12 = od(X12);
          po = (((x0 + kb) | (x8 + kb)) & ((x4 + kb) | (x12 + kb))) | (((kd - x0) | (kd - x8)) & ((kd - x4) | (kd - x12)));
        }
        pass4 = ((pe >> 15) & 1u) | ((po >> 14) & 2u) | ((pe >> 29) & 4u) | ((po >> 28) & 8u);
        const int ncol = A.nc - cg;   // pixels of this group inside the image
        if (ncol < 4) pass4 &= (1u << ncol) - 1u;
        if (has_mask && pass4) {
          // (round 6) a pixel whose mask byte is 0 cannot be a corner (planes & 0, below): it never becomes a candidate.  One dword of mask per group, where a group has
          // a candidate at all, instead of a byte load per corner behind the ring test — the tracker's mask is 0 almost everywhere (a square around every keypoint).
          const uint8_t* mp = M.row<uint8_t>(r) + cg;
          uint32_t m4 = 0;
          if (ncol >= 4) __builtin_memcpy(&m4, mp, 4);
          else for (int k = 0; k < ncol; k++) m4 |= (uint32_t)mp[k] << (8 * k);
          pass4 &= ((m4 & 0xFFu) ? 1u : 0u) | ((m4 & 0xFF00u) ? 2u : 0u) | ((m4 & 0xFF0000u) ? 4u : 0u) | ((m4 & 0xFF000000u) ? 8u : 0u);
        }
        if (MODE == VPP_FAST9_LOCAL_MAXIMA) {
          uint16_t* fr = F.row<uint16_t>(r) + cg;
          if (ncol >= 4) *(uint2*)fr = make_uint2(0u, 0u);
          else for (int k = 0; k < ncol; k++) fr[k] = 0;
          if (cg == 0) fr[-1] = 0;
          if (cg + 4 >= A.nc) F.row<uint16_t>(r)[A.nc] = 0;
        }
        if (MODE == kDenseU8) {   // (the launcher checks the destination's 4- / 16-byte alignment)
          uint8_t* fr = F.row<uint8_t>(r) + cg;
          if (ncol >= 4) *(uint32_t*)fr = 0u; else for (int k = 0; k < ncol; k++) fr[k] = 0;
        }
        if (MODE == kDenseI32) {
          int32_t* fr = F.row<int32_t>(r) + cg;
          if (ncol >= 4) *(uint4*)fr = make_uint4(0u, 0u, 0u, 0u); else for (int k = 0; k < ncol; k++) fr[k] = 0;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool pk = (pass4 >> k) & 1u;
        const unsigned long long m = __ballot(pk);
        if (pk) cand[wv][ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(j * TW + 4 * g + k);
        ncand += __popcll(m);
      }
    }
  }
  // ---- phase 2: ring test + score on the compacted candidates ----
  // (round 6) The four waves' candidate lists are walked as ONE list, 256 candidates per step: a wave used to walk its own list, 64 per step — 70-80 candidates per wave
  // on the bench frame are two steps with the second one a fifth full; pooled, the workgroup's ~300 are 1.2 steps per wave.
  __shared__ int s_ncand[4];
  if (lane == 0) s_ncand[wv] = ncand;
  __syncthreads();
  const int nc0 = s_ncand[0], nc1 = nc0 + s_ncand[1], nc2 = nc1 + s_ncand[2], nctot = nc2 + s_ncand[3];
  for (int base = 0; base < nctot; base += 256) {
    const int kk = base + (int)threadIdx.x;
    if (kk < nctot) {
      const int cw = (kk >= nc0) + (kk >= nc1) + (kk >= nc2);   // the wave whose list holds candidate kk, and its place there
      const int k = kk - (cw == 0 ? 0 : (cw == 1 ? nc0 : (cw == 2 ? nc1 : nc2)));
      const int id = cand[cw][k], j = id / TW, col = id - j * TW;
      const int lr = cw * (TH / 4) + j;
      const uint8_t* p = tile + (lr + HALO) * LP + col + 4;
      const int v = p[0];
      const int vhi = min(v + thc, 255), vlo = max(v - thc, 0);
      int x[16];
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = p[ring_dr<REF>(i) * LP + ring_dc(i)];
      uint32_t mb = 0, md = 0;
#pragma unroll
      for (int i = 15; i >= 0; i--) { mb = push_sign(mb, vhi - x[i]); md = push_sign(md, x[i] - vlo); }
      int planes = (nine_contiguous(mb) ? 0x10 : 0) | (nine_contiguous(md) ? 0x01 : 0);
      if (has_mask && planes) planes &= M.row<uint8_t>(r0 + lr)[c0 + col];
      if (planes && MODE == kDenseU8) F.row<uint8_t>(r0 + lr)[c0 + col] = 1;
      else if (planes && MODE == kDenseI32) F.row<int32_t>(r0 + lr)[c0 + col] = 1;
      else if (planes) {
        if (REF) {  // only a4 / a12 differ from the samples the detector used
          x[4] = p[3]; x[12] = p[-3];
          mb = (mb & ~0x1010u) | ((uint32_t)(vhi - x[4]) >> 31 << 4) | ((uint32_t)(vhi - x[12]) >> 31 << 12);
          md = (md & ~0x1010u) | ((uint32_t)(x[4] - vlo) >> 31 << 4) | ((uint32_t)(x[12] - vlo) >> 31 << 12);
        }
        const uint32_t dnc = (uint32_t)max(v - th, 0), upc = (uint32_t)min(v + th, 255);
        const uint32_t dn4 = dnc * 0x01010101u, up4 = upc * 0x01010101u;
        uint32_t sx = 0, sdn = 0, sup = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t X = (uint32_t)x[4 * q] | ((uint32_t)x[4 * q + 1] << 8) | ((uint32_t)x[4 * q + 2] << 16) | ((uint32_t)x[4 * q + 3] << 24);
          sx = __builtin_amdgcn_sad_u8(X, 0u, sx); sdn = __builtin_amdgcn_sad_u8(X, dn4, sdn); sup = __builtin_amdgcn_sad_u8(X, up4, sup);
        }
        const int over_sup = (int)(sdn + 16u * dnc - sx) >> 1, over_inf = (int)(sup + sx - 16u * upc) >> 1;
        const uint32_t f = (uint32_t)max(th * __popc(md) + over_sup, th * __popc(mb) + over_inf) + 1u;
        if (MODE == VPP_FAST9_BLOCKWISE) {
          const uint32_t s16 = (f - 1u) >> 4, r = (uint32_t)(r0 + lr), cc = (uint32_t)(c0 + col);   // stored(): score / 16, 0 never wins (fast.hpp:693,770-789)
          if (s16) {
            const uint32_t br = bs_magic ? __umulhi(r, bs_magic) : r, bc = bs_magic ? __umulhi(cc, bs_magic) : cc;
            atomicMax(&blkkey[(size_t)br * nbc + bc], ((unsigned long long)s16 << 32) | (0xFFFFFFFFu - ((r << 16) | cc)));
          }
        } else {
          if (FUSED) fsc[cw][id] = (uint16_t)f;
          else F.row<uint16_t>(r0 + lr)[c0 + col] = (uint16_t)f;
          atomicOr(&words[cw][j], 1ull << col);
        }
      }
    }
  }
  if (MODE == VPP_FAST9_BLOCKWISE || MODE == kDenseU8 || MODE == kDenseI32) return;
  __syncthreads();   // (a wave's words / scores are written by all four)
  if constexpr (FUSED) {
    __shared__ uint32_t wtot[4], s_last, s_excl;
    const int band = blockIdx.y, bx = blockIdx.x;
    // ---- this tile's counts, and its records staged tile-major (rows in order, a row's corners in column order)
    uint32_t nrow = 0;
    if (lane < TH / 4) {   // rows past the frame's last one have no corners: their counts are written all the same (nobody reads a stale byte)
      nrow = (uint32_t)__popcll(words[wv][lane]);
      __hip_atomic_store(fz.rowcnt + (size_t)(band * TH + wv * (TH / 4) + lane) * fz.ntcp + bx, (uint8_t)nrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t inc = nrow;   // inclusive scan over the wave's 8 rows (lanes 0 .. 7)
    { uint32_t t = __shfl_up(inc, 1); if (lane >= 1) inc += t; t = __shfl_up(inc, 2); if (lane >= 2) inc += t; t = __shfl_up(inc, 4); if (lane >= 4) inc += t; }
    if (lane == TH / 4 - 1) wtot[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += wtot[w];
    uint32_t* stg = fz.stage + (size_t)(band * ntc + bx) * kStagePerTile;
#pragma unroll 1
    for (int j = 0; j < TH / 4; j++) {
      const unsigned long long w = words[wv][j];
      const uint32_t rb = wbase + __shfl(inc - nrow, j);   // records of the tile's rows above this one
      if ((w >> lane) & 1ull)
        __hip_atomic_store(stg + rb + (uint32_t)__popcll(w & ((1ull << lane) - 1ull)), ((uint32_t)fsc[wv][j * TW + lane] << 16) | (uint32_t)((wv * (TH / 4) + j) << 6) | (uint32_t)lane,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's counts and records have been performed (write-through) before the workgroup arrives
    __syncthreads();
    unsigned long long* const bctl = fz.band + (size_t)band * kBandStride;
    if (threadIdx.x == 0) {
      const uint32_t ttot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
      const unsigned long long old = __hip_atomic_fetch_add(bctl, ((unsigned long long)ttot << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (uint32_t)old == (uint32_t)ntc - 1u ? 1u : 0u;
      s_excl = (uint32_t)(old >> 32) + ttot;   // (the last arriver's: the band's corners)
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the band is complete; this workgroup places its records.  First the corners of the bands above, by decoupled look-back.
    const uint32_t btot = s_excl;
    __syncthreads();
    // Everything this workgroup will read is requested NOW, before the look-back's wait: the first 64 staged records of the wave's tiles (wv, wv + 4, ...) and
    // this thread's share of the count matrix — one memory round trip for the whole gather instead of one per step (the gather of the launch's last band is its tail)
    constexpr int kTilesPerWave = kFuseMaxTiles / 4;
    uint32_t rec[kTilesPerWave];
#pragma unroll
    for (int u = 0; u < kTilesPerWave; u++) {
      const int t = wv + 4 * u;
      rec[u] = t < ntc ? ld_u32_sc1(fz.stage + (size_t)(band * ntc + t) * kStagePerTile + lane) : 0u;   // (lanes past the tile's count read stale words and drop them)
    }
    static_assert(TH * (kFuseMaxTiles / 8) == 256, "one 8-byte word of the count matrix per thread");
    unsigned long long cword = 0;
    { const int j = threadIdx.x / (kFuseMaxTiles / 8), q = threadIdx.x - j * (kFuseMaxTiles / 8);
      if (q < fz.ntcp / 8) cword = ld_u64_sc1((const unsigned long long*)(fz.rowcnt + (size_t)(band * TH + j) * fz.ntcp) + q); }   // (bytes past ntc: zeroed with the control block, never written)
    if (wv == 0) {
      if (lane == 0 && band > 0) __hip_atomic_store(bctl + 1, (1ull << 32) | btot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t excl = 0;
      bool gave_up = false;
      for (int b = band - 1; b >= 0 && !gave_up;) {
        const int mb = b - lane;
        unsigned long long st = 2ull << 32;   // lanes past band 0: an inclusive prefix of 0
        if (mb >= 0) {
          unsigned spin = 0;
          while (((st = ld_u64_sc1(fz.band + (size_t)mb * kBandStride + 1)) >> 32) == 0ull && ++spin < kFuseSpinLimit) __builtin_amdgcn_s_sleep(2);
          if (spin >= kFuseSpinLimit) gave_up = true;
        }
        gave_up = __ballot(gave_up) != 0ull;
        const unsigned long long pm = __ballot((st >> 32) == 2ull);
        const int first = pm ? __ffsll((long long)pm) - 1 : 64;   // the nearest band above whose inclusive prefix is known ends the walk
        uint32_t v = lane <= first ? (uint32_t)st : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        excl += v;
        if (pm) break;
        b -= 64;
      }
      if (gave_up && lane == 0 && fz.err) __hip_atomic_fetch_or(fz.err, (unsigned)kDevErrFastFuse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (lane == 0) {
        __hip_atomic_store(bctl + 1, (2ull << 32) | (unsigned long long)(excl + btot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (band == (int)gridDim.y - 1) *fz.total = excl + btot;
        s_excl = excl;
      }
    }
    // ---- the band's tables (the detect phase's LDS is free now: the count matrix and the two prefix tables take its place)
    static_assert(sizeof(tile) >= TH * kFuseMaxTiles && sizeof(fsc) >= TH * kFuseMaxTiles * 2 && sizeof(cand) >= TH * kFuseMaxTiles * 2, "the band tables overlay the detect phase's LDS");
    __syncthreads();   // (every wave is done with words / fsc)
    uint8_t (*cnt)[kFuseMaxTiles] = (uint8_t (*)[kFuseMaxTiles])tile;       // corners of (row, tile)
    uint16_t (*left)[kFuseMaxTiles] = (uint16_t (*)[kFuseMaxTiles])fsc;     // corners of the row in the tiles left of this one
    uint16_t (*top)[kFuseMaxTiles] = (uint16_t (*)[kFuseMaxTiles])cand;     // corners of the tile in the rows above this one (= the record's place in the tile's staging block)
    __shared__ uint32_t rowbase[TH], rowtot[TH];
    { const int j = threadIdx.x / (kFuseMaxTiles / 8), q = threadIdx.x - j * (kFuseMaxTiles / 8);
      *(unsigned long long*)&cnt[j][8 * q] = cword; }
    __syncthreads();
#pragma unroll 1
    for (int jj = 0; jj < TH / 4; jj++) {   // a wave per row: exclusive scan along the tiles
      const int j = wv * (TH / 4) + jj;
      uint32_t carry = 0;
      for (int t0 = 0; t0 < ntc; t0 += 64) {
        const int t = t0 + lane;
        const uint32_t c = t < ntc ? cnt[j][t] : 0u;
        uint32_t sc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(sc, d); if (lane >= d) sc += u; }
        if (t < ntc) left[j][t] = (uint16_t)(carry + sc - c);
        carry += __shfl(sc, 63);
      }
      if (lane == 0) rowtot[j] = carry;
    }
    if ((int)threadIdx.x < ntc) {   // a lane per tile: down its rows
      uint32_t acc = 0;
#pragma unroll 4
      for (int j = 0; j < TH; j++) { top[j][threadIdx.x] = (uint16_t)acc; acc += cnt[j][threadIdx.x]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = s_excl; for (int j = 0; j < TH; j++) { rowbase[j] = acc; acc += rowtot[j]; } }
    __syncthreads();
    // ---- move the records: a wave takes tiles wv, wv + 4, ... (their first 64 records are in registers already)
#pragma unroll
    for (int u = 0; u < kTilesPerWave; u++) {
      const int t = wv + 4 * u;
      if (t >= ntc) break;
      const int tt = (int)top[TH - 1][t] + (int)cnt[TH - 1][t];   // the tile's records
      for (int i0 = 0; i0 < tt; i0 += 64) {
        const int idx = i0 + lane;
        if (idx < tt) {
          const uint32_t x = i0 == 0 ? rec[u] : ld_u32_sc1(fz.stage + (size_t)(band * ntc + t) * kStagePerTile + idx);
          const int j = (x >> 6) & (TH - 1), col = x & 63;
          const uint32_t k = rowbase[j] + left[j][t] + (uint32_t)(idx - (int)top[j][t]);
          if ((int)k < fz.capacity) {
            fz.out_rc[2 * (size_t)k] = band * TH + j; fz.out_rc[2 * (size_t)k + 1] = t * TW + col;
            if (fz.out_scores) fz.out_scores[k] = (int32_t)(x >> 16) - 1;
          }
        }
      }
    }
    // ---- hand the control block back: the launch's last band zeroes it (every band has read what it needed of the bands above)
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(fz.finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1u ? 2u : 1u;
    __syncthreads();
    if (s_last == 2u) {
      for (int b = threadIdx.x; b < (int)gridDim.y; b += 256) {
        __hip_atomic_store(fz.band + (size_t)b * kBandStride, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fz.band + (size_t)b * kBandStride + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (threadIdx.x == 0) __hip_atomic_store(fz.finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  uint32_t nrow = 0;
  if (lane < TH / 4) {
    const int r = r0 + wv * (TH / 4) + lane;
    if (r < A.nr) {
      const unsigned long long w = words[wv][lane];
      bitmap[(size_t)r * ntc + blockIdx.x] = w;
      if (MODE == VPP_FAST9_RAW) { nrow = (uint32_t)__popcll(w); rowcnt[(size_t)r * ntc + blockIdx.x] = (uint8_t)nrow; }   // <= 64
    }
  }
  if (MODE == VPP_FAST9_RAW) {
    // RAW keeps every corner, so the ordered write needs no count pass of its own: this tile's corner total (and the per-row counts above) are
    // all the offsets are made of — fast9_write_rows_kernel sums the tiles of the bands above its row and the rows above it in its band
    __shared__ uint32_t wtot[4];
    nrow += __shfl_xor(nrow, 1); nrow += __shfl_xor(nrow, 2); nrow += __shfl_xor(nrow, 4);
    if (lane == 0) wtot[wv] = nrow;
    __syncthreads();
    if (threadIdx.x == 0) tiletot[(size_t)blockIdx.y * ntc + blockIdx.x] = (uint16_t)(wtot[0] + wtot[1] + wtot[2] + wtot[3]);   // <= 2048
  }
}

__device__ __forceinline__ int fast9_score_px(const DImg& A, int r, int c, int th) { return fast9_score_at(A, r, c, th); }  // fast.hpp:38-77 (tracker_device.hpp)

__global__ __launch_bounds__(256) void fast9_scores_list_kernel(DImg A, int th, const int32_t* __restrict__ rc, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = fast9_score_px(A, rc[2 * i], rc[2 * i + 1], th);  // fast.hpp:643-652
}

__global__ __launch_bounds__(256) void fast9_scores_moved_kernel(DImg A, int th, const int32_t* __restrict__ moved, const int32_t* __restrict__ prev,
                                                                 int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int r = moved[2 * i], c = moved[2 * i + 1];
  if (!A.has(r, c)) { r = prev[2 * i]; c = prev[2 * i + 1]; }  // video_extruder.hpp:50-53: an out-of-frame match removes the keypoint where it was
  out[i] = fast9_score_px(A, r, c, th);
}

// ---- ordered selection -----------------------------------------------------------------------------------
// Units are laid out in the reference's serial output order (row-major 64-px mask words for RAW / LOCAL_MAXIMA, row-major
// blocks for BLOCKWISE), one unit per thread, so a flat exclusive scan of the per-unit counts is the output index.
__device__ __forceinline__ uint32_t stored(uint32_t f) { return f ? (f - 1) >> 4 : 0; }  // scores_img value: score / 16 (fast.hpp:693)

// exclusive prefix of `v` over the 256 threads of the block (thread order); total returned through *tot
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* tot) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wv; w++) off += wsum[w];
  *tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return off + inc - v;
}

// Output offset of workgroup `nb` = sum of the per-workgroup counts before it.  Every write workgroup sums them itself (the
// array is a few KB and sits in L2), which replaces a separate single-workgroup scan launch between the two passes.  Measured
// alternatives (round 2, 4K RAW, 2025 workgroups): "last workgroup to finish scans" behind one atomic ticket: +40 us (same-address
// atomics); per-32-workgroup group sums by atomicAdd: count pass 5.0 -> 12.4 us, write pass unchanged — the re-summation is not
// what the write pass waits for.
__device__ __forceinline__ void publish_count(uint32_t* __restrict__ unit_count, uint32_t tot) {
  if (threadIdx.x == 0) unit_count[blockIdx.x] = tot;
}
__device__ __forceinline__ uint32_t group_offset(const uint32_t* __restrict__ unit_count, int nb) {
  __shared__ uint32_t gsum[4];
  uint32_t s = 0;
  for (int i = threadIdx.x; i < nb; i += 256) s += unit_count[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) gsum[threadIdx.x >> 6] = s;
  __syncthreads();
  const uint32_t r = gsum[0] + gsum[1] + gsum[2] + gsum[3];
  __syncthreads();
  return r;
}

// RAW / LOCAL_MAXIMA unit: one 16-px segment of the corner bitmap (a u16 of the u64 word the detect wave wrote), so the
// serial per-bit chains stay short and ~8k waves are in flight.
constexpr int SEG = 16;

// LOCAL_MAXIMA: keep the bits of a segment whose stored score is a strict maximum of its 8 neighbours (fast.hpp:907-921)
__device__ __forceinline__ uint32_t refine_local_maxima(const DImg& F, int r, int cbase, uint32_t m) {
  uint32_t keep = 0;
  const uint16_t *fm = F.row<uint16_t>(r - 1), *f0 = F.row<uint16_t>(r), *fp = F.row<uint16_t>(r + 1);
  while (m) {
    const int b = __ffs(m) - 1;
    m &= m - 1;
    const int c = cbase + b;
    const uint32_t a = stored(f0[c]);
    int is_max = 1;
    is_max &= a > stored(fm[c - 1]); is_max &= a > stored(fm[c]); is_max &= a > stored(fm[c + 1]);
    is_max &= a > stored(f0[c - 1]); is_max &= a > stored(f0[c + 1]);
    is_max &= a > stored(fp[c - 1]); is_max &= a > stored(fp[c]); is_max &= a > stored(fp[c + 1]);
    if (is_max) keep |= 1u << b;
  }
  return keep;
}

// pass 1, RAW / LOCAL_MAXIMA: per-segment keypoint count (LOCAL rewrites the segment with the surviving bits)
template <int MODE>
__global__ __launch_bounds__(256) void fast9_count_segs_kernel(DImg F, uint16_t* __restrict__ segs, int nsc, int nsegs,
                                                               uint32_t* __restrict__ unit_count) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  uint32_t m = u < nsegs ? segs[u] : 0;
  if (MODE == 1 && m) {
    const int r = u / nsc, sc = u - r * nsc;
    m = refine_local_maxima(F, r, sc * SEG, m);
    segs[u] = (uint16_t)m;
  }
  uint32_t tot;
  block_exscan((uint32_t)__popc(m), &tot);
  publish_count(unit_count, tot);
}

// pass 3, RAW / LOCAL_MAXIMA: ordered write
template <int MODE>
__global__ __launch_bounds__(256) void fast9_write_segs_kernel(DImg F, const uint16_t* __restrict__ segs, int nsc, int nsegs,
                                                               const uint32_t* __restrict__ unit_count, uint32_t* __restrict__ total,
                                                               int32_t* __restrict__ out_rc, int32_t* __restrict__ out_scores, int capacity) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  uint32_t m = u < nsegs ? segs[u] : 0;
  uint32_t tot;
  const uint32_t base = group_offset(unit_count, blockIdx.x);
  uint32_t k = base + block_exscan((uint32_t)__popc(m), &tot);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = base + tot;
  if (!m) return;
  const int r = u / nsc, cbase = (u - r * nsc) * SEG;
  const uint16_t* f0 = F.row<uint16_t>(r);   // (both 16-byte halves of the segment loaded ahead of the loop measured slower: 12.5 vs 11.2 us)
  while (m) {
    const int c = cbase + __ffs(m) - 1;
    m &= m - 1;
    if ((int)k < capacity) {
      out_rc[2 * (size_t)k] = r; out_rc[2 * (size_t)k + 1] = c;
      if (out_scores) out_scores[k] = MODE == 0 ? (int32_t)f0[c] - 1 : (int32_t)stored(f0[c]);
    }
    k++;
  }
}

// RAW after the two-phase detect kernel, which left the corner count of every (row, 64-px tile) in rowcnt and of every TH-row tile in
// tiletot: one workgroup per image row computes its output offset from those (the tiles of the bands above: <= 8 KB of u16 at 4K; the rows
// above it inside its band: <= 31 x ntc bytes) and writes the row's corners in column order — raw detection is two launches.
__global__ __launch_bounds__(256) void fast9_write_rows_kernel(DImg F, const uint16_t* __restrict__ segs, int nsc, int ntc, const uint8_t* __restrict__ rowcnt,
                                                               const uint16_t* __restrict__ tiletot, uint32_t* __restrict__ total,
                                                               int32_t* __restrict__ out_rc, int32_t* __restrict__ out_scores, int capacity) {
  __shared__ uint32_t gsum[4];
  const int r = blockIdx.x, band = r / TH;
  uint32_t s = 0;
  // (round 6) Everything the row needs first — its own corner words, two 16-byte pieces of the tile totals above, one of the row counts above — is requested in ONE memory
  // round trip (clamped addresses, results masked: no branch in front of a load); frames up to 4K need no more than that.  The loops used to follow each other, each
  // waiting for its own loads, and the corner words were only asked for behind the block sum: four dependent round trips in a launch that is little else.
  const int n16 = band * ntc;   // u16 tile totals of the bands above
  const int b0 = band * TH * ntc, n8 = r * ntc - b0;   // u8 counts of the rows above this one inside its band
  const uint4* t4 = (const uint4*)tiletot;
  const uint4* c4 = (const uint4*)(rowcnt + b0);
  const int t = threadIdx.x;
  const uint32_t m_first = segs[(size_t)r * nsc + min(t, nsc - 1)];
  const int nt4 = n16 / 8, nc4 = n8 / 16;
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  uint4 va = t4[min(t, max(nt4 - 1, 0))], vb = t4[min(t + 256, max(nt4 - 1, 0))], vc = c4[min(t, max(nc4 - 1, 0))];   // (tiletot / rowcnt: at least 16 bytes each)
  if (t >= nt4) va = z4;
  if (t + 256 >= nt4) vb = z4;
  if (t >= nc4) vc = z4;
  {  // both ranges start 16-byte aligned (the arrays are 256-byte aligned, TH * ntc is a multiple of 16): 16-byte loads, SWAR sums
    auto sum16 = [&](const uint4& v) { s += (v.x & 0xFFFFu) + (v.x >> 16) + (v.y & 0xFFFFu) + (v.y >> 16) + (v.z & 0xFFFFu) + (v.z >> 16) + (v.w & 0xFFFFu) + (v.w >> 16); };
    auto sum8 = [&](const uint4& v) { s = __builtin_amdgcn_sad_u8(v.x, 0u, s); s = __builtin_amdgcn_sad_u8(v.y, 0u, s); s = __builtin_amdgcn_sad_u8(v.z, 0u, s); s = __builtin_amdgcn_sad_u8(v.w, 0u, s); };
    sum16(va); sum16(vb); sum8(vc);
    for (int i = 512 + t; i < nt4; i += 256) sum16(t4[i]);
    for (int i = (n16 & ~7) + t; i < n16; i += 256) s += tiletot[i];
    for (int i = 256 + t; i < nc4; i += 256) sum8(c4[i]);
    for (int i = (n8 & ~15) + t; i < n8; i += 256) s += rowcnt[b0 + i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) gsum[threadIdx.x >> 6] = s;
  __syncthreads();
  uint32_t base = gsum[0] + gsum[1] + gsum[2] + gsum[3];
  const uint16_t* f0 = F.row<uint16_t>(r);
  for (int u0 = 0; u0 < nsc; u0 += 256) {   // 256 segments = 4096 px per step (one step up to 4K frames)
    const int u = u0 + threadIdx.x;
    uint32_t m = u < nsc ? (u0 == 0 ? m_first : segs[(size_t)r * nsc + u]) : 0;
    uint32_t tot;
    uint32_t k = base + block_exscan((uint32_t)__popc(m), &tot);
    base += tot;
    const int cbase = u * SEG;
    while (m) {
      const int c = cbase + __ffs(m) - 1;
      m &= m - 1;
      if ((int)k < capacity) {
        out_rc[2 * (size_t)k] = r; out_rc[2 * (size_t)k + 1] = c;
        if (out_scores) out_scores[k] = (int32_t)f0[c] - 1;
      }
      k++;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = base;
}

// pass 1, BLOCKWISE: first strict maximum of each bs x bs block in scan order (fast.hpp:770-789).  Only corner pixels (bits of
// the detect bitmap) can raise the maximum, so a lane takes one block row through its mask words; the rows of a block meet in
// an LDS max over the key (score << 32 | ~position): largest score first, then the earliest row-major position.
// A workgroup owns G = 256 / RB consecutive blocks (RB = min(bs, 256) lanes per block).
__global__ __launch_bounds__(256) void fast9_count_blocks_kernel(DImg F, const uint64_t* __restrict__ bitmap, int ntc, int bs, int nbc,
                                                                 int nblocks, int RB, int G, uint2* __restrict__ blkres,
                                                                 uint32_t* __restrict__ unit_count) {
  __shared__ unsigned long long key[256];
  key[threadIdx.x] = 0;
  __syncthreads();
  const int g = threadIdx.x / RB, row = threadIdx.x - g * RB;
  const int b = blockIdx.x * G + g;
  if (g < G && b < nblocks) {
    const int br = b / nbc, bc = b - br * nbc;
    const int r0 = br * bs, c0 = bc * bs, r1 = min(r0 + bs, F.nr), c1 = min(c0 + bs, F.nc);
    const int w0 = c0 >> 6, w1 = (c1 - 1) >> 6;
    unsigned long long best = 0;
    for (int r = r0 + row; r < r1; r += RB) {
      const uint16_t* f = F.row<uint16_t>(r);
      for (int w = w0; w <= w1; w++) {
        uint64_t m = bitmap[(size_t)r * ntc + w];
        const int lo = max(c0 - w * 64, 0), hi = min(c1 - w * 64, 64);  // bit range [lo, hi) of this word
        m >>= lo;
        if (hi - lo < 64) m &= (1ull << (hi - lo)) - 1;
        while (m) {
          const int c = w * 64 + lo + __ffsll((unsigned long long)m) - 1;
          m &= m - 1;
          const unsigned long long k = ((unsigned long long)stored(f[c]) << 32) | (0xFFFFFFFFu - (((uint32_t)r << 16) | (uint32_t)c));
          if ((k >> 32) && k > best) best = k;
        }
      }
    }
    if (best) atomicMax(&key[g], best);
  }
  __syncthreads();
  uint32_t has = 0;
  if ((int)threadIdx.x < G && blockIdx.x * G + (int)threadIdx.x < nblocks) {
    const unsigned long long k = key[threadIdx.x];
    has = k ? 1u : 0u;
    blkres[blockIdx.x * G + threadIdx.x] = make_uint2(0xFFFFFFFFu - (uint32_t)k, (uint32_t)(k >> 32));
  }
  uint32_t tot;
  block_exscan(has, &tot);
  publish_count(unit_count, tot);
}

// BLOCKWISE after the atomic-key detect: the units are the blocks themselves, 256 per workgroup
__global__ __launch_bounds__(256) void fast9_count_keys_kernel(const unsigned long long* __restrict__ blkkey, int nblocks, uint32_t* __restrict__ unit_count) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  uint32_t tot;
  block_exscan((b < nblocks && blkkey[b]) ? 1u : 0u, &tot);
  publish_count(unit_count, tot);
}
__global__ __launch_bounds__(256) void fast9_write_keys_kernel(unsigned long long* __restrict__ blkkey, int nblocks, const uint32_t* __restrict__ unit_count,
                                                               uint32_t* __restrict__ total, int32_t* __restrict__ out_rc,
                                                               int32_t* __restrict__ out_scores, int capacity) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long key = b < nblocks ? blkkey[b] : 0ull;
  uint32_t tot;
  const uint32_t base = group_offset(unit_count, blockIdx.x);
  const uint32_t k = base + block_exscan(key ? 1u : 0u, &tot);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = base + tot;
  if (key) blkkey[b] = 0ull;   // consumed: the next call finds the keys zeroed
  if (key && (int)k < capacity) {
    const uint32_t pos = 0xFFFFFFFFu - (uint32_t)key;
    out_rc[2 * (size_t)k] = (int32_t)(pos >> 16); out_rc[2 * (size_t)k + 1] = (int32_t)(pos & 0xFFFFu);
    if (out_scores) out_scores[k] = (int32_t)(key >> 32);
  }
}

__global__ __launch_bounds__(256) void fast9_write_blocks_kernel(const uint2* __restrict__ blkres, int nblocks, int G,
                                                                 const uint32_t* __restrict__ unit_count, uint32_t* __restrict__ total,
                                                                 int32_t* __restrict__ out_rc, int32_t* __restrict__ out_scores, int capacity) {
  const int b = blockIdx.x * G + threadIdx.x;
  const uint2 res = ((int)threadIdx.x < G && b < nblocks) ? blkres[b] : make_uint2(0, 0);
  uint32_t tot;
  const uint32_t base = group_offset(unit_count, blockIdx.x);
  const uint32_t k = base + block_exscan(res.y > 0 ? 1u : 0u, &tot);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = base + tot;
  if (res.y > 0 && (int)k < capacity) {
    out_rc[2 * (size_t)k] = (int32_t)(res.x >> 16); out_rc[2 * (size_t)k + 1] = (int32_t)(res.x & 0xFFFFu);
    if (out_scores) out_scores[k] = (int32_t)res.y;
  }
}

// ---- FAST_internals::fast_detector9(A, B, th) (fast.hpp:511-551): dense corner flags on the TRUE ring, plain int compares ----
// (a > v + th) / (a < v - th) without saturation; the reference's 2-bit interleaved code + fast9_check_code (fast.hpp:25-35) is
// "9 contiguous of 16, circular" on each plane, which is what nine_contiguous tests on the two 16-bit masks.
template <class U>
__global__ __launch_bounds__(256) void fast9_dense_kernel(DImg B, DImg A, int th) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (c >= A.nc || r >= A.nr) return;
  const uint8_t* p = A.row<uint8_t>(r) + c;
  const int v = p[0], hi = v + th, lo = v - th;
  uint32_t mb = 0, md = 0;
#pragma unroll
  for (int i = 15; i >= 0; i--) {
    const int a = p[(ptrdiff_t)ring_dr<false>(i) * A.pitch + ring_dc(i)];
    mb = push_sign(mb, hi - a);  // a > v + th
    md = push_sign(md, a - lo);  // a < v - th
  }
  B.row<U>(r)[c] = (U)((nine_contiguous(mb) || nine_contiguous(md)) ? 1 : 0);
}

// ---- blockwise_maxima_filter(A, block_size) (fast.hpp:577-614), in place: every pixel of a block is zeroed, then the FIRST
// strict maximum (> 0, row-major scan of the block) is written back.  One thread per block.
template <class V>
__global__ __launch_bounds__(256) void blockwise_maxima_kernel(DImg A, int bs, int nbr, int nbc) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nbr * nbc) return;
  const int br = b / nbc, bc = b - br * nbc;
  const int r0 = br * bs, c0 = bc * bs, r1 = min(r0 + bs, A.nr), c1 = min(c0 + bs, A.nc);
  V vmax = 0; int pr = 0, pc = 0;
  for (int r = r0; r < r1; r++) {
    V* row = A.row<V>(r);
    for (int c = c0; c < c1; c++) {
      const V v = row[c];
      row[c] = 0;
      if (v > vmax) { vmax = v; pr = r; pc = c; }
    }
  }
  if (vmax > 0) A.row<V>(pr)[pc] = vmax;
}

// The same filter with the image's rows read and written as rows (round 6).  A workgroup owns one row of blocks x a span of BPW whole blocks (<= 256 dwords of a row):
// a thread takes one dword — 4 / 2 / 1 pixels — of each of the block row's bs rows, coalesced; every positive pixel raises its block's key {value bits, ~position in
// the block's scan order} with one LDS atomic max (largest value, earliest position: the reference's strict `>` in row-major order; positive floats order like their
// bits); after a barrier the thread writes its dwords back — zero but for a winner that falls into them.  One lane per block read a 4K frame's 10 x 10 blocks as ten 10-byte
// segments 3 840 bytes apart and wrote them the same way: 23 us for 16.6 MB.
template <class V>
__global__ __launch_bounds__(256) void blockwise_maxima_rows_kernel(DImg A, int bs, int nbc, int BPW) {
  constexpr int NPX = 4 / (int)sizeof(V);
  __shared__ unsigned long long keys[256];   // BPW <= 256 NPX / bs <= 256 (bs >= 4)
  const int br = blockIdx.y, r0 = br * bs, r1 = min(r0 + bs, A.nr);
  const int bc0 = blockIdx.x * BPW, nb = min(BPW, nbc - bc0);          // this workgroup's blocks
  const int span0 = bc0 * bs, span1 = min(span0 + nb * bs, A.nc);      // and their columns
  const int t = threadIdx.x, c = span0 + t * NPX;
  if (t < nb) keys[t] = 0ull;
  __syncthreads();
  const int n = min(NPX, span1 - c);                                    // pixels of this thread's dword inside the span (<= 0: none)
  int lb[NPX], lc[NPX];                                                 // per pixel: block within the span, column within the block
#pragma unroll
  for (int k = 0; k < NPX; k++) { lb[k] = (t * NPX + k) / bs; lc[k] = t * NPX + k - lb[k] * bs; }
  auto bits = [](V v) -> uint32_t { if constexpr (std::is_same<V, float>::value) return __float_as_uint(v); else return (uint32_t)v; };
  if (n > 0) {
    // rows in batches of 8: the batch's loads are all requested before the first is looked at (a runtime-trip loop of load -> test -> atomic ran one memory round
    // trip per row; keeping a thread's per-block maxima in registers instead of the LDS atomics measured slower: 10.4 -> 12.4 us on a 4K u8 frame)
    for (int rb = r0; rb < r1; rb += 8) {
      V v[8][NPX];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const V* row = A.row<V>(min(rb + q, r1 - 1)) + c;
        if (n == NPX) __builtin_memcpy(v[q], row, 4);
        else { for (int k = 0; k < NPX; k++) v[q][k] = k < n ? row[k] : V(0); }
      }
#pragma unroll
      for (int q = 0; q < 8; q++)
#pragma unroll
        for (int k = 0; k < NPX; k++)
          if (rb + q < r1 && k < n && v[q][k] > V(0))
            atomicMax(&keys[lb[k]], ((unsigned long long)bits(v[q][k]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)((rb + q - r0) * bs + lc[k])));
    }
  }
  __syncthreads();
  if (n <= 0) return;
  // the winners that fall into this thread's dword: at most one per pixel's block
  int wr[NPX]; uint32_t wv[NPX];
#pragma unroll
  for (int k = 0; k < NPX; k++) {
    wr[k] = -1; wv[k] = 0;
    if (k < n) {
      const unsigned long long key = keys[lb[k]];
      if (key) { const uint32_t pos = 0xFFFFFFFFu - (uint32_t)key; const int pr = (int)(pos / (uint32_t)bs), pc = (int)pos - pr * bs; if (pc == lc[k]) { wr[k] = pr; wv[k] = (uint32_t)(key >> 32); } }
    }
  }
  auto from_bits = [](uint32_t b) -> V { if constexpr (std::is_same<V, float>::value) return __uint_as_float(b); else return (V)b; };
  for (int r = r0; r < r1; r++) {
    V* row = A.row<V>(r) + c;
    V v[NPX];
#pragma unroll
    for (int k = 0; k < NPX; k++) v[k] = wr[k] == r - r0 ? from_bits(wv[k]) : V(0);
    if (n == NPX) __builtin_memcpy(row, v, 4);
    else { for (int k = 0; k < n; k++) row[k] = v[k]; }
  }
}

thread_local Scratch g_scratch;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" {

int vpp_fast9_scores(const vpp_image_desc* src, int th, const int32_t* rc, int n, int32_t* out_scores, void* stream) {
  VPP_REQUIRE(valid_desc(src) && rc && out_scores && n >= 0, VPP_ERR_INVALID_ARG, "vpp_fast9_scores: invalid argument");
  VPP_REQUIRE(src->dtype == VPP_U8 && src->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_fast9_scores: u8 x1 only");
  VPP_REQUIRE(src->border >= 3, VPP_ERR_BORDER_TOO_SMALL, "Image need a border of 3px at least for the FAST detector");
  if (n == 0) return VPP_OK;
  fast9_scores_list_kernel<<<(n + 255) / 256, 256, 0, as_stream(stream)>>>(dimg(src), th, rc, n, out_scores);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_fast9_dense(const vpp_image_desc* dst, const vpp_image_desc* src, int th, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(src) && same_domain(dst, src), VPP_ERR_INVALID_ARG, "vpp_fast9_dense: invalid descriptors / domain mismatch");
  VPP_REQUIRE(src->dtype == VPP_U8 && src->channels == 1 && dst->channels == 1 && (dst->dtype == VPP_U8 || dst->dtype == VPP_I32), VPP_ERR_UNSUPPORTED,
              "vpp_fast9_dense: u8 x1 source, u8 or int32 x1 destination");
  VPP_REQUIRE(src->border >= 3, VPP_ERR_BORDER_TOO_SMALL, "vpp_fast9_dense: the ring reads 3 px beyond the domain (fast.hpp:520-541), border is %d", src->border);
  VPP_REQUIRE(th > -(1 << 30) && th < (1 << 30), VPP_ERR_INVALID_ARG, "vpp_fast9_dense: threshold out of range");
  dim3 grid((src->ncols + 63) / 64, (src->nrows + 3) / 4);
  // thresholds 0 ... 255 (where plain and saturated compares agree) on dword- / 16-byte-aligned destinations: the detector's two-phase kernel with the flag image as its
  // result — the 4-sample pre-test on 4 packed pixels per lane, the ring test on the LDS-compacted survivors only (4K: 28.6 us with the ring test on every pixel)
  if (th >= 0 && th <= 255 && tuning("fast9.dense_two_phase", 1) &&
      ((uintptr_t)dst->first_pixel % (dst->dtype == VPP_U8 ? 4 : 16)) == 0 && dst->pitch % (dst->dtype == VPP_U8 ? 4 : 16) == 0) {
    const dim3 g2((unsigned)((src->ncols + TW - 1) / TW), (unsigned)((src->nrows + TH - 1) / TH));
    const DImg A = dimg(src), D = dimg(dst);
    if (dst->dtype == VPP_U8) fast9_detect2_kernel<false, kDenseU8><<<g2, 256, 0, as_stream(stream)>>>(A, A, 0, th, D, nullptr, (int)g2.x, nullptr, 0u, 0, nullptr, nullptr, RawFuse{});
    else fast9_detect2_kernel<false, kDenseI32><<<g2, 256, 0, as_stream(stream)>>>(A, A, 0, th, D, nullptr, (int)g2.x, nullptr, 0u, 0, nullptr, nullptr, RawFuse{});
    VPP_LAUNCH_CHECK();
    return VPP_OK;
  }
  if (dst->dtype == VPP_U8) fast9_dense_kernel<uint8_t><<<grid, 256, 0, as_stream(stream)>>>(dimg(dst), dimg(src), th);
  else fast9_dense_kernel<int32_t><<<grid, 256, 0, as_stream(stream)>>>(dimg(dst), dimg(src), th);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_blockwise_maxima_filter(const vpp_image_desc* img, int block_size, void* stream) {
  VPP_REQUIRE(valid_desc(img) && block_size > 0, VPP_ERR_INVALID_ARG, "vpp_blockwise_maxima_filter: invalid argument");
  VPP_REQUIRE(img->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_blockwise_maxima_filter: scalar images only");
  const int nbr = (img->nrows + block_size - 1) / block_size, nbc = (img->ncols + block_size - 1) / block_size;
  const unsigned grid = (unsigned)(((long long)nbr * nbc + 255) / 256);
  DImg A = dimg(img);
  hipStream_t st = as_stream(stream);
  {  // rows as rows: blocks of 4 ... 256 dwords' worth of pixels (a workgroup's span is whole blocks); narrower / wider blocks keep one lane per block
    const int npx = 4 / dtype_size(img->dtype);
    if (block_size >= 4 && block_size <= 256 * npx && nbr <= 65535 && tuning("blockwise_maxima.rows", 1)) {
      const int BPW = (256 * npx) / block_size;
      const dim3 g((unsigned)((nbc + BPW - 1) / BPW), (unsigned)nbr);
      switch (img->dtype) {
        case VPP_U8: blockwise_maxima_rows_kernel<uint8_t><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        case VPP_I8: blockwise_maxima_rows_kernel<int8_t><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        case VPP_U16: blockwise_maxima_rows_kernel<uint16_t><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        case VPP_I16: blockwise_maxima_rows_kernel<int16_t><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        case VPP_I32: blockwise_maxima_rows_kernel<int32_t><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        case VPP_U32: blockwise_maxima_rows_kernel<uint32_t><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        case VPP_F32: blockwise_maxima_rows_kernel<float><<<g, 256, 0, st>>>(A, block_size, nbc, BPW); break;
        default: VPP_REQUIRE(false, VPP_ERR_UNSUPPORTED, "vpp_blockwise_maxima_filter: unsupported dtype %d", img->dtype);
      }
      VPP_LAUNCH_CHECK();
      return VPP_OK;
    }
  }
  switch (img->dtype) {
    case VPP_U8: blockwise_maxima_kernel<uint8_t><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    case VPP_I8: blockwise_maxima_kernel<int8_t><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    case VPP_U16: blockwise_maxima_kernel<uint16_t><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    case VPP_I16: blockwise_maxima_kernel<int16_t><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    case VPP_I32: blockwise_maxima_kernel<int32_t><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    case VPP_U32: blockwise_maxima_kernel<uint32_t><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    case VPP_F32: blockwise_maxima_kernel<float><<<grid, 256, 0, st>>>(A, block_size, nbr, nbc); break;
    default: VPP_REQUIRE(false, VPP_ERR_UNSUPPORTED, "vpp_blockwise_maxima_filter: unsupported dtype %d", img->dtype);
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_fast9_scores_moved(const vpp_image_desc* src, int th, const int32_t* rc_moved, const int32_t* rc_prev, int n, int32_t* out_scores, void* stream) {
  VPP_REQUIRE(valid_desc(src) && rc_moved && rc_prev && out_scores && n >= 0, VPP_ERR_INVALID_ARG, "vpp_fast9_scores_moved: invalid argument");
  VPP_REQUIRE(src->dtype == VPP_U8 && src->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_fast9_scores_moved: u8 x1 only");
  VPP_REQUIRE(src->border >= 3, VPP_ERR_BORDER_TOO_SMALL, "Image need a border of 3px at least for the FAST detector");
  if (n == 0) return VPP_OK;
  fast9_scores_moved_kernel<<<(n + 255) / 256, 256, 0, as_stream(stream)>>>(dimg(src), th, rc_moved, rc_prev, n, out_scores);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

// Everything of a detection up to and including the ordered write, queued on `stream`; the keypoint total (not clamped to the capacity) lands
// in *total_dev, a device-visible word (HBM or pinned host memory).  No synchronisation.
static int fast9_enqueue(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size, int compat,
                         int32_t* out_rc, int32_t* out_scores, int capacity, uint32_t* total_dev, void* stream) {
  VPP_REQUIRE(valid_desc(src) && total_dev && capacity >= 0 && (out_rc || capacity == 0), VPP_ERR_INVALID_ARG, "vpp_fast9_detect: invalid argument");
  VPP_REQUIRE(src->dtype == VPP_U8 && src->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_fast9_detect: u8 x1 only");
  VPP_REQUIRE(src->border >= 3, VPP_ERR_BORDER_TOO_SMALL, "Image need a border of 3px at least for the FAST detector");  // fast.hpp:937-938
  VPP_REQUIRE(src->nrows < 65536 && src->ncols < 65536, VPP_ERR_UNSUPPORTED, "vpp_fast9_detect: image larger than 65535 px");
  VPP_REQUIRE(mode >= VPP_FAST9_RAW && mode <= VPP_FAST9_BLOCKWISE, VPP_ERR_INVALID_ARG, "vpp_fast9_detect: bad mode %d", mode);
  VPP_REQUIRE(mode != VPP_FAST9_BLOCKWISE || block_size > 0, VPP_ERR_INVALID_ARG, "vpp_fast9_detect: block_size must be > 0");
  if (mask) {
    VPP_REQUIRE(valid_desc(mask) && mask->dtype == VPP_U8 && mask->channels == 1 && same_domain(mask, src), VPP_ERR_INVALID_ARG,
                "vpp_fast9_detect: mask must be u8 x1 over the same domain");
  }
  hipStream_t st = as_stream(stream);
  const int nr = src->nrows, nc = src->ncols;
  // scratch layout: [header][F: u16 map with border 1][bitmap: one u64 per 64-px row segment][blkres / block keys][unit_count]
  int32_t fpitch; size_t fbytes, ffirst;
  vpp_image_layout(nr, nc, 2, 1, 16, &fpitch, &fbytes, &ffirst);
  const int ntc = (nc + TW - 1) / TW, nwords = nr * ntc;
  const int nbr = mode == VPP_FAST9_BLOCKWISE ? (nr + block_size - 1) / block_size : 0;
  const int nbc = mode == VPP_FAST9_BLOCKWISE ? (nc + block_size - 1) / block_size : 0;
  const long long nblocks_ll = (long long)nbr * nbc;
  VPP_REQUIRE(nblocks_ll < (1ll << 31), VPP_ERR_UNSUPPORTED, "vpp_fast9_detect: too many blocks");
  const int nblocks = (int)nblocks_ll;
  const int RB = block_size < 256 ? (block_size > 0 ? block_size : 1) : 256, G = 256 / RB;   // BLOCKWISE: lanes per block, blocks per workgroup
  const int nsc = ntc * (TW / SEG), nsegs = nr * nsc;                                        // RAW / LOCAL: 16-px segments
  const int ngroups = mode == VPP_FAST9_BLOCKWISE ? (nblocks + G - 1) / G : (nsegs + 255) / 256;
  const int impl = tuning("fast9.impl", 2);   // 2 = two-phase (compacted candidates), 1 = one lane per pixel all the way
  const bool keyed = impl == 2 && mode == VPP_FAST9_BLOCKWISE && tuning("fast9.block_keys", 1);
  const size_t off_f = 256;
  const size_t off_bm = off_f + align_up(fbytes, 256), off_br = off_bm + align_up((size_t)nwords * 8, 256);
  const size_t off_uc = off_br + align_up((size_t)nblocks * 8, 256);
  const bool fused_raw = impl == 2 && mode == VPP_FAST9_RAW && tuning("fast9.raw_rows", 1);   // RAW: the detect kernel leaves the counts, no count pass
  const int nby = (nr + TH - 1) / TH;
  const bool raw2 = impl == 2 && mode == VPP_FAST9_RAW;   // the RAW instance of the two-phase kernel always writes its counts
  const size_t off_rc = off_uc + align_up((size_t)ngroups * 4, 256), off_tt = off_rc + align_up(raw2 ? (size_t)nwords : 0, 256);
  // RAW in ONE launch (round 6): the detect kernel writes the ordered list itself (fast9_detect2_kernel<.., RAW, true>); its control block + row counts follow the rest
  // OFF by default — measured (round 6, 4K frame with 183 k corners, same box, synchronous / 50 recorded calls): two launches 41.7 / 29.3 us per call; this form 98 us
  // with every tile waiting for its band's prefix, 64 / 54 us as stage + band gather with the bands' counters side by side, 51 / 38 us with one atomic per tile on
  // words 256 bytes apart, 47 / 35 us (incl. a 4.5 us fill node when recorded) with all of the gather's loads requested up front.  The last band's gather — one
  // workgroup, a look-back, a table build and ~2 700 records — is a tail as long as the write launch it replaces, and every tile pays a returning atomic before it
  // frees its slot.  Kept, parity-tested (tests/test_gpu_algos.py), as tuning fast9.raw_fused = 1.
  const bool fused_write = raw2 && ntc <= kFuseMaxTiles && tuning("fast9.raw_fused", 0);
  const int ntcp = (ntc + 7) / 8 * 8;
  const size_t off_fz = off_tt + align_up(raw2 ? (size_t)nby * ntc * 2 : 0, 256);
  // [control block | row counts] are kept zeroed between calls; the staging area behind them (8 KB per tile, touched only where there are corners) is not
  const size_t fz_ctl = 256 + (size_t)nby * kBandStride * 8, fz_bytes = fused_write ? fz_ctl + align_up((size_t)nby * TH * ntcp, 256) : 0;
  const size_t off_stage = off_fz + fz_bytes;
  const size_t total_bytes = off_stage + (fused_write ? (size_t)nby * ntc * kStagePerTile * 4 : 0);
  int rc = g_scratch.ensure(total_bytes, st);
  if (rc != VPP_OK) return rc;
  uint8_t* base = (uint8_t*)g_scratch.p;
  uint32_t* d_total = total_dev;
  DImg F{base + off_f + ffirst, nr, nc, fpitch, 1, VPP_U16, 1};
  uint64_t* bitmap = (uint64_t*)(base + off_bm);
  uint2* blkres = (uint2*)(base + off_br);
  uint32_t* unit_count = (uint32_t*)(base + off_uc);
  uint8_t* rowcnt = base + off_rc;
  uint16_t* tiletot = (uint16_t*)(base + off_tt);
  DImg A = dimg(src), M = mask ? dimg(mask) : A;
  dim3 grid(ntc, (nr + TH - 1) / TH);
  unsigned long long* blkkey = (unsigned long long*)blkres;
  // a note kept with this scratch buffer (cleared when it is reallocated): user[0] = signature of the key area that the last keyed call
  // left all-zero (fast9_write_keys_kernel zeroes every key it consumes, so back-to-back blockwise calls need no memset node); any other
  // call may lay the buffer out differently and clears it
  Scratch::Slot& sl = *g_scratch.cur;
  const unsigned long long key_sig = keyed ? (((unsigned long long)off_br << 24) ^ (unsigned long long)nblocks) + 1ull : 0ull;
  // under stream capture nothing runs now: a graph must carry its own memset (it may be replayed after any other call has used the key
  // area) and must not leave a note about a state it did not produce
  // (Scratch::ensure has looked: a buffer that a capture recorded work on — now or ever — takes and believes no notes)
  const bool keys_clean = keyed && sl.note(0, key_sig);
  sl.set_note(0, 0);   // until this call's write pass is queued (an error return in between must not leave a wrong note)
  const uint32_t bs_magic = block_size >= 2 ? (uint32_t)((1ull << 32) / (unsigned)block_size) + 1u : 0u;   // x / bs = umulhi(x, magic), exact for x < 2^16
  if (keyed && !keys_clean) { const int rf = device_fill(blkkey, 0, (size_t)nblocks * 8, st); if (rf != VPP_OK) return rf; }
  if (fused_write) {
    // user[1]: the control block (and the padding bytes of the row counts) of THIS layout are all-zero — the launch's last tile hands them back that way, so
    // back-to-back raw calls need no fill; any other layout, a recorded call (notes are not believed on a buffer a capture has recorded on) or a device-side fault
    // (check_device_error bumps the notes' epoch) gets the fill
    const unsigned long long fz_sig = (((unsigned long long)off_fz << 24) ^ ((unsigned long long)nby << 12) ^ (unsigned long long)ntcp) + 1ull;
    const bool fz_clean = sl.note(1, fz_sig);
    sl.set_note(1, 0);
    if (!fz_clean) { const int rf = device_fill(base + off_fz, 0, fz_bytes, st); if (rf != VPP_OK) return rf; }
    uint8_t* q = base + off_fz;
    RawFuse fz{};
    fz.finished = (uint32_t*)q; q += 256;
    fz.band = (unsigned long long*)q; q += (size_t)nby * kBandStride * 8;
    fz.rowcnt = q; fz.ntcp = ntcp;
    fz.stage = (uint32_t*)(base + off_stage);
    fz.total = d_total; fz.out_rc = out_rc; fz.out_scores = out_scores; fz.capacity = capacity; fz.err = device_error_word();
    if (compat == VPP_FAST9_REFERENCE) fast9_detect2_kernel<true, VPP_FAST9_RAW, true><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc, blkkey, bs_magic, nbc, rowcnt, tiletot, fz);
    else fast9_detect2_kernel<false, VPP_FAST9_RAW, true><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc, blkkey, bs_magic, nbc, rowcnt, tiletot, fz);
    VPP_LAUNCH_CHECK();
    sl.set_note(1, fz_sig);
    return VPP_OK;
  }
  if (impl == 2) {
#define VPP_FAST_DETECT2(R)                                                                                                                             \
    if (keyed) fast9_detect2_kernel<R, VPP_FAST9_BLOCKWISE><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc, blkkey, bs_magic, nbc, rowcnt, tiletot, RawFuse{});           \
    else if (mode == VPP_FAST9_RAW) fast9_detect2_kernel<R, VPP_FAST9_RAW><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc, blkkey, bs_magic, nbc, rowcnt, tiletot, RawFuse{}); \
    else fast9_detect2_kernel<R, VPP_FAST9_LOCAL_MAXIMA><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc, blkkey, bs_magic, nbc, rowcnt, tiletot, RawFuse{});
    if (compat == VPP_FAST9_REFERENCE) { VPP_FAST_DETECT2(true) } else { VPP_FAST_DETECT2(false) }
#undef VPP_FAST_DETECT2
  } else if (compat == VPP_FAST9_REFERENCE) fast9_detect_kernel<true><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc);
  else fast9_detect_kernel<false><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F, bitmap, ntc);
  VPP_LAUNCH_CHECK();
  if (keyed) {
    const int ng = (nblocks + 255) / 256;   // <= ngroups (G <= 256): unit_count is large enough
    fast9_count_keys_kernel<<<ng, 256, 0, st>>>(blkkey, nblocks, unit_count);
    fast9_write_keys_kernel<<<ng, 256, 0, st>>>(blkkey, nblocks, unit_count, d_total, out_rc, out_scores, capacity);
    if (hipPeekAtLastError() == hipSuccess) sl.set_note(0, key_sig);   // every key that was raised is zero again once this kernel has run
  } else if (mode == VPP_FAST9_BLOCKWISE) {
    fast9_count_blocks_kernel<<<ngroups, 256, 0, st>>>(F, bitmap, ntc, block_size, nbc, nblocks, RB, G, blkres, unit_count);
    fast9_write_blocks_kernel<<<ngroups, 256, 0, st>>>(blkres, nblocks, G, unit_count, d_total, out_rc, out_scores, capacity);
  } else if (fused_raw) {
    fast9_write_rows_kernel<<<nr, 256, 0, st>>>(F, (const uint16_t*)bitmap, nsc, ntc, rowcnt, tiletot, d_total, out_rc, out_scores, capacity);
  } else if (mode == VPP_FAST9_RAW) {
    fast9_count_segs_kernel<0><<<ngroups, 256, 0, st>>>(F, (uint16_t*)bitmap, nsc, nsegs, unit_count);
    fast9_write_segs_kernel<0><<<ngroups, 256, 0, st>>>(F, (const uint16_t*)bitmap, nsc, nsegs, unit_count, d_total, out_rc, out_scores, capacity);
  } else {
    fast9_count_segs_kernel<1><<<ngroups, 256, 0, st>>>(F, (uint16_t*)bitmap, nsc, nsegs, unit_count);
    fast9_write_segs_kernel<1><<<ngroups, 256, 0, st>>>(F, (const uint16_t*)bitmap, nsc, nsegs, unit_count, d_total, out_rc, out_scores, capacity);
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_fast9_detect_async(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size, int compat,
                           int32_t* out_rc, int32_t* out_scores, int capacity, uint32_t* count_dev, void* stream) {
  return fast9_enqueue(src, th, mask, mode, block_size, compat, out_rc, out_scores, capacity, count_dev, stream);
}

int vpp_fast9_detect(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size, int compat,
                     int32_t* out_rc, int32_t* out_scores, int capacity, int* count, void* stream) {
  VPP_REQUIRE(count, VPP_ERR_INVALID_ARG, "vpp_fast9_detect: invalid argument");
  // the keypoint total lands in a pinned, device-visible host word (one per host thread): a stream sync replaces the 4-byte D2H copy
  thread_local uint32_t* t_total = nullptr;
  if (!t_total) { void* h = nullptr; int hs = vpp_malloc_host(64, &h); if (hs != VPP_OK) return hs; t_total = (uint32_t*)h; }
  uint32_t* d_total = t_total;
  const int rc = fast9_enqueue(src, th, mask, mode, block_size, compat, out_rc, out_scores, capacity, d_total, stream);
  if (rc != VPP_OK) return rc;
  VPP_HIP_TRY(hipStreamSynchronize(as_stream(stream)));   // the total was written straight into pinned host memory by the scan kernel: no copy
  const uint32_t total = *(volatile uint32_t*)d_total;
  *count = (int)total;
  if ((int)total > capacity) {
    set_error("vpp_fast9_detect: %u keypoints found, output capacity %d", total, capacity);
    return VPP_ERR_CAPACITY;
  }
  return VPP_OK;
}

}  // extern "C"
