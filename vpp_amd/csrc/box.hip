// box.hip — K2: neighbourhood (box_nbh2d / relative_access) mean filter.
// Reference: vpp/core/pixel_wise.hpp:14-25,57-63 (relative_access range), vpp/core/relative_accessor.hh:26-33
// (nbh(dr,dc) = line[dr][col+dc]), kernel lambda of benchmarks/box_5x5_filter2.cc:73-80 / examples/box_filter.cc:23-32.
//
// Two kernels:
//  * box5x5_u8_lds_kernel<CH> — the headline path (vuchar3 4K).  A row of CH-interleaved u8 pixels is a row of bytes in
//    which the horizontal neighbour sits CH bytes away, so the filter is a 1-channel 5x5 stencil with column stride CH.
//    A 256-thread workgroup stages a (TH+4) x (1024+32) byte tile in LDS with 16-B coalesced loads issued all at once
//    (one HBM round trip per workgroup, vertical halo shared by the four waves), then each wave produces TH/4 rows:
//    one lane owns 16 consecutive output bytes (one 16-B store per row) and keeps the running 5-row column sums of its
//    32-byte window as packed u16 pairs (even/odd bytes split with 0x00FF00FF masks); rows enter / leave the sum from
//    LDS (ds_read_b64 + b128 + b64 per row).  Horizontal taps are register selects + v_alignbit on those pairs; packed
//    sums never exceed 25*255 = 6375 < 2^16 so plain 32-bit adds/subs act as two independent 16-bit lanes.  Exact
//    truncating /25: (x * 671089) >> 24 for x <= 6375.  HBM traffic: 1 read + 1 write per byte (6 B/px for vuchar3);
//    the (TH+4)/TH vertical re-read is served by L2/MALL; row blocks are XCD-remapped so vertical neighbours share an L2.
//  * box_generic_kernel<T,S> — any dtype / window: LDS tile (+halo), taps summed in row-major order in the promoted
//    type (order matters for float), C++ `/ (R*C)`.
#include "common.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <map>
#include <type_traits>
#include <vector>
using namespace vpp_amd;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- fast path -------------------------------------------------------------------------------------------
// packed u16 pairs over the 32-B window: e[i] = bytes (4i, 4i+2), o[i] = bytes (4i+1, 4i+3)

// Dword at byte offset `off` of a row whose addressable bytes are [lo, hi), hi - lo >= 4: bytes outside read as 0.
// Branch-free: one (possibly unaligned) dword load from the offset clamped into [lo, hi - 4], then the loaded bytes are
// shifted to where they belong; a dword wholly outside shifts to 0.
__device__ __forceinline__ uint32_t ld_dword_guarded(const uint8_t* __restrict__ p, int off, int lo, int hi) {
  const int a = min(max(off, lo), hi - 4);
  uint32_t v;
  __builtin_memcpy(&v, p + a, 4);
  const int sh = a - off;                       // > 0: loaded later bytes -> move them up; < 0: loaded earlier bytes -> move down
  const uint32_t up = sh >= 4 ? 0u : v << (8 * (sh & 3)), dn = sh <= -4 ? 0u : v >> (8 * (-sh & 3));
  return sh >= 0 ? up : dn;
}

// 16-byte chunk at byte offset gx of a row; GUARD: dwords not wholly inside [lo, hi) are assembled bytewise.
template <bool GUARD>
__device__ __forceinline__ u32x4 ld_chunk(const uint8_t* __restrict__ row, int gx, int lo, int hi) {
  if (!GUARD || (gx >= lo && gx + 16 <= hi)) return *(const u32x4*)(row + gx);
  u32x4 v;
  v.x = ld_dword_guarded(row, gx, lo, hi); v.y = ld_dword_guarded(row, gx + 4, lo, hi);
  v.z = ld_dword_guarded(row, gx + 8, lo, hi); v.w = ld_dword_guarded(row, gx + 12, lo, hi);
  return v;
}

// packed pair (col[q], col[q+2]) of the window's column sums, q compile-time.
template <int Q> __device__ __forceinline__ uint32_t pair_at(const uint32_t* E, const uint32_t* O) {
  static_assert(Q >= 0 && Q + 2 < 32, "window overrun");
  constexpr int m = Q / 4;
  if constexpr (Q % 4 == 0) return E[m];
  else if constexpr (Q % 4 == 1) return O[m];
  else if constexpr (Q % 4 == 2) return __builtin_amdgcn_alignbit(E[m + 1], E[m], 16);
  else return __builtin_amdgcn_alignbit(O[m + 1], O[m], 16);
}
// horizontal sum of KC taps at stride CH around window position Q (packed pairs)
template <int Q, int CH, int KC = 5, int T = 0> __device__ __forceinline__ uint32_t hsum(const uint32_t* E, const uint32_t* O) {
  if constexpr (T == KC) return 0u;
  else return pair_at<Q + (T - KC / 2) * CH>(E, O) + hsum<Q, CH, KC, T + 1>(E, O);
}
// x * M with the exact quotient floor(x / N) in byte 3, N = window area: M = ceil(2^24 / N); for x <= 255 N the product stays
// below 2^32 and floor(x M / 2^24) == x / N (checked exhaustively for N = 3, 5, 7, 9, 15, 21, 25, 35, 49; N = 25: 671089).
template <int N> __device__ __forceinline__ uint32_t mul_div(uint32_t x) {
  constexpr uint32_t M = ((1u << 24) + N - 1) / N;
  static_assert((unsigned long long)M * N - (1ull << 24) < (1ull << 24) / (255ull * N), "rounding error of the reciprocal reaches the quotient");
  static_assert(255ull * N * M < (1ull << 32) && M < (1u << 24), "product overflows");
  return (uint32_t)__umul24(x, M);
}
template <int I, int CH, int KC = 5, int N = 25> __device__ __forceinline__ uint32_t out_dword(const uint32_t* E, const uint32_t* O) {
  // output bytes 4I..4I+3 of the lane's 16 (window positions 8+4I ..)
  const uint32_t se = hsum<8 + 4 * I, CH, KC>(E, O);      // sums for bytes (4I, 4I+2)
  const uint32_t so = hsum<8 + 4 * I + 1, CH, KC>(E, O);  // sums for bytes (4I+1, 4I+3)
  const uint32_t qa = mul_div<N>(se & 0xFFFFu), qb = mul_div<N>(so & 0xFFFFu), qc = mul_div<N>(se >> 16), qd = mul_div<N>(so >> 16);
  // gather byte 3 of each product: v_perm_b32 selects from {S0 = bytes 7..4, S1 = bytes 3..0}; 0x0c = constant 0
  return __builtin_amdgcn_perm(qb, qa, 0x0c0c0703u) | __builtin_amdgcn_perm(qd, qc, 0x07030c0cu);
}

constexpr int kTileW = 1024;               // output bytes per workgroup row = 64 lanes x 16 B
constexpr int kLdsPitch = kTileW + 32;      // + 16 B halo chunk on each side
constexpr int kChunksPerRow = kLdsPitch / 16;

// Reading a few bytes past a row's border is safe whenever the row is not the first / last row of the allocation (the
// neighbouring bytes belong to the adjacent row of the same buffer) and those bytes never feed the arithmetic (only
// window positions [8-2CH, 24+2CH) do).  GUARD=true is used by the row blocks that touch the allocation's first / last
// row when src.border == 2.
template <int CH, int TH, bool NT, bool GUARD>
__device__ __forceinline__ void box5x5_u8_tile(uint8_t* __restrict__ dp, const uint8_t* __restrict__ sp, int dpitch, int spitch,
                                               int nrows, int row_bytes, int border_bytes, int x0, int r0, uint8_t* lds) {
  constexpr int kRows = TH + 4;
  constexpr int kTotal = kRows * kChunksPerRow;
  constexpr int kIters = (kTotal + 255) / 256;
  const int lo = -border_bytes, hi = row_bytes + border_bytes;
  const int tid = threadIdx.x;
  // ---- stage: all global loads first, then the LDS writes
  u32x4 stage[kIters];
#pragma unroll
  for (int it = 0; it < kIters; it++) {
    const int c = tid + it * 256;
    const int rr = c / kChunksPerRow, cc = c - rr * kChunksPerRow;
    const int gx = x0 - 16 + cc * 16;
    const int r = r0 - 2 + rr;
    stage[it] = u32x4{0, 0, 0, 0};
    if (c < kTotal && r <= nrows + 1 && gx < row_bytes + 16)
      stage[it] = ld_chunk<GUARD>(sp + (ptrdiff_t)r * spitch, gx, lo, hi);
  }
#pragma unroll
  for (int it = 0; it < kIters; it++) {
    const int c = tid + it * 256;
    if (c < kTotal) *(u32x4*)(lds + c * 16) = stage[it];  // chunk c sits at row (c / 66), col16 (c % 66): linear
  }
  __syncthreads();

  // ---- compute: wave w produces tile rows [w*RW, w*RW+RW)
  constexpr int RW = TH / 4;
  const int lane = tid & 63, wv = tid >> 6;
  const int x = x0 + lane * 16;
  if (x >= row_bytes) return;
  const bool full_store = x + 16 <= row_bytes;
  const uint8_t* lw = lds + 8 + lane * 16;  // window byte 0 (= x-8) of tile row 0
  auto load_unpack = [&](int t, uint32_t* e, uint32_t* o) {
    const uint8_t* q = lw + t * kLdsPitch;
    const u32x2 l = *(const u32x2*)q; const u32x4 m = *(const u32x4*)(q + 8); const u32x2 rr = *(const u32x2*)(q + 24);
    const uint32_t w[8] = {l.x, l.y, m.x, m.y, m.z, m.w, rr.x, rr.y};
#pragma unroll
    for (int i = 0; i < 8; i++) { e[i] = __builtin_amdgcn_perm(0u, w[i], 0x0c020c00u); o[i] = __builtin_amdgcn_perm(0u, w[i], 0x0c030c01u); }
  };
  uint32_t E[8], O[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { E[i] = 0; O[i] = 0; }
  const int j0 = wv * RW;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t e[8], o[8];
    load_unpack(j0 + k, e, o);
#pragma unroll
    for (int i = 0; i < 8; i++) { E[i] += e[i]; O[i] += o[i]; }
  }
#pragma unroll
  for (int j = 0; j < RW; j++) {
    const int r = r0 + j0 + j;
    if (r >= nrows) break;
    {
      uint32_t e[8], o[8];
      load_unpack(j0 + j + 4, e, o);
#pragma unroll
      for (int i = 0; i < 8; i++) { E[i] += e[i]; O[i] += o[i]; }
    }
    u32x4 res;
    res.x = out_dword<0, CH>(E, O); res.y = out_dword<1, CH>(E, O); res.z = out_dword<2, CH>(E, O); res.w = out_dword<3, CH>(E, O);
    uint8_t* drow = dp + (ptrdiff_t)r * dpitch + x;
    if (full_store) {
      if (NT) __builtin_nontemporal_store(res, (u32x4*)drow); else *(u32x4*)drow = res;
    } else {
      const int n = row_bytes - x;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const uint32_t d = k < 4 ? res.x : k < 8 ? res.y : k < 12 ? res.z : res.w;
        if (k < n) drow[k] = (uint8_t)(d >> (8 * (k & 3)));
      }
    }
    if (j + 1 < RW) {
      uint32_t e[8], o[8];
      load_unpack(j0 + j, e, o);  // the row leaving the 5-row window (re-read from LDS instead of a register ring)
#pragma unroll
      for (int i = 0; i < 8; i++) { E[i] -= e[i]; O[i] -= o[i]; }
    }
  }
}

template <int CH, int TH, bool NT>
__global__ __launch_bounds__(256) void box5x5_u8_lds_kernel(uint8_t* __restrict__ dp, const uint8_t* __restrict__ sp, int dpitch,
                                                            int spitch, int nrows, int row_bytes, int border_bytes, int nblk_x,
                                                            int nblk_y, int guard_ends) {
  static_assert(CH >= 1 && CH <= 4, "window holds 2*CH <= 8 halo bytes");
  static_assert(TH % 4 == 0, "four waves split the tile rows");
  __shared__ __attribute__((aligned(16))) uint8_t lds[(TH + 4) * kLdsPitch];
  // logical block id, XCD-remapped so that vertically adjacent row blocks run on the same XCD / L2
  const unsigned nb = (unsigned)nblk_x * (unsigned)nblk_y;
  const unsigned lb = xcd_remap(blockIdx.x, nb);
  const int by = lb / nblk_x, bx = lb - by * nblk_x;
  const int x0 = bx * kTileW, r0 = by * TH;
  if (guard_ends && (by == 0 || by == nblk_y - 1))
    box5x5_u8_tile<CH, TH, NT, true>(dp, sp, dpitch, spitch, nrows, row_bytes, border_bytes, x0, r0, lds);
  else
    box5x5_u8_tile<CH, TH, NT, false>(dp, sp, dpitch, spitch, nrows, row_bytes, border_bytes, x0, r0, lds);
}

// ---- streaming variant: no LDS, halo column sums exchanged between neighbouring lanes with DPP wave shifts ----------
// One wave owns a strip of 64 x 16 input bytes; lanes 1..62 produce output (992 B per row), lanes 0 and 63 only feed
// their neighbours.  Each lane keeps the 5-row column sums of ITS 16 bytes only (half the vertical work of a 32-byte
// window) and reads the 8 halo pairs it needs from lane-1 / lane+1 (v_mov_b32_dpp wave_shr:1 / wave_shl:1).  All
// RW+4 row loads of a wave are issued up front, so rows are consumed as they arrive and compute overlaps the stream.
constexpr int kStripOut = 62 * 16;

__device__ __forceinline__ uint32_t from_left(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true); }   // lane i <- lane i-1
__device__ __forceinline__ uint32_t from_right(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xf, 0xf, true); }  // lane i <- lane i+1

template <int CH, int KR, int KC, int RW, bool NT, bool GUARD, int PROBE>
__device__ __forceinline__ void box_u8_stream_body(uint8_t* __restrict__ dp, const uint8_t* __restrict__ sp, int dpitch, int spitch,
                                                   int nrows, int row_bytes, int border_bytes, int x, int r0, bool writer) {
  static_assert((KR & 1) && (KC & 1) && KR <= 7 && (KC / 2) * CH <= 8, "the 32-byte window holds 8 halo bytes per side; 7 rows of 255 fit a u16 column sum");
  constexpr int HR = KR / 2;
  const int lo = -border_bytes, hi = row_bytes + border_bytes;
  u32x4 raw[RW + KR - 1];
  const bool in_reach = x + 16 > lo && x < row_bytes + 16;
#pragma unroll
  for (int k = 0; k < RW + KR - 1; k++) {
    const int r = r0 - HR + k;
    raw[k] = u32x4{0, 0, 0, 0};
    if (in_reach && r <= nrows - 1 + HR) raw[k] = ld_chunk<GUARD>(sp + (ptrdiff_t)r * spitch, x, lo, hi);
  }
  auto unpack = [](const u32x4& w, uint32_t* e, uint32_t* o) {
    const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { e[i] = __builtin_amdgcn_perm(0u, d[i], 0x0c020c00u); o[i] = __builtin_amdgcn_perm(0u, d[i], 0x0c030c01u); }
  };
  uint32_t E[4] = {0, 0, 0, 0}, O[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < KR - 1; k++) {
    uint32_t e[4], o[4];
    unpack(raw[k], e, o);
#pragma unroll
    for (int i = 0; i < 4; i++) { E[i] += e[i]; O[i] += o[i]; }
  }
  const bool full_store = x + 16 <= row_bytes;
#pragma unroll
  for (int j = 0; j < RW; j++) {
    const int r = r0 + j;
    if (r >= nrows) break;
    {
      uint32_t e[4], o[4];
      unpack(raw[j + KR - 1], e, o);
#pragma unroll
      for (int i = 0; i < 4; i++) { E[i] += e[i]; O[i] += o[i]; }
    }
    // 32-byte window of column sums: [left lane's bytes 8..15 | own 16 | right lane's bytes 0..7]
    const uint32_t WE[8] = {from_left(E[2]), from_left(E[3]), E[0], E[1], E[2], E[3], from_right(E[0]), from_right(E[1])};
    const uint32_t WO[8] = {from_left(O[2]), from_left(O[3]), O[0], O[1], O[2], O[3], from_right(O[0]), from_right(O[1])};
    u32x4 res;
    res.x = out_dword<0, CH, KC, KR * KC>(WE, WO); res.y = out_dword<1, CH, KC, KR * KC>(WE, WO);
    res.z = out_dword<2, CH, KC, KR * KC>(WE, WO); res.w = out_dword<3, CH, KC, KR * KC>(WE, WO);
    if (PROBE == 1) res = raw[j + HR];  // measurement probe (tools/probe_box.py): same loads / stores, no arithmetic
    if (writer) {
      uint8_t* drow = dp + (ptrdiff_t)r * dpitch + x;
      if (full_store) {
        if (NT) __builtin_nontemporal_store(res, (u32x4*)drow); else *(u32x4*)drow = res;
      } else {
        const int n = row_bytes - x;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const uint32_t d = k < 4 ? res.x : k < 8 ? res.y : k < 12 ? res.z : res.w;
          if (k < n) drow[k] = (uint8_t)(d >> (8 * (k & 3)));
        }
      }
    }
    if (j + 1 < RW) {
      uint32_t e[4], o[4];
      unpack(raw[j], e, o);  // the row leaving the window
#pragma unroll
      for (int i = 0; i < 4; i++) { E[i] -= e[i]; O[i] -= o[i]; }
    }
  }
}

template <int CH, int KR, int KC, int RW, bool NT, int PROBE>
__global__ __launch_bounds__(256, (RW <= 2 ? 8 : RW <= 4 ? 6 : 4)) void box_u8_stream_kernel(uint8_t* __restrict__ dp, const uint8_t* __restrict__ sp, int dpitch,
                                                               int spitch, int nrows, int row_bytes, int border_bytes, int nstrips,
                                                               int nblk_y, int guard_ends) {
  static_assert(CH >= 1 && CH <= 4, "window holds 2*CH <= 8 halo bytes");
  const unsigned nb = (unsigned)nstrips * (unsigned)nblk_y;
  const unsigned lb = xcd_remap(blockIdx.x, nb);  // consecutive logical blocks = vertically adjacent row blocks of one strip
  const int s = lb / nblk_y, by = lb - s * nblk_y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = s * kStripOut - 16 + lane * 16;
  const int r0 = (by * (int)(blockDim.x >> 6) + wv) * RW;
  if (r0 >= nrows) return;
  const bool writer = lane >= 1 && lane <= 62 && x < row_bytes;
  if (guard_ends && (r0 == 0 || r0 + RW >= nrows))
    box_u8_stream_body<CH, KR, KC, RW, NT, true, PROBE>(dp, sp, dpitch, spitch, nrows, row_bytes, border_bytes, x, r0, writer);
  else
    box_u8_stream_body<CH, KR, KC, RW, NT, false, PROBE>(dp, sp, dpitch, spitch, nrows, row_bytes, border_bytes, x, r0, writer);
}

// ---- line-aligned variant of the streaming kernel ------------------------------------------------------------------
// Same arithmetic, different geometry: a wave owns 1024 OUTPUT bytes per row (all 64 lanes store 16 B, so every wave-row
// store covers eight whole 128-B lines of an aligned destination instead of splitting a sector with the neighbouring
// strip).  The 8 halo bytes the edge lanes miss come from one extra 8-B load per input row issued by lanes 0 and 63 only
// (the line belongs to the neighbouring wave of the same workgroup, which requests it at the same time); their column
// sums live in the same two register pairs (left halo in lane 0, right halo in lane 63) and enter the window as the
// `old` operand of the DPP wave shifts, which lane 0 / lane 63 keep because their source lane does not exist.
// A workgroup is WX waves side by side x 4/WX row blocks; ORDER 0 walks the block grid row-major (the WX-wide blocks
// of one row band run together on one XCD), ORDER 1 strip-major like the kernel above.
__device__ __forceinline__ uint32_t from_left_or(uint32_t old, uint32_t v) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t from_right_or(uint32_t old, uint32_t v) { return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false); }

enum { kStoreDefault = 0, kStoreNT = 1, kStoreSC1 = 2, kStoreSC01 = 3, kStoreNTSC1 = 4 };
// cache-policy bits of the buffer instructions (aux operand): sc0 = 1, nt = 2, sc1 = 16
enum { kAuxDefault = 0, kAuxSC0 = 1, kAuxNT = 2, kAuxSC1 = 16 };

// Buffer descriptors instead of flat pointers: (1) the row term of every address is a scalar offset and the lane term ONE
// 32-bit register shared by all row loads — no 64-bit address arithmetic or address register pairs; (2) the hardware range
// check (gfx950: per dword, scalar offset included, a negative offset reads as 0 — tools/buftest.hip) replaces every guard:
// the source descriptor spans exactly [16 bytes before the first addressable row, end of the last addressable 16-byte
// granule], so rows below the bottom border and chunks past the last row read as 0 and never fault, and there is no
// separate code path for the row blocks that touch the allocation's first / last row.
struct BoxGeom {
  const uint8_t* sbase;   // 16 bytes before byte 0 of row -border
  uint8_t* dbase;         // byte 0 of row 0
  uint32_t sbytes, dbytes;
  int spitch, dpitch, nrows, row_bytes, srow0;  // srow0: first addressable row (-border) as a row index offset: row r sits at (r + srow0) * spitch + 16
  int nbx, nby, nhi, order;
};

// HALO = true: 1024-B line-aligned strips with halo loads (above); HALO = false: the 992-B strips of the first streaming
// kernel (lanes 0 / 63 only feed their neighbours).  `rw` <= RW is the number of output rows of THIS wave (wave-uniform):
// the launcher mixes row blocks of RW and RW - 1 rows so that the whole grid is resident in one round.
template <int CH, int KR, int KC, int RW, int SAUX, int LAUX, bool HALO, int PROBE>
__device__ __forceinline__ void box_u8_wide_body(const BoxGeom& g, int x0, int r0, int rw, int lane) {
  constexpr int HR = KR / 2, NR = RW + KR - 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.sbase, 0, g.sbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)g.dbase, 0, g.dbytes, 0x00020000);
  const int x = HALO ? x0 + lane * 16 : x0 - 16 + lane * 16;
  const uint32_t vo = (uint32_t)(min(x, (g.row_bytes + 15) & ~15) + 16);   // lanes past the chunk that holds the right border re-read it (their sums feed nobody)
  const int srow = (r0 - HR + g.srow0) * g.spitch;               // scalar offset of the wave's first input row
  u32x4 raw[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    if (k < NR - 1 || rw == RW) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, srow + k * g.spitch, LAUX);   // the last row only feeds output row RW - 1
    else raw[k] = u32x4{0, 0, 0, 0};
  }
  u32x2 hraw[HALO ? NR : 1];
  if constexpr (HALO) {
    // halo chunk of the edge lanes: bytes [x0 - 8, x0) for lane 0, [x0 + 1024, x0 + 1032) for lane 63
    const int hx = lane == 0 ? x0 - 8 : x0 + 1024;
    const bool halo_lane = PROBE != 2 && (lane == 0 || lane == 63) && hx < g.row_bytes + 8;
#pragma unroll
    for (int k = 0; k < NR; k++) hraw[k] = u32x2{0, 0};
    if (halo_lane) {
#pragma unroll
      for (int k = 0; k < NR; k++) hraw[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)(hx + 16), srow + k * g.spitch, 0);
    }
  }
  auto unpack = [](const u32x4& w, uint32_t* e, uint32_t* o) {
    const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { e[i] = __builtin_amdgcn_perm(0u, d[i], 0x0c020c00u); o[i] = __builtin_amdgcn_perm(0u, d[i], 0x0c030c01u); }
  };
  auto unpack2 = [](const u32x2& w, uint32_t* e, uint32_t* o) {
    e[0] = __builtin_amdgcn_perm(0u, w.x, 0x0c020c00u); o[0] = __builtin_amdgcn_perm(0u, w.x, 0x0c030c01u);
    e[1] = __builtin_amdgcn_perm(0u, w.y, 0x0c020c00u); o[1] = __builtin_amdgcn_perm(0u, w.y, 0x0c030c01u);
  };
  uint32_t E[4] = {0, 0, 0, 0}, O[4] = {0, 0, 0, 0}, HE[2] = {0, 0}, HO[2] = {0, 0};
#pragma unroll
  for (int k = 0; k < KR - 1; k++) {
    uint32_t e[4], o[4];
    unpack(raw[k], e, o);
#pragma unroll
    for (int i = 0; i < 4; i++) { E[i] += e[i]; O[i] += o[i]; }
    if constexpr (HALO) {
      uint32_t he[2], ho[2];
      unpack2(hraw[k], he, ho);
#pragma unroll
      for (int i = 0; i < 2; i++) { HE[i] += he[i]; HO[i] += ho[i]; }
    }
  }
  const bool writer = HALO ? x < g.row_bytes : (lane >= 1 && lane <= 62 && x < g.row_bytes);
  const bool full_store = x + 16 <= g.row_bytes;
#pragma unroll
  for (int j = 0; j < RW; j++) {
    const int r = r0 + j;
    if (j >= rw || r >= g.nrows) break;
    {
      uint32_t e[4], o[4];
      unpack(raw[j + KR - 1], e, o);
#pragma unroll
      for (int i = 0; i < 4; i++) { E[i] += e[i]; O[i] += o[i]; }
      if constexpr (HALO) {
        uint32_t he[2], ho[2];
        unpack2(hraw[j + KR - 1], he, ho);
#pragma unroll
        for (int i = 0; i < 2; i++) { HE[i] += he[i]; HO[i] += ho[i]; }
      }
    }
    // 32-byte window of column sums: [left neighbour's bytes 8..15 (lane 0: its halo) | own 16 | right neighbour's bytes 0..7 (lane 63: its halo)]
    uint32_t WE[8], WO[8];
    if constexpr (HALO) {
      WE[0] = from_left_or(HE[0], E[2]); WE[1] = from_left_or(HE[1], E[3]); WE[6] = from_right_or(HE[0], E[0]); WE[7] = from_right_or(HE[1], E[1]);
      WO[0] = from_left_or(HO[0], O[2]); WO[1] = from_left_or(HO[1], O[3]); WO[6] = from_right_or(HO[0], O[0]); WO[7] = from_right_or(HO[1], O[1]);
    } else {
      WE[0] = from_left(E[2]); WE[1] = from_left(E[3]); WE[6] = from_right(E[0]); WE[7] = from_right(E[1]);
      WO[0] = from_left(O[2]); WO[1] = from_left(O[3]); WO[6] = from_right(O[0]); WO[7] = from_right(O[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { WE[2 + i] = E[i]; WO[2 + i] = O[i]; }
    u32x4 res;
    res.x = out_dword<0, CH, KC, KR * KC>(WE, WO); res.y = out_dword<1, CH, KC, KR * KC>(WE, WO);
    res.z = out_dword<2, CH, KC, KR * KC>(WE, WO); res.w = out_dword<3, CH, KC, KR * KC>(WE, WO);
    if (PROBE == 1) res = raw[j + HR];  // measurement probes: 1 = copy (the unused row loads are dropped by the compiler), 2 = arithmetic without the halo loads, 3 = all loads, no arithmetic
    if (PROBE == 3) { res = raw[j + HR]; for (int k = 0; k < NR; k++) { res.x ^= raw[k].y; if constexpr (HALO) res.x ^= hraw[k].x ^ hraw[k].y; } }
    if (writer) {
      if (full_store) __builtin_amdgcn_raw_buffer_store_b128(res, rd, (uint32_t)x, r * g.dpitch, SAUX);
      else {  // the row's last, partial chunk (row_bytes % 16 != 0): dwords, then bytes
        const int n = g.row_bytes - x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t d = k == 0 ? res.x : k == 1 ? res.y : k == 2 ? res.z : res.w;
          if (4 * k + 4 <= n) __builtin_amdgcn_raw_buffer_store_b32(d, rd, (uint32_t)(x + 4 * k), r * g.dpitch, 0);
          else
            for (int b = 0; b < 3; b++)
              if (4 * k + b < n) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(d >> (8 * b)), rd, (uint32_t)(x + 4 * k + b), r * g.dpitch, 0);
        }
      }
    }
    if (j + 1 < RW) {
      uint32_t e[4], o[4];
      unpack(raw[j], e, o);  // the row leaving the window
#pragma unroll
      for (int i = 0; i < 4; i++) { E[i] -= e[i]; O[i] -= o[i]; }
      if constexpr (HALO) {
        uint32_t he[2], ho[2];
        unpack2(hraw[j], he, ho);
#pragma unroll
        for (int i = 0; i < 2; i++) { HE[i] -= he[i]; HO[i] -= ho[i]; }
      }
    }
  }
}

// nby row blocks per strip: the first nhi have RW rows, the others RW - 1 (nhi == nby: all RW).
// One launch serves a BATCH of nframes frames of one geometry (vpp_box_filter_batch; vpp_box_filter = a batch of one): the block grid is the
// frames' grids back to back, walked XCD-major as a whole, so the chip never drains between frames — consecutive single-frame launches each
// pay their own ramp and tail (a 50 MB frame is only ~6 rounds of resident waves) and, from a host that submits few launches per
// synchronisation, the submission latency once per frame.
constexpr int kBoxBatchMax = 64;
struct BoxBatch { const uint8_t* sbase[kBoxBatchMax]; uint8_t* dbase[kBoxBatchMax]; };
template <int CH, int KR, int KC, int RW, int WX, int SAUX, int LAUX, bool HALO, int OCC, int PROBE, int NW = 4>
__global__ __launch_bounds__(64 * NW, OCC * 4 / NW) void box_u8_wide_kernel(const BoxGeom g0, const BoxBatch frames, int nframes) {
  static_assert(CH >= 1 && CH <= 4 && NW % WX == 0, "window holds 2*CH <= 8 halo bytes; NW waves per workgroup, WX of them side by side");
  constexpr int WY = NW / WX;
  const unsigned nb = (unsigned)g0.nbx * (unsigned)g0.nby;
  const unsigned L = (g0.order & 2) ? blockIdx.x : xcd_remap(blockIdx.x, nb * (unsigned)nframes);
  const unsigned f = L / nb, lb = L - f * nb;
  BoxGeom g = g0;
  g.sbase = frames.sbase[f]; g.dbase = frames.dbase[f];
  int bx, by;
  if ((g.order & 1) == 0) { by = lb / g.nbx; bx = lb - by * g.nbx; } else { bx = lb / g.nby; by = lb - bx * g.nby; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wx = wv % WX, wy = wv / WX;
  // the wave id is uniform but the compiler cannot prove it: readfirstlane keeps everything derived from it in SGPRs
  const int x0 = __builtin_amdgcn_readfirstlane((bx * WX + wx) * (HALO ? 1024 : kStripOut));
  const int yb = __builtin_amdgcn_readfirstlane(by * WY + wy);       // row block of this wave
  const int r0 = yb * (RW - 1) + min(yb, g.nhi);
  const int rw = yb < g.nhi ? RW : RW - 1;
  if (x0 >= g.row_bytes || r0 >= g.nrows) return;
  box_u8_wide_body<CH, KR, KC, RW, SAUX, LAUX, HALO, PROBE>(g, x0, r0, rw, lane);
}

// ---- 32-bit 5x5: int / unsigned (the element type of the reference's own benchmark, benchmarks/box_5x5_filter.cc:165-171,187-191)
//      and float (taps added in the reference's order) ----
// Same shape as the u8 streaming kernel with one pixel per dword: a lane owns 4 consecutive pixels (16 B), keeps their
// 5-row column sums (integer adds are order-independent, so the running add / subtract is exact, wrap-around included),
// takes the two halo sums of each side from lane -1 / +1 over DPP, and lanes 1..62 store 4 truncating quotients (`/ 25` in T).
// All RW + 4 row loads are issued up front through a buffer descriptor over the source's addressable bytes: nothing outside the
// allocation is read (out-of-range dwords come back 0 from the range check), so border == 2 needs no separate guarded body.
constexpr int kW32StripOut = 62 * 4;

template <class T, int KR, int KC, int RW, bool NT>
__global__ __launch_bounds__(256) void box_w32_stream_kernel(T* __restrict__ dp, const T* __restrict__ sp, int dpitch, int spitch,
                                                                int nrows, int ncols, int border, int nstrips, int nblk_y, uint32_t sbytes) {
  // geometry of the round-2 u8 kernel (measured there: 9.4 -> 8.5 us): a workgroup = its waves side by side (adjacent strips of one
  // row block), the block grid walked row-major inside each XCD, so the waves that split a store sector / share a halo line run together
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = (int)(blockDim.x >> 6);
  const int nbx = (nstrips + wpb - 1) / wpb;
  const unsigned nb = (unsigned)nbx * (unsigned)nblk_y;
  const unsigned lb = xcd_remap(blockIdx.x, nb);
  const int by = lb / nbx, bx = lb - by * nbx;
  const int s = __builtin_amdgcn_readfirstlane(bx * wpb + wv);
  if (s >= nstrips) return;
  const int x = s * kW32StripOut - 4 + lane * 4;   // first pixel of this lane's chunk
  const int r0 = by * RW;
  if (r0 >= nrows) return;
  static_assert((KR & 1) && (KC & 1) && KR <= 7 && KC <= 5, "two halo pixels per side come over DPP");
  constexpr int HR = KR / 2, C0 = 2 - KC / 2;  // first window column of output pixel 0 within the 8-pixel register window
  const bool writer = lane >= 1 && lane <= 62 && x < ncols;
  typedef uint32_t U;  // raw dwords: integer sums wrap like the hardware's adds (no signed-overflow UB in the source); floats are bit-cast
  // Buffer descriptor over the source's addressable bytes (16 bytes before row -border .. the last addressable 16-byte granule): the row
  // term of every address is a scalar offset, the lane term one register, the RW + KR - 1 row loads go out back to back without any
  // clamping or per-element form — rows past the bottom border and chunks past the last row read 0 (hardware range check, tools/buftest.hip)
  // and feed only outputs that are not stored; a chunk that straddles a row end reads the row's padding / the next row, which no window uses.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint8_t*)sp - (ptrdiff_t)border * spitch - 16), 0, sbytes, 0x00020000);
  const uint32_t vo = (uint32_t)(16 + min(x, (ncols + 3) & ~3) * 4);
  const int srow = (r0 - HR + border) * spitch;
  U raw[RW + KR - 1][4];
#pragma unroll
  for (int k = 0; k < RW + KR - 1; k++) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, srow + k * spitch, 0);
    raw[k][0] = v.x; raw[k][1] = v.y; raw[k][2] = v.z; raw[k][3] = v.w;
  }
  const bool full_store = x + 4 <= ncols;
  if constexpr (std::is_floating_point<T>::value) {
    // float: the 25 taps are added one by one in the reference's row-major order (the sum is not associative), starting from 0;
    // each input row's halo (2 pixels per side) is exchanged once, the window of an output row is then 5 x 8 registers
    float win[RW + KR - 1][8];
#pragma unroll
    for (int k = 0; k < RW + KR - 1; k++) {
      const U w8[8] = {from_left(raw[k][2]), from_left(raw[k][3]), raw[k][0], raw[k][1], raw[k][2], raw[k][3], from_right(raw[k][0]), from_right(raw[k][1])};
#pragma unroll
      for (int i = 0; i < 8; i++) win[k][i] = __builtin_bit_cast(float, w8[i]);
    }
#pragma unroll
    for (int j = 0; j < RW; j++) {
      const int r = r0 + j;
      if (r >= nrows) break;
      U out[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float sum = 0.f;
#pragma unroll
        for (int dr = 0; dr < KR; dr++)
#pragma unroll
          for (int dc = 0; dc < KC; dc++) sum += win[j + dr][i + C0 + dc];
        out[i] = __builtin_bit_cast(U, sum / (KR * KC));   // C++ `/ (R*C)` on the promoted type: float / int -> IEEE float division
      }
      if (writer) {
        U* drow = (U*)((uint8_t*)dp + (ptrdiff_t)r * dpitch) + x;
        if (full_store) {
          const u32x4 v = {out[0], out[1], out[2], out[3]};
          if (NT) __builtin_nontemporal_store(v, (u32x4*)drow); else *(u32x4*)drow = v;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) if (x + i < ncols) drow[i] = out[i];
        }
      }
    }
    return;
  }
  U V[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < KR - 1; k++)
#pragma unroll
    for (int i = 0; i < 4; i++) V[i] += raw[k][i];
#pragma unroll
  for (int j = 0; j < RW; j++) {
    const int r = r0 + j;
    if (r >= nrows) break;
#pragma unroll
    for (int i = 0; i < 4; i++) V[i] += raw[j + KR - 1][i];
    const U W[8] = {from_left(V[2]), from_left(V[3]), V[0], V[1], V[2], V[3], from_right(V[0]), from_right(V[1])};
    U out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      U hs = 0;
#pragma unroll
      for (int t = 0; t < KC; t++) hs += W[i + C0 + t];
      out[i] = (U)((T)hs / (T)(KR * KC));
    }
    if (writer) {
      U* drow = (U*)((uint8_t*)dp + (ptrdiff_t)r * dpitch) + x;
      if (full_store) {
        const u32x4 v = {out[0], out[1], out[2], out[3]};
        if (NT) __builtin_nontemporal_store(v, (u32x4*)drow); else *(u32x4*)drow = v;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) if (x + i < ncols) drow[i] = out[i];
      }
    }
    if (j + 1 < RW) {
#pragma unroll
      for (int i = 0; i < 4; i++) V[i] -= raw[j][i];
    }
  }
}

template <class T, int KR = 5, int KC = 5> int launch_w32(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st) {
  // measured 4K (round 2, descriptor loads): int rows 2 -> 10.6 us, 4 -> 10.9; float (25 adds per pixel) rows 2 -> 13.2, 4 -> 12.4
  const int rows = tuning("box.rows32", std::is_floating_point<T>::value ? 4 : 2), nt = tuning("box.nt", 1);
  int wpb = tuning("box.waves_per_block", 4);
  if (wpb != 1 && wpb != 2) wpb = 4;
  const int nstrips = (dst->ncols + kW32StripOut - 1) / kW32StripOut;
  auto go = [&](auto RWc, auto NTc) {
    constexpr int RW = decltype(RWc)::value; constexpr bool NT = decltype(NTc)::value;
    // one buffer descriptor addresses < 4 GiB: larger sources go out as row bands (a band's upper / lower neighbours are real rows of
    // the same image, its descriptor simply ends 'border' rows below the band)
    const size_t row_tail = 16 + (size_t)((((dst->ncols + src->border) * 4) + 15) & ~15);
    const int band_tune = tuning("box.band_rows32", 0);   // tests: force small bands
    const int band_max = band_tune > 0 ? std::max(RW, band_tune / RW * RW)
                                       : (int)std::min<size_t>((size_t)dst->nrows, std::max<size_t>(RW, (((size_t)1 << 31) / (size_t)src->pitch) / RW * RW));
    for (int b0 = 0; b0 < dst->nrows; b0 += band_max) {
      const int band = std::min(band_max, dst->nrows - b0), nblk_y = (band + RW - 1) / RW;
      const uint32_t sbytes = (uint32_t)((size_t)(band - 1 + 2 * src->border) * src->pitch + row_tail);
      box_w32_stream_kernel<T, KR, KC, RW, NT><<<((nstrips + wpb - 1) / wpb) * nblk_y, 64 * wpb, 0, st>>>(
          (T*)((uint8_t*)dst->first_pixel + (ptrdiff_t)b0 * dst->pitch), (const T*)((const uint8_t*)src->first_pixel + (ptrdiff_t)b0 * src->pitch), dst->pitch, src->pitch, band,
          dst->ncols, src->border, nstrips, nblk_y, sbytes);
    }
  };
  auto pick = [&](auto NTc) {
    if constexpr (KR == 5 && KC == 5) {
      switch (rows) {
        case 1: go(std::integral_constant<int, 1>(), NTc); return;
        case 2: go(std::integral_constant<int, 2>(), NTc); return;
        case 8: go(std::integral_constant<int, 8>(), NTc); return;
      }
    }
    go(std::integral_constant<int, 4>(), NTc);  // the other windows: 4 rows per wave only
  };
  if constexpr (KR == 5 && KC == 5) { if (nt) pick(std::true_type()); else pick(std::false_type()); }
  else pick(std::true_type());
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
// 32-bit single-channel images: odd windows up to 7 rows x 5 columns through the streaming kernel; -1 = not one of them
template <class T> int launch_w32_windows(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, hipStream_t st) {
  switch (R * 10 + C) {
    case 33: return launch_w32<T, 3, 3>(dst, src, st);
    case 35: return launch_w32<T, 3, 5>(dst, src, st);
    case 53: return launch_w32<T, 5, 3>(dst, src, st);
    case 55: return launch_w32<T, 5, 5>(dst, src, st);
    case 73: return launch_w32<T, 7, 3>(dst, src, st);
    case 75: return launch_w32<T, 7, 5>(dst, src, st);
  }
  return -1;
}

// ---- generic path ----------------------------------------------------------------------------------------
// Tile of TW x TH output components; LDS holds (TH + R - 1) x (TW + (C-1)*ch) components in the promoted type.
template <class T, class S, int TW, int TH>
__global__ __launch_bounds__(256) void box_generic_kernel(DImg dst, DImg src, int R, int C, int ncomp, int lds_w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  S* tile = (S*)smem_raw;
  const int ch = dst.ch, hr = R / 2, hc = C / 2;
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH;
  const int lds_h = TH + R - 1;
  for (int idx = threadIdx.x; idx < lds_w * lds_h; idx += 256) {
    const int ly = idx / lds_w, lx = idx - ly * lds_w;
    const int r = r0 + ly - hr, c = c0 + lx - hc * ch;
    S v = 0;
    if (r < dst.nr + hr && c < ncomp + hc * ch) v = (S)src.row<T>(r)[c];  // r >= -hr, c >= -hc*ch always: inside the border
    tile[idx] = v;
  }
  __syncthreads();
  const int div = R * C;
  if constexpr (!std::is_floating_point<S>::value) {
    // integer sums do not depend on the order of the taps: R-row column sums first (one thread per tile column, running
    // down the tile), then C of them per output — R + C LDS reads per output instead of R * C
    S* vs = tile + lds_w * lds_h;  // [TH][lds_w]
    for (int lx = threadIdx.x; lx < lds_w; lx += 256) {
      S run = 0;
      for (int dr = 0; dr < R - 1; dr++) run += tile[dr * lds_w + lx];
      for (int ty = 0; ty < TH; ty++) {
        run += tile[(ty + R - 1) * lds_w + lx];
        vs[ty * lds_w + lx] = run;
        run -= tile[ty * lds_w + lx];
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TW * TH; idx += 256) {
      const int ty = idx / TW, tx = idx - ty * TW;
      const int r = r0 + ty, c = c0 + tx;
      if (r >= dst.nr || c >= ncomp) continue;
      S sum = 0;
      for (int dc = 0; dc < C; dc++) sum += vs[ty * lds_w + tx + dc * ch];
      dst.row<T>(r)[c] = (T)(sum / div);
    }
    return;
  }
  for (int idx = threadIdx.x; idx < TW * TH; idx += 256) {  // float: taps in the reference's row-major order
    const int ty = idx / TW, tx = idx - ty * TW;
    const int r = r0 + ty, c = c0 + tx;
    if (r >= dst.nr || c >= ncomp) continue;
    S sum = 0;
    for (int dr = 0; dr < R; dr++)
      for (int dc = 0; dc < C; dc++) sum += tile[(ty + dr) * lds_w + tx + dc * ch];
    dst.row<T>(r)[c] = (T)(sum / div);
  }
}

template <class T, class S>
int launch_generic(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, hipStream_t st) {
  constexpr int TW = 128, TH = 16;
  const int ncomp = dst->ncols * dst->channels;
  const int lds_w = TW + (C - 1) * dst->channels;
  const size_t smem = (size_t)lds_w * (TH + R - 1 + (std::is_floating_point<S>::value ? 0 : TH)) * sizeof(S);
  VPP_REQUIRE(smem <= 64 * 1024, VPP_ERR_UNSUPPORTED, "vpp_box_filter: window %dx%d too large for the LDS tile", R, C);
  dim3 grid((ncomp + TW - 1) / TW, (dst->nrows + TH - 1) / TH);
  box_generic_kernel<T, S, TW, TH><<<grid, 256, smem, st>>>(dimg(dst), dimg(src), R, C, ncomp, lds_w);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

// second streaming kernel (5x5): geometry and cache policy from the tuning knobs; the lab build (tools/boxlab.hip,
// -DVPP_BOX_LAB) instantiates the sweep for CH == 3, the product only the configuration the sweep picked.
// rows: RW rows per wave; mix = 1: row blocks of RW and RW - 1 rows mixed so that nstrips * nblocks <= `slots` waves (the whole
// grid resident in one round: slots = CUs x 32 wave slots at 8 waves / SIMD).
// Why starting a row's first chunk 16 bytes before pixel (row, 0) can never fault, whatever memory the caller handed in: these kernels
// only run on images with first_pixel and pitch 16-byte aligned (aligned16) and border >= 1, so pixel (-border, 0) is 16-byte aligned and
// the byte just before it is the image's own left border, i.e. addressable.  The chunk [pixel - 16, pixel) is the aligned 16-byte granule
// that holds that byte: same page, same mapping — at worst its first 16 - border * elem_bytes bytes belong to whatever precedes the image
// in memory; they are loaded and never used (no window reaches them).  The far end is covered by the descriptor's range check.
inline bool fits_descriptor(const vpp_image_desc* dst, const vpp_image_desc* src) {
  const size_t lim = 0xFFFFFF00u;
  return (size_t)(src->nrows + 2 * src->border) * (size_t)src->pitch + 64 < lim && (size_t)dst->nrows * (size_t)dst->pitch < lim;
}
template <int CH, int RW, int WX, bool HALO, int NW>
BoxGeom wide_geometry(const vpp_image_desc* dst, const vpp_image_desc* src, int order, int mix, int slots);
template <int CH, int RW, int WX, int SAUX, bool HALO, int OCC, int PROBE, int KR = 5, int KC = 5, int NW = 4, int LAUX = kAuxDefault>
void launch_wide_cfg(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st, int order, int mix, int slots, int n = 1) {
  const BoxGeom g = wide_geometry<CH, RW, WX, HALO, NW>(&dst[0], &src[0], order, mix, slots);   // n frames of one geometry: dst[k] <- src[k]
  for (int b0 = 0; b0 < n; b0 += kBoxBatchMax) {
    const int nb = std::min(kBoxBatchMax, n - b0);
    BoxBatch fr{};
    for (int k = 0; k < nb; k++) {
      fr.sbase[k] = (const uint8_t*)src[b0 + k].first_pixel - (ptrdiff_t)src[b0 + k].border * src[b0 + k].pitch - 16;
      fr.dbase[k] = (uint8_t*)dst[b0 + k].first_pixel;
    }
    if (tuning("box.anyorder", 0))   // experiment (tools/overlap_lab.hip): the AQL packet without its barrier bit — may start before the stream's previous packet has completed
      hipExtLaunchKernelGGL((box_u8_wide_kernel<CH, KR, KC, RW, WX, SAUX, LAUX, HALO, OCC, PROBE, NW>), dim3(g.nbx * g.nby * nb), dim3(64 * NW), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, g, fr, nb);
    else
      box_u8_wide_kernel<CH, KR, KC, RW, WX, SAUX, LAUX, HALO, OCC, PROBE, NW><<<g.nbx * g.nby * nb, 64 * NW, 0, st>>>(g, fr, nb);
  }
}
template <int CH, int RW, int WX, bool HALO, int NW>
BoxGeom wide_geometry(const vpp_image_desc* dst, const vpp_image_desc* src, int order, int mix, int slots) {
  const int row_bytes = dst->ncols * CH, strip = HALO ? 1024 : kStripOut, WY = NW / WX, bb = src->border * CH;
  const int nstrips = (row_bytes + strip - 1) / strip;
  BoxGeom g;
  g.nbx = (nstrips + WX - 1) / WX;
  int nblk = (dst->nrows + RW - 1) / RW;  // row blocks per strip, all of RW rows
  g.nhi = nblk;
  if (mix && RW > 1) {
    // as few blocks as keep the grid within `slots` waves, but never more than RW rows per block: nblk blocks of RW - 1 or RW rows
    const int want = max(1, slots / (g.nbx * WX));
    const int lo_blocks = (dst->nrows + RW - 1) / RW, hi_blocks = (dst->nrows + RW - 2) / (RW - 1);
    nblk = min(max(want / WY * WY, lo_blocks), hi_blocks);
    g.nhi = dst->nrows - nblk * (RW - 1);                   // nhi * RW + (nblk - nhi) * (RW - 1) == nrows
  }
  g.nby = (nblk + WY - 1) / WY;
  g.order = order;
  g.sbase = (const uint8_t*)src->first_pixel - (ptrdiff_t)src->border * src->pitch - 16;
  g.dbase = (uint8_t*)dst->first_pixel;
  g.sbytes = (uint32_t)((size_t)(dst->nrows - 1 + 2 * src->border) * src->pitch + 16 + (size_t)((row_bytes + bb + 15) & ~15));
  g.dbytes = (uint32_t)((size_t)(dst->nrows - 1) * dst->pitch + row_bytes);
  g.spitch = src->pitch; g.dpitch = dst->pitch; g.nrows = dst->nrows; g.row_bytes = row_bytes; g.srow0 = src->border;
  return g;
}
template <int CH> int launch_wide(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st, int n = 1) {
  // measured on MI355X (tools/boxlab, 4K vuchar3): 992-B strips, 2 rows per wave, workgroup = 4 strips side by side, block grid
  // walked row-major per XCD, non-temporal stores: 8.6 us; strip-major order 8.9, one strip per workgroup 9.4, no XCD remap 15.5
  // (A batch is one XCD remap over all its frames: an XCD owns whole frames.  Splitting EVERY frame over the 8 XCDs instead, frame after frame, measured faster on one
  // box — 8 frames per launch 9.98 -> 9.35 us per frame — and 1-1.5 % slower on four others: not kept.)
  const int rows = tuning("box.rows", 2), wx = tuning("box.wx", 4), sp = tuning("box.sp", kAuxNT), order = tuning("box.order", 0);
  const int halo = tuning("box.halo", 0), mix = tuning("box.mix", 0), slots = tuning("box.slots", 8192);
#ifdef VPP_BOX_LAB
  const int probe = tuning("box.probe", 0), occ = tuning("box.occ", 8);
  bool done = false;
  auto try_cfg = [&](auto RWc, auto WXc, auto SPc, auto HALOc, auto OCCc, auto PRc) {
    if (!done && rows == decltype(RWc)::value && wx == decltype(WXc)::value && sp == decltype(SPc)::value && halo == (int)decltype(HALOc)::value && occ == decltype(OCCc)::value && probe == decltype(PRc)::value) {
      launch_wide_cfg<CH, decltype(RWc)::value, decltype(WXc)::value, decltype(SPc)::value, decltype(HALOc)::value, decltype(OCCc)::value, decltype(PRc)::value>(dst, src, st, order, mix, slots, n);
      done = true;
    }
  };
  auto for_rw = [&](auto WXc, auto SPc, auto HALOc, auto OCCc, auto PRc) {
    try_cfg(std::integral_constant<int, 2>(), WXc, SPc, HALOc, OCCc, PRc); try_cfg(std::integral_constant<int, 3>(), WXc, SPc, HALOc, OCCc, PRc);
    try_cfg(std::integral_constant<int, 4>(), WXc, SPc, HALOc, OCCc, PRc); try_cfg(std::integral_constant<int, 5>(), WXc, SPc, HALOc, OCCc, PRc);
    try_cfg(std::integral_constant<int, 6>(), WXc, SPc, HALOc, OCCc, PRc);
  };
  auto for_wx = [&](auto SPc, auto HALOc, auto OCCc, auto PRc) {
    for_rw(std::integral_constant<int, 1>(), SPc, HALOc, OCCc, PRc); for_rw(std::integral_constant<int, 2>(), SPc, HALOc, OCCc, PRc); for_rw(std::integral_constant<int, 4>(), SPc, HALOc, OCCc, PRc);
  };
  auto for_occ = [&](auto SPc, auto HALOc, auto PRc) {
    for_wx(SPc, HALOc, std::integral_constant<int, 8>(), PRc); for_wx(SPc, HALOc, std::integral_constant<int, 6>(), PRc); for_wx(SPc, HALOc, std::integral_constant<int, 4>(), PRc);
  };
  if constexpr (CH == 3) {
    for_occ(std::integral_constant<int, kAuxNT>(), std::false_type(), std::integral_constant<int, 0>());
    for_occ(std::integral_constant<int, kAuxNT>(), std::true_type(), std::integral_constant<int, 0>());
    for_occ(std::integral_constant<int, kAuxNT>(), std::false_type(), std::integral_constant<int, 1>());
    for_occ(std::integral_constant<int, kAuxNT>(), std::false_type(), std::integral_constant<int, 3>());
    for_occ(std::integral_constant<int, kAuxNT>(), std::true_type(), std::integral_constant<int, 3>());
    for_occ(std::integral_constant<int, kAuxNT | kAuxSC1>(), std::false_type(), std::integral_constant<int, 0>());
  }
  const int shape = tuning("box.shape", 0), laux = tuning("box.laux", 0);
  if (CH == 3 && !done && (shape || laux || (sp != kAuxNT && sp != (kAuxNT | kAuxSC1)))) {
    // policy / workgroup-shape sweep around the chosen geometry (992-B strips, 2 rows per wave)
    auto pol = [&](auto SPc, auto LAc) {
      if (!done && shape == 0 && sp == decltype(SPc)::value && laux == decltype(LAc)::value) { launch_wide_cfg<CH, 2, 4, decltype(SPc)::value, false, 8, 0, 5, 5, 4, decltype(LAc)::value>(dst, src, st, order, mix, slots, n); done = true; }
    };
    auto pols = [&](auto LAc) {
      pol(std::integral_constant<int, 0>(), LAc); pol(std::integral_constant<int, 1>(), LAc); pol(std::integral_constant<int, 2>(), LAc); pol(std::integral_constant<int, 3>(), LAc);
      pol(std::integral_constant<int, 16>(), LAc); pol(std::integral_constant<int, 17>(), LAc); pol(std::integral_constant<int, 18>(), LAc); pol(std::integral_constant<int, 19>(), LAc);
    };
    pols(std::integral_constant<int, 0>()); pols(std::integral_constant<int, 1>()); pols(std::integral_constant<int, 2>()); pols(std::integral_constant<int, 16>());
    if (!done && sp == kAuxNT && laux == 0) {
      done = true;
      switch (shape * 10 + rows) {
        case 12: launch_wide_cfg<CH, 2, 4, kAuxNT, false, 8, 0, 5, 5, 8>(dst, src, st, order, mix, slots, n); break;   // 512 threads: 4 strips x 2 row blocks
        case 13: launch_wide_cfg<CH, 3, 4, kAuxNT, false, 8, 0, 5, 5, 8>(dst, src, st, order, mix, slots, n); break;
        case 22: launch_wide_cfg<CH, 2, 6, kAuxNT, false, 8, 0, 5, 5, 6>(dst, src, st, order, mix, slots, n); break;   // 384 threads: 6 strips side by side
        case 32: launch_wide_cfg<CH, 2, 12, kAuxNT, false, 8, 0, 5, 5, 12>(dst, src, st, order, mix, slots, n); break; // 768 threads: a whole 4K row
        case 42: launch_wide_cfg<CH, 2, 3, kAuxNT, false, 8, 0, 5, 5, 3>(dst, src, st, order, mix, slots, n); break;   // 192 threads: 3 strips
        case 52: launch_wide_cfg<CH, 2, 4, kAuxNT, false, 8, 0, 5, 5, 16>(dst, src, st, order, mix, slots, n); break;  // 1024 threads: 4 strips x 4 row blocks
        case 62: launch_wide_cfg<CH, 2, 2, kAuxNT, false, 8, 0, 5, 5, 8>(dst, src, st, order, mix, slots, n); break;   // 512 threads: 2 strips x 4 row blocks
        case 72: launch_wide_cfg<CH, 2, 2, kAuxNT, false, 8, 0, 5, 5, 2>(dst, src, st, order, mix, slots, n); break;   // 128 threads: 2 strips side by side
        case 82: launch_wide_cfg<CH, 2, 1, kAuxNT, false, 8, 0, 5, 5, 1>(dst, src, st, order, mix, slots, n); break;   // 64 threads: one strip per workgroup
        case 92: launch_wide_cfg<CH, 2, 1, kAuxNT, false, 8, 0, 5, 5, 2>(dst, src, st, order, mix, slots, n); break;   // 128 threads: 1 strip x 2 row blocks
        default: done = false;
      }
    }
  }
  if (!done) { set_error("boxlab: configuration rows=%d wx=%d sp=%d halo=%d occ=%d probe=%d not instantiated", rows, wx, sp, halo, occ, probe); return VPP_ERR_UNSUPPORTED; }
#else
  (void)rows; (void)wx; (void)sp; (void)halo;
  // Round 3, sources really from HBM (tools/boxlab sweep6-9, 32 rotating frame sets; us per 4K vuchar3 frame at 1 / 2 / 4 / 8 / 16 / 32 frames per
  // launch): 2 rows per wave 13.4 / 11.5 / 10.6 / 10.1 / 9.4 / 9.2, 3 rows 13.3 / 11.3 / 10.4 / 9.8 / 9.0 / 8.8, 6 rows at 4 waves per SIMD
  // 13.7 / 11.3 / 10.2 / 9.4 / 8.6 / 8.4 — more new bytes in flight per wave (10 row loads for 6 rows instead of 6 for 2) once there are
  // enough waves; a plain copy of the same geometry: 11.8 / 10.2 / 9.1 / 8.6 / 8.1 / 8.0.
  // (8 rows per wave measured: 8.90 / 8.72 us per frame at 32 / 64 frames per launch against 8.93 / 8.73 for 6 — not worth an instance; 64 frames per launch, kBoxBatchMax,
  // are worth 2.3 % over 32: 0.697 -> 0.713 of the HBM peak on the box of that run)
  if (n >= 4) launch_wide_cfg<CH, 6, 4, kAuxNT, false, 4, 0>(dst, src, st, order, mix, slots, n);
  else if (n >= 2) launch_wide_cfg<CH, 3, 4, kAuxNT, false, 8, 0>(dst, src, st, order, mix, slots, n);
  else launch_wide_cfg<CH, 2, 4, kAuxNT, false, 8, 0>(dst, src, st, order, mix, slots, n);
#endif
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

template <int CH> int launch_fast(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st) {
  const int row_bytes = dst->ncols * CH;
  const int th = tuning("box.rows", 2);
  const int nt = tuning("box.nt", 1);
  // 2 = streaming DPP kernel on buffer descriptors (default), 1 = its flat-pointer predecessor (also the fallback when a
  // descriptor's 32-bit byte count cannot span the image), 0 = LDS-staged tile kernel
  int impl = tuning("box.impl", 2);
  if (impl == 2 && !fits_descriptor(dst, src)) impl = 1;
  uint8_t* dp = (uint8_t*)dst->first_pixel; const uint8_t* sp = (const uint8_t*)src->first_pixel;
  const int guard = src->border == 2 ? 1 : 0;
  if (impl == 2) return launch_wide<CH>(dst, src, st);
  if (impl == 1) {
    const int nstrips = (row_bytes + kStripOut - 1) / kStripOut;
    auto go = [&](auto RWc, auto NTc) {
      constexpr int RW = decltype(RWc)::value; constexpr bool NT = decltype(NTc)::value;
      int wpb = tuning("box.waves_per_block", 4);
      if (wpb != 1 && wpb != 2) wpb = 4;
      const int nblk_y = (dst->nrows + wpb * RW - 1) / (wpb * RW);
      if (CH == 3 && tuning("box.probe", 0))  // data-movement probe, never used by the product path
        box_u8_stream_kernel<CH, 5, 5, RW, NT, 1><<<nstrips * nblk_y, 64 * wpb, 0, st>>>(dp, sp, dst->pitch, src->pitch, dst->nrows, row_bytes, src->border * CH, nstrips, nblk_y, guard);
      else
        box_u8_stream_kernel<CH, 5, 5, RW, NT, 0><<<nstrips * nblk_y, 64 * wpb, 0, st>>>(dp, sp, dst->pitch, src->pitch, dst->nrows, row_bytes, src->border * CH, nstrips, nblk_y, guard);
    };
    auto pick = [&](auto NTc) {
      switch (th) {
        case 1: go(std::integral_constant<int, 1>(), NTc); break;
        case 2: go(std::integral_constant<int, 2>(), NTc); break;
        case 8: go(std::integral_constant<int, 8>(), NTc); break;
        case 16: go(std::integral_constant<int, 16>(), NTc); break;
        case 4: go(std::integral_constant<int, 4>(), NTc); break;
        default: go(std::integral_constant<int, 2>(), NTc); break;
      }
    };
    if (nt) pick(std::true_type()); else pick(std::false_type());
  } else {
    const int nblk_x = (row_bytes + kTileW - 1) / kTileW;
    auto go = [&](auto RW, auto NTc) {
      constexpr int TH = decltype(RW)::value; constexpr bool NT = decltype(NTc)::value;
      const int nblk_y = (dst->nrows + TH - 1) / TH;
      box5x5_u8_lds_kernel<CH, TH, NT><<<nblk_x * nblk_y, 256, 0, st>>>(dp, sp, dst->pitch, src->pitch, dst->nrows, row_bytes, src->border * CH, nblk_x,
                                                                        nblk_y, guard);
    };
    auto pick = [&](auto NTc) {
      switch (th) {
        case 16: go(std::integral_constant<int, 16>(), NTc); break;
        case 32: go(std::integral_constant<int, 32>(), NTc); break;
        default: go(std::integral_constant<int, 8>(), NTc); break;
      }
    };
    if (nt) pick(std::true_type()); else pick(std::false_type());
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

}  // namespace

// other odd windows up to 7 x 7 on 8-bit images through the same streaming kernel (2 output rows per wave, non-temporal stores)
template <int CH, int KR, int KC> int launch_stream_window(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st) {
  if (fits_descriptor(dst, src)) {
    launch_wide_cfg<CH, 2, 4, kAuxNT, false, 8, 0, KR, KC>(dst, src, st, 0, 0, 0);
    VPP_LAUNCH_CHECK();
    return VPP_OK;
  }
  const int row_bytes = dst->ncols * CH;
  const int nstrips = (row_bytes + kStripOut - 1) / kStripOut;
  constexpr int RW = 2, WPB = 4;
  const int nblk_y = (dst->nrows + WPB * RW - 1) / (WPB * RW);
  const int guard = src->border == KR / 2 ? 1 : 0;  // the window's first / last row is the allocation's: no slack before / after it
  box_u8_stream_kernel<CH, KR, KC, RW, true, 0><<<nstrips * nblk_y, 64 * WPB, 0, st>>>((uint8_t*)dst->first_pixel, (const uint8_t*)src->first_pixel, dst->pitch, src->pitch,
                                                                                      dst->nrows, row_bytes, src->border * CH, nstrips, nblk_y, guard);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
template <int CH> int launch_stream_windows(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, hipStream_t st) {
  const int key = R * 10 + C;
  switch (key) {
    case 33: return launch_stream_window<CH, 3, 3>(dst, src, st);
    case 35: return launch_stream_window<CH, 3, 5>(dst, src, st);
    case 53: return launch_stream_window<CH, 5, 3>(dst, src, st);
    case 73: return launch_stream_window<CH, 7, 3>(dst, src, st);
    case 75: return launch_stream_window<CH, 7, 5>(dst, src, st);
  }
  if constexpr (CH <= 2) {  // 7 taps at stride CH reach 3 CH <= 8 halo bytes
    switch (key) {
      case 37: return launch_stream_window<CH, 3, 7>(dst, src, st);
      case 57: return launch_stream_window<CH, 5, 7>(dst, src, st);
      case 77: return launch_stream_window<CH, 7, 7>(dst, src, st);
    }
  }
  return -1;  // not a streaming window: the caller falls back to the LDS-tiled generic kernel
}

#ifndef VPP_BOX_LAB
// ---- the per-frame call form on a recorded stream ----------------------------------------------------------------------------------------------------------
// The reference filters one frame per call (benchmarks/box_5x5_filter2.cc:43-69).  Eagerly that is one launch per call, and a 50 MB launch reaches 47 % of
// the HBM peak (ramp + drain behind a kernel boundary; tools/overlap_lab.hip: the AQL barrier bit cannot be dropped on this chip, two queues reach 59 %).
// While a stream is being RECORDED through vpp_graph_begin nothing runs yet, so vpp_box_filter holds its frame back in the calling thread's window (common.hpp:
// the mechanism of the *_deferred entry points) and a window that closes — 64 frames, a frame related to a pending one, any other call, vpp_graph_end — records
// ONE node of the batched kernel.  A recorded loop of per-frame calls thereby replays as the launches of vpp_box_filter_batch, with the results of the calls in
// sequence.  (Rounds 4-5 re-parameterised the previous call's node instead — hipGraphKernelNodeSetParams on a graph under capture; see LABNOTES.md, round 6.)
namespace {
inline bool box_batchable(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C) {   // the frames the batched streaming kernel serves (the test of vpp_box_filter_batch)
  return dst->dtype == VPP_U8 && R == 5 && C == 5 && dst->channels <= 4 && src->border >= 2 && aligned16(dst) && aligned16(src) && fits_descriptor(dst, src) &&
         !tuning("box.force_generic", 0) && tuning("box.impl", 2) == 2 && tuning("box.batch", 1);
}
}  // namespace
#endif

// n frames of one geometry, one launch (u8 5x5 images the streaming kernel serves; anything else goes out as n calls of vpp_box_filter)
extern "C" int vpp_box_filter(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, void* stream);
extern "C" int vpp_box_filter_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, int R, int C, void* stream) {
  VPP_REQUIRE(n >= 0 && (n == 0 || (dst && src)), VPP_ERR_INVALID_ARG, "vpp_box_filter_batch: invalid argument");
  if (n == 0) return VPP_OK;
  bool same = true;
  for (int k = 0; k < n; k++) {
    VPP_REQUIRE(valid_desc(&dst[k]) && valid_desc(&src[k]), VPP_ERR_INVALID_ARG, "vpp_box_filter_batch: invalid descriptor %d", k);
    same = same && same_domain(&dst[k], &dst[0]) && same_type(&dst[k], &dst[0]) && dst[k].pitch == dst[0].pitch && same_domain(&src[k], &dst[0]) && same_type(&src[k], &dst[0]) &&
           src[k].pitch == src[0].pitch && src[k].border == src[0].border && dst[k].first_pixel != src[k].first_pixel;
  }
  const bool wide = same && n > 1 && R == 5 && C == 5 && dst[0].dtype == VPP_U8 && dst[0].channels <= 4 && src[0].border >= 2 && !tuning("box.force_generic", 0) &&
                    tuning("box.impl", 2) == 2 && tuning("box.batch", 1);
  bool ok16 = wide;
  for (int k = 0; ok16 && k < n; k++) ok16 = aligned16(&dst[k]) && aligned16(&src[k]) && fits_descriptor(&dst[k], &src[k]);
  // "the results of n calls one after the other": a frame whose result is another frame's source (frames of a ring, a chain) or overlaps another
  // result is only that in sequence — such a batch goes out as the n calls
  if (ok16) { const vpp_image_desc* srcs[1] = {src}; ok16 = !batch_frames_interfere(n, dst, srcs, 1); }
  if (ok16) {
    hipStream_t st = as_stream(stream);
    switch (dst[0].channels) {
      case 1: return launch_wide<1>(dst, src, st, n);
      case 2: return launch_wide<2>(dst, src, st, n);
      case 3: return launch_wide<3>(dst, src, st, n);
      case 4: return launch_wide<4>(dst, src, st, n);
    }
  }
  for (int k = 0; k < n; k++) { const int rc = vpp_box_filter(&dst[k], &src[k], R, C, stream); if (rc) return rc; }
  return VPP_OK;
}

// diagnostics (not part of include/vpp_amd.h): the data movement of the batched 4K vuchar3 kernel alone — the same instance (6 rows per wave, 4 strips per
// workgroup, one XCD remap over the batch, the same descriptor loads and non-temporal stores) with the arithmetic compiled out: every lane stores the
// centre row's 16 bytes it loaded.  bench.py times it beside the headline so that a roofline fraction can be read against what a plain copy of the
// same geometry reaches on the same box in the same run (roofline.copy_frac).
extern "C" int vpp_debug_box_copy_batch(const vpp_image_desc* dst, const vpp_image_desc* src, int n, void* stream) {
  VPP_REQUIRE(n >= 1 && dst && src, VPP_ERR_INVALID_ARG, "vpp_debug_box_copy_batch: invalid argument");
  for (int k = 0; k < n; k++)
    VPP_REQUIRE(valid_desc(&dst[k]) && valid_desc(&src[k]) && dst[k].dtype == VPP_U8 && dst[k].channels == 3 && same_domain(&dst[k], &dst[0]) && same_domain(&src[k], &dst[0]) &&
                same_type(&src[k], &dst[0]) && src[k].pitch == src[0].pitch && dst[k].pitch == dst[0].pitch && src[k].border == src[0].border && src[k].border >= 2 &&
                aligned16(&dst[k]) && aligned16(&src[k]) && fits_descriptor(&dst[k], &src[k]), VPP_ERR_UNSUPPORTED, "vpp_debug_box_copy_batch: frame %d: 16-byte aligned vuchar3 frames of one geometry", k);
  launch_wide_cfg<3, 6, 4, kAuxNT, false, 4, 1>(dst, src, as_stream(stream), tuning("box.order", 0), 0, 8192, n);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

#ifndef VPP_BOX_LAB
// The per-frame call form without its per-frame launch (common.hpp, "held-back per-frame calls"): the frame joins the calling thread's window; argument errors
// are reported here, at the call, as vpp_box_filter reports them.  Frames the batched streaming kernel does not serve go out at once — behind the window, which
// as_stream() launches first.
extern "C" int vpp_box_filter_deferred(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(src), VPP_ERR_INVALID_ARG, "vpp_box_filter: invalid descriptor");
  VPP_REQUIRE(same_domain(dst, src) && same_type(dst, src), VPP_ERR_INVALID_ARG, "vpp_box_filter: domain/type mismatch");
  VPP_REQUIRE(R > 0 && C > 0 && (R & 1) && (C & 1), VPP_ERR_INVALID_ARG, "vpp_box_filter: window must be odd x odd");
  VPP_REQUIRE(src->border >= (R > C ? R : C) / 2, VPP_ERR_BORDER_TOO_SMALL, "vpp_box_filter: src border %d < %d", src->border, (R > C ? R : C) / 2);
  VPP_REQUIRE(dst->first_pixel != src->first_pixel, VPP_ERR_INVALID_ARG, "vpp_box_filter: in-place not supported");
  // (a stream captured by other means than vpp_graph_begin ends its capture where this library cannot see it: nothing is held back there)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  const bool hold = !g_defer_bypass && tuning("defer", 1) && box_batchable(dst, src, R, C) &&
                    (cap == hipStreamCaptureStatusNone || (defer_recording(stream) && tuning("box.coalesce", 1)));
  if (!hold) return vpp_box_filter(dst, src, R, C, stream);
  return defer_call(kDeferBox, R, C, stream, dst, src, nullptr);
}
#endif

extern "C" int vpp_box_filter(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(src), VPP_ERR_INVALID_ARG, "vpp_box_filter: invalid descriptor");
  VPP_REQUIRE(same_domain(dst, src) && same_type(dst, src), VPP_ERR_INVALID_ARG, "vpp_box_filter: domain/type mismatch");
  VPP_REQUIRE(R > 0 && C > 0 && (R & 1) && (C & 1), VPP_ERR_INVALID_ARG, "vpp_box_filter: window must be odd x odd");
  VPP_REQUIRE(src->border >= (R > C ? R : C) / 2, VPP_ERR_BORDER_TOO_SMALL, "vpp_box_filter: src border %d < %d", src->border, (R > C ? R : C) / 2);
  VPP_REQUIRE(dst->first_pixel != src->first_pixel, VPP_ERR_INVALID_ARG, "vpp_box_filter: in-place not supported");
#ifndef VPP_BOX_LAB
  // on a stream this thread records through vpp_graph_begin: the frame is held back and recorded as part of ONE batched node (above)
  if (!g_defer_bypass && defer_recording(stream) && tuning("box.coalesce", 1) && box_batchable(dst, src, R, C)) return defer_call(kDeferBox, R, C, stream, dst, src, nullptr);
#endif
  hipStream_t st = as_stream(stream);
#ifndef VPP_BOX_LAB
  // while the stream is recorded into a launch graph: calls on unrelated images become sibling nodes (common.hpp, IndependentCall)
  const Extent wr = extent_of(*dst), rd = extent_of(*src);
  IndependentCall side_by_side(st, &wr, 1, &rd, 1);
#endif
  if (dst->dtype == VPP_U8 && R == 5 && C == 5 && dst->channels <= 4 && aligned16(dst) && aligned16(src) && !tuning("box.force_generic", 0)) {
    switch (dst->channels) {
      case 1: return launch_fast<1>(dst, src, st);
      case 2: return launch_fast<2>(dst, src, st);
      case 3: return launch_fast<3>(dst, src, st);
      case 4: return launch_fast<4>(dst, src, st);
    }
  }
  if (dst->dtype == VPP_U8 && R <= 7 && C <= 7 && R > 1 && C > 1 && dst->channels <= 4 && aligned16(dst) && aligned16(src) && !tuning("box.force_generic", 0)) {
    int rc = -1;
    switch (dst->channels) {
      case 1: rc = launch_stream_windows<1>(dst, src, R, C, st); break;
      case 2: rc = launch_stream_windows<2>(dst, src, R, C, st); break;
      case 3: rc = launch_stream_windows<3>(dst, src, R, C, st); break;
      case 4: rc = launch_stream_windows<4>(dst, src, R, C, st); break;
    }
    if (rc >= 0) return rc;
  }
  if ((dst->dtype == VPP_I32 || dst->dtype == VPP_U32 || dst->dtype == VPP_F32) && dst->channels == 1 && aligned16(dst) && aligned16(src) && !tuning("box.force_generic", 0)) {
    const int rc = dst->dtype == VPP_I32 ? launch_w32_windows<int32_t>(dst, src, R, C, st)
                 : dst->dtype == VPP_U32 ? launch_w32_windows<uint32_t>(dst, src, R, C, st) : launch_w32_windows<float>(dst, src, R, C, st);
    if (rc >= 0) return rc;
  }
  switch (dst->dtype) {
    case VPP_U8: return launch_generic<uint8_t, int>(dst, src, R, C, st);
    case VPP_I8: return launch_generic<int8_t, int>(dst, src, R, C, st);
    case VPP_U16: return launch_generic<uint16_t, int>(dst, src, R, C, st);
    case VPP_I16: return launch_generic<int16_t, int>(dst, src, R, C, st);
    case VPP_I32: return launch_generic<int32_t, int32_t>(dst, src, R, C, st);
    case VPP_U32: return launch_generic<uint32_t, uint32_t>(dst, src, R, C, st);
    case VPP_F32: return launch_generic<float, float>(dst, src, R, C, st);
  }
  return VPP_ERR_UNSUPPORTED;
}
