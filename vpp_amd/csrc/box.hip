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
#include <type_traits>
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

// ---- 32-bit 5x5: int / unsigned (the element type of the reference's own benchmark, benchmarks/box_5x5_filter.cc:165-171,187-191)
//      and float (taps added in the reference's order) ----
// Same shape as the u8 streaming kernel with one pixel per dword: a lane owns 4 consecutive pixels (16 B), keeps their
// 5-row column sums (integer adds are order-independent, so the running add / subtract is exact, wrap-around included),
// takes the two halo sums of each side from lane -1 / +1 over DPP, and lanes 1..62 store 4 truncating quotients (`/ 25` in T).
// All RW + 4 row loads are issued up front.  Elements outside [-border, ncols + border) x [-border, nrows + border) are
// never read (a straddling chunk takes predicated dword loads), so border == 2 needs no separate guarded body.
constexpr int kW32StripOut = 62 * 4;

template <class T, int KR, int KC, int RW, bool NT>
__global__ __launch_bounds__(256) void box_w32_stream_kernel(T* __restrict__ dp, const T* __restrict__ sp, int dpitch, int spitch,
                                                                int nrows, int ncols, int border, int nstrips, int nblk_y) {
  const unsigned nb = (unsigned)nstrips * (unsigned)nblk_y;
  const unsigned lb = xcd_remap(blockIdx.x, nb);  // consecutive logical blocks = vertically adjacent row blocks of one strip
  const int s = lb / nblk_y, by = lb - s * nblk_y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = s * kW32StripOut - 4 + lane * 4;   // first pixel of this lane's chunk
  const int r0 = (by * (int)(blockDim.x >> 6) + wv) * RW;
  if (r0 >= nrows) return;
  static_assert((KR & 1) && (KC & 1) && KR <= 7 && KC <= 5, "two halo pixels per side come over DPP");
  constexpr int HR = KR / 2, C0 = 2 - KC / 2;  // first window column of output pixel 0 within the 8-pixel register window
  const int lo = -border, hi = ncols + border;
  const bool in_reach = x + 4 > lo && x < ncols + 4, inside = x >= lo && x + 4 <= hi;
  const bool writer = lane >= 1 && lane <= 62 && x < ncols;
  typedef uint32_t U;  // raw dwords: integer sums wrap like the hardware's adds (no signed-overflow UB in the source); floats are bit-cast
  // Every load is unconditional so that the RW + 4 rows are requested back to back: rows past the bottom border are clamped
  // (they only feed output rows >= nrows, which are not stored), lanes outside the strip's reach read a valid chunk that
  // nobody consumes, and only the waves that contain a chunk straddling a row end (first / last strip) take the
  // per-element form (clamped address + select), as a wave-uniform choice.
  const bool vector_wave = __all(inside || !in_reach);
  U raw[RW + KR - 1][4];
#pragma unroll
  for (int k = 0; k < RW + KR - 1; k++) {
    const int r = min(r0 - HR + k, nrows - 1 + border);
    const U* row = (const U*)((const uint8_t*)sp + (ptrdiff_t)r * spitch);
    if (vector_wave) {
      const u32x4 v = *(const u32x4*)(row + (in_reach ? x : 0));
      raw[k][0] = v.x; raw[k][1] = v.y; raw[k][2] = v.z; raw[k][3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) { const int c = x + i; const U v = row[min(max(c, lo), hi - 1)]; raw[k][i] = (c >= lo && c < hi) ? v : 0; }
    }
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
  const int rows = tuning("box.rows32", 4), nt = tuning("box.nt", 1);   // measured 4K int: 1 -> 20.4, 2 -> 17.4, 4 -> 16.6, 8 -> 21.8 us
  int wpb = tuning("box.waves_per_block", 4);
  if (wpb != 1 && wpb != 2) wpb = 4;
  const int nstrips = (dst->ncols + kW32StripOut - 1) / kW32StripOut;
  auto go = [&](auto RWc, auto NTc) {
    constexpr int RW = decltype(RWc)::value; constexpr bool NT = decltype(NTc)::value;
    const int nblk_y = (dst->nrows + wpb * RW - 1) / (wpb * RW);
    box_w32_stream_kernel<T, KR, KC, RW, NT><<<nstrips * nblk_y, 64 * wpb, 0, st>>>((T*)dst->first_pixel, (const T*)src->first_pixel, dst->pitch, src->pitch,
                                                                              dst->nrows, dst->ncols, src->border, nstrips, nblk_y);
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

template <int CH> int launch_fast(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st) {
  const int row_bytes = dst->ncols * CH;
  const int th = tuning("box.rows", 2);
  const int nt = tuning("box.nt", 1);
  const int impl = tuning("box.impl", 1);  // 1 = streaming DPP kernel, 0 = LDS-staged tile kernel
  uint8_t* dp = (uint8_t*)dst->first_pixel; const uint8_t* sp = (const uint8_t*)src->first_pixel;
  const int guard = src->border == 2 ? 1 : 0;
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

extern "C" int vpp_box_filter(const vpp_image_desc* dst, const vpp_image_desc* src, int R, int C, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(src), VPP_ERR_INVALID_ARG, "vpp_box_filter: invalid descriptor");
  VPP_REQUIRE(same_domain(dst, src) && same_type(dst, src), VPP_ERR_INVALID_ARG, "vpp_box_filter: domain/type mismatch");
  VPP_REQUIRE(R > 0 && C > 0 && (R & 1) && (C & 1), VPP_ERR_INVALID_ARG, "vpp_box_filter: window must be odd x odd");
  VPP_REQUIRE(src->border >= (R > C ? R : C) / 2, VPP_ERR_BORDER_TOO_SMALL, "vpp_box_filter: src border %d < %d", src->border, (R > C ? R : C) / 2);
  VPP_REQUIRE(dst->first_pixel != src->first_pixel, VPP_ERR_INVALID_ARG, "vpp_box_filter: in-place not supported");
  hipStream_t st = as_stream(stream);
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
