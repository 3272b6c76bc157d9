// box.hip — K2: neighbourhood (box_nbh2d / relative_access) mean filter.
// Reference: vpp/core/pixel_wise.hpp:14-25,57-63 (relative_access range), vpp/core/relative_accessor.hh:26-33
// (nbh(dr,dc) = line[dr][col+dc]), kernel lambda of benchmarks/box_5x5_filter2.cc:73-80 / examples/box_filter.cc:23-32.
//
// Two kernels:
//  * box5x5_u8_kernel<CH>  — the headline path (vuchar3 4K).  A row of CH-interleaved u8 pixels is a row of bytes in
//    which the horizontal neighbour sits CH bytes away, so the filter is a 1-channel 5x5 stencil with column stride CH.
//    One lane owns 16 consecutive output bytes (one 16-B store per row) and marches down ROWS rows keeping, in registers,
//    the running 5-row column sums of a 32-byte window as packed u16 pairs (even/odd bytes split with 0x00FF00FF masks).
//    Horizontal taps are register selects + v_alignbit on those pairs; packed sums never exceed 25*255 = 6375 < 2^16 so
//    plain 32-bit adds/subs act as two independent 16-bit lanes.  Exact truncating /25: (x * 671089) >> 24 for x <= 6375.
//    HBM traffic: 1 read + 1 write per byte (6 B/px for vuchar3); the (ROWS+4)/ROWS vertical re-read and the 2x8 B
//    horizontal halo are served by L2/MALL.  Row blocks are XCD-remapped so vertical neighbours share an L2.
//  * box_generic_kernel<T,S> — any dtype / window: LDS tile (+halo), taps summed in row-major order in the promoted
//    type (order matters for float), C++ `/ (R*C)`.
#include "common.hpp"
#include <type_traits>
using namespace vpp_amd;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- fast path -------------------------------------------------------------------------------------------
struct Win { uint32_t e[8], o[8]; };  // packed u16 pairs: e[i] = bytes (4i, 4i+2), o[i] = bytes (4i+1, 4i+3) of the 32-B window

__device__ __forceinline__ uint32_t ld_bytes_guarded(const uint8_t* p, int off, int lo, int hi) {
  // little-endian dword assembled from bytes p[off..off+3] that fall inside [lo, hi); others 0
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int q = off + k;
    if (q >= lo && q < hi) v |= (uint32_t)p[q] << (8 * k);
  }
  return v;
}

// Loads the window bytes [x-8, x+24) of one row.  [lo, hi) = addressable bytes of the row relative to its pixel 0.
// Each of the three pieces (8 B left halo, 16 B body, 8 B right halo) is one vector load when it lies inside [lo, hi),
// else it is assembled from the bytes that do (only the first / last lane of a row, and ragged row ends).
__device__ __forceinline__ void load_window(const uint8_t* row, int x, int lo, int hi, uint32_t w[8]) {
  if (x - 8 >= lo) { u32x2 l = *(const u32x2*)(row + x - 8); w[0] = l.x; w[1] = l.y; }
  else { w[0] = ld_bytes_guarded(row, x - 8, lo, hi); w[1] = ld_bytes_guarded(row, x - 4, lo, hi); }
  if (x + 16 <= hi) { u32x4 m = *(const u32x4*)(row + x); w[2] = m.x; w[3] = m.y; w[4] = m.z; w[5] = m.w; }
  else {
#pragma unroll
    for (int i = 0; i < 4; i++) w[2 + i] = ld_bytes_guarded(row, x + 4 * i, lo, hi);
  }
  if (x + 24 <= hi) { u32x2 r = *(const u32x2*)(row + x + 16); w[6] = r.x; w[7] = r.y; }
  else { w[6] = ld_bytes_guarded(row, x + 16, lo, hi); w[7] = ld_bytes_guarded(row, x + 20, lo, hi); }
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
template <int Q, int CH> __device__ __forceinline__ uint32_t hsum5(const uint32_t* E, const uint32_t* O) {
  return pair_at<Q - 2 * CH>(E, O) + pair_at<Q - CH>(E, O) + pair_at<Q>(E, O) + pair_at<Q + CH>(E, O) + pair_at<Q + 2 * CH>(E, O);
}
__device__ __forceinline__ uint32_t div25(uint32_t x) { return __umul24(x, 671089u) >> 24; }  // exact for x <= 6375
template <int I, int CH> __device__ __forceinline__ uint32_t out_dword(const uint32_t* E, const uint32_t* O) {
  // output bytes 4I..4I+3 of the lane's 16 (window positions 8+4I ..)
  const uint32_t se = hsum5<8 + 4 * I, CH>(E, O);      // sums for bytes (4I, 4I+2)
  const uint32_t so = hsum5<8 + 4 * I + 1, CH>(E, O);  // sums for bytes (4I+1, 4I+3)
  return div25(se & 0xFFFFu) | (div25(so & 0xFFFFu) << 8) | (div25(se >> 16) << 16) | (div25(so >> 16) << 24);
}

template <int CH, int ROWS>
__global__ __launch_bounds__(256) void box5x5_u8_kernel(DImg dst, DImg src, int row_bytes, int nblk_x, int nblk_y) {
  static_assert(CH >= 1 && CH <= 4, "window holds 2*CH <= 8 halo bytes");
  // logical block id, XCD-remapped so that vertically adjacent row blocks run on the same XCD / L2
  const unsigned nb = (unsigned)nblk_x * (unsigned)nblk_y;
  const unsigned lb = xcd_remap(blockIdx.x, nb);
  const int by = lb / nblk_x, bx = lb - by * nblk_x;
  const int x = (bx * 256 + threadIdx.x) * 16;
  if (x >= row_bytes) return;
  const int r0 = by * ROWS;
  const int lo = -src.border * CH, hi = row_bytes + src.border * CH;
  const bool full_store = x + 16 <= row_bytes;

  uint32_t ring_e[5][8], ring_o[5][8];  // unpacked rows r-2..r+2 (register ring, fully unrolled)
  uint32_t E[8], O[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { E[i] = 0; O[i] = 0; }

  auto load_unpack = [&](int r, uint32_t* e, uint32_t* o) {
    uint32_t w[8];
    load_window(src.p0 + (ptrdiff_t)r * src.pitch, x, lo, hi, w);
#pragma unroll
    for (int i = 0; i < 8; i++) { e[i] = w[i] & 0x00FF00FFu; o[i] = (w[i] >> 8) & 0x00FF00FFu; }
  };

  // prologue: rows r0-2 .. r0+1
#pragma unroll
  for (int k = 0; k < 4; k++) {
    load_unpack(r0 - 2 + k, ring_e[k], ring_o[k]);
#pragma unroll
    for (int i = 0; i < 8; i++) { E[i] += ring_e[k][i]; O[i] += ring_o[k][i]; }
  }
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int r = r0 + j;
    if (r >= dst.nr) break;
    const int slot_new = (j + 4) % 5;  // slot of row r+2; it previously held row r-3
    if (j > 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) { E[i] -= ring_e[slot_new][i]; O[i] -= ring_o[slot_new][i]; }
    }
    load_unpack(r + 2, ring_e[slot_new], ring_o[slot_new]);
#pragma unroll
    for (int i = 0; i < 8; i++) { E[i] += ring_e[slot_new][i]; O[i] += ring_o[slot_new][i]; }

    u32x4 res;
    res.x = out_dword<0, CH>(E, O); res.y = out_dword<1, CH>(E, O); res.z = out_dword<2, CH>(E, O); res.w = out_dword<3, CH>(E, O);
    uint8_t* drow = dst.p0 + (ptrdiff_t)r * dst.pitch + x;
    if (full_store) *(u32x4*)drow = res;
    else {
      union { u32x4 v; uint8_t b[16]; } u; u.v = res;
      for (int k = 0; k < row_bytes - x; k++) drow[k] = u.b[k];
    }
  }
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
  for (int idx = threadIdx.x; idx < TW * TH; idx += 256) {
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
  const size_t smem = (size_t)lds_w * (TH + R - 1) * sizeof(S);
  VPP_REQUIRE(smem <= 64 * 1024, VPP_ERR_UNSUPPORTED, "vpp_box_filter: window %dx%d too large for the LDS tile", R, C);
  dim3 grid((ncomp + TW - 1) / TW, (dst->nrows + TH - 1) / TH);
  box_generic_kernel<T, S, TW, TH><<<grid, 256, smem, st>>>(dimg(dst), dimg(src), R, C, ncomp, lds_w);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

template <int CH> int launch_fast(const vpp_image_desc* dst, const vpp_image_desc* src, hipStream_t st) {
  const int row_bytes = dst->ncols * CH;
  const int rows = tuning("box.rows", 8);
  const int nblk_x = (row_bytes + 256 * 16 - 1) / (256 * 16);
  auto go = [&](auto RW) {
    constexpr int ROWS = decltype(RW)::value;
    const int nblk_y = (dst->nrows + ROWS - 1) / ROWS;
    box5x5_u8_kernel<CH, ROWS><<<nblk_x * nblk_y, 256, 0, st>>>(dimg(dst), dimg(src), row_bytes, nblk_x, nblk_y);
  };
  switch (rows) {
    case 4: go(std::integral_constant<int, 4>()); break;
    case 16: go(std::integral_constant<int, 16>()); break;
    case 32: go(std::integral_constant<int, 32>()); break;
    default: go(std::integral_constant<int, 8>()); break;
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

}  // namespace

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
