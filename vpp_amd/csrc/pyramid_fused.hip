// pyramid_fused.hip — a whole factor-2 pyramid in ONE launch.
// Reference: pyramid<V,N>::update / propagate_level0 (vpp/core/pyramid.hh:169-198: copy into level 0, fill_border_mirror, then per
// level antialiasing_lowpass_filter :12-59 + subsample2 :62-81 + fill_border_mirror :182) and, for the gradient pyramid of
// pyrlk / lucas_kanade, scharr (vpp/algorithms/filters/scharr.hh:46-87) into level 0 first (pyrlk_opencv_comparison.cc:56-60,
// lucas_kanade.hpp:151-157).
//
// The per-level launches of pyramid.hip are 4-10 us kernels of a few hundred workgroups each plus a border launch per level: a
// frame's pyramids were ~10 launches and launch-latency bound.  Here a workgroup owns a T x T tile of the COARSEST level and
// produces everything underneath it: it loads the level-0 pixels the tile depends on into LDS (for a 3-level pyramid a
// (4T+9)^2 patch), runs the reference's two passes per level inside LDS —
//     H(rr, 2c) = V(tap5 of level l around column 2c, columns past an edge = the level's mirror-filled border),
//     L(r, c)   = (2r < nr && 2c < nc) ? V(tap5 of H over rows 2r-2 .. 2r+2, rows past an edge mirrored as tmp's border is) : 0
//   (the zero is the reference's unfilled temp border) —
// and writes, for every level, the region it owns (level l: 2^(L-1-l) T rows / columns) together with the mirrored copies of
// the pixels that sit within `border` of an edge (fill_border_mirror).  Every value is computed by the same expression, in
// the same type and order, from the same truncated intermediate values as the chain of per-level kernels, so the result is
// bit-identical; the halo recomputation costs (2T+3)^2 / (2T)^2 per level, all of it on chip.
#include "common.hpp"
#include <type_traits>
#include "sdof_tail.hpp"
using namespace vpp_amd;

namespace {

template <class T> struct Promo { typedef int type; };
template <> struct Promo<float> { typedef float type; };
template <> struct Promo<uint32_t> { typedef uint32_t type; };

template <class T, class S> __device__ __forceinline__ T tap5(S a, S b, S c, S d, S e) { return (T)((1 * a + 4 * b + 6 * c + 4 * d + 1 * e) / 16); }
__device__ __forceinline__ int mirror_idx(int r, int n) {
  const int m = r < 0 ? -r - 1 : (r >= n ? 2 * n - r - 1 : r);
  return min(max(m, 0), n - 1);  // only differs for extents below 2, where the reference reads outside its buffer
}

struct Rect { int r0, c0, h, w; };  // domain rows [r0, r0 + h), columns [c0, c0 + w) of one level, held in LDS row-major

constexpr int kMaxLevels = 3;
struct Chain {
  DImg lv[kMaxLevels];   // the pyramid's levels (lv[0] = finest)
  DImg src;              // the image level 0 is made from (copied, or differentiated)
  int nlevels;
};

// Level-0 producers.  fill(): the workgroup loads / computes the level-0 rect `rt` (columns dword-aligned, possibly a few past
// the right edge: those cells are never read) into LDS `top` (row pitch rt.w pixels); `aux` is producer-private LDS.
template <class T, int CH> struct CopySrc {   // level 0 = the source's domain pixels (pyramid.hh:196 copy)
  static constexpr int kAuxBytes = 16;
  template <int PITCH> static __device__ __forceinline__ void fill(const DImg& src, const Rect& rt, T* top, uint8_t*) {
    constexpr int PXB = (int)sizeof(T) * CH;
    static_assert((PITCH * PXB) % 4 == 0, "LDS rows are dword multiples");
    const bool vec = ((((uintptr_t)src.p0) | (uintptr_t)src.pitch) & 3) == 0 && (rt.w * PXB) % 4 == 0 && (rt.c0 * PXB) % 4 == 0;
    if (vec) {  // dword loads (the padded columns stay inside the row pitch: pitch % 4 == 0 and pitch >= ncols * PXB)
      constexpr int NDW = PITCH * PXB / 4;
      const int ndw = rt.w * PXB / 4;
      for (int idx = threadIdx.x; idx < rt.h * NDW; idx += blockDim.x) {
        const int y = idx / NDW, x = idx - y * NDW;
        if (x < ndw) ((uint32_t*)top)[idx] = ((const uint32_t*)(src.p0 + (ptrdiff_t)(rt.r0 + y) * src.pitch + (ptrdiff_t)rt.c0 * PXB))[x];
      }
    } else {
      for (int idx = threadIdx.x; idx < rt.h * PITCH; idx += blockDim.x) {
        const int y = idx / PITCH, x = idx - y * PITCH;
        if (x >= rt.w) continue;
        const int c = min(rt.c0 + x, src.nc - 1);
        const T* p = src.row<T>(rt.r0 + y) + c * CH;
#pragma unroll
        for (int k = 0; k < CH; k++) top[(size_t)idx * CH + k] = p[k];
      }
    }
  }
};
// level 0 = rgb_to_graylevel<uchar>(frame) (colorspace_conversions.hh:10-33; 4-channel :36-48 ignores the 4th component): the frame ingest of the
// reference's video loop (examples/video_extruder.cc:46-48) fused with the pyramid it feeds.  A thread turns the 4 * CH source bytes under four
// level-0 pixels (dword loads: the rect's columns are multiples of 4, so the byte offset 4 * CH * k is dword aligned) into one dword of LDS;
// the quotient (s * 43691) >> 17 is exact for s <= 765.
template <int CH> struct GraySrc {
  static constexpr int kAuxBytes = 16;
  template <int PITCH> static __device__ __forceinline__ void fill(const DImg& src, const Rect& rt, uint8_t* top, uint8_t*) {
    static_assert(PITCH % 4 == 0, "LDS rows are dword multiples");
    constexpr int NG = PITCH / 4;
    const int ng = rt.w / 4;
    const bool vec = ((((uintptr_t)src.p0) | (uintptr_t)src.pitch) & 3) == 0 && rt.c0 + rt.w <= src.nc;
    auto gray = [](uint32_t a, uint32_t b, uint32_t c) { return ((a + b + c) * 43691u) >> 17; };
    for (int idx = threadIdx.x; idx < rt.h * NG; idx += blockDim.x) {
      const int y = idx / NG, x = idx - y * NG;
      if (x >= ng) continue;
      const uint8_t* row = src.p0 + (ptrdiff_t)(rt.r0 + y) * src.pitch;
      uint32_t o = 0;
      if (vec) {
        const uint32_t* q = (const uint32_t*)(row + (ptrdiff_t)(rt.c0 + 4 * x) * CH);
        if constexpr (CH == 3) {
          const uint32_t a = q[0], b = q[1], c = q[2];
          o = gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u) | gray(a >> 24, b & 255u, (b >> 8) & 255u) << 8 |
              gray((b >> 16) & 255u, b >> 24, c & 255u) << 16 | gray((c >> 8) & 255u, (c >> 16) & 255u, c >> 24) << 24;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) { const uint32_t a = q[k]; o |= gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u) << (8 * k); }
        }
      } else {   // the rect's padding columns past the right edge (never read back) are clamped onto the last pixel; foreign pitches
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint8_t* px = row + (ptrdiff_t)min(rt.c0 + 4 * x + k, src.nc - 1) * CH;
          o |= gray(px[0], px[1], px[2]) << (8 * k);
        }
      }
      ((uint32_t*)top)[idx] = o;
    }
  }
};
template <class V> struct ScharrSrc {  // level 0 = scharr(img) (scharr.hh:46-87: arithmetic in V, `/ 32.f`, conversion to V); reads img's border
  static constexpr int kAuxPitch = 96;   // bytes per staged u8 row: rect width (<= 80) + 1-pixel halo, dword aligned
  static constexpr int kAuxBytes = kAuxPitch * 84;
  template <int PITCH> static __device__ __forceinline__ void fill(const DImg& in, const Rect& rt, V* top, uint8_t* aux) {
    // stage the u8 patch [r0 - 1, r0 + h] x [c0 - 1, c0 + w] (clipped to what img's border holds) with dword loads, then differentiate from LDS
    const int a0 = (rt.c0 - 1) & ~3, wbytes = ((rt.c0 + rt.w + 1 + 3) & ~3) - a0, ndw = wbytes / 4, rows = rt.h + 2;
    // a0 can reach column -4 and the far end column nc + 7: inside the border / the row's padding of any vpp-allocated image (border bytes are
    // rounded up to the alignment, >= 16), checked here for foreign pitches
    const bool vec = ((((uintptr_t)in.p0) | (uintptr_t)in.pitch) & 3) == 0 && in.border >= 4 && in.pitch >= in.nc + in.border + 8;
    if (vec) {
      for (int idx = threadIdx.x; idx < rows * ndw; idx += blockDim.x) {
        const int y = idx / ndw, x = idx - y * ndw;
        *(uint32_t*)(aux + y * kAuxPitch + 4 * x) = *(const uint32_t*)(in.p0 + (ptrdiff_t)(rt.r0 - 1 + y) * in.pitch + a0 + 4 * x);
      }
    } else {
      for (int idx = threadIdx.x; idx < rows * wbytes; idx += blockDim.x) {
        const int y = idx / wbytes, x = idx - y * wbytes;
        const int c = min(max(a0 + x, -in.border), in.nc + in.border - 1);
        aux[y * kAuxPitch + x] = in.row<uint8_t>(rt.r0 - 1 + y)[c];
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rt.h * PITCH; idx += blockDim.x) {
      const int y = idx / PITCH, x = idx - y * PITCH;
      if (x >= rt.w) continue;
      const uint8_t *row1 = aux + y * kAuxPitch + (rt.c0 + x - a0), *row2 = row1 + kAuxPitch, *row3 = row2 + kAuxPitch;
      const V a1 = (V)row1[-1], b1 = (V)row1[0], c1 = (V)row1[1], a2 = (V)row2[-1], c2 = (V)row2[1];
      const V a3 = (V)row3[-1], b3 = (V)row3[0], c3 = (V)row3[1];
      top[(size_t)idx * 2] = (V)((3 * a3 + 10 * b3 + 3 * c3 - 3 * a1 - 10 * b1 - 3 * c1) / 32.f);
      top[(size_t)idx * 2 + 1] = (V)((3 * c1 + 10 * c2 + 3 * c3 - 3 * a1 - 10 * a2 - 3 * a3) / 32.f);
    }
  }
};

// the mirrored copies of domain pixel (r, c) of `img` in its border (fill.hh:60-83); requires border <= nrows, ncols
template <class T, int CH> __device__ __forceinline__ void store_border_copies(const DImg& img, int r, int c, const T* v) {
  const int b = img.border, nr = img.nr, nc = img.nc;
  if (b == 0 || (r >= b && r < nr - b && c >= b && c < nc - b)) return;
  auto put = [&](int rr, int cc) { T* p = img.row<T>(rr) + cc * CH;
#pragma unroll
    for (int k = 0; k < CH; k++) p[k] = v[k]; };
  const int mr = r < b ? -r - 1 : (r >= nr - b ? 2 * nr - r - 1 : r);
  const int mr2 = (r < b && r >= nr - b) ? 2 * nr - r - 1 : mr;   // a row within b of both ends mirrors both ways
  const int mc = c < b ? -c - 1 : (c >= nc - b ? 2 * nc - c - 1 : c);
  const int mc2 = (c < b && c >= nc - b) ? 2 * nc - c - 1 : mc;
  if (mc != c) put(r, mc);
  if (mc2 != mc) put(r, mc2);
  if (mr != r) { put(mr, c); if (mc != c) put(mr, mc); if (mc2 != mc) put(mr, mc2); }
  if (mr2 != mr) { put(mr2, c); if (mc != c) put(mr2, mc); if (mc2 != mc) put(mr2, mc2); }
}
template <class T, int CH> __device__ __forceinline__ void store_with_border(const DImg& img, int r, int c, const T* v) {
  T* p = img.row<T>(r) + c * CH;
#pragma unroll
  for (int k = 0; k < CH; k++) p[k] = v[k];
  store_border_copies<T, CH>(img, r, c, v);
}
// the part [r0, r1) x [c0, c1) (at most WMAX wide) of an LDS rect (row pitch PITCH pixels) to `img`, rows as dword stores when
// everything is dword aligned, plus the border copies
template <class T, int CH, int PITCH, int WMAX> __device__ __forceinline__ void store_region(const DImg& img, const T* lds, const Rect& rc, int r0, int r1, int c0, int c1) {
  constexpr int PXB = (int)sizeof(T) * CH;
  const int w = c1 - c0, h = r1 - r0;
  if (w <= 0 || h <= 0) return;
  const bool vec = ((((uintptr_t)img.p0) | (uintptr_t)img.pitch) & 3) == 0 && (c0 * PXB) % 4 == 0 && (w * PXB) % 4 == 0 && ((c0 - rc.c0) * PXB) % 4 == 0 && (PITCH * PXB) % 4 == 0;
  if (vec) {
    constexpr int NDW = (WMAX * PXB + 3) / 4;
    const int ndw = w * PXB / 4;
    for (int idx = threadIdx.x; idx < h * NDW; idx += blockDim.x) {
      const int y = idx / NDW, x = idx - y * NDW;
      if (x < ndw) ((uint32_t*)(img.p0 + (ptrdiff_t)(r0 + y) * img.pitch + (ptrdiff_t)c0 * PXB))[x] = ((const uint32_t*)(lds + ((size_t)(r0 + y - rc.r0) * PITCH + (c0 - rc.c0)) * CH))[x];
    }
  } else {
    for (int idx = threadIdx.x; idx < h * WMAX; idx += blockDim.x) {
      const int y = idx / WMAX, x = idx - y * WMAX;
      if (x >= w) continue;
      const T* v = lds + ((size_t)(r0 + y - rc.r0) * PITCH + (c0 + x - rc.c0)) * CH;
      T* p = img.row<T>(r0 + y) + (c0 + x) * CH;
#pragma unroll
      for (int k = 0; k < CH; k++) p[k] = v[k];
    }
  }
  const int b = img.border;
  if (b > 0 && (r0 < b || r1 > img.nr - b || c0 < b || c1 > img.nc - b))   // the region touches the band that is mirrored into the border
    for (int idx = threadIdx.x; idx < h * WMAX; idx += blockDim.x) {
      const int y = idx / WMAX, x = idx - y * WMAX;
      if (x < w) store_border_copies<T, CH>(img, r0 + y, c0 + x, lds + ((size_t)(r0 + y - rc.r0) * PITCH + (c0 + x - rc.c0)) * CH);
    }
}

// one level down inside LDS: `in` (rect ri of a level with extent nr x nc, row pitch PIN) -> `out` (rect ro of the next level, row
// pitch POUT), hb = scratch for the horizontal pass (ri.h rows, row pitch POUT).  EDGE = false: the rect is away from every edge
// of the level, so no tap is mirrored and no output falls into the zero cells — plain offsets.
template <class T, class S, int CH, int PIN, int POUT, int HIN, bool EDGE>
__device__ __forceinline__ void down_stage_impl(const T* in, const Rect& ri, int nr, int nc, T* hb, T* out, const Rect& ro) {
  for (int idx = threadIdx.x; idx < HIN * POUT; idx += blockDim.x) {   // H(rr, 2c) for every held row rr and every wanted column c
    const int y = idx / POUT, x = idx - y * POUT;
    if (y >= ri.h || x >= ro.w) continue;
    const int cc = 2 * (ro.c0 + x);
    T* o = hb + (size_t)idx * CH;
    const T* row = in + (size_t)y * PIN * CH;
    if (!EDGE) {
      const T* p = row + (cc - 2 - ri.c0) * CH;
#pragma unroll
      for (int k = 0; k < CH; k++) o[k] = tap5<T, S>((S)p[k], (S)p[CH + k], (S)p[2 * CH + k], (S)p[3 * CH + k], (S)p[4 * CH + k]);
    } else if (cc < nc) {
      const int i0 = mirror_idx(cc - 2, nc) - ri.c0, i1 = mirror_idx(cc - 1, nc) - ri.c0, i2 = cc - ri.c0, i3 = mirror_idx(cc + 1, nc) - ri.c0, i4 = mirror_idx(cc + 2, nc) - ri.c0;
#pragma unroll
      for (int k = 0; k < CH; k++) o[k] = tap5<T, S>((S)row[i0 * CH + k], (S)row[i1 * CH + k], (S)row[i2 * CH + k], (S)row[i3 * CH + k], (S)row[i4 * CH + k]);
    } else {
#pragma unroll
      for (int k = 0; k < CH; k++) o[k] = 0;
    }
  }
  __syncthreads();
  constexpr int HOUT = (HIN - 3) / 2 + 1;   // rows of the next level a rect of HIN rows can feed
  for (int idx = threadIdx.x; idx < HOUT * POUT; idx += blockDim.x) {
    const int y = idx / POUT, x = idx - y * POUT;
    if (y >= ro.h || x >= ro.w) continue;
    const int r = ro.r0 + y, c = ro.c0 + x;
    T* o = out + (size_t)idx * CH;
    if (!EDGE) {
      const T* h = hb + ((size_t)(2 * r - 2 - ri.r0) * POUT + x) * CH;
#pragma unroll
      for (int k = 0; k < CH; k++) o[k] = tap5<T, S>((S)h[k], (S)h[POUT * CH + k], (S)h[2 * POUT * CH + k], (S)h[3 * POUT * CH + k], (S)h[4 * POUT * CH + k]);
    } else if (2 * r < nr && 2 * c < nc) {
      const T* h0 = hb + ((size_t)(mirror_idx(2 * r - 2, nr) - ri.r0) * POUT + x) * CH;
      const T* h1 = hb + ((size_t)(mirror_idx(2 * r - 1, nr) - ri.r0) * POUT + x) * CH;
      const T* h2 = hb + ((size_t)(2 * r - ri.r0) * POUT + x) * CH;
      const T* h3 = hb + ((size_t)(mirror_idx(2 * r + 1, nr) - ri.r0) * POUT + x) * CH;
      const T* h4 = hb + ((size_t)(mirror_idx(2 * r + 2, nr) - ri.r0) * POUT + x) * CH;
#pragma unroll
      for (int k = 0; k < CH; k++) o[k] = tap5<T, S>((S)h0[k], (S)h1[k], (S)h2[k], (S)h3[k], (S)h4[k]);
    } else {
#pragma unroll
      for (int k = 0; k < CH; k++) o[k] = 0;
    }
  }
  __syncthreads();
}
template <class T, class S, int CH, int PIN, int POUT, int HIN>
__device__ __forceinline__ void down_stage(const T* in, const Rect& ri, int nr, int nc, T* hb, T* out, const Rect& ro) {
  // interior: every tap 2c-2 .. 2c+2 / 2r-2 .. 2r+2 of every wanted output lies inside the level
  const bool interior = 2 * ro.r0 - 2 >= 0 && 2 * (ro.r0 + ro.h - 1) + 2 < nr && 2 * ro.c0 - 2 >= 0 && 2 * (ro.c0 + ro.w - 1) + 2 < nc;
  if (interior) down_stage_impl<T, S, CH, PIN, POUT, HIN, false>(in, ri, nr, nc, hb, out, ro);
  else down_stage_impl<T, S, CH, PIN, POUT, HIN, true>(in, ri, nr, nc, hb, out, ro);
}

// rows / columns of level l that the stage above needs from it: [max(0, 2a - 2), min(n - 1, 2b + 2)] for wanted [a, b] of level l + 1
__device__ __forceinline__ void needed(int a, int b, int n, int& lo, int& len) { lo = max(0, 2 * a - 2); len = min(n - 1, 2 * b + 2) - lo + 1; }

// T0 x T0 = tile of the coarsest level per workgroup.  LDS: the level rects + one horizontal-pass scratch.
template <class T, class S, int CH, int T0, class SRC, int NL, bool WRITE0>
__device__ __forceinline__ void chain_tile(const Chain& ch, int ty, int tx) {
  static_assert(NL == 2 || NL == 3, "two or three levels per launch");
  // rect extents (rows) and LDS row pitches (pixels): level NL-2 holds 2 T0 + 3, level 0 of a 3-level chain 2 (2 T0 + 3) + 3; the level-0
  // pitch leaves room for the dword alignment of its columns and is a dword multiple for every pixel size
  constexpr int E1 = 2 * T0 + 3, H0 = NL == 3 ? 2 * E1 + 3 : E1, P0 = (H0 + 3 + 3) & ~3, P1 = NL == 3 ? E1 : T0;
  __shared__ __attribute__((aligned(16))) T s_top[H0 * P0 * CH];                             // level 0 rect
  __shared__ __attribute__((aligned(16))) T s_mid[(NL == 3 ? E1 * E1 : T0 * T0) * CH];      // level 1 rect (NL == 3) or the level-1 tile (NL == 2)
  __shared__ __attribute__((aligned(16))) T s_low[(NL == 3 ? T0 * T0 : 1) * CH];            // level 2 tile
  __shared__ __attribute__((aligned(16))) T s_h[H0 * P1 * CH];                               // horizontal pass of the larger stage
  __shared__ __attribute__((aligned(16))) uint8_t s_aux[SRC::kAuxBytes];

  const DImg& last = ch.lv[NL - 1];
  Rect rl;  // the tile of the coarsest level
  rl.r0 = ty * T0; rl.c0 = tx * T0; rl.h = min(T0, last.nr - rl.r0); rl.w = min(T0, last.nc - rl.c0);
  Rect rm = rl, rt;  // NL == 3: level-1 rect; rt = level-0 rect
  if (NL == 3) { needed(rl.r0, rl.r0 + rl.h - 1, ch.lv[1].nr, rm.r0, rm.h); needed(rl.c0, rl.c0 + rl.w - 1, ch.lv[1].nc, rm.c0, rm.w); }
  needed(rm.r0, rm.r0 + rm.h - 1, ch.lv[0].nr, rt.r0, rt.h); needed(rm.c0, rm.c0 + rm.w - 1, ch.lv[0].nc, rt.c0, rt.w);
  { const int end = (rt.c0 + rt.w + 3) & ~3; rt.c0 &= ~3; rt.w = end - rt.c0; }   // dword-aligned columns; cells past the right edge are loaded but never read

  // ---- level 0 into LDS; the owned part goes to lv[0]
  SRC::template fill<P0>(ch.src, rt, s_top, s_aux);
  __syncthreads();
  if (WRITE0) {   // (false: level 0 already exists — the source IS level 0, only the coarser levels are produced)
    const int own_r0 = rl.r0 << (NL - 1), own_c0 = rl.c0 << (NL - 1);
    store_region<T, CH, P0, (T0 << (NL - 1))>(ch.lv[0], s_top, rt, own_r0, min(own_r0 + (T0 << (NL - 1)), ch.lv[0].nr), own_c0, min(own_c0 + (T0 << (NL - 1)), ch.lv[0].nc));
  }
  if (NL == 3) {
    down_stage<T, S, CH, P0, E1, H0>(s_top, rt, ch.lv[0].nr, ch.lv[0].nc, s_h, s_mid, rm);
    store_region<T, CH, E1, 2 * T0>(ch.lv[1], s_mid, rm, rl.r0 * 2, min(rl.r0 * 2 + 2 * T0, ch.lv[1].nr), rl.c0 * 2, min(rl.c0 * 2 + 2 * T0, ch.lv[1].nc));
    down_stage<T, S, CH, E1, T0, E1>(s_mid, rm, ch.lv[1].nr, ch.lv[1].nc, s_h, s_low, rl);
    store_region<T, CH, T0, T0>(ch.lv[2], s_low, rl, rl.r0, rl.r0 + rl.h, rl.c0, rl.c0 + rl.w);
  } else {
    down_stage<T, S, CH, P0, T0, H0>(s_top, rt, ch.lv[0].nr, ch.lv[0].nc, s_h, s_mid, rl);
    store_region<T, CH, T0, T0>(ch.lv[1], s_mid, rl, rl.r0, rl.r0 + rl.h, rl.c0, rl.c0 + rl.w);
  }
}
template <class T, class S, int CH, int T0, class SRC, int NL, bool WRITE0>
__global__ __launch_bounds__(256) void pyramid_chain_kernel(Chain ch) {
  const int tiles_x = (ch.lv[NL - 1].nc + T0 - 1) / T0;
  const int ty = blockIdx.x / tiles_x;
  chain_tile<T, S, CH, T0, SRC, NL, WRITE0>(ch, ty, blockIdx.x - ty * tiles_x);
}

// ---- three u8 x1 levels with packed arithmetic (round 3) ----------------------------------------------------------------------------
// The tile kernel above spends ~64 lane-operations per level-0 pixel (a byte per lane and tap, index arithmetic and edge tests per item) and
// is issue / latency bound: 25 us for a 4K pyramid whose bytes take 4 us.  Away from the frame's edges none of the edge handling is needed,
// and the 5-tap passes vectorise over bytes: a lane produces FOUR outputs per item from dwords — the horizontal pass of a decimating level
// takes 16 consecutive bytes, splits them into their even and odd bytes as packed 16-bit pairs (v_perm), lines the taps up with v_alignbit
// and evaluates a + 4 b + 6 c + 4 d + e on two pixels per 32-bit operation (sums < 4096: no carry between the halves); the vertical pass
// does the same on the five rows' dwords.  `>> 4` and the byte pack are one shift and one v_perm.  Same integer arithmetic, same truncated
// intermediate (the horizontal pass' u8), so the levels are bit-identical to the tile kernel's.
// A workgroup owns an 8 x 16 tile of level 2 (16 x 32 of level 1, 32 x 64 of level 0) and holds a 41 x 88 byte patch of level 0.  Tiles whose
// patch, or whose owned pixels' mirror copies, touch a frame edge (of any level) take the tile kernel's code path, as two workgroups of one 8 x 8
// tile each; they are numbered first so that the slower workgroups start first.
__device__ __forceinline__ uint32_t even_u16(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c020c00u); }   // bytes 0, 2 as two u16 fields
__device__ __forceinline__ uint32_t odd_u16(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c01u); }    // bytes 1, 3
// bytes x[0 .. 15] = w0 .. w3: four outputs k = 0 .. 3, output k centred on x[2 k + 4]
__device__ __forceinline__ uint32_t hpass4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  const uint32_t e0 = even_u16(w0), e1 = even_u16(w1), e2 = even_u16(w2), e3 = even_u16(w3);   // e_i = (x[4i], x[4i + 2])
  const uint32_t o0 = odd_u16(w0), o1 = odd_u16(w1), o2 = odd_u16(w2);                           // o_i = (x[4i + 1], x[4i + 3])
  const uint32_t a01 = __builtin_amdgcn_alignbit(e1, e0, 16), b01 = __builtin_amdgcn_alignbit(o1, o0, 16);   // (x2, x4), (x3, x5)
  const uint32_t f01 = __builtin_amdgcn_alignbit(e2, e1, 16), b23 = __builtin_amdgcn_alignbit(o2, o1, 16);   // (x6, x8), (x7, x9)
  const uint32_t f23 = __builtin_amdgcn_alignbit(e3, e2, 16);                                                // (x10, x12)
  const uint32_t s01 = __umul24(e1, 6u) + (a01 + f01) + ((b01 + o1) << 2);   // outputs 0, 1: x2 + 4 x3 + 6 x4 + 4 x5 + x6 | x4 + ... + x8
  const uint32_t s23 = __umul24(e2, 6u) + (f01 + f23) + ((b23 + o2) << 2);   // outputs 2, 3
  return __builtin_amdgcn_perm(s23 >> 4, s01 >> 4, 0x06040200u);
}
// the same tap over five rows, four columns packed in each dword
__device__ __forceinline__ uint32_t vpass4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t r4) {
  const uint32_t se = __umul24(even_u16(r2), 6u) + (even_u16(r0) + even_u16(r4)) + ((even_u16(r1) + even_u16(r3)) << 2);
  const uint32_t so = __umul24(odd_u16(r2), 6u) + (odd_u16(r0) + odd_u16(r4)) + ((odd_u16(r1) + odd_u16(r3)) << 2);
  return __builtin_amdgcn_perm(so >> 4, se >> 4, 0x06020400u);
}
// level-0 producers of the packed path: the dword of four level-0 pixels at (r, c), c a multiple of 4
struct CopyFast {
  static __device__ __forceinline__ uint32_t load4(const DImg& src, int r, int c) { return *(const uint32_t*)(src.p0 + (ptrdiff_t)r * src.pitch + c); }
};
template <int CH> struct GrayFast {   // rgb_to_graylevel of four pixels (GraySrc's arithmetic)
  static __device__ __forceinline__ uint32_t load4(const DImg& src, int r, int c) {
    const uint32_t* q = (const uint32_t*)(src.p0 + (ptrdiff_t)r * src.pitch + (ptrdiff_t)c * CH);
    auto gray = [](uint32_t a, uint32_t b, uint32_t cc) { return ((a + b + cc) * 43691u) >> 17; };
    if constexpr (CH == 3) {
      const uint32_t a = q[0], b = q[1], cc = q[2];
      return gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u) | gray(a >> 24, b & 255u, (b >> 8) & 255u) << 8 |
             gray((b >> 16) & 255u, b >> 24, cc & 255u) << 16 | gray((cc >> 8) & 255u, (cc >> 16) & 255u, cc >> 24) << 24;
    } else {
      uint32_t o = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint32_t a = q[k]; o |= gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u) << (8 * k); }
      return o;
    }
  }
};
struct Swar3 { Chain ch; int TX, TY, tx_lo, tx_hi, ty_lo, ty_hi, tiles_x8, wide;   // wide: source and level 0 are 16-byte aligned (the patch goes in as 16-byte pieces)
               int row_tiles, tx8_lo, tx8_hi, wt, xcd, per_wg; };   // row_tiles (2 / 4 / 8, else 0): the interior as ROW tiles (pyramid_swarw_body), columns [tx8_lo, tx8_hi) of 8 x 8 tiles, wt tiles per row

// WIDE (round 6, copy producer, 16-byte aligned source and level 0): the patch is 41 rows of 96 bytes from column 64 tx - 16 — six 16-byte pieces per row, 246 loads, ONE
// pass of the workgroup — instead of 41 x 22 dwords from column 64 tx - 12 in 3.5 passes with a division, two tests and a 4-byte store per dword: the staging was a third of
// the kernel's instructions.  The four bytes more on either side are read and never used.
constexpr int kSwarLds = 41 * 72 + 41 * 34 + 19 * 34 + 19 * 16 + 12;   // dwords: the packed path's stage buffers (one allocation for all the variants; the largest: the row tiles of 8 x 64)
template <class FAST, class SRC, bool WIDE>
__device__ __forceinline__ void pyramid_swar3_body_(const Swar3& a, const int bid, uint32_t* __restrict__ lds) {
  constexpr int R0 = 41, W0 = WIDE ? 24 : 22, XO = WIDE ? 1 : 0, G1 = 10, R1 = 19, G2 = 4;   // level-0 patch rows / dwords per row (XO: dword of the patch's column 64 tx - 12), level-1 groups per row / patch rows, level-2 groups
  int ty, tx;
  {
    // blocks [0, 2 n_edge): the edge tiles, one 8 x 8 tile of the tile kernel each (two per 8 x 16 tile, so that no workgroup runs two in a row —
    // that chain was the whole kernel's critical path at 1080p); then the interior tiles
    const int wi = a.tx_hi - a.tx_lo, n_top = a.ty_lo * a.TX, n_mid = (a.ty_hi - a.ty_lo) * (a.TX - wi), n_bot = (a.TY - a.ty_hi) * a.TX;
    const int n_edge = n_top + n_mid + n_bot;
    const bool interior = bid >= 2 * n_edge;
    int q = interior ? bid - 2 * n_edge : bid >> 1;
    if (interior) { const int row = q / wi; ty = a.ty_lo + row; tx = a.tx_lo + (q - row * wi); }
    else if (q < n_top) { ty = q / a.TX; tx = q - ty * a.TX; }
    else if (q < n_top + n_mid) { q -= n_top; const int row = q / (a.TX - wi), k = q - row * (a.TX - wi); ty = a.ty_lo + row; tx = k < a.tx_lo ? k : k + wi; }
    else { q -= n_top + n_mid; const int row = q / a.TX; ty = a.ty_hi + row; tx = q - row * a.TX; }
    if (!interior) {   // a tile that touches an edge: the tile kernel's path
      const int tx8 = 2 * tx + (bid & 1);
      if (tx8 < a.tiles_x8) chain_tile<uint8_t, int, 1, 8, SRC, 3, true>(a.ch, ty, tx8);
      return;
    }
  }
  constexpr int O1 = (R0 * W0 + 3) & ~3, O2 = (O1 + R0 * G1 + 3) & ~3, O3 = (O2 + R1 * G1 + 3) & ~3;   // the four stage buffers, 16-byte aligned, in the caller's one allocation
  static_assert(O3 + R1 * G2 <= kSwarLds, "the stages' LDS");
  uint32_t* const s0 = lds; uint32_t* const sh0 = lds + O1; uint32_t* const s1 = lds + O2; uint32_t* const sh1 = lds + O3;
  const DImg L0 = a.ch.lv[0], L1 = a.ch.lv[1], L2 = a.ch.lv[2], src = a.ch.src;
  const int r2 = 8 * ty, c2 = 16 * tx;
  const int pr0 = 4 * r2 - 6, pc0 = 4 * c2 - 12;   // origin of the level-0 patch; level-1 patch: rows from 2 r2 - 2, columns from 2 c2 - 4
  if constexpr (WIDE) {
    for (int idx = threadIdx.x; idx < R0 * 6; idx += 256) {
      const int y = idx / 6, x = idx - y * 6;
      const ptrdiff_t col = pc0 - 4 + 16 * x;
      const uint4 v = *(const uint4*)(src.p0 + (ptrdiff_t)(pr0 + y) * src.pitch + col);
      *(uint4*)&s0[y * W0 + 4 * x] = v;
      if (y >= 6 && y < 38 && x >= 1 && x < 5) *(uint4*)(L0.p0 + (ptrdiff_t)(pr0 + y) * L0.pitch + col) = v;   // the owned 32 x 64 pixels
    }
  } else
  for (int idx = threadIdx.x; idx < R0 * W0; idx += 256) {
    const int y = idx / W0, x = idx - y * W0;
    const uint32_t v = FAST::load4(src, pr0 + y, pc0 + 4 * x);
    s0[idx] = v;
    if (y >= 6 && y < 38 && x >= 3 && x < 19) *(uint32_t*)(L0.p0 + (ptrdiff_t)(pr0 + y) * L0.pitch + (pc0 + 4 * x)) = v;   // the owned 32 x 64 pixels
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < R0 * G1; idx += 256) {   // horizontal pass of level 0: level-1 columns 2 c2 - 4 + 4 g + k from patch bytes 8 g .. 8 g + 15
    const int y = idx / G1, g = idx - y * G1;
    const uint32_t* q = &s0[y * W0 + XO + 2 * g];
    sh0[idx] = hpass4(q[0], q[1], q[2], q[3]);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < R1 * G1; idx += 256) {   // vertical pass: level-1 row 2 r2 - 2 + i from patch rows 2 i .. 2 i + 4
    const int i = idx / G1, g = idx - i * G1;
    const uint32_t* p = &sh0[2 * i * G1 + g];
    const uint32_t v = vpass4(p[0], p[G1], p[2 * G1], p[3 * G1], p[4 * G1]);
    s1[idx] = v;
    if (i >= 2 && i < 18 && g >= 1 && g < 9) *(uint32_t*)(L1.p0 + (ptrdiff_t)(2 * r2 - 2 + i) * L1.pitch + (2 * c2 - 4 + 4 * g)) = v;   // the owned 16 x 32
  }
  __syncthreads();
  if (threadIdx.x < R1 * G2) {   // horizontal pass of level 1: level-2 columns c2 + 4 g + k
    const int i = threadIdx.x >> 2, g = threadIdx.x & 3;
    const uint2 lo = *(const uint2*)&s1[i * G1 + 2 * g], hi = *(const uint2*)&s1[i * G1 + 2 * g + 2];
    sh1[threadIdx.x] = hpass4(lo.x, lo.y, hi.x, hi.y);
  }
  __syncthreads();
  if (threadIdx.x < 8 * G2) {    // vertical pass: the 8 x 16 tile of level 2
    const int i = threadIdx.x >> 2, g = threadIdx.x & 3;
    const uint32_t* p = &sh1[2 * i * G2 + g];
    *(uint32_t*)(L2.p0 + (ptrdiff_t)(r2 + i) * L2.pitch + (c2 + 4 * g)) = vpass4(p[0], p[G2], p[2 * G2], p[3 * G2], p[4 * G2]);
  }
}

// ROW tiles (round 6, copy producer, 16-byte aligned source and level 0): the same passes on a tile of TH2 x 64 pixels of level 2 (4 TH2 x 256 of level 0; TH2 = 8 by
// default, 2 / 4 by tuning) instead of 8 x 16.  Measured with the arithmetic taken out (stage + level-0 store only, interior tiles only, 4K): 32 x 64-pixel tiles 9.1 us,
// 16 x 128 8.2 us, 8 x 256 6.7 us — about a copy's time — although the flat tile loads 2.4 x its pixels instead of 1.9 x: a 64-byte row of a tile is half a cache line,
// written by one workgroup and completed by another one later, and its 96-byte patch rows straddle two lines each; 256-byte rows are whole lines.  With the arithmetic
// (4K, border 3, per pyramid): 8 x 16 tiles 12.3 us, 2 x 64 11.3, 4 x 64 10.0, 8 x 64 9.2 (1.28 x halo; 1 012 workgroups); the flow's pair + tail launch 25.6 -> 20.8 us.
// The frame's edge ring stays with the tile kernel's code on 8 x 8 tiles of level 2 (one workgroup each, numbered first); the interior is the rectangle of those tiles that
// the packed arithmetic may own, cut into row tiles from its left edge, the last tile of a row moved left to end at the rectangle's right edge (the overlap is computed
// twice, to the same bytes).  Counters (tools/pyr_pmc.sh): 2.7 M VALU wave-instructions per 4K pyramid against 3.4 M before, ~1 M of them in the 374 edge workgroups.
template <class FAST, class SRC, int TH2>
__device__ __forceinline__ void pyramid_swarw_body(const Swar3& a, const int bid, uint32_t* __restrict__ lds) {
  constexpr int TW2 = 64, R0 = 4 * TH2 + 9, NP = (4 * TW2 + 32) / 16, W0 = 4 * NP, G1 = (2 * TW2 + 8) / 4, R1 = 2 * TH2 + 3, G2 = TW2 / 4;
  const int TX8 = a.tiles_x8, wi = a.tx8_hi - a.tx8_lo;
  const int n_top = a.ty_lo * TX8, n_mid = (a.ty_hi - a.ty_lo) * (TX8 - wi), n_bot = (a.TY - a.ty_hi) * TX8, n_edge = n_top + n_mid + n_bot;
  if (bid < n_edge) {   // a tile of the edge ring: the tile kernel's path
    int q = bid, ty, tx;
    if (q < n_top) { ty = q / TX8; tx = q - ty * TX8; }
    else if (q < n_top + n_mid) { q -= n_top; const int row = q / (TX8 - wi), k = q - row * (TX8 - wi); ty = a.ty_lo + row; tx = k < a.tx8_lo ? k : k + wi; }
    else { q -= n_top + n_mid; const int row = q / TX8; ty = a.ty_hi + row; tx = q - row * TX8; }
    chain_tile<uint8_t, int, 1, 8, SRC, 3, true>(a.ch, ty, tx);
    return;
  }
  // A workgroup takes `per_wg` consecutive tiles (down a column of tiles where the order is XCD-aware; 1 by default) and requests the next tile's patch — three 16-byte
  // pieces per thread at most, in registers — before it starts on the passes of the current one.  Measured no faster with 2 or 3 tiles per workgroup (4K: 9.4 -> 9.8 us
  // with 8-row tiles, 10.3 -> 10.3 / 11.3 with 4-row tiles): the kernel is bound by what it issues, not by the patch's round trip.
  const int nrt = (a.ty_hi - a.ty_lo) * (8 / TH2), ntiles = nrt * a.wt;   // tile rows of the interior, tiles
  int w = bid - n_edge;
  if (a.xcd) {   // consecutive workgroups go to the 8 XCDs in turn: an XCD takes a contiguous run of tiles, ordered DOWN the columns (a tile's patch shares 9 of its rows with the tile below)
    const int nw = (ntiles + a.per_wg - 1) / a.per_wg, n8 = nw & ~7;
    if (w < n8) w = (w & 7) * (n8 >> 3) + (w >> 3);
  }
  const int t0 = w * a.per_wg, t1 = min(t0 + a.per_wg, ntiles);
  constexpr int O1 = R0 * W0, O2 = (O1 + R0 * G1 + 3) & ~3, O3 = (O2 + R1 * G1 + 3) & ~3;
  static_assert(O1 % 4 == 0 && O3 + R1 * G2 <= kSwarLds && G1 % 2 == 0 && G1 >= 2 * G2 + 2, "the stages' LDS");
  uint32_t* const s0 = lds; uint32_t* const sh0 = lds + O1; uint32_t* const s1 = lds + O2; uint32_t* const sh1 = lds + O3;
  const DImg L0 = a.ch.lv[0], L1 = a.ch.lv[1], L2 = a.ch.lv[2], src = a.ch.src;
  constexpr int NLD = (R0 * NP + 255) / 256;   // pieces per thread
  // this thread's pieces: patch row / piece of each, as byte offsets into the source / level 0 (same for every tile) and the LDS dword
  int poff[NLD], loff[NLD], ldw[NLD]; bool own[NLD];
#pragma unroll
  for (int k = 0; k < NLD; k++) {
    const int idx = min((int)threadIdx.x + 256 * k, R0 * NP - 1), y = idx / NP, x = idx - y * NP;
    poff[k] = std::is_same<FAST, CopyFast>::value ? y * src.pitch + 16 * x : (y << 16 | 16 * x);   // (a producer other than the copy gets the piece's patch row and column)
    loff[k] = y * L0.pitch + 16 * x; ldw[k] = y * W0 + 4 * x;
    own[k] = (int)threadIdx.x + 256 * k < R0 * NP && y >= 6 && y < 6 + 4 * TH2 && x >= 1 && x < NP - 1;
  }
  auto origin = [&](int t, int& r2, int& c2) {
    int trow, tcol;
    if (a.xcd) { tcol = t / nrt; trow = t - tcol * nrt; } else { trow = t / a.wt; tcol = t - trow * a.wt; }
    r2 = 8 * a.ty_lo + TH2 * trow; c2 = min(8 * a.tx8_lo + TW2 * tcol, 8 * a.tx8_hi - TW2);
  };
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (an array of HIP's uint4 stays in scratch memory)
  u32x4 v[NLD];
  // the patch of the tile at (r2, c2) of level 2: origin (4 r2 - 6, 4 c2 - 16) of level 0, R0 rows of W0 dwords; level-1 patch: rows from 2 r2 - 2, columns from 2 c2 - 4
  auto request = [&](int r2, int c2) {
    if constexpr (std::is_same<FAST, CopyFast>::value) {
      const uint8_t* p = src.p0 + (ptrdiff_t)(4 * r2 - 6) * src.pitch + (4 * c2 - 16);
#pragma unroll
      for (int k = 0; k < NLD; k++) v[k] = *(const u32x4*)(p + poff[k]);
    } else {   // (the ingest as producer: four pixels per dword from 12 / 16 source bytes)
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        const int r = 4 * r2 - 6 + (poff[k] >> 16), c = 4 * c2 - 16 + (poff[k] & 0xFFFF);
        v[k] = u32x4{FAST::load4(src, r, c), FAST::load4(src, r, c + 4), FAST::load4(src, r, c + 8), FAST::load4(src, r, c + 12)};
      }
    }
  };
  int r2, c2;
  origin(t0, r2, c2);
  request(r2, c2);
  for (int t = t0; t < t1; t++) {
    {
      uint8_t* o = L0.p0 + (ptrdiff_t)(4 * r2 - 6) * L0.pitch + (4 * c2 - 16);
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        if (k < NLD - 1 || (int)threadIdx.x + 256 * k < R0 * NP) *(u32x4*)&s0[ldw[k]] = v[k];
        if (own[k]) *(u32x4*)(o + loff[k]) = v[k];   // the owned 4 TH2 x 256 pixels
      }
    }
    __syncthreads();
    const int cr2 = r2, cc2 = c2;
    if (t + 1 < t1) {   // the next tile's patch: in flight during this tile's passes
      origin(t + 1, r2, c2);
      request(r2, c2);
    }
    for (int idx = threadIdx.x; idx < R0 * G1; idx += 256) {   // horizontal pass of level 0: level-1 columns 2 c2 - 4 + 4 g + k from patch bytes 4 + 8 g .. 4 + 8 g + 15
      const int y = idx / G1, g = idx - y * G1;
      const uint32_t* q4 = &s0[y * W0 + 1 + 2 * g];
      sh0[idx] = hpass4(q4[0], q4[1], q4[2], q4[3]);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < R1 * G1; idx += 256) {   // vertical pass: level-1 row 2 r2 - 2 + i from patch rows 2 i .. 2 i + 4
      const int i = idx / G1, g = idx - i * G1;
      const uint32_t* p = &sh0[2 * i * G1 + g];
      const uint32_t x = vpass4(p[0], p[G1], p[2 * G1], p[3 * G1], p[4 * G1]);
      s1[idx] = x;
      if (i >= 2 && i < 2 + 2 * TH2 && g >= 1 && g < G1 - 1) *(uint32_t*)(L1.p0 + (ptrdiff_t)(2 * cr2 - 2 + i) * L1.pitch + (2 * cc2 - 4 + 4 * g)) = x;   // the owned 2 TH2 x 128
    }
    __syncthreads();
    for (int u = threadIdx.x; u < R1 * G2; u += 256) {   // horizontal pass of level 1: level-2 columns c2 + 4 g + k
      const int i = u / G2, g = u - i * G2;
      const uint2 lo = *(const uint2*)&s1[i * G1 + 2 * g], hi = *(const uint2*)&s1[i * G1 + 2 * g + 2];
      sh1[u] = hpass4(lo.x, lo.y, hi.x, hi.y);
    }
    __syncthreads();
    static_assert(TH2 * G2 <= 256, "one pass");
    if (threadIdx.x < TH2 * G2) {    // vertical pass: the TH2 x 64 tile of level 2
      const int i = threadIdx.x / G2, g = threadIdx.x - i * G2;
      const uint32_t* p = &sh1[2 * i * G2 + g];
      *(uint32_t*)(L2.p0 + (ptrdiff_t)(cr2 + i) * L2.pitch + (cc2 + 4 * g)) = vpass4(p[0], p[G2], p[2 * G2], p[3 * G2], p[4 * G2]);
    }
    // (no barrier at the loop's end: the next tile's writes to a stage buffer come at least one barrier after this tile's last reads of it)
  }
}

template <class FAST, class SRC>
__device__ __forceinline__ void pyramid_swar3_body(const Swar3& a, const int bid) {
  __shared__ __attribute__((aligned(16))) uint32_t s_swar[kSwarLds];
  if (a.row_tiles == 8) { pyramid_swarw_body<FAST, SRC, 8>(a, bid, s_swar); return; }
  if constexpr (std::is_same<FAST, CopyFast>::value) {
    if (a.row_tiles == 4) { pyramid_swarw_body<FAST, SRC, 4>(a, bid, s_swar); return; }
    if (a.row_tiles == 2) { pyramid_swarw_body<FAST, SRC, 2>(a, bid, s_swar); return; }
    if (a.wide) { pyramid_swar3_body_<FAST, SRC, true>(a, bid, s_swar); return; }
  }
  pyramid_swar3_body_<FAST, SRC, false>(a, bid, s_swar);
}

// interior tile ranges of the packed path along one axis: tiles [lo, hi) of `step2` level-2 pixels whose level-0 patch (from 4 t0 - lead0, `span0`
// pixels) lies inside the level and whose owned pixels are at least a border away from both ends on every level
void swar3_axis(int n0, int n1, int n2, int b0, int b1, int b2, int step2, int lead0, int span0, int ntiles, int* lo, int* hi) {
  auto ok = [&](int t) {
    const int p2 = t * step2, p1 = 2 * p2, p0 = 4 * p2;
    return p0 - lead0 >= 0 && p0 - lead0 + span0 <= n0 && p0 >= b0 && p0 + 4 * step2 <= n0 - b0 && p1 >= b1 && p1 + 2 * step2 <= n1 - b1 && p2 >= b2 &&
           p2 + step2 <= n2 - b2;
  };
  int l = 0;
  while (l < ntiles && !ok(l)) l++;
  int h = l;
  while (h < ntiles && ok(h)) h++;
  *lo = l; *hi = h;
}
template <class FAST, class SRC>
__global__ __launch_bounds__(256) void pyramid_swar3_kernel(Swar3 a) { pyramid_swar3_body<FAST, SRC>(a, (int)blockIdx.x); }
// two pyramids of one geometry in one launch (the two frames of a flow call): the second pyramid's tiles follow the first's
template <class FAST, class SRC>
__global__ __launch_bounds__(256) void pyramid_swar3_pair_kernel(Swar3 a, Swar3 b, int blocks_a) {
  if ((int)blockIdx.x < blocks_a) pyramid_swar3_body<FAST, SRC>(a, (int)blockIdx.x);
  else pyramid_swar3_body<FAST, SRC>(b, (int)blockIdx.x - blocks_a);
}

// the pair + `tail_blocks` blocks of the flow's map reset and claims (sdof_tail.hpp): independent work in one launch.  The tail's blocks — short, memory-only workgroups —
// come LAST since the pyramid tiles are row tiles (they fill the launch's drain: 21.5 -> 20.4 us, the 4K pair -1.5 us; bit 31 of `tail_blocks`, tuning sdof.tail_last);
// with the 8 x 16 tiles of before (VALU-bound, ~25 us at 4K) they came first and were gone by the time the tiles had ramped up
template <class FAST, class SRC>
__global__ __launch_bounds__(256) void pyramid_swar3_pair_tail_kernel(Swar3 a, Swar3 b, int blocks_a, unsigned tail_blocks, ResetClaimTail t) {
  int bid;
  if (tail_blocks & 0x80000000u) {   // (tuning sdof.tail_last) the tail's blocks behind the pyramids' tiles
    const unsigned npyr = gridDim.x - (tail_blocks & 0x7FFFFFFFu);
    if (blockIdx.x >= npyr) { reset_claim_block(t.a, t.kps, t.n, t.patch, t.c, blockIdx.x - npyr); return; }
    bid = (int)blockIdx.x;
  } else {
    if (blockIdx.x < tail_blocks) { reset_claim_block(t.a, t.kps, t.n, t.patch, t.c, blockIdx.x); return; }
    bid = (int)(blockIdx.x - tail_blocks);
  }
  if (bid < blocks_a) pyramid_swar3_body<FAST, SRC>(a, bid);
  else pyramid_swar3_body<FAST, SRC>(b, bid - blocks_a);
}

// the packed kernel's arguments for one pyramid; false: unaligned levels or no interior tile (the tile kernel takes the pyramid); *blocks = its grid
inline bool swar3_args(const vpp_image_desc* levels, const vpp_image_desc* src, Swar3* out, int* blocks, bool copy_producer = true) {
  auto al4 = [](const vpp_image_desc& d) { return ((uintptr_t)d.first_pixel & 3) == 0 && (d.pitch & 3) == 0; };
  if (!(al4(levels[0]) && al4(levels[1]) && al4(levels[2]) && al4(*src))) return false;
  Swar3& a = *out;
  for (int l = 0; l < 3; l++) a.ch.lv[l] = dimg(&levels[l]);
  a.ch.src = dimg(src); a.ch.nlevels = 3;
  a.TY = (levels[2].nrows + 7) / 8; a.TX = (levels[2].ncols + 15) / 16; a.tiles_x8 = (levels[2].ncols + 7) / 8;
  swar3_axis(levels[0].nrows, levels[1].nrows, levels[2].nrows, levels[0].border, levels[1].border, levels[2].border, 8, 6, 41, a.TY, &a.ty_lo, &a.ty_hi);
  auto al16 = [](const vpp_image_desc& d) { return ((uintptr_t)d.first_pixel & 15) == 0 && (d.pitch & 15) == 0; };
  a.wide = copy_producer && tuning("pyr.wide", 1) && al16(levels[0]) && al16(*src) ? 1 : 0;
  swar3_axis(levels[0].ncols, levels[1].ncols, levels[2].ncols, levels[0].border, levels[1].border, levels[2].border, 16, a.wide ? 16 : 12, a.wide ? 96 : 88, a.TX, &a.tx_lo, &a.tx_hi);
  if (a.ty_hi <= a.ty_lo || a.tx_hi <= a.tx_lo) return false;   // no interior tile: the tile kernel
  const int n_int = (a.ty_hi - a.ty_lo) * (a.tx_hi - a.tx_lo), n_edge = a.TY * a.TX - n_int;
  *blocks = 2 * n_edge + n_int;
  a.row_tiles = 0; a.tx8_lo = a.tx8_hi = a.wt = 0; a.per_wg = 1; a.xcd = (int)tuning("pyr.xcd", 1);
  const int rt = (int)tuning("pyr.row_tiles", 8);
  // (the ingest as producer: level 0 16-byte aligned is all the row tiles need — its source is read through the producer's own dword loads; 8-row tiles only)
  const bool gray_rows = !copy_producer && rt == 8 && tuning("pyr.gray_row_tiles", 1) && al16(levels[0]);
  if ((a.wide && (rt == 2 || rt == 4 || rt == 8)) || gray_rows) {   // the interior as row tiles of rt x 64 pixels of level 2 (pyramid_swarw_body): its columns in units of the edge ring's 8 x 8 tiles
    swar3_axis(levels[0].ncols, levels[1].ncols, levels[2].ncols, levels[0].border, levels[1].border, levels[2].border, 8, 16, 64, a.tiles_x8, &a.tx8_lo, &a.tx8_hi);
    if (a.tx8_hi - a.tx8_lo >= 8) {
      a.row_tiles = rt; a.wt = (a.tx8_hi - a.tx8_lo + 7) / 8;
      a.per_wg = std::max(1, std::min(8, (int)tuning("pyr.tiles_per_wg", 1)));
      *blocks = a.TY * a.tiles_x8 - (a.ty_hi - a.ty_lo) * (a.tx8_hi - a.tx8_lo) + ((a.ty_hi - a.ty_lo) * (8 / rt) * a.wt + a.per_wg - 1) / a.per_wg;
    }
  }
  return true;
}
template <class FAST, class SRC>
bool launch_swar3(const vpp_image_desc* levels, const vpp_image_desc* src, hipStream_t st) {
  Swar3 a; int blocks = 0;
  if (!swar3_args(levels, src, &a, &blocks, std::is_same<FAST, CopyFast>::value)) return false;
  pyramid_swar3_kernel<FAST, SRC><<<blocks, 256, 0, st>>>(a);
  return true;
}

bool chain_shape_ok(const vpp_image_desc* levels, int nlevels) {
  for (int l = 0; l < nlevels; l++) {
    if (levels[l].border > levels[l].nrows || levels[l].border > levels[l].ncols) return false;  // the fused border writes assume the mirror stays inside the level
    if (l > 0 && (levels[l].nrows != 1 + levels[l - 1].nrows / 2 || levels[l].ncols != 1 + levels[l - 1].ncols / 2)) return false;
  }
  return true;
}

template <class T, class S, int CH, int T0, class SRC, bool WRITE0 = true>
void launch_chain(const vpp_image_desc* levels, int nlevels, const vpp_image_desc* src, hipStream_t st) {
  Chain c;
  for (int l = 0; l < nlevels; l++) c.lv[l] = dimg(&levels[l]);
  c.src = dimg(src); c.nlevels = nlevels;
  const vpp_image_desc& last = levels[nlevels - 1];
  const int tiles = ((last.nrows + T0 - 1) / T0) * ((last.ncols + T0 - 1) / T0);
  // (128- and 64-thread workgroups measured: 4K 25.9 / 28.9 us against 26.1, 1080p 13.8 / 18.8 against 12.5)
  if (nlevels == 3) pyramid_chain_kernel<T, S, CH, T0, SRC, 3, WRITE0><<<tiles, 256, 0, st>>>(c);
  else pyramid_chain_kernel<T, S, CH, T0, SRC, 2, WRITE0><<<tiles, 256, 0, st>>>(c);
}

}  // namespace

namespace vpp_amd { int vpp_scharr_bordered(const vpp_image_desc* out, const vpp_image_desc* in, void* stream); }

extern "C" {

// pyramid2d<V>(src, nlevels, 2, _border = levels[0].border) (pyramid.hh:146-198): level 0 = copy of src's domain + mirror
// border, then propagate_level0.  u8 x1 pyramids of 2 or 3 levels go through the fused kernel, everything else through the
// per-level kernels (vpp_copy + vpp_fill_border + vpp_pyr_down).
int vpp_pyramid_build(const vpp_image_desc* levels, int nlevels, const vpp_image_desc* src, void* stream) {
  VPP_REQUIRE(levels && src && nlevels >= 1, VPP_ERR_INVALID_ARG, "vpp_pyramid_build: invalid argument");
  for (int l = 0; l < nlevels; l++) VPP_REQUIRE(valid_desc(&levels[l]) && same_type(&levels[l], src), VPP_ERR_INVALID_ARG, "vpp_pyramid_build: level %d: invalid descriptor / element type", l);
  VPP_REQUIRE(valid_desc(src) && same_domain(&levels[0], src), VPP_ERR_INVALID_ARG, "vpp_pyramid_build: level 0 and the source differ in size");
  hipStream_t st = as_stream(stream);
  if (src->dtype == VPP_U8 && src->channels == 1 && (nlevels == 2 || nlevels == 3) && chain_shape_ok(levels, nlevels) && tuning("pyr.fused", 1)) {
    // three levels: the packed kernel (its edge tiles run the tile kernel's code); two levels, unaligned images, frames without interior tiles: the tile kernel
    // (8 x 8 coarse tiles: 16 x 16 — a 64 x 64 level-0 tile per workgroup, 1.3x instead of 1.6x halo — measured 24.7 vs 26.1 us at 4K and 17.5 vs 12.8 us at 1080p)
    if (!(nlevels == 3 && tuning("pyr.swar", 1) && launch_swar3<CopyFast, CopySrc<uint8_t, 1>>(levels, src, st)))
      launch_chain<uint8_t, int, 1, 8, CopySrc<uint8_t, 1>>(levels, nlevels, src, st);
    VPP_LAUNCH_CHECK();
    return VPP_OK;
  }
  int rc = vpp_copy(&levels[0], src, 0, stream);
  if (rc != VPP_OK) return rc;
  rc = vpp_fill_border(&levels[0], VPP_BORDER_MIRROR, nullptr, stream);
  for (int l = 1; l < nlevels && rc == VPP_OK; l++) rc = vpp_pyr_down(&levels[l], &levels[l - 1], stream);
  return rc;
}

// Two pyramids of one geometry (the two frames of a flow call) in ONE launch when both take the packed kernel; otherwise two vpp_pyramid_build calls.
// Internal (sdof.hip): not part of include/vpp_amd.h.
int vpp_pyramid_build_pair(const vpp_image_desc* levels_a, const vpp_image_desc* src_a, const vpp_image_desc* levels_b, const vpp_image_desc* src_b, int nlevels, void* stream) {
  hipStream_t st = as_stream(stream);
  if (nlevels == 3 && levels_a && levels_b && src_a && src_b && tuning("pyr.fused", 1) && tuning("pyr.swar", 1) && tuning("pyr.pair", 1)) {
    bool ok = true;
    for (int l = 0; l < 3 && ok; l++) ok = valid_desc(&levels_a[l]) && valid_desc(&levels_b[l]) && same_type(&levels_a[l], src_a) && same_type(&levels_b[l], src_b);
    ok = ok && valid_desc(src_a) && valid_desc(src_b) && src_a->dtype == VPP_U8 && src_a->channels == 1 && src_b->dtype == VPP_U8 && src_b->channels == 1 &&
         same_domain(&levels_a[0], src_a) && same_domain(&levels_b[0], src_b) && chain_shape_ok(levels_a, 3) && chain_shape_ok(levels_b, 3);
    Swar3 a, b; int na = 0, nb = 0;
    if (ok && swar3_args(levels_a, src_a, &a, &na) && swar3_args(levels_b, src_b, &b, &nb)) {
      pyramid_swar3_pair_kernel<CopyFast, CopySrc<uint8_t, 1>><<<na + nb, 256, 0, st>>>(a, b, na);
      VPP_LAUNCH_CHECK();
      return VPP_OK;
    }
  }
  int rc = vpp_pyramid_build(levels_a, nlevels, src_a, stream);
  if (rc == VPP_OK) rc = vpp_pyramid_build(levels_b, nlevels, src_b, stream);
  return rc;
}

}  // extern "C"
namespace vpp_amd {
int pyramid_pair_with_tail(const vpp_image_desc* levels_a, const vpp_image_desc* src_a, const vpp_image_desc* levels_b, const vpp_image_desc* src_b, int nlevels,
                           const ResetClaimTail& tail, unsigned tail_blocks, hipStream_t st, bool* fused) {
  *fused = false;
  if (!(nlevels == 3 && levels_a && levels_b && src_a && src_b && tuning("pyr.fused", 1) && tuning("pyr.swar", 1) && tuning("pyr.pair", 1))) return VPP_OK;
  bool ok = true;
  for (int l = 0; l < 3 && ok; l++) ok = valid_desc(&levels_a[l]) && valid_desc(&levels_b[l]) && same_type(&levels_a[l], src_a) && same_type(&levels_b[l], src_b);
  ok = ok && valid_desc(src_a) && valid_desc(src_b) && src_a->dtype == VPP_U8 && src_a->channels == 1 && src_b->dtype == VPP_U8 && src_b->channels == 1 &&
       same_domain(&levels_a[0], src_a) && same_domain(&levels_b[0], src_b) && chain_shape_ok(levels_a, 3) && chain_shape_ok(levels_b, 3);
  Swar3 a, b; int na = 0, nb = 0;
  if (!(ok && swar3_args(levels_a, src_a, &a, &na) && swar3_args(levels_b, src_b, &b, &nb))) return VPP_OK;
  pyramid_swar3_pair_tail_kernel<CopyFast, CopySrc<uint8_t, 1>><<<tail_blocks + (unsigned)(na + nb), 256, 0, st>>>(a, b, na, tail_blocks | (tuning("sdof.tail_last", 1) ? 0x80000000u : 0u), tail);
  VPP_LAUNCH_CHECK();
  *fused = true;
  return VPP_OK;
}
// ONE pyramid (the new frame of a tracker that keeps the previous frame's; a gray frame, or an rgb / rgba frame through the ingest) with the tail's blocks in its launch:
// the pair kernel with an empty second pyramid
int pyramid_one_with_tail(const vpp_image_desc* levels, const vpp_image_desc* src, int nlevels, const ResetClaimTail& tail, unsigned tail_blocks, hipStream_t st, bool* fused) {
  *fused = false;
  if (!(nlevels == 3 && levels && src && tuning("pyr.fused", 1) && tuning("pyr.swar", 1) && tuning("pyr.pair", 1))) return VPP_OK;
  bool ok = valid_desc(src) && src->dtype == VPP_U8 && (src->channels == 1 || src->channels == 3 || src->channels == 4);
  for (int l = 0; l < 3 && ok; l++) ok = valid_desc(&levels[l]) && levels[l].dtype == VPP_U8 && levels[l].channels == 1;
  ok = ok && same_domain(&levels[0], src) && chain_shape_ok(levels, 3);
  Swar3 a; int na = 0;
  if (!(ok && swar3_args(levels, src, &a, &na, src->channels == 1))) return VPP_OK;
  const unsigned tb = tail_blocks | (tuning("sdof.tail_last", 1) ? 0x80000000u : 0u), grid = tail_blocks + (unsigned)na;
  if (src->channels == 1) pyramid_swar3_pair_tail_kernel<CopyFast, CopySrc<uint8_t, 1>><<<grid, 256, 0, st>>>(a, a, na, tb, tail);
  else if (src->channels == 3) pyramid_swar3_pair_tail_kernel<GrayFast<3>, GraySrc<3>><<<grid, 256, 0, st>>>(a, a, na, tb, tail);   // (the ingest as producer: vpp_rgb_pyramid_build's kernel)
  else pyramid_swar3_pair_tail_kernel<GrayFast<4>, GraySrc<4>><<<grid, 256, 0, st>>>(a, a, na, tb, tail);
  VPP_LAUNCH_CHECK();
  *fused = true;
  return VPP_OK;
}
}  // namespace vpp_amd
extern "C" {

// Frame ingest fused with the pyramid it feeds (examples/video_extruder.cc:46-48 + pyramid.hh:169-198): levels[0] = rgb_to_graylevel<uchar> of
// the u8 x3 / x4 frame `rgb` (its border is not read) with a mirror-filled border — exactly what vpp_rgb_to_graylevel(mirror = 1) into a gray
// frame followed by vpp_pyramid_build from that frame leaves in the levels — then the coarser levels; one launch, the gray frame is written
// once.  Pyramids of 2 or 3 levels take the fused kernel; other depths go through the two-call chain with levels[0] as the gray frame.
int vpp_rgb_pyramid_build(const vpp_image_desc* levels, int nlevels, const vpp_image_desc* rgb, void* stream) {
  VPP_REQUIRE(levels && rgb && nlevels >= 1 && valid_desc(rgb), VPP_ERR_INVALID_ARG, "vpp_rgb_pyramid_build: invalid argument");
  VPP_REQUIRE(rgb->dtype == VPP_U8 && (rgb->channels == 3 || rgb->channels == 4), VPP_ERR_UNSUPPORTED, "vpp_rgb_pyramid_build: the frame must be u8 x3 or x4");
  for (int l = 0; l < nlevels; l++)
    VPP_REQUIRE(valid_desc(&levels[l]) && levels[l].dtype == VPP_U8 && levels[l].channels == 1, VPP_ERR_INVALID_ARG, "vpp_rgb_pyramid_build: level %d: u8 x1 expected", l);
  VPP_REQUIRE(same_domain(&levels[0], rgb), VPP_ERR_INVALID_ARG, "vpp_rgb_pyramid_build: level 0 and the frame differ in size");
  hipStream_t st = as_stream(stream);
  if ((nlevels == 2 || nlevels == 3) && chain_shape_ok(levels, nlevels) && tuning("pyr.fused", 1)) {
    const bool swar = nlevels == 3 && tuning("pyr.swar", 1);
    if (rgb->channels == 3) { if (!(swar && launch_swar3<GrayFast<3>, GraySrc<3>>(levels, rgb, st))) launch_chain<uint8_t, int, 1, 8, GraySrc<3>>(levels, nlevels, rgb, st); }
    else if (!(swar && launch_swar3<GrayFast<4>, GraySrc<4>>(levels, rgb, st))) launch_chain<uint8_t, int, 1, 8, GraySrc<4>>(levels, nlevels, rgb, st);
    VPP_LAUNCH_CHECK();
    return VPP_OK;
  }
  int rc = vpp_rgb_to_graylevel(&levels[0], rgb, 1, stream);
  for (int l = 1; l < nlevels && rc == VPP_OK; l++) rc = vpp_pyr_down(&levels[l], &levels[l - 1], stream);
  return rc;
}

// The gradient pyramid of pyrlk_match / lucas_kanade: scharr(img, grad[0]); fill_border_mirror(grad[0]); propagate_level0
// (pyrlk_opencv_comparison.cc:56-60, lucas_kanade.hpp:151-157).  img: u8 x1 with a filled border >= 1; grad: f32 x2 or i32 x2.
int vpp_scharr_pyramid_build(const vpp_image_desc* grad, int nlevels, const vpp_image_desc* img, void* stream) {
  VPP_REQUIRE(grad && img && nlevels >= 1 && valid_desc(img), VPP_ERR_INVALID_ARG, "vpp_scharr_pyramid_build: invalid argument");
  VPP_REQUIRE(img->dtype == VPP_U8 && img->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_scharr_pyramid_build: the image must be u8 x1");
  VPP_REQUIRE(img->border >= 1, VPP_ERR_BORDER_TOO_SMALL, "vpp_scharr_pyramid_build: the image needs border >= 1 (scharr.hh:48)");
  for (int l = 0; l < nlevels; l++)
    VPP_REQUIRE(valid_desc(&grad[l]) && grad[l].channels == 2 && (grad[l].dtype == VPP_F32 || grad[l].dtype == VPP_I32) && grad[l].dtype == grad[0].dtype, VPP_ERR_UNSUPPORTED,
                "vpp_scharr_pyramid_build: gradient levels must be f32 x2 or i32 x2");
  VPP_REQUIRE(same_domain(&grad[0], img), VPP_ERR_INVALID_ARG, "vpp_scharr_pyramid_build: level 0 and the image differ in size");

  if (chain_shape_ok(grad, nlevels) && tuning("pyr.fused", 1)) {
    // measured (tools/time_pyramids.py, 1080p x 3 levels, f32 x2): scharr inside the tile kernel 34 us, wide scharr + tile kernel for
    // the coarser levels 33 us, the 4-launch chain 32 us — 8-byte pixels make the LDS tiles slower than the per-level gathers.
    // What pays is dropping the separate border launch: scharr writes the mirrored copies itself, then the per-level kernels
    // (which fuse their own borders).
    int rc = vpp_scharr_bordered(&grad[0], img, stream);
    for (int l = 1; l < nlevels && rc == VPP_OK; l++) rc = vpp_pyr_down(&grad[l], &grad[l - 1], stream);
    return rc;
  }
  int rc = vpp_scharr(&grad[0], img, stream);
  if (rc != VPP_OK) return rc;
  rc = vpp_fill_border(&grad[0], VPP_BORDER_MIRROR, nullptr, stream);
  for (int l = 1; l < nlevels && rc == VPP_OK; l++) rc = vpp_pyr_down(&grad[l], &grad[l - 1], stream);
  return rc;
}

}  // extern "C"
