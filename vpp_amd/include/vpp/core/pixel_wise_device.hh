// pixel_wise_device.hh — GPU evaluation of pixel_wise(ranges...) | <opaque callable> when the USER's translation unit is
// compiled by hipcc (single-source mode: -DVPP_AMD_DEVICE, --offload-arch=gfx950).  Included by pixel_wise.hh.
// Reference semantics: vpp/core/pixel_wise.hpp:68-105 (process_row), :146-165 (run), :188-213 (operator|); call sites such as
// benchmarks/image_add.cc:51-57 and benchmarks/box_5x5_filter2.cc:71-81 compile unchanged.
//
// hipcc makes a plain C++ lambda callable from device code, so the callable itself is the kernel body: one generic, hand-written
// __global__ function templated on the callable and on one accessor per range.
//  * image ranges: a lane owns NPX consecutive pixels of a row — as many as make every range's chunk a multiple of 16 bytes —
//    loads them with 16-byte accesses into registers, applies the callable to references to those registers and writes a chunk
//    back with 16-byte stores unless the compiler can prove the callable left it untouched (so `a = b + c` moves 12 B / px: a
//    chunk that is overwritten without being read is not loaded, one that is only read is not stored);
//  * relative_access / box_nbh2d ranges: a functor over global memory (row pitch + column offset), L1/L2 serve the overlap;
//  * box2d ranges: the coordinates.
// What stays on the host (pixel_wise.hh decides): callables with state (a by-reference capture would hold host addresses; use
// vpp::ops tags or the _device option to vouch for a by-value capture), non-default traversal options (_no_threads, _right_to_left,
// _bottom_to_top... imply an order), ranges that alias each other's storage, pixel types that are not trivially copyable.
#pragma once
#if defined(VPP_AMD_DEVICE) && defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <tuple>
#include <type_traits>

namespace vpp {
namespace pwdev {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- device-side accessors, built on the host from the ranges (mirror pointers) ----------------------------------
template <class V> struct image_acc { V* p0; int pitch; };                    // p0 = pixel (0, 0) in HBM
struct box_acc {};
template <class V> struct nbh_acc { V* p0; int pitch; const char *lo, *hi; };   // + the byte range of the whole buffer's mirror: what a neighbourhood of a view may reach (the LDS tile path)
template <class V, int R, int C> struct boxnbh_acc { V* p0; int pitch; const char *lo, *hi; };

template <class V> struct nbh_px {  // relative_access_kernel on the device (relative_accessor.hh:26-33)
  V* p; int pitch;
  __device__ V& operator()(int dr, int dc) const { return *(V*)((char*)p + (ptrdiff_t)dr * pitch + (ptrdiff_t)dc * (int)sizeof(V)); }
  __device__ V& operator()(vint2 d) const { return (*this)(d[0], d[1]); }
};
template <class V, int R, int C> struct boxnbh_px {  // box_nbh2d<V,R,C> at a point
  V* p; int pitch;
  __device__ V& operator()(int dr, int dc) const { return *(V*)((char*)p + (ptrdiff_t)dr * pitch + (ptrdiff_t)dc * (int)sizeof(V)); }
  __device__ V& north() const { return (*this)(-1, 0); }
  __device__ V& south() const { return (*this)(1, 0); }
  __device__ V& east() const { return (*this)(0, 1); }
  __device__ V& west() const { return (*this)(0, -1); }
  template <class F> __device__ void for_all(F f) const {
    for (int dr = -(R / 2); dr <= R / 2; dr++)
      for (int dc = -(C / 2); dc <= C / 2; dc++) f((*this)(dr, dc));
  }
};

// The same two accessors over an LDS tile (pixel_wise_tile_kernel): the row pitch is a compile-time constant, so the taps of an unrolled window are
// ds_read instructions with immediate offsets from ONE base register.  READ-ONLY by type (const V&): the tile is a copy, a write through it would never reach
// the image — so this path is only taken under the caller's `_nbh_read_only` option (launch), and a callable that assigns through it anyway does not compile.
// A4 (round 6): the lane's chunk starts 4-byte aligned in the tile and V is a 3-byte pixel (vuchar3).  A tap is then returned BY VALUE, cut out of the ALIGNED dwords that
// hold it (offset from the chunk's start: 3 i + dr LP + 3 dc — a constant once the callable's window loops are unrolled, so the compiler knows which dwords and which
// shift, and reads every dword of the window's rows once for the lane's four pixels: 7 ds_read_b32 per row instead of 120 ds_read_u8).  Through a `const V&` the
// callable's per-component reads are three byte reads per tap — 75 per pixel of a 5 x 5 window, which is what the tile kernel was bound by.  (One unaligned 4-byte
// read per tap was measured first: 115 us against the byte reads' 50 — LDS dwords at odd addresses are slow.)
template <class V, bool A4> struct tile_tap {
  typedef const V& type;
  static __device__ type get(const char* b, int off) { return *(const V*)(b + off); }
};
template <> struct tile_tap<vector<unsigned char, 3>, true> {
  typedef vector<unsigned char, 3> type;
  static __device__ type get(const char* b, int off) {
    const unsigned* w = (const unsigned*)__builtin_assume_aligned(b, 4);
    const int wi = off >> 2, sh = off & 3;
    const unsigned lo = w[wi], hi = sh >= 2 ? w[wi + 1] : 0u;
    type r;   // component j is byte sh + j of the dword pair: one bit-field extract (or a byte-select operand of the callable's own add) each
#pragma unroll
    for (int j = 0; j < 3; j++) r.v[j] = (unsigned char)(sh + j < 4 ? __builtin_amdgcn_ubfe(lo, 8u * (unsigned)(sh + j), 8u) : __builtin_amdgcn_ubfe(hi, 8u * (unsigned)(sh + j - 4), 8u));
    return r;
  }
};
template <class V, int LP, bool A4 = false> struct nbh_tile_px {
  const char* b; int o;   // the tap (dr, dc) is at b + o + dr LP + dc sizeof(V); A4: b = the lane's chunk (4-byte aligned), o = the pixel's offset in it
  __device__ typename tile_tap<V, A4>::type operator()(int dr, int dc) const { return tile_tap<V, A4>::get(b, o + dr * LP + dc * (int)sizeof(V)); }
  __device__ typename tile_tap<V, A4>::type operator()(vint2 d) const { return (*this)(d[0], d[1]); }
};
template <class V, int R, int C, int LP, bool A4 = false> struct boxnbh_tile_px {
  const char* b; int o;
  __device__ typename tile_tap<V, A4>::type operator()(int dr, int dc) const { return tile_tap<V, A4>::get(b, o + dr * LP + dc * (int)sizeof(V)); }
  __device__ typename tile_tap<V, A4>::type north() const { return (*this)(-1, 0); }
  __device__ typename tile_tap<V, A4>::type south() const { return (*this)(1, 0); }
  __device__ typename tile_tap<V, A4>::type east() const { return (*this)(0, 1); }
  __device__ typename tile_tap<V, A4>::type west() const { return (*this)(0, -1); }
  template <class F> __device__ void for_all(F f) const {
    for (int dr = -(R / 2); dr <= R / 2; dr++)
      for (int dc = -(C / 2); dc <= C / 2; dc++) f((*this)(dr, dc));
  }
};

template <class V, int LP, bool A4 = false> struct nbh_tile_acc { char* p00; int r00, c00; };                      // p00: the tile byte of pixel (r00, c00)
template <class V, int R, int C, int LP, bool A4 = false> struct boxnbh_tile_acc { char* p00; int r00, c00; };

template <class A> struct is_image_acc : std::false_type {};
template <class V> struct is_image_acc<image_acc<V>> : std::true_type {};

// per-lane staging of one range: NPX pixels of an image in registers; nothing for the other range kinds
template <class A, int NPX> struct stage { __device__ void load(const A&, int, int) {} __device__ void store(const A&, int, int) {} };
template <class V, int NPX> struct stage<image_acc<V>, NPX> {
  typedef typename std::remove_const<V>::type T;
  static constexpr int kBytes = NPX * (int)sizeof(T);
  static constexpr bool kVec = kBytes % 16 == 0;   // NPX > 1 chunks are 16-byte multiples and 16-byte aligned (the launcher checks)
  union buf { T px[NPX]; u32x4 q[(kBytes + 15) / 16]; unsigned char b[kBytes]; __device__ buf() {} } now, old;
  __device__ char* addr(const image_acc<V>& a, int r, int c) const { return (char*)a.p0 + (ptrdiff_t)r * a.pitch + (ptrdiff_t)c * (int)sizeof(T); }
  __device__ void load(const image_acc<V>& a, int r, int c) {
    const char* p = addr(a, r, c);
    if constexpr (kVec) {
#pragma unroll
      for (int i = 0; i < kBytes / 16; i++) now.q[i] = ((const u32x4*)p)[i];
    } else {
#pragma unroll
      for (int i = 0; i < NPX; i++) now.px[i] = ((const T*)p)[i];
    }
#pragma unroll
    for (int i = 0; i < kBytes; i++) old.b[i] = now.b[i];
  }
  __device__ void store(const image_acc<V>& a, int r, int c) {
    if (std::is_const<V>::value) return;
    // Unchanged chunks are not written.  When the compiler can see that the callable never assigns through this reference the
    // comparison folds to a constant and the store disappears; when it cannot tell at compile time, the chunk is stored
    // unconditionally — `old` is then dead, and with it the load of a chunk that the callable only overwrites.
    bool same = true;
#pragma unroll
    for (int i = 0; i < kBytes; i++) same = same && (now.b[i] == old.b[i]);
    if (__builtin_constant_p(same) && same) return;
    char* p = addr(a, r, c);
    if constexpr (kVec) {
#pragma unroll
      for (int i = 0; i < kBytes / 16; i++) __builtin_nontemporal_store(now.q[i], &((u32x4*)p)[i]);   // (streamed out like the library's own kernels' results: 4K `int` 5 x 5 lambda 21.7 -> 20.3 us)
    } else {
#pragma unroll
      for (int i = 0; i < NPX; i++) ((T*)p)[i] = now.px[i];
    }
  }
};

// the argument the callable sees for pixel i of the lane's chunk
template <class V, int NPX> __device__ V& arg(stage<image_acc<V>, NPX>& s, const image_acc<V>&, int, int, int i) { return (V&)s.now.px[i]; }
template <int NPX> __device__ vint2 arg(stage<box_acc, NPX>&, const box_acc&, int r, int c, int i) { return vint2(r, c + i); }
template <class V, int NPX> __device__ nbh_px<V> arg(stage<nbh_acc<V>, NPX>&, const nbh_acc<V>& a, int r, int c, int i) {
  return nbh_px<V>{(V*)((char*)a.p0 + (ptrdiff_t)r * a.pitch) + c + i, a.pitch};
}
template <class V, int R, int C, int NPX> __device__ boxnbh_px<V, R, C> arg(stage<boxnbh_acc<V, R, C>, NPX>&, const boxnbh_acc<V, R, C>& a, int r, int c, int i) {
  return boxnbh_px<V, R, C>{(V*)((char*)a.p0 + (ptrdiff_t)r * a.pitch) + c + i, a.pitch};
}
template <class V, int LP, bool A4, int NPX> __device__ nbh_tile_px<V, LP, A4> arg(stage<nbh_tile_acc<V, LP, A4>, NPX>&, const nbh_tile_acc<V, LP, A4>& a, int r, int c, int i) {
  return nbh_tile_px<V, LP, A4>{a.p00 + (r - a.r00) * LP + (c - a.c00) * (int)sizeof(V), i * (int)sizeof(V)};
}
template <class V, int R, int C, int LP, bool A4, int NPX> __device__ boxnbh_tile_px<V, R, C, LP, A4> arg(stage<boxnbh_tile_acc<V, R, C, LP, A4>, NPX>&, const boxnbh_tile_acc<V, R, C, LP, A4>& a, int r, int c, int i) {
  return boxnbh_tile_px<V, R, C, LP, A4>{a.p00 + (r - a.r00) * LP + (c - a.c00) * (int)sizeof(V), i * (int)sizeof(V)};
}
template <class F, class... X> __device__ __forceinline__ void call_lvalues(F& f, X&&... x) { f(x...); }  // kernels may take `auto&`

// NPX consecutive pixels of row r starting at column c: stage, apply, write back
template <int NPX, class F, class... A> __device__ __forceinline__ void pixel_step(F& f, int r, int c, const A&... acc) {
  std::tuple<stage<A, NPX>...> st;
  std::apply([&](auto&... s) { (void)std::initializer_list<int>{(s.load(acc, r, c), 0)...}; }, st);
#pragma unroll
  for (int i = 0; i < NPX; i++) std::apply([&](auto&... s) { call_lvalues(f, arg(s, acc, r, c, i)...); }, st);
  std::apply([&](auto&... s) { (void)std::initializer_list<int>{(s.store(acc, r, c), 0)...}; }, st);
}

// One lane = NPX consecutive pixels of one row (the row's ragged tail: pixel by pixel).  grid.x covers the chunks of a row,
// grid.y the rows.
template <int NPX, class F, class... A>
__global__ __launch_bounds__(256) void pixel_wise_kernel(F f, int r0, int c0, int nrows, int ncols, A... acc) {
  const int chunk = blockIdx.x * 256 + threadIdx.x;
  const int c = c0 + chunk * NPX;
  if (chunk * NPX >= ncols) return;
  const int n = min(NPX, ncols - chunk * NPX);
  for (int r = r0 + blockIdx.y; r < r0 + nrows; r += gridDim.y) {
    if (n == NPX) pixel_step<NPX>(f, r, c, acc...);
    else
      for (int i = 0; i < n; i++) pixel_step<1>(f, r, c + i, acc...);
  }
}

// ---- neighbourhoods out of LDS ------------------------------------------------------------------------------------------------------------------------
// A callable that reads a neighbourhood (benchmarks/box_5x5_filter2.cc:71-81: 25 taps per pixel) issued every tap as a global load: 4K vuchar3 5 x 5 mean 64 us
// against 13 us for the hand-written kernel.  Here a workgroup first stages the source rows of its TH x (64 NPX) pixel tile plus a halo of H pixels in LDS
// (16-byte loads, every byte fetched once per tile), and the callable's taps read the tile.  H = 4 covers windows up to 9 x 9.  The tile is a COPY: the launcher
// takes this path only under the caller's `_nbh_read_only` option (the callable reads through the neighbourhood, never writes, and no tap reaches further than H
// pixels — in the image's interior a tap may legally reach further than the border, and a neighbourhood is a V& the reference's own code writes through,
// distance_transforms.hh), for a box_nbh2d<V, R, C> only when R / 2 and C / 2 are at most H, when the source's pitch is a multiple of 16 and there is exactly one
// neighbourhood range.
constexpr int kTileH = 4, kTileRows = 16;
template <class V, int NPXK> struct tile_geom {
  static constexpr int ES = (int)sizeof(V), TW = 64 * NPXK;
  static constexpr int LP = (((TW + 2 * kTileH) * ES + 15 + 15) / 16) * 16 + 16;   // row bytes + the alignment shift, in 16-byte units, + 16: consecutive rows start 4 banks apart
};
template <class A> struct nbh_traits { static constexpr bool value = false; static constexpr int reach = 0; };
template <class V> struct nbh_traits<nbh_acc<V>> {
  static constexpr bool value = true; typedef V pixel;
  static constexpr int reach = 0;   // unknown: vouched for by `_nbh_read_only`
  template <int LP, bool A4 = false> using tile = nbh_tile_acc<V, LP, A4>;
};
template <class V, int R, int C> struct nbh_traits<boxnbh_acc<V, R, C>> {
  static constexpr bool value = true; typedef V pixel;
  static constexpr int reach = (R / 2 > C / 2 ? R / 2 : C / 2);
  template <int LP, bool A4 = false> using tile = boxnbh_tile_acc<V, R, C, LP, A4>;
};
template <class... A> struct first_nbh;
template <class A0, class... A> struct first_nbh<A0, A...> {
  typedef typename std::conditional<nbh_traits<A0>::value, A0, typename first_nbh<A...>::type>::type type;
  __device__ __host__ static const type& get(const A0& a0, const A&... a) { if constexpr (nbh_traits<A0>::value) return a0; else return first_nbh<A...>::get(a...); }
};
template <> struct first_nbh<> { typedef void type; };
// the tile accessor in place of the neighbourhood range, everything else as it is
template <class A, class T> __device__ __forceinline__ const typename std::conditional<nbh_traits<A>::value, T, A>::type& tile_swap(const A& a, const T& t) {
  if constexpr (nbh_traits<A>::value) return t; else return a;
}

// A4: every lane's chunk of 4 pixels starts 4-byte aligned in the tile and the rows have no ragged tail (the launcher checks): 3-byte pixels are tapped out of aligned dwords (tile_tap)
template <int NPXK, bool A4, class F, class... A>
__global__ __launch_bounds__(256) void pixel_wise_tile_kernel(F f, int r0, int c0, int nrows, int ncols, A... acc) {
  typedef typename first_nbh<A...>::type NA;
  typedef typename nbh_traits<NA>::pixel V;
  typedef tile_geom<V, NPXK> G;
  constexpr int ES = G::ES, TW = G::TW, LP = G::LP, H = kTileH, TH = kTileRows, ROWS = TH + 2 * H, CPR = LP / 16;
  __shared__ __attribute__((aligned(16))) char lds[ROWS * LP + 16];   // (+ 16: tile_tap<.., true> reads the whole dword that holds a pixel's last byte)
  const NA& nb = first_nbh<A...>::get(acc...);
  const int tr = r0 + blockIdx.y * TH, tc = c0 + blockIdx.x * TW;
  // ---- stage: tile row rr holds image row tr - H + rr from pixel column tc - H on, at the byte offset `shift` (its global address modulo 16)
  const char* g0 = (const char*)nb.p0 + (ptrdiff_t)(tr - H) * nb.pitch + (ptrdiff_t)(tc - H) * ES;
  const int shift = (int)((size_t)g0 & 15);
  const char* ga = g0 - shift;
  // Staged: every byte of the tile's rows that lies inside the buffer's mirror [nb.lo, nb.hi) — for a view (sub-image) that is more than the view's own
  // bordered area, as on the host, where a tap may reach whatever the parent image holds there; bytes outside the buffer are never touched (nor legally tapped).
  const int want_hi = shift + (TW + 2 * H) * ES;
  // (round 6) A tile whose staged bytes all lie inside the buffer — one test per workgroup — requests its chunks back to back, without a test or a branch in front of any
  // load, and stores them afterwards: in the guarded loop below every pass of the workgroup waits for its own load before the next one is requested.
  constexpr int NIT = (ROWS * CPR + 255) / 256;
  if (nb.pitch > 0 && ga >= nb.lo && ga + (ptrdiff_t)(ROWS - 1) * nb.pitch + ((want_hi + 15) & ~15) <= nb.hi) {
    u32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = min((int)threadIdx.x + 256 * it, ROWS * CPR - 1), rr = k / CPR, off = (k - rr * CPR) * 16;
      v[it] = *(const u32x4*)(ga + (ptrdiff_t)rr * nb.pitch + (off < want_hi ? off : 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int k = (int)threadIdx.x + 256 * it, rr = k / CPR, off = (k - rr * CPR) * 16;
      if (k < ROWS * CPR && off < want_hi) *(u32x4*)(lds + rr * LP + off) = v[it];
    }
  } else
  for (int k = threadIdx.x; k < ROWS * CPR; k += 256) {
    const int rr = k / CPR, off = (k - rr * CPR) * 16;
    if (off >= want_hi) continue;
    const char* src = ga + (ptrdiff_t)rr * nb.pitch + off;
    char* dst = lds + rr * LP + off;
    if (src >= nb.lo && src + 16 <= nb.hi) *(u32x4*)dst = *(const u32x4*)src;
    else if (src + 16 > nb.lo && src < nb.hi)
      for (int b = 0; b < 16; b++) if (src + b >= nb.lo && src + b < nb.hi) dst[b] = src[b];   // a chunk cut by the buffer's first / last byte
  }
  __syncthreads();
  // ---- compute: wave w takes rows [w TH/4, (w + 1) TH/4) of the tile, a lane NPXK consecutive pixels
  const typename nbh_traits<NA>::template tile<LP, A4> ta{lds + shift + H * LP + H * ES, tr, tc};
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = tc + lane * NPXK;
  if (c >= c0 + ncols) return;
  const int n = min(NPXK, c0 + ncols - c);
#pragma unroll 1
  for (int j = 0; j < TH / 4; j++) {
    const int r = tr + wv * (TH / 4) + j;
    if (r >= r0 + nrows) break;
    if (n == NPXK) pixel_step<NPXK>(f, r, c, tile_swap(acc, ta)...);
    else
      for (int i = 0; i < n; i++) pixel_step<1>(f, r, c + i, tile_swap(acc, ta)...);
  }
}

// ---- neighbourhoods out of a register window (round 6: dword pixel types under `_nbh_read_only`) -----------------------------------------------------------
// For 4-byte pixels the LDS tile LOSES to the global taps (39 vs 27 us on the 4K `int` 5 x 5 mean: dword taps already come out of L1 / L2 near the streaming
// rate, the staging pass and its barrier only add) — but 25 taps per pixel are still 25 load instructions per pixel.  Here a lane owns 4 consecutive pixels of a
// row and marches DOWN kWinRows rows: it keeps the rows r - 4 .. r + 4 of the columns c - 4 .. c + 7 (three 16-byte chunks per row) in registers, loads ONE new
// row per output row and shifts the window (register renaming once the row loop is unrolled); the callable's taps index that window with constants once its
// window loops are unrolled, so they cost no instruction at all, and the window entries a callable never reads are never loaded (dead loads: a 5 x 5 body keeps
// 5 rows x 3 chunks live).  12 B of 16-byte loads per pixel instead of 100 B of dword loads.  READ-ONLY by type (taps are returned by value): taken only under
// `_nbh_read_only` (reach <= 4, no writes through the neighbourhood), for one neighbourhood range of a 4-byte pixel type on a 16-byte aligned image.
constexpr int kWinRows = 8;
template <class V> struct win_regs { V px[2 * kTileH + 1][12]; };   // rows r - 4 .. r + 4, columns c - 4 .. c + 7 of the lane's 4-pixel chunk at (r, c)
template <class V> struct nbh_win_px {
  const win_regs<V>* w; int i;
  __device__ __forceinline__ V operator()(int dr, int dc) const { return w->px[dr + kTileH][4 + i + dc]; }
  __device__ __forceinline__ V operator()(vint2 d) const { return (*this)(d[0], d[1]); }
};
template <class V, int R, int C> struct boxnbh_win_px {
  const win_regs<V>* w; int i;
  __device__ __forceinline__ V operator()(int dr, int dc) const { return w->px[dr + kTileH][4 + i + dc]; }
  __device__ __forceinline__ V north() const { return (*this)(-1, 0); }
  __device__ __forceinline__ V south() const { return (*this)(1, 0); }
  __device__ __forceinline__ V east() const { return (*this)(0, 1); }
  __device__ __forceinline__ V west() const { return (*this)(0, -1); }
  template <class F> __device__ __forceinline__ void for_all(F f) const {
#pragma unroll
    for (int dr = -(R / 2); dr <= R / 2; dr++)
#pragma unroll
      for (int dc = -(C / 2); dc <= C / 2; dc++) f((*this)(dr, dc));
  }
};
template <class V> struct nbh_win_acc { const win_regs<V>* w; };
template <class V, int R, int C> struct boxnbh_win_acc { const win_regs<V>* w; };
template <class V, int NPX> __device__ __forceinline__ nbh_win_px<V> arg(stage<nbh_win_acc<V>, NPX>&, const nbh_win_acc<V>& a, int, int, int i) { return nbh_win_px<V>{a.w, i}; }
template <class V, int R, int C, int NPX> __device__ __forceinline__ boxnbh_win_px<V, R, C> arg(stage<boxnbh_win_acc<V, R, C>, NPX>&, const boxnbh_win_acc<V, R, C>& a, int, int, int i) {
  return boxnbh_win_px<V, R, C>{a.w, i};
}
template <class A> struct win_of;
template <class V> struct win_of<nbh_acc<V>> { typedef nbh_win_acc<V> type; };
template <class V, int R, int C> struct win_of<boxnbh_acc<V, R, C>> { typedef boxnbh_win_acc<V, R, C> type; };

// one row of the window: the three 16-byte chunks at columns c - 4, c, c + 4 of image row r; bytes outside the buffer's mirror [lo, hi) are never touched
// INSIDE: every chunk of the wave's window is known to lie inside the buffer (checked once per wave): plain loads, no test and no branch in front of any of them — behind a
// test each load sat in a branch of its own and was waited for at that branch's end (36 loads, 48 waits and 274 branches in the 5 x 5 `int` kernel's code)
template <class V, bool INSIDE = false> __device__ __forceinline__ void win_load_row(V (&dst)[12], const char* p0, int pitch, int r, int c, const char* lo, const char* hi) {
  const char* src = p0 + (ptrdiff_t)r * pitch + (ptrdiff_t)(c - 4) * 4;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const char* s = src + 16 * q;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (INSIDE || (s >= lo && s + 16 <= hi)) v = *(const u32x4*)s;   // (plain: the neighbouring waves' windows re-read these rows out of L2 — non-temporal loads measured 20.4 -> 25.5 us)
    else if (s + 16 > lo && s < hi) {   // a chunk cut by the buffer's first / last byte
      unsigned int e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 4; k++) if (s + 4 * k >= lo && s + 4 * k + 4 <= hi) e[k] = *(const unsigned int*)(s + 4 * k);
      v = u32x4{e[0], e[1], e[2], e[3]};
    }
    __builtin_memcpy(&dst[4 * q], &v, 16);
  }
}

// (ragged tail of pixel_wise_window_kernel: ranges staged one pixel at a time — index 0 — while the window accessor still needs the pixel's place in the lane's chunk)
template <class S, class A> __device__ __forceinline__ decltype(auto) win_arg(S& s, const A& a, int r, int c, int) { return arg(s, a, r, c, 0); }
template <class V> __device__ __forceinline__ nbh_win_px<V> win_arg(stage<nbh_win_acc<V>, 1>&, const nbh_win_acc<V>& a, int, int, int i) { return nbh_win_px<V>{a.w, i}; }
template <class V, int R, int C> __device__ __forceinline__ boxnbh_win_px<V, R, C> win_arg(stage<boxnbh_win_acc<V, R, C>, 1>&, const boxnbh_win_acc<V, R, C>& a, int, int, int i) {
  return boxnbh_win_px<V, R, C>{a.w, i};
}

template <bool INSIDE, class F, class... A>
__device__ __forceinline__ void pixel_wise_window_body(F& f, int r0, int c0, int nrows, int ncols, A&... acc) {
  typedef typename first_nbh<A...>::type NA;
  typedef typename nbh_traits<NA>::pixel V;
  static_assert(sizeof(V) == 4, "the register window holds 4-byte pixels");
  const NA& nb = first_nbh<A...>::get(acc...);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = c0 + (blockIdx.x * 64 + lane) * 4;
  const int rw = r0 + (blockIdx.y * 4 + wv) * kWinRows;          // the wave's first output row
  if (rw >= r0 + nrows) return;
  const bool col_ok = c < c0 + ncols;
  const int n = col_ok ? min(4, c0 + ncols - c) : 0;
  const int cc = col_ok ? c : c0;                                  // (lanes past the row's end load in-range columns and drop them)
  win_regs<V> w;
#pragma unroll
  for (int k = 0; k < 2 * kTileH; k++) win_load_row<V, INSIDE>(w.px[k + 1], (const char*)nb.p0, nb.pitch, rw - kTileH + k, cc, nb.lo, nb.hi);
  const typename win_of<NA>::type wa{&w};
#pragma unroll
  for (int j = 0; j < kWinRows; j++) {
    const int r = rw + j;
#pragma unroll
    for (int k = 0; k < 2 * kTileH; k++)
#pragma unroll
      for (int q = 0; q < 12; q++) w.px[k][q] = w.px[k + 1][q];
    if (INSIDE || r < r0 + nrows) win_load_row<V, INSIDE>(w.px[2 * kTileH], (const char*)nb.p0, nb.pitch, r + kTileH, cc, nb.lo, nb.hi);
    if (r < r0 + nrows && n) {
      if (n == 4) pixel_step<4>(f, r, c, tile_swap(acc, wa)...);
      else {
#pragma unroll
        for (int i = 0; i < 3; i++) if (i < n) {   // the row's ragged tail, pixel by pixel (the window is the lane's: pixel i reads it at i)
          const typename win_of<NA>::type wi{&w};
          std::tuple<stage<typename std::conditional<nbh_traits<A>::value, typename win_of<NA>::type, A>::type, 1>...> st1;
          std::apply([&](auto&... s_) { (void)std::initializer_list<int>{(s_.load(tile_swap(acc, wi), r, c + i), 0)...}; }, st1);
          std::apply([&](auto&... s_) { call_lvalues(f, win_arg(s_, tile_swap(acc, wi), r, c + i, i)...); }, st1);
          std::apply([&](auto&... s_) { (void)std::initializer_list<int>{(s_.store(tile_swap(acc, wi), r, c + i), 0)...}; }, st1);
        }
      }
    }
  }
}

template <class F, class... A>
__global__ __launch_bounds__(256) void pixel_wise_window_kernel(F f, int r0, int c0, int nrows, int ncols, A... acc) {
  typedef typename first_nbh<A...>::type NA;
  const NA& nb = first_nbh<A...>::get(acc...);
  // the wave's whole window — rows rw - 4 .. rw + kWinRows + 3, the 16-byte chunks from column (first lane's) - 4 to (last lane's) + 8 — inside the buffer: the fast path
  const int wv = threadIdx.x >> 6;
  const int rw = r0 + (blockIdx.y * 4 + wv) * kWinRows, cw = c0 + (int)blockIdx.x * 256;   // the wave's first output row / the workgroup's first column
  const char* first = (const char*)nb.p0 + (ptrdiff_t)(rw - kTileH) * nb.pitch + (ptrdiff_t)(c0 - 4) * 4;   // (lanes past the row's end load from column c0)
  const char* last = (const char*)nb.p0 + (ptrdiff_t)(rw + kWinRows - 1 + kTileH) * nb.pitch + (ptrdiff_t)(cw + 63 * 4 + 8) * 4;
  if (nb.pitch > 0 && first >= nb.lo && last <= nb.hi) pixel_wise_window_body<true>(f, r0, c0, nrows, ncols, acc...);
  else pixel_wise_window_body<false>(f, r0, c0, nrows, ncols, acc...);
}

constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
constexpr int lcm_(int a, int b) { return a / gcd_(a, b) * b; }
template <class A> struct npx_of { enum { value = 1 }; };
template <class V> struct npx_of<image_acc<V>> { enum { value = 16 / gcd_(16, (int)sizeof(V)) }; };
template <class... A> struct npx_all;
template <> struct npx_all<> { enum { value = 1 }; };
template <class A0, class... A> struct npx_all<A0, A...> { enum { value = lcm_(npx_of<A0>::value, npx_all<A...>::value) }; };

template <class A> inline bool aligned16(const A&, int) { return true; }
template <class V> inline bool aligned16(const image_acc<V>& a, int c0) { return ((size_t)((char*)a.p0 + (ptrdiff_t)c0 * (int)sizeof(V)) % 16) == 0 && a.pitch % 16 == 0; }

template <class A> struct is_nbh_acc : std::false_type {};
template <class V> struct is_nbh_acc<nbh_acc<V>> : std::true_type {};
template <class V, int R, int C> struct is_nbh_acc<boxnbh_acc<V, R, C>> : std::true_type {};

// NBH_RO: the call carries `_nbh_read_only` (see pixel_wise_tile_kernel)
template <bool NBH_RO, class F, class... A> void launch(F f, int r0, int c0, int nrows, int ncols, A... acc) {
  if (nrows <= 0 || ncols <= 0) return;
  device::flush_held_back();   // these kernels are launched by the caller's TU, not through the ABI: frames the tagged functors have held back go first
  constexpr int NPX = npx_all<A...>::value;
  constexpr bool kNbh = (is_nbh_acc<A>::value || ...);
  bool al = true;
  (void)std::initializer_list<int>{(al = al && aligned16(acc, c0), 0)...};
  const int gy = nrows < 65535 ? nrows : 65535;
  if constexpr (NBH_RO && kNbh && ((nbh_traits<A>::value ? 1 : 0) + ...) == 1) {   // one read-only neighbourhood range: its taps out of an LDS tile
   if constexpr (nbh_traits<typename first_nbh<A...>::type>::reach <= kTileH) {
    const auto& nb = first_nbh<A...>::get(acc...);
    static const bool off = [] { const char* e = getenv("VPP_PW_TILE"); return e && e[0] == '0'; }();   // A/B switch for the tests and the benchmark
    // Measured (4K 5 x 5 mean through the opaque lambda, synchronous calls, same box): vuchar3 64.2 us with global taps -> 56.9 us out of the tile (its taps are
    // byte loads: the tile turns 75 of them per pixel into a handful of wide LDS reads, the rest is the callable's own per-component arithmetic); `int` 27.1 us
    // with global taps -> 39.3 us out of the tile (dword taps already come out of L1 / L2 at near the streaming rate; the staging pass, the barrier and the
    // 1.5 x halo rows only add).  So: pixel types that are not dword multiples take the tile, the others keep the global taps.
    typedef typename nbh_traits<typename first_nbh<A...>::type>::pixel PV;
    if constexpr (sizeof(PV) == 4 && NPX == 4) {   // 4-byte pixels, every image range 4-byte too: the register window (pixel_wise_window_kernel)
      static const bool woff = [] { const char* e = getenv("VPP_PW_WINDOW"); return e && e[0] == '0'; }();   // A/B switch for the tests and the benchmark
      if (!woff && al && nb.pitch % 16 == 0 && ((size_t)((const char*)nb.p0 + (ptrdiff_t)c0 * 4) % 16) == 0) {
        dim3 grid((ncols + 255) / 256, (nrows + 4 * kWinRows - 1) / (4 * kWinRows));
        hipLaunchKernelGGL((pixel_wise_window_kernel<F, A...>), grid, dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) throw std::runtime_error(std::string("pixel_wise (device, register window): launch failed: ") + hipGetErrorString(e));
        device::call_done();   // queued, not drained: vpp/core/device.hh
        return;
      }
    }
    if (!off && sizeof(PV) % 4 != 0 && nb.pitch % 16 == 0) {
      constexpr int NPXK = NPX % 4 == 0 ? 4 : 1;
      const bool vec = al && NPXK > 1;
      dim3 grid((ncols + 64 * (vec ? NPXK : 1) - 1) / (64 * (vec ? NPXK : 1)), (nrows + kTileRows - 1) / kTileRows);
      // 3-byte pixels, 4 per lane: a chunk starts 4-byte aligned in the tile when its first byte does in memory (16-byte aligned rows, c0 a multiple of 4: 12-byte chunks)
      constexpr bool kA4 = sizeof(PV) == 3 && NPXK == 4 && kTileH % 4 == 0;
      static const bool a4off = [] { const char* e = getenv("VPP_PW_TILE_A4"); return e && e[0] == '0'; }();   // A/B switch for the tests and the benchmark
      if (kA4 && vec && !a4off && ((size_t)nb.p0 % 16) == 0 && c0 % 4 == 0 && ncols % 4 == 0)
        hipLaunchKernelGGL((pixel_wise_tile_kernel<NPXK, kA4, F, A...>), grid, dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
      else if (vec) hipLaunchKernelGGL((pixel_wise_tile_kernel<NPXK, false, F, A...>), grid, dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
      else hipLaunchKernelGGL((pixel_wise_tile_kernel<1, false, F, A...>), grid, dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) throw std::runtime_error(std::string("pixel_wise (device, tiled): launch failed: ") + hipGetErrorString(e));
      device::call_done();   // queued, not drained: vpp/core/device.hh
      return;
    }
   }
  }
  // A callable that reads a neighbourhood issues its taps per pixel, so with NPX pixels per lane the lanes of a wave sit NPX pixels apart and every
  // tap of the wave is spread over NPX times as many cache lines.  For pixel types whose 16-byte chunk is many pixels (vuchar3: 16 px = 48 B per lane,
  // a byte load per component and tap) four pixels per lane (12 B, moved pixel by pixel: packing them into dwords made the compiler keep the
  // chunk in LDS, 175 us) are the measured optimum — 4K vuchar3 5 x 5 mean through the
  // opaque lambda, synchronous calls: 16 px per lane 369 us, 1 px 107 us, 4 px 63 us (ops::box_mean<5, 5>: 19 us; `int` pixels are 4 px per lane anyway)
  if (kNbh && al && NPX > 4 && NPX % 4 == 0) {
    const int chunks = (ncols + 3) / 4;
    hipLaunchKernelGGL((pixel_wise_kernel<(NPX % 4 == 0 ? 4 : 1), F, A...>), dim3((chunks + 255) / 256, gy), dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
  } else if (al && NPX > 1) {
    const int chunks = (ncols + NPX - 1) / NPX;
    hipLaunchKernelGGL((pixel_wise_kernel<NPX, F, A...>), dim3((chunks + 255) / 256, gy), dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
  } else {  // unaligned view (sub-image at an odd column, foreign pitch): one pixel per lane
    hipLaunchKernelGGL((pixel_wise_kernel<1, F, A...>), dim3((ncols + 255) / 256, gy), dim3(256), 0, (hipStream_t)device::stream(), f, r0, c0, nrows, ncols, acc...);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("pixel_wise (device): launch failed: ") + hipGetErrorString(e));
  device::call_done();   // queued, not drained: vpp/core/device.hh
}

// ---- block_wise on the device (vpp/core/block_wise.hh:26-56): one lane per block, the callable sees one view per range --------
// A view is what a sub-image is to host code, reduced to what device code can hold: pixel (0, 0) of the block, the pitch and the
// block's extent (clipped to the domain).  view(r, c), view(vint2), view[r] (row pointer), nrows(), ncols().
template <class V> struct block_view {
  V* p0; int pitch, nr, nc;
  __device__ int nrows() const { return nr; }
  __device__ int ncols() const { return nc; }
  __device__ V* operator[](int r) const { return (V*)((char*)p0 + (ptrdiff_t)r * pitch); }
  __device__ V& operator()(int r, int c) const { return (*this)[r][c]; }
  __device__ V& operator()(vint2 p) const { return (*this)[p[0]][p[1]]; }
};
// fill(view, value) inside a block_wise callable (tests/block_wise.cc:104 `[] (auto si) { fill(si, 1); }`)
template <class V, class U> __device__ void fill(const block_view<V>& v, U value) {
  for (int r = 0; r < v.nr; r++) for (int c = 0; c < v.nc; c++) v(r, c) = V(value);
}
struct box_view {  // a box2d range: the block's corners in the coordinates of the first range
  int r0, c0, r1, c1;
  __device__ vint2 p1() const { return vint2(r0, c0); }
  __device__ vint2 p2() const { return vint2(r1, c1); }
  __device__ int nrows() const { return r1 - r0 + 1; }
  __device__ int ncols() const { return c1 - c0 + 1; }
};
template <class V> __device__ block_view<V> block_arg(const image_acc<V>& a, int r0, int c0, int r1, int c1) {
  return block_view<V>{(V*)((char*)a.p0 + (ptrdiff_t)r0 * a.pitch) + c0, a.pitch, r1 - r0 + 1, c1 - c0 + 1};
}
__device__ inline box_view block_arg(const box_acc&, int r0, int c0, int r1, int c1) { return box_view{r0, c0, r1, c1}; }

template <class F, class... A>
__global__ __launch_bounds__(64) void block_wise_kernel(F f, int rstart, int cstart, int rend, int cend, int bsr, int bsc, int gr, int gc, A... acc) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= gr * gc) return;
  const int br = b / gc, bc = b - br * gc;
  const int r0 = rstart + br * bsr, c0 = cstart + bc * bsc;
  const int r1 = min(r0 + bsr - 1, rend), c1 = min(c0 + bsc - 1, cend);
  call_lvalues(f, block_arg(acc, r0, c0, r1, c1)...);
}
template <class F, class... A> void launch_blocks(F f, int rstart, int cstart, int rend, int cend, int bsr, int bsc, A... acc) {
  const int gr = (rend - rstart) / bsr + 1, gc = (cend - cstart) / bsc + 1;   // block_wise.hh:37-38
  if (gr <= 0 || gc <= 0) return;
  device::flush_held_back();   // (see launch)
  hipLaunchKernelGGL((block_wise_kernel<F, A...>), dim3((gr * gc + 63) / 64), dim3(64), 0, (hipStream_t)device::stream(), f, rstart, cstart, rend, cend, bsr, bsc, gr, gc, acc...);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("block_wise (device): launch failed: ") + hipGetErrorString(e));
  device::call_done();   // queued, not drained: vpp/core/device.hh
}

}  // namespace pwdev
}  // namespace vpp
#endif
