// pixelwise.hip — K1: pixel_wise maps (reference: vpp/core/pixel_wise.hpp:68-105,146-165) for arithmetic kernels,
// plus copy / fill (vpp/core/copy.hh:10-27, vpp/core/fill.hh:12-29).
//
// HBM-bound streaming: one lane moves 16 B per access (64 lanes = 1 KiB per wave instruction), UNROLL independent
// accesses in flight per lane.  Row-pitch aware: images whose rows are contiguous are walked as one flat byte range,
// otherwise one grid row per image row.  Algorithmic bytes: (2 reads + 1 write) * sizeof(V) per pixel (12 B/px for int).
#include "common.hpp"
#include <algorithm>
#include <map>
#include <vector>
#include <algorithm>
#include <cstring>
#include <type_traits>
using namespace vpp_amd;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int OP, class T> __device__ __forceinline__ T op_scalar(T a, T b) {
  if constexpr (sizeof(T) == 4 && !__is_floating_point(T)) {
    typedef uint32_t U;  // wrap-around in unsigned arithmetic
    if constexpr (OP == VPP_OP_ADD) return (T)((U)a + (U)b);
    if constexpr (OP == VPP_OP_SUB) return (T)((U)a - (U)b);
    if constexpr (OP == VPP_OP_MUL) return (T)((U)a * (U)b);
    if constexpr (OP == VPP_OP_MIN) return a < b ? a : b;
    if constexpr (OP == VPP_OP_MAX) return a > b ? a : b;
    return a > b ? (T)((U)a - (U)b) : (T)((U)b - (U)a);
  } else if constexpr (__is_floating_point(T)) {
    if constexpr (OP == VPP_OP_ADD) return a + b;
    if constexpr (OP == VPP_OP_SUB) return a - b;
    if constexpr (OP == VPP_OP_MUL) return a * b;
    if constexpr (OP == VPP_OP_MIN) return a < b ? a : b;
    if constexpr (OP == VPP_OP_MAX) return a > b ? a : b;
    return a > b ? a - b : b - a;
  } else {
    int x = a, y = b;  // integer promotion, then conversion back to T (modulo 2^bits)
    if constexpr (OP == VPP_OP_ADD) return (T)(x + y);
    if constexpr (OP == VPP_OP_SUB) return (T)(x - y);
    if constexpr (OP == VPP_OP_MUL) return (T)(x * y);
    if constexpr (OP == VPP_OP_MIN) return a < b ? a : b;
    if constexpr (OP == VPP_OP_MAX) return a > b ? a : b;
    return (T)(a > b ? x - y : y - x);
  }
}

// ---- 8-bit elements, four per dword, without unpacking to scalars (the per-byte form is ~4 VALU ops per byte and turns the
// stream VALU-bound at 2.7 TB/s).  Results are modulo 256, exactly the conversion back to T of op_scalar. ----
template <int OP, bool SIGNED> __device__ __forceinline__ uint32_t op_bytes(uint32_t x, uint32_t y) {
  constexpr uint32_t H = 0x80808080u, L = 0x7f7f7f7fu, E = 0x00ff00ffu;
  if constexpr (OP == VPP_OP_ADD) return ((x & L) + (y & L)) ^ ((x ^ y) & H);             // carries cannot cross a byte
  else if constexpr (OP == VPP_OP_SUB) return ((x | H) - (y & L)) ^ ((x ^ ~y) & H);       // borrows cannot cross a byte
  else {
    if constexpr (SIGNED && OP != VPP_OP_MUL) { x ^= H; y ^= H; }                         // order-preserving map to unsigned bytes
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; u16x2 v; } xe, xo, ye, yo, re, ro;
    xe.u = x & E; xo.u = (x >> 8) & E; ye.u = y & E; yo.u = (y >> 8) & E;                  // even / odd bytes as packed u16 pairs
    if constexpr (OP == VPP_OP_MUL) { re.v = xe.v * ye.v; ro.v = xo.v * yo.v; }
    else if constexpr (OP == VPP_OP_MIN) { re.v = __builtin_elementwise_min(xe.v, ye.v); ro.v = __builtin_elementwise_min(xo.v, yo.v); }
    else if constexpr (OP == VPP_OP_MAX) { re.v = __builtin_elementwise_max(xe.v, ye.v); ro.v = __builtin_elementwise_max(xo.v, yo.v); }
    else { re.v = __builtin_elementwise_max(xe.v, ye.v) - __builtin_elementwise_min(xe.v, ye.v);
           ro.v = __builtin_elementwise_max(xo.v, yo.v) - __builtin_elementwise_min(xo.v, yo.v); }
    uint32_t r = (re.u & E) | ((ro.u & E) << 8);
    if constexpr (SIGNED && (OP == VPP_OP_MIN || OP == VPP_OP_MAX)) r ^= H;
    return r;
  }
}

template <int OP, class T> __device__ __forceinline__ u32x4 op_vec(u32x4 a, u32x4 b) {
  if constexpr (sizeof(T) == 1) {
    constexpr bool S = std::is_signed<T>::value;
    return u32x4{op_bytes<OP, S>(a.x, b.x), op_bytes<OP, S>(a.y, b.y), op_bytes<OP, S>(a.z, b.z), op_bytes<OP, S>(a.w, b.w)};
  } else if constexpr (sizeof(T) == 2) {  // packed 16-bit VALU ops (v_pk_add_u16, v_pk_min_i16, ...), one dword = two elements
    typedef T t16x2 __attribute__((ext_vector_type(2)));
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    auto one = [](uint32_t x, uint32_t y) -> uint32_t {
      t16x2 tx, ty; u16x2 ux, uy, ur;
      __builtin_memcpy(&tx, &x, 4); __builtin_memcpy(&ty, &y, 4); __builtin_memcpy(&ux, &x, 4); __builtin_memcpy(&uy, &y, 4);
      if constexpr (OP == VPP_OP_ADD) ur = ux + uy;
      else if constexpr (OP == VPP_OP_SUB) ur = ux - uy;
      else if constexpr (OP == VPP_OP_MUL) ur = ux * uy;
      else {
        const t16x2 hi = __builtin_elementwise_max(tx, ty), lo = __builtin_elementwise_min(tx, ty);
        u16x2 uh, ul;
        __builtin_memcpy(&uh, &hi, 4); __builtin_memcpy(&ul, &lo, 4);
        ur = OP == VPP_OP_MIN ? ul : (OP == VPP_OP_MAX ? uh : (u16x2)(uh - ul));
      }
      uint32_t r;
      __builtin_memcpy(&r, &ur, 4);
      return r;
    };
    return u32x4{one(a.x, b.x), one(a.y, b.y), one(a.z, b.z), one(a.w, b.w)};
  } else {
    // 32-bit elements: per component on plain dwords (a union with an element array made the unrolled kernel keep 456 VGPRs for
    // float and run at one wave per SIMD)
    static_assert(sizeof(T) == 4, "element sizes 1, 2 and 4");
    auto one = [](uint32_t x, uint32_t y) -> uint32_t {
      T tx, ty;
      __builtin_memcpy(&tx, &x, 4); __builtin_memcpy(&ty, &y, 4);
      const T tr = op_scalar<OP, T>(tx, ty);
      uint32_t r;
      __builtin_memcpy(&r, &tr, 4);
      return r;
    };
    return u32x4{one(a.x, b.x), one(a.y, b.y), one(a.z, b.z), one(a.w, b.w)};
  }
}

// Flat range of nvec 16-byte vectors (+ tail bytes handled by the scalar kernel).
template <int OP, class T, int UNROLL, bool NT>
__device__ __forceinline__ void binary_flat_body(u32x4* __restrict__ d, const u32x4* __restrict__ a, const u32x4* __restrict__ b, size_t nvec, unsigned blk) {
  const size_t base = (size_t)blk * (256 * UNROLL) + threadIdx.x;
  if constexpr (std::is_integral<T>::value && sizeof(T) == 4) {
    // 32-bit integers (the 4K add of the benchmark): per-access guards.  This form compiles to 40 VGPRs and measures 15.7 us on the
    // 99.5 MB add; the unguarded form below makes the scheduler hold more loads back (56-66 VGPRs) and measures 16.6 us.
    u32x4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const size_t i = base + (size_t)u * 256;
      if (i < nvec) {
        va[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
        vb[u] = NT ? __builtin_nontemporal_load(b + i) : b[i];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const size_t i = base + (size_t)u * 256;
      if (i < nvec) {
        const u32x4 r = op_vec<OP, T>(va[u], vb[u]);
        if (NT) __builtin_nontemporal_store(r, d + i); else d[i] = r;
      }
    }
    return;
  }
  // other element types: the guarded form makes clang build 32-wide register tuples for float (456 VGPRs, one wave per SIMD)
  if (base + (size_t)(UNROLL - 1) * 256 < nvec) {  // whole block in range (every block but the last): no per-access guards
    u32x4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const size_t i = base + (size_t)u * 256;
      va[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
      vb[u] = NT ? __builtin_nontemporal_load(b + i) : b[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const size_t i = base + (size_t)u * 256;
      const u32x4 r = op_vec<OP, T>(va[u], vb[u]);
      if (NT) __builtin_nontemporal_store(r, d + i); else d[i] = r;
    }
    return;
  }
#pragma unroll 1
  for (int u = 0; u < UNROLL; u++) {  // ragged last block
    const size_t i = base + (size_t)u * 256;
    if (i < nvec) d[i] = op_vec<OP, T>(a[i], b[i]);
  }
}

template <int OP, class T, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void binary_flat_kernel(u32x4* __restrict__ d, const u32x4* __restrict__ a,
                                                          const u32x4* __restrict__ b, size_t nvec) {
  binary_flat_body<OP, T, UNROLL, NT>(d, a, b, nvec, blockIdx.x);
}
// n image triples of one size in ONE launch (vpp_pixelwise_binary_batch): the frames' block ranges back to back, no drain between frames
constexpr int kPwBatchMax = 16;
struct PwBatch { u32x4* d[kPwBatchMax]; const u32x4* a[kPwBatchMax]; const u32x4* b[kPwBatchMax]; };
template <int OP, class T, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void binary_flat_batch_kernel(const PwBatch fr, size_t nvec, unsigned blocks_per_frame) {
  const unsigned f = blockIdx.x / blocks_per_frame, blk = blockIdx.x - f * blocks_per_frame;
  binary_flat_body<OP, T, UNROLL, NT>(fr.d[f], fr.a[f], fr.b[f], nvec, blk);
}

// Pitched rows: blockIdx.y = row, x covers the row's 16-byte vectors; row tail (< 16 B) done scalar by one lane.
template <int OP, class T>
__global__ __launch_bounds__(256) void binary_rows_kernel(DImg d, DImg a, DImg b, int row_bytes) {
  const int r = blockIdx.y;
  const int nvec = row_bytes >> 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint8_t* dr = d.p0 + (ptrdiff_t)r * d.pitch;
  const uint8_t* ar = a.p0 + (ptrdiff_t)r * a.pitch;
  const uint8_t* br = b.p0 + (ptrdiff_t)r * b.pitch;
  if (i < nvec) ((u32x4*)dr)[i] = op_vec<OP, T>(((const u32x4*)ar)[i], ((const u32x4*)br)[i]);
  if (i == nvec) {
    const int n0 = (nvec << 4) / (int)sizeof(T), n1 = row_bytes / (int)sizeof(T);
    for (int k = n0; k < n1; k++) ((T*)dr)[k] = op_scalar<OP, T>(((const T*)ar)[k], ((const T*)br)[k]);
  }
}

// Fully general fallback (unaligned pointers / pitches): one component per lane.
template <int OP, class T>
__global__ __launch_bounds__(256) void binary_scalar_kernel(DImg d, DImg a, DImg b, int ncomp) {
  const int r = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < ncomp) d.row<T>(r)[c] = op_scalar<OP, T>(a.row<T>(r)[c], b.row<T>(r)[c]);
}

template <int OP, class T>
int launch_binary(const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, hipStream_t st) {
  const int row_bytes = dst->ncols * elem_bytes(dst);
  const bool al = aligned16(dst) && aligned16(a) && aligned16(b);
  const bool flat = al && dst->pitch == row_bytes && a->pitch == row_bytes && b->pitch == row_bytes;
  if (flat) {
    const size_t total = (size_t)row_bytes * dst->nrows;
    const size_t nvec = total >> 4;
    const int unroll = tuning("add.unroll", 8);
    const int nt = tuning("add.nt", 1);
    auto go = [&](auto U, auto NTc) {
      constexpr int UN = decltype(U)::value; constexpr bool N = decltype(NTc)::value;
      const size_t per_block = 256 * UN;
      const unsigned blocks = (unsigned)((nvec + per_block - 1) / per_block);
      if (blocks) binary_flat_kernel<OP, T, UN, N><<<blocks, 256, 0, st>>>((u32x4*)dst->first_pixel, (const u32x4*)a->first_pixel, (const u32x4*)b->first_pixel, nvec);
    };
    auto pick = [&](auto NTc) {
      switch (unroll) {
        case 1: go(std::integral_constant<int, 1>(), NTc); break;
        case 2: go(std::integral_constant<int, 2>(), NTc); break;
        case 4: go(std::integral_constant<int, 4>(), NTc); break;
        default: go(std::integral_constant<int, 8>(), NTc); break;
      }
    };
    if (nt) pick(std::true_type()); else pick(std::false_type());
    const size_t tail = total - (nvec << 4);
    if (tail) {  // < 16 bytes: reuse the scalar kernel on a 1-row view
      DImg dd{(uint8_t*)dst->first_pixel + (nvec << 4), 1, 0, 0, 0, dst->dtype, 1}, aa = dd, bb = dd;
      aa.p0 = (uint8_t*)a->first_pixel + (nvec << 4); bb.p0 = (uint8_t*)b->first_pixel + (nvec << 4);
      binary_scalar_kernel<OP, T><<<dim3(1, 1), 256, 0, st>>>(dd, aa, bb, (int)(tail / sizeof(T)));
    }
  } else if (al) {
    const int nvec = row_bytes >> 4;
    dim3 grid((nvec + 1 + 255) / 256, dst->nrows);
    binary_rows_kernel<OP, T><<<grid, 256, 0, st>>>(dimg(dst), dimg(a), dimg(b), row_bytes);
  } else {
    const int ncomp = dst->ncols * dst->channels;
    dim3 grid((ncomp + 255) / 256, dst->nrows);
    binary_scalar_kernel<OP, T><<<grid, 256, 0, st>>>(dimg(dst), dimg(a), dimg(b), ncomp);
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

template <class T> int dispatch_op(int op, const vpp_image_desc* d, const vpp_image_desc* a, const vpp_image_desc* b, hipStream_t st) {
  switch (op) {
    case VPP_OP_ADD: return launch_binary<VPP_OP_ADD, T>(d, a, b, st);
    case VPP_OP_SUB: return launch_binary<VPP_OP_SUB, T>(d, a, b, st);
    case VPP_OP_MUL: return launch_binary<VPP_OP_MUL, T>(d, a, b, st);
    case VPP_OP_MIN: return launch_binary<VPP_OP_MIN, T>(d, a, b, st);
    case VPP_OP_MAX: return launch_binary<VPP_OP_MAX, T>(d, a, b, st);
    case VPP_OP_ABSDIFF: return launch_binary<VPP_OP_ABSDIFF, T>(d, a, b, st);
  }
  set_error("vpp_pixelwise_binary: unknown op %d", op);
  return VPP_ERR_INVALID_ARG;
}

// copy rows [r0, r1) x byte range [b0, b1) relative to first_pixel; 1 lane = 1..16 bytes.
__global__ __launch_bounds__(256) void copy_rows_kernel(DImg d, DImg s, int r0, int byte0, int nbytes, int vec_ok) {
  const int r = r0 + blockIdx.y;
  uint8_t* dr = d.p0 + (ptrdiff_t)r * d.pitch + byte0;
  const uint8_t* sr = s.p0 + (ptrdiff_t)r * s.pitch + byte0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (vec_ok) {
    const int nvec = nbytes >> 4;
    if (i < nvec) ((u32x4*)dr)[i] = ((const u32x4*)sr)[i];
    else if (i == nvec) for (int k = nvec << 4; k < nbytes; k++) dr[k] = sr[k];
  } else {
    for (int k = i * 16; k < min(nbytes, i * 16 + 16); k++) dr[k] = sr[k];
  }
}

template <int ES>
__global__ __launch_bounds__(256) void fill_kernel(DImg d, int r0, int c0, int ncols, const uint8_t* __restrict__ val) {
  const int r = r0 + blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  uint8_t* p = d.p0 + (ptrdiff_t)r * d.pitch + (ptrdiff_t)(c0 + c) * ES;
#pragma unroll
  for (int k = 0; k < ES; k++) p[k] = val[k];
}
struct FillVal { uint8_t b[16]; };
template <int ES>
__global__ __launch_bounds__(256) void fill_kernel_v(DImg d, int r0, int c0, int ncols, FillVal v) {
  const int r = r0 + blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  uint8_t* p = d.p0 + (ptrdiff_t)r * d.pitch + (ptrdiff_t)(c0 + c) * ES;
#pragma unroll
  for (int k = 0; k < ES; k++) p[k] = v.b[k];
}

}  // namespace

namespace {
// n triples of one size in one launch: the batched form of the flat 32-bit integer / float case (contiguous 16-byte aligned images — what
// imageNd::allocate gives a border-less image whose row is a multiple of the alignment); anything else goes out as n single calls.
template <int OP, class T> int launch_flat_batch(const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, int n, hipStream_t st) {
  const size_t total = (size_t)dst[0].ncols * elem_bytes(&dst[0]) * dst[0].nrows, nvec = total >> 4;
  constexpr int UN = 8;
  const unsigned bpf = (unsigned)((nvec + 256 * UN - 1) / (256 * UN));
  for (int b0 = 0; b0 < n; b0 += kPwBatchMax) {
    const int nb = std::min(kPwBatchMax, n - b0);
    PwBatch fr{};
    for (int k = 0; k < nb; k++) { fr.d[k] = (u32x4*)dst[b0 + k].first_pixel; fr.a[k] = (const u32x4*)a[b0 + k].first_pixel; fr.b[k] = (const u32x4*)b[b0 + k].first_pixel; }
    binary_flat_batch_kernel<OP, T, UN, true><<<bpf * nb, 256, 0, st>>>(fr, nvec, bpf);
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
}  // namespace

// ---- the per-call form on a recorded stream / without its per-call launch (see box.hip and common.hpp: the same window for `A = B + C` on flat 32-bit images) ----
namespace {
inline bool pw_batchable(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b) {   // the frames launch_flat_batch serves
  if (!((op == VPP_OP_ADD || op == VPP_OP_SUB) && (dst->dtype == VPP_I32 || dst->dtype == VPP_U32) && tuning("add.batch", 1) && tuning("add.coalesce", 1) &&
        tuning("add.unroll", 8) == 8 && tuning("add.nt", 1) == 1)) return false;
  const vpp_image_desc* t[3] = {dst, a, b};
  for (const vpp_image_desc* d : t) {
    const int row_bytes = d->ncols * elem_bytes(d);
    if (!(aligned16(d) && d->pitch == row_bytes && ((size_t)row_bytes * d->nrows) % 16 == 0)) return false;
  }
  return true;
}
}  // namespace

extern "C" {

int vpp_pixelwise_binary(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(a) && valid_desc(b), VPP_ERR_INVALID_ARG, "vpp_pixelwise_binary: invalid descriptor");
  VPP_REQUIRE(same_domain(dst, a) && same_domain(dst, b), VPP_ERR_INVALID_ARG, "vpp_pixelwise_binary: domains differ");
  VPP_REQUIRE(same_type(dst, a) && same_type(dst, b), VPP_ERR_UNSUPPORTED, "vpp_pixelwise_binary: mixed element types");
  // on a stream this thread records through vpp_graph_begin: the triple is held back and recorded as part of ONE batched node (box.hip, common.hpp)
  if (!g_defer_bypass && defer_recording(stream) && pw_batchable(op, dst, a, b) && dst->first_pixel != a->first_pixel && dst->first_pixel != b->first_pixel)
    return defer_call(kDeferBinary, op, 0, stream, dst, a, b);
  hipStream_t st = as_stream(stream);
  // while the stream is recorded into a launch graph: calls on unrelated images become sibling nodes (common.hpp, IndependentCall)
  const Extent wr = extent_of(*dst), rd[2] = {extent_of(*a), extent_of(*b)};
  IndependentCall side_by_side(st, &wr, 1, rd, 2);
  switch (dst->dtype) {
    case VPP_U8: return dispatch_op<uint8_t>(op, dst, a, b, st);
    case VPP_I8: return dispatch_op<int8_t>(op, dst, a, b, st);
    case VPP_U16: return dispatch_op<uint16_t>(op, dst, a, b, st);
    case VPP_I16: return dispatch_op<int16_t>(op, dst, a, b, st);
    case VPP_I32: return dispatch_op<int32_t>(op, dst, a, b, st);
    case VPP_U32: return dispatch_op<uint32_t>(op, dst, a, b, st);
    case VPP_F32: return dispatch_op<float>(op, dst, a, b, st);
  }
  return VPP_ERR_UNSUPPORTED;
}

int vpp_pixelwise_binary_batch(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, int n, void* stream) {
  VPP_REQUIRE(n >= 0 && (n == 0 || (dst && a && b)), VPP_ERR_INVALID_ARG, "vpp_pixelwise_binary_batch: invalid argument");
  if (n == 0) return VPP_OK;
  bool flat = n > 1 && tuning("add.batch", 1) && op >= VPP_OP_ADD && op <= VPP_OP_ABSDIFF;
  for (int k = 0; k < n && flat; k++) {
    const vpp_image_desc* t[3] = {&dst[k], &a[k], &b[k]};
    for (const vpp_image_desc* d : t) {
      const int row_bytes = d->ncols * elem_bytes(d);
      flat = flat && valid_desc(d) && same_domain(d, &dst[0]) && same_type(d, &dst[0]) && aligned16(d) && d->pitch == row_bytes && ((size_t)row_bytes * d->nrows) % 16 == 0;
    }
  }
  flat = flat && (dst[0].dtype == VPP_I32 || dst[0].dtype == VPP_U32);
  // one launch only when no triple's result is another triple's operand or result (else: the n calls in sequence, whose results are the contract)
  if (flat) { const vpp_image_desc* srcs[2] = {a, b}; flat = !batch_frames_interfere(n, dst, srcs, 2); }
  if (flat) {
    hipStream_t st = as_stream(stream);
    switch (op) {
      case VPP_OP_ADD: return launch_flat_batch<VPP_OP_ADD, int32_t>(dst, a, b, n, st);
      case VPP_OP_SUB: return launch_flat_batch<VPP_OP_SUB, int32_t>(dst, a, b, n, st);
    }
  }
  for (int k = 0; k < n; k++) { const int rc = vpp_pixelwise_binary(op, &dst[k], &a[k], &b[k], stream); if (rc) return rc; }
  return VPP_OK;
}

// The per-call form without its per-call launch (common.hpp, "deferred per-frame calls"): the triple joins the calling thread's window; argument errors are
// reported here, at the call, as vpp_pixelwise_binary reports them.  Triples the batched kernel does not serve go out at once — behind the window, which
// as_stream() launches first.
int vpp_pixelwise_binary_deferred(int op, const vpp_image_desc* dst, const vpp_image_desc* a, const vpp_image_desc* b, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(a) && valid_desc(b), VPP_ERR_INVALID_ARG, "vpp_pixelwise_binary: invalid descriptor");
  VPP_REQUIRE(same_domain(dst, a) && same_domain(dst, b), VPP_ERR_INVALID_ARG, "vpp_pixelwise_binary: domains differ");
  VPP_REQUIRE(same_type(dst, a) && same_type(dst, b), VPP_ERR_UNSUPPORTED, "vpp_pixelwise_binary: mixed element types");
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if ((cap != hipStreamCaptureStatusNone && !defer_recording(stream)) || g_defer_bypass || !tuning("defer", 1) || !pw_batchable(op, dst, a, b) || dst->first_pixel == a->first_pixel ||
      dst->first_pixel == b->first_pixel)
    return vpp_pixelwise_binary(op, dst, a, b, stream);
  return defer_call(kDeferBinary, op, 0, stream, dst, a, b);
}

int vpp_copy(const vpp_image_desc* dst, const vpp_image_desc* src, int with_border, void* stream) {
  VPP_REQUIRE(valid_desc(dst) && valid_desc(src), VPP_ERR_INVALID_ARG, "vpp_copy: invalid descriptor");
  VPP_REQUIRE(same_domain(dst, src) && same_type(dst, src), VPP_ERR_INVALID_ARG, "vpp_copy: domain/type mismatch");
  const int b = with_border ? src->border : 0;
  VPP_REQUIRE(dst->border >= b, VPP_ERR_BORDER_TOO_SMALL, "vpp_copy: dst border %d < src border %d (copy.hh:25)", dst->border, b);
  const int es = elem_bytes(dst);
  const int byte0 = -b * es, nbytes = (dst->ncols + 2 * b) * es;
  const bool vec_ok = ((uintptr_t)((uint8_t*)dst->first_pixel + byte0) % 16 == 0) && ((uintptr_t)((uint8_t*)src->first_pixel + byte0) % 16 == 0) &&
                      dst->pitch % 16 == 0 && src->pitch % 16 == 0;
  dim3 grid(((nbytes + 15) / 16 + 1 + 255) / 256, dst->nrows + 2 * b);
  copy_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(dimg(dst), dimg(src), -b, byte0, nbytes, vec_ok ? 1 : 0);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_fill(const vpp_image_desc* img, const void* value, int with_border, void* stream) {
  VPP_REQUIRE(valid_desc(img) && value, VPP_ERR_INVALID_ARG, "vpp_fill: invalid argument");
  const int es = elem_bytes(img);
  VPP_REQUIRE(es <= 16, VPP_ERR_UNSUPPORTED, "vpp_fill: element of %d bytes", es);
  const int b = with_border ? img->border : 0;
  FillVal v; memcpy(v.b, value, es);
  dim3 grid((img->ncols + 2 * b + 255) / 256, img->nrows + 2 * b);
  hipStream_t st = as_stream(stream);
  DImg d = dimg(img);
#define VPP_FILL_CASE(ES) case ES: fill_kernel_v<ES><<<grid, 256, 0, st>>>(d, -b, -b, img->ncols + 2 * b, v); break;
  switch (es) {
    VPP_FILL_CASE(1) VPP_FILL_CASE(2) VPP_FILL_CASE(3) VPP_FILL_CASE(4) VPP_FILL_CASE(6) VPP_FILL_CASE(8) VPP_FILL_CASE(12) VPP_FILL_CASE(16)
    default: set_error("vpp_fill: unsupported element size %d", es); return VPP_ERR_UNSUPPORTED;
  }
#undef VPP_FILL_CASE
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

}  // extern "C"
