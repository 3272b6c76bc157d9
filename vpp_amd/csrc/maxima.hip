// maxima.hip — local_maxima_filter (reference: vpp/algorithms/fast_detector/fast.hpp:555-575), in place on a scalar image.
// The reference maps a lambda over (A, relative_access(A)) IN PLACE: a pixel survives iff it is strictly greater than its eight
// neighbours, of which the four above / left of it (raster order) have already been filtered.  Serial order is the defined result
// (the reference's test build has no OpenMP; its OpenMP build races on those four neighbours, SURVEY Q-list) and it is what this
// kernel reproduces, without walking the image serially:
//   for pixel p with value v: a neighbour q that is LATER in raster order (E, SW, S, SE) or in the border still holds its input
//   a(q): p needs v > a(q).  An EARLIER neighbour q (NW, N, NE, W) holds a(q) if q survived and 0 if it was zeroed: p needs
//   v > a(q) in the first case and v > 0 in the second.  If both hold (or both fail) q's fate does not matter; otherwise p waits
//   for q.  Pass 1 decides every pixel that needs no earlier neighbour's fate (on score images: almost all) and lists the rest;
//   the list is then resolved in rounds — a pixel whose earlier neighbours are all decided is decided — by one workgroup when the
//   list is short (each round a barrier, not a launch), by grid-wide rounds otherwise.  Raster order is a DAG, so every round
//   decides at least the first listed pixel.  The input is not modified until every pixel is decided (pass 3 writes the zeros).
#include "common.hpp"
using namespace vpp_amd;

namespace {
enum : uint8_t { kUnknown = 0, kKept = 1, kZeroed = 2 };

template <class V> struct Img2 {
  uint8_t* p0; int pitch, nr, nc;
  __device__ __forceinline__ V at(int r, int c) const { return ((const V*)(p0 + (ptrdiff_t)r * pitch))[c]; }
};

// fate of pixel (r, c) given what is known of its earlier neighbours; kUnknown if it has to wait
template <class V, class ST> __device__ __forceinline__ uint8_t decide(const Img2<V>& a, int r, int c, ST state_of) {
  const V v = a.at(r, c);
  // later neighbours and everything outside the domain: input values
  const int lr[4] = {0, 1, 1, 1}, lc[4] = {1, -1, 0, 1};
#pragma unroll
  for (int k = 0; k < 4; k++) if (!(v > a.at(r + lr[k], c + lc[k]))) return kZeroed;
  const int er[4] = {-1, -1, -1, 0}, ec[4] = {-1, 0, 1, -1};
  bool wait = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int qr = r + er[k], qc = c + ec[k];
    const bool gt_input = v > a.at(qr, qc);
    if (qr < 0 || qc < 0 || qc >= a.nc) { if (!gt_input) return kZeroed; continue; }   // border: never filtered
    const bool gt_zero = v > V(0);
    if (gt_input && gt_zero) continue;
    if (!gt_input && !gt_zero) return kZeroed;
    const uint8_t s = state_of(qr, qc);
    if (s == kUnknown) { wait = true; continue; }
    if (!(s == kKept ? gt_input : gt_zero)) return kZeroed;
  }
  return wait ? kUnknown : kKept;
}

template <class V>
__global__ __launch_bounds__(256) void lm_classify_kernel(Img2<V> a, uint8_t* __restrict__ state, int32_t* __restrict__ list, int32_t* __restrict__ count) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= a.nc) return;
  // pass 1 sees no neighbour's fate: anything that would need one is listed
  const uint8_t s = decide<V>(a, r, c, [](int, int) { return (uint8_t)kUnknown; });
  state[(size_t)r * a.nc + c] = s;
  if (s == kUnknown) list[atomicAdd(count, 1)] = r * a.nc + c;
}

// one round over the list: decided pixels publish their state, the others go to the next list
template <class V>
__global__ __launch_bounds__(256) void lm_round_kernel(Img2<V> a, uint8_t* state, const int32_t* __restrict__ list, int n, int32_t* __restrict__ next, int32_t* __restrict__ next_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int idx = list[i], r = idx / a.nc, c = idx - r * a.nc;
  // states written in THIS round by other lanes may or may not be visible: either way the value read is a valid (possibly older) state
  const uint8_t s = decide<V>(a, r, c, [&](int qr, int qc) { return __hip_atomic_load(&state[(size_t)qr * a.nc + qc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
  if (s == kUnknown) next[atomicAdd(next_count, 1)] = idx;
  else __hip_atomic_store(&state[(size_t)r * a.nc + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the whole list in one workgroup: rounds separated by barriers until nothing is left
template <class V>
__global__ __launch_bounds__(1024) void lm_resolve_kernel(Img2<V> a, uint8_t* state, int32_t* list, int n) {
  __shared__ int remaining;
  for (;;) {
    if (threadIdx.x == 0) remaining = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
      const int idx = list[i];
      if (idx < 0) continue;
      const int r = idx / a.nc, c = idx - r * a.nc;
      const uint8_t s = decide<V>(a, r, c, [&](int qr, int qc) { return __hip_atomic_load(&state[(size_t)qr * a.nc + qc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
      if (s == kUnknown) atomicAdd(&remaining, 1);
      else { __hip_atomic_store(&state[(size_t)r * a.nc + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); list[i] = -1; }
    }
    __threadfence();
    __syncthreads();
    if (remaining == 0) break;
    __syncthreads();
  }
}

template <class V>
__global__ __launch_bounds__(256) void lm_apply_kernel(Img2<V> a, const uint8_t* __restrict__ state) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= a.nc) return;
  if (state[(size_t)r * a.nc + c] == kZeroed) ((V*)(a.p0 + (ptrdiff_t)r * a.pitch))[c] = V(0);
}

thread_local Scratch g_scratch;

template <class V> int run(const vpp_image_desc* img, hipStream_t st) {
  const size_t npx = (size_t)img->nrows * img->ncols;
  VPP_REQUIRE(npx < (size_t)1 << 31, VPP_ERR_UNSUPPORTED, "vpp_local_maxima_filter: image too large");
  // scratch: state bytes | two lists of pixel indices | two counters
  const size_t off_list0 = (npx + 255) / 256 * 256, off_list1 = off_list0 + npx * 4, off_cnt = off_list1 + npx * 4;
  int rc = g_scratch.ensure(off_cnt + 256, st);
  if (rc != VPP_OK) return rc;
  uint8_t* base = (uint8_t*)g_scratch.p;
  uint8_t* state = base;
  int32_t *list[2] = {(int32_t*)(base + off_list0), (int32_t*)(base + off_list1)}, *cnt = (int32_t*)(base + off_cnt);
  Img2<V> a{(uint8_t*)img->first_pixel, img->pitch, img->nrows, img->ncols};
  { const int rf = device_fill(cnt, 0, 8, st); if (rf != VPP_OK) return rf; }
  dim3 grid((img->ncols + 255) / 256, img->nrows);
  lm_classify_kernel<V><<<grid, 256, 0, st>>>(a, state, list[0], cnt);
  int n = 0, cur = 0;
  VPP_HIP_TRY(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, st));
  VPP_HIP_TRY(hipStreamSynchronize(st));
  while (n > 0) {
    if (n <= 65536) { lm_resolve_kernel<V><<<1, 1024, 0, st>>>(a, state, list[cur], n); break; }
    { const int rf = device_fill(cnt + (1 - cur), 0, 4, st); if (rf != VPP_OK) return rf; }
    lm_round_kernel<V><<<(n + 255) / 256, 256, 0, st>>>(a, state, list[cur], n, list[1 - cur], cnt + (1 - cur));
    cur = 1 - cur;
    VPP_HIP_TRY(hipMemcpyAsync(&n, cnt + cur, 4, hipMemcpyDeviceToHost, st));
    VPP_HIP_TRY(hipStreamSynchronize(st));
  }
  lm_apply_kernel<V><<<grid, 256, 0, st>>>(a, state);
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}
}  // namespace

extern "C" int vpp_local_maxima_filter(const vpp_image_desc* img, void* stream) {
  VPP_REQUIRE(valid_desc(img) && img->channels == 1, VPP_ERR_INVALID_ARG, "vpp_local_maxima_filter: a scalar image is required");
  VPP_REQUIRE(img->border >= 1, VPP_ERR_BORDER_TOO_SMALL, "vpp_local_maxima_filter: the image needs border >= 1 (fast.hpp:558-571 reads the 8 neighbours)");
  hipStream_t st = as_stream(stream);
  switch (img->dtype) {
    case VPP_U8: return run<uint8_t>(img, st);
    case VPP_I8: return run<int8_t>(img, st);
    case VPP_U16: return run<uint16_t>(img, st);
    case VPP_I16: return run<int16_t>(img, st);
    case VPP_I32: return run<int32_t>(img, st);
    case VPP_U32: return run<uint32_t>(img, st);
    case VPP_F32: return run<float>(img, st);
  }
  return VPP_ERR_UNSUPPORTED;
}
