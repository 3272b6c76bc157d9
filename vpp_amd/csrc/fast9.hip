// fast9.hip — K7-K9: FAST-9 detector, score and non-maximum suppression.
// Reference: vpp/algorithms/fast_detector/fast.hpp:254-508 (fast_detector9_simd), :38-77 (fast9_score),
// :676-707 (fast_detector9_maxima), :745-799 (blockwise maxima), :889-928 (local maxima), :931-955 (front end fast9).
//
// Pipeline (all on one stream, no host round trip until the final count):
//   1. fast9_detect_kernel   LDS tile (64x64 px + 3 px halo, dword loads), one lane per pixel column marching down 16 rows.
//                            Segment test per lane: saturated thresholds, 4 cardinal samples first (any 9-arc holds >= 2
//                            of ring indices {0,4,8,12}), then 16-bit brighter/darker masks and a shift-and "9 circularly
//                            contiguous" test.  corner = mask_byte & (0x10*B9 | 0x01*D9) != 0 (fast.hpp:120-126,312,333).
//                            Corner lanes then evaluate fast9_score on the TRUE ring (fast.hpp:52-74) from the same LDS tile
//                            and every lane writes the dense u16 map F(p) = score + 1 (0 = not a corner) — the reference's
//                            scores_img (:685-694) without the intermediate keypoint list (a single global append counter
//                            saturates at ~90 atomics/us, which measured 790 us on a 4K frame; the dense write costs 16.6 MB).
//   2. count / scan / write  per image row (RAW, LOCAL_MAXIMA) or per row of bs x bs blocks (BLOCKWISE): decisions are
//                            recomputed in the write pass so the output is in the serial reference order (row-major
//                            pixels / row-major blocks), which the OpenMP reference itself does not guarantee.
// VALU-bound (the ring test is ~100 lane-ops per pixel against 1 B/px of HBM traffic); see DESIGN.md.
#include "common.hpp"
#include <mutex>
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

constexpr int TW = 64, TH = 64, HALO = 3;
constexpr int LP = 72;                       // LDS row pitch: cols [c0-4, c0+68)
constexpr int LROWS = TH + 2 * HALO;         // 70
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

// shift a comparison result into a ring mask without a compare: (m << 1) | (d < 0), one v_alignbit_b32
__device__ __forceinline__ uint32_t push_sign(uint32_t m, int d) { return __builtin_amdgcn_alignbit(m, (uint32_t)d, 31); }

template <bool REF>
__global__ __launch_bounds__(256) void fast9_detect_kernel(DImg A, DImg M, int has_mask, int th, DImg F) {
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
    F.row<uint16_t>(r)[c] = (uint16_t)f;
  }
}

__device__ __forceinline__ int fast9_score_px(const DImg& A, int r, int c, int th) {  // fast.hpp:38-77
  const int v = A.row<uint8_t>(r)[c];
  int sum_inf = 0, sum_sup = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int a = A.row<uint8_t>(r + ring_dr<false>(i))[c + ring_dc(i)];
    const int diff = v - a;
    if (diff < -th) sum_inf -= diff;
    else if (diff > th) sum_sup += diff;
  }
  return max(sum_sup, sum_inf);
}

__global__ __launch_bounds__(256) void fast9_scores_list_kernel(DImg A, int th, const int32_t* __restrict__ rc, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = fast9_score_px(A, rc[2 * i], rc[2 * i + 1], th);  // fast.hpp:643-652
}

// ---- ordered selection -----------------------------------------------------------------------------------
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

// MODE 0 RAW, 1 LOCAL_MAXIMA: one workgroup per image row, thread t owns columns [t*K, t*K+K).
template <int MODE, bool WRITE>
__global__ __launch_bounds__(256) void fast9_select_rows_kernel(DImg F, int K, uint32_t* __restrict__ unit_count,
                                                                const uint32_t* __restrict__ unit_off, int32_t* __restrict__ out_rc,
                                                                int32_t* __restrict__ out_scores, int capacity) {
  const int r = blockIdx.x;
  const uint16_t* f0 = F.row<uint16_t>(r);
  const int cbeg = threadIdx.x * K, cend = min(cbeg + K, F.nc);
  auto decide = [&](int c, uint32_t& score) -> bool {
    const uint32_t f = f0[c];
    if (!f) return false;
    if (MODE == 0) { score = f - 1; return true; }
    const uint16_t *fm = F.row<uint16_t>(r - 1), *fp = F.row<uint16_t>(r + 1);
    const uint32_t a = stored(f);
    int is_max = 1;  // fast.hpp:907-921, strict >
    is_max &= a > stored(fm[c - 1]); is_max &= a > stored(fm[c]); is_max &= a > stored(fm[c + 1]);
    is_max &= a > stored(f0[c - 1]); is_max &= a > stored(f0[c + 1]);
    is_max &= a > stored(fp[c - 1]); is_max &= a > stored(fp[c]); is_max &= a > stored(fp[c + 1]);
    score = a;
    return is_max != 0;
  };
  uint32_t cnt = 0, sc;
  for (int c = cbeg; c < cend; c++) cnt += decide(c, sc) ? 1u : 0u;
  uint32_t tot;
  const uint32_t ex = block_exscan(cnt, &tot);
  if (!WRITE) { if (threadIdx.x == 0) unit_count[r] = tot; return; }
  uint32_t k = unit_off[r] + ex;
  for (int c = cbeg; c < cend; c++)
    if (decide(c, sc)) {
      if ((int)k < capacity) { out_rc[2 * k] = r; out_rc[2 * k + 1] = c; if (out_scores) out_scores[k] = (int32_t)sc; }
      k++;
    }
}

// BLOCKWISE: one workgroup per row of bs x bs blocks; thread t owns block columns [t*K, t*K+K).
template <bool WRITE>
__global__ __launch_bounds__(256) void fast9_select_blocks_kernel(DImg F, int bs, int nbc, int K, uint32_t* __restrict__ unit_count,
                                                                  const uint32_t* __restrict__ unit_off, int32_t* __restrict__ out_rc,
                                                                  int32_t* __restrict__ out_scores, int capacity) {
  const int r = blockIdx.x * bs;
  const int bbeg = threadIdx.x * K, bend = min(bbeg + K, nbc);
  auto decide = [&](int b, int& pr, int& pc, uint32_t& vmax) -> bool {  // fast.hpp:770-789: first strict max in scan order
    const int c = b * bs;
    vmax = 0; pr = 0; pc = 0;
    for (int br = 0; br < bs; br++) {
      if (r + br >= F.nr) break;
      const uint16_t* f = F.row<uint16_t>(r + br);
      for (int bc = c; bc < c + bs && bc < F.nc; bc++) {
        const uint32_t v = stored(f[bc]);
        if (v > vmax) { vmax = v; pr = br; pc = bc; }
      }
    }
    return vmax > 0;
  };
  uint32_t cnt = 0, vm; int pr, pc;
  for (int b = bbeg; b < bend; b++) cnt += decide(b, pr, pc, vm) ? 1u : 0u;
  uint32_t tot;
  const uint32_t ex = block_exscan(cnt, &tot);
  if (!WRITE) { if (threadIdx.x == 0) unit_count[blockIdx.x] = tot; return; }
  uint32_t k = unit_off[blockIdx.x] + ex;
  for (int b = bbeg; b < bend; b++)
    if (decide(b, pr, pc, vm)) {
      if ((int)k < capacity) { out_rc[2 * k] = r + pr; out_rc[2 * k + 1] = pc; if (out_scores) out_scores[k] = (int32_t)vm; }
      k++;
    }
}

// exclusive scan of n unit counts (single workgroup), total -> *total
__global__ __launch_bounds__(256) void scan_units_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off, int n, uint32_t* __restrict__ total) {
  const int per = (n + 255) / 256;
  const int b = threadIdx.x * per, e = min(b + per, n);
  uint32_t s = 0;
  for (int i = b; i < e; i++) s += cnt[i];
  uint32_t tot;
  uint32_t ex = block_exscan(s, &tot);
  for (int i = b; i < e; i++) { off[i] = ex; ex += cnt[i]; }
  if (threadIdx.x == 0) *total = tot;
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

int vpp_fast9_detect(const vpp_image_desc* src, int th, const vpp_image_desc* mask, int mode, int block_size, int compat,
                     int32_t* out_rc, int32_t* out_scores, int capacity, int* count, void* stream) {
  VPP_REQUIRE(valid_desc(src) && count && capacity >= 0 && (out_rc || capacity == 0), VPP_ERR_INVALID_ARG, "vpp_fast9_detect: invalid argument");
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
  // scratch layout: [counters: total][F: u16 map with border 1][unit_count][unit_off]
  int32_t fpitch; size_t fbytes, ffirst;
  vpp_image_layout(nr, nc, 2, 1, 16, &fpitch, &fbytes, &ffirst);
  const int nunits = mode == VPP_FAST9_BLOCKWISE ? (nr + block_size - 1) / block_size : nr;
  const size_t off_f = 256, off_uc = off_f + align_up(fbytes, 256);
  const size_t off_uo = off_uc + align_up((size_t)nunits * 4, 256), total_bytes = off_uo + align_up((size_t)nunits * 4, 256);
  int rc = g_scratch.ensure(total_bytes, st);
  if (rc != VPP_OK) return rc;
  uint8_t* base = (uint8_t*)g_scratch.p;
  uint32_t* counters = (uint32_t*)base;
  DImg F{base + off_f + ffirst, nr, nc, fpitch, 1, VPP_U16, 1};
  uint32_t* unit_count = (uint32_t*)(base + off_uc);
  uint32_t* unit_off = (uint32_t*)(base + off_uo);
  // zero the counters and F's 1-px border (the detect kernel writes every domain pixel): first / last row + the strips between rows
  VPP_HIP_TRY(hipMemsetAsync(base, 0, off_f + (size_t)fpitch + ffirst, st));
  VPP_HIP_TRY(hipMemsetAsync(base + off_f + (size_t)(nr + 1) * fpitch, 0, fpitch, st));
  VPP_HIP_TRY(hipMemset2DAsync(base + off_f + ffirst + (size_t)nc * 2, fpitch, 0, (size_t)fpitch - (size_t)nc * 2, (size_t)nr, st));
  DImg A = dimg(src), M = mask ? dimg(mask) : A;
  dim3 grid((nc + TW - 1) / TW, (nr + TH - 1) / TH);
  if (compat == VPP_FAST9_REFERENCE) fast9_detect_kernel<true><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F);
  else fast9_detect_kernel<false><<<grid, 256, 0, st>>>(A, M, mask ? 1 : 0, th, F);
  VPP_LAUNCH_CHECK();
  if (mode == VPP_FAST9_BLOCKWISE) {
    const int nbc = (nc + block_size - 1) / block_size, K = (nbc + 255) / 256;
    fast9_select_blocks_kernel<false><<<nunits, 256, 0, st>>>(F, block_size, nbc, K, unit_count, unit_off, out_rc, out_scores, capacity);
    scan_units_kernel<<<1, 256, 0, st>>>(unit_count, unit_off, nunits, counters + 1);
    fast9_select_blocks_kernel<true><<<nunits, 256, 0, st>>>(F, block_size, nbc, K, unit_count, unit_off, out_rc, out_scores, capacity);
  } else {
    const int K = (nc + 255) / 256;
    if (mode == VPP_FAST9_RAW) {
      fast9_select_rows_kernel<0, false><<<nunits, 256, 0, st>>>(F, K, unit_count, unit_off, out_rc, out_scores, capacity);
      scan_units_kernel<<<1, 256, 0, st>>>(unit_count, unit_off, nunits, counters + 1);
      fast9_select_rows_kernel<0, true><<<nunits, 256, 0, st>>>(F, K, unit_count, unit_off, out_rc, out_scores, capacity);
    } else {
      fast9_select_rows_kernel<1, false><<<nunits, 256, 0, st>>>(F, K, unit_count, unit_off, out_rc, out_scores, capacity);
      scan_units_kernel<<<1, 256, 0, st>>>(unit_count, unit_off, nunits, counters + 1);
      fast9_select_rows_kernel<1, true><<<nunits, 256, 0, st>>>(F, K, unit_count, unit_off, out_rc, out_scores, capacity);
    }
  }
  VPP_LAUNCH_CHECK();
  uint32_t total = 0;
  VPP_HIP_TRY(hipMemcpyAsync(&total, counters + 1, sizeof total, hipMemcpyDeviceToHost, st));
  VPP_HIP_TRY(hipStreamSynchronize(st));
  *count = (int)total;
  if ((int)total > capacity) {
    set_error("vpp_fast9_detect: %u keypoints found, output capacity %d", total, capacity);
    return VPP_ERR_CAPACITY;
  }
  return VPP_OK;
}

}  // extern "C"
