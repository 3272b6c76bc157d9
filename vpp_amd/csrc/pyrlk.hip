// pyrlk.hip — K10: pyramidal Lucas-Kanade.
// Reference: vpp/algorithms/pyrlk/pyrlk_match.hh:15-55 (coarse-to-fine driver),
//            vpp/algorithms/pyrlk/lk.hh:43-175 (lk_match_point_square_win<WS>::operator()),
//            vpp/algorithms/lucas_kanade/lucas_kanade.hpp:12-131,159-183 (lk_internals::match + serial driver),
//            vpp/core/imageNd.hpp:280-300 (linear_interpolate, result cast back to the pixel type),
//            vpp/core/keypoint_container.hpp:136-167 (move / remove).
//
// Float parity: every sum runs over the window in the reference's row-major order with the reference's operation
// order, compiled with -ffp-contract=off; IEEE add/mul/div/sqrt are correctly rounded on gfx950 (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), so displacements are bit-identical to the serial oracle except for the
// min-eigenvalue gate (closed form instead of Eigen's EigenSolver; only a threshold compare).
//
// Mapping: LPK lanes per keypoint (1, 8 or 16, chosen from the keypoint count), all pyramid levels in one launch
// (keypoints are independent, pyrlk_match.hh:24).  With LPK > 1 each lane owns every LPK-th window offset: it does the
// bilinear sampling for its offsets and stages the per-offset terms in LDS; every lane of the group then accumulates
// ALL terms in the reference's row-major order, so each sum is the same left-to-right float chain as the serial code —
// bit-identical results at ~1/LPK of the per-keypoint latency (10 k keypoints are only 157 waves at LPK = 1).
// The window cache gs[]/as[] of lk.hh:90-112 lives in registers (or scratch for large windows at LPK = 1).
// Not bandwidth bound: ~1 KB touched per keypoint-level, L2-resident; the metric is tracks/s.
#include "common.hpp"
#include <cfloat>
#include <vector>
using namespace vpp_amd;

namespace {

constexpr int kMaxLevels = 8;
struct Pyr { DImg l[kMaxLevels]; };

// imageNd::linear_interpolate (imageNd.hpp:280-300).  Reads (x, x+row, x+col, x+row+col) with no bounds check in the
// reference; SAFE=false clamps the four taps into the bordered area so that a wandering estimate can never fault
// (identical values whenever the reference's own read was inside the border).
template <class T, int CH, bool SAFE>
__device__ __forceinline__ void interp(const DImg& I, float p0, float p1, T* out) {
  const int x0 = (int)p0, x1 = (int)p1;
  const float a0 = p0 - x0, a1 = p1 - x1;
  int r0 = x0, r1 = x0 + 1, c0 = x1, c1 = x1 + 1;
  if (!SAFE) {
    const int rlo = -I.border, rhi = I.nr + I.border - 1, clo = -I.border, chi = I.nc + I.border - 1;
    r0 = min(max(r0, rlo), rhi); r1 = min(max(r1, rlo), rhi); c0 = min(max(c0, clo), chi); c1 = min(max(c1, clo), chi);
  }
  // Unsigned 32-bit byte offsets from the start of the bordered area (wave-uniform base, every tap of either form lies at or
  // after it): a load is `global_load v, v_offset, s[base]` with no 64-bit address arithmetic per tap, and the row product is
  // one 24-bit multiply-add (rows < 2^23, pitch < 2^23).
  const int es = (int)sizeof(T) * CH;
  const uint8_t* base = I.p0 - ((ptrdiff_t)I.border * I.pitch + (ptrdiff_t)I.border * es);
  const uint32_t o00 = (uint32_t)(__mul24(r0 + I.border, I.pitch) + (c0 + I.border) * es);
  // SAFE: r1 = r0 + 1 and c1 = c0 + 1 exactly — one add each instead of a second multiply-add (which the compiler cannot fold: 24-bit wrap)
  const uint32_t o10 = SAFE ? o00 + (uint32_t)I.pitch : (uint32_t)(__mul24(r1 + I.border, I.pitch) + (c0 + I.border) * es);
  const uint32_t o01 = SAFE ? o00 + (uint32_t)es : (uint32_t)(__mul24(r0 + I.border, I.pitch) + (c1 + I.border) * es);
  const uint32_t o11 = SAFE ? o10 + (uint32_t)es : (uint32_t)(__mul24(r1 + I.border, I.pitch) + (c1 + I.border) * es);
  const float w00 = (1 - a0) * (1 - a1), w10 = a0 * (1 - a1), w01 = (1 - a0) * a1, w11 = a0 * a1;
  if constexpr (SAFE && sizeof(T) == 1 && CH == 1) {
    // the two taps of a row are adjacent bytes: one (unaligned) 16-bit load per row instead of two byte loads — the kernel
    // issues ~2700 tap loads per wave and is bound by how fast the texture path takes them, not by bytes
    uint16_t t0, t1;
    __builtin_memcpy(&t0, base + o00, 2);
    __builtin_memcpy(&t1, base + o10, 2);
    const float v = w00 * (float)(T)(t0 & 255) + w10 * (float)(T)(t1 & 255) + w01 * (float)(T)(t0 >> 8) + w11 * (float)(T)(t1 >> 8);
    out[0] = (T)v;
    return;
  }
#pragma unroll
  for (int k = 0; k < CH; k++) {
    const float v = w00 * (float)((const T*)(base + o00))[k] + w10 * (float)((const T*)(base + o10))[k] + w01 * (float)((const T*)(base + o01))[k] +
                    w11 * (float)((const T*)(base + o11))[k];
    out[k] = (T)v;  // vpp::cast<V>: truncation for integer V
  }
}

// true when every tap of a WS x WS window of bilinear reads centred on (v0,v1) stays inside the bordered area
__device__ __forceinline__ bool window_inside(const DImg& I, float v0, float v1, int hws) {
  const float m = (float)(hws + 2);
  return v0 - m >= (float)(-I.border) && v0 + m <= (float)(I.nr + I.border - 1) && v1 - m >= (float)(-I.border) && v1 + m <= (float)(I.nc + I.border - 1);
}

struct Match { float f0, f1, err; };

template <int WS, class GT, bool PYRLK>
__device__ Match lk_match(float p0, float p1, float tr0, float tr1, const DImg& A, const DImg& B, const DImg& Ag, float min_ev_th,
                          int max_it, float delta) {
  constexpr int hws = WS / 2, N = WS * WS;
  const bool a_safe = window_inside(A, p0, p1, hws);  // A and Ag share the domain and (by contract) the border
  float G00 = 0, G01 = 0, G10 = 0, G11 = 0;
  int cpt = 0;
  float gs0[N], gs1[N];
  int as[N];
  // lk.hh:56-72 (structure tensor) and :99-112 (cache) visit the same offsets with the same predicate (A and Ag share
  // their domain), so one pass fills both; unset cache entries are 0 (canonical value).
#pragma unroll(N <= 49 ? N : 1)
  for (int i = 0; i < N; i++) {
    const int r = i / WS - hws, c = i % WS - hws;
    const float n0 = p0 + (float)r, n1 = p1 + (float)c;
    gs0[i] = 0.f; gs1[i] = 0.f; as[i] = 0;
    if (A.has((int)n0, (int)n1)) {
      GT g[2]; uint8_t a;
      if (a_safe) { interp<GT, 2, true>(Ag, n0, n1, g); interp<uint8_t, 1, true>(A, n0, n1, &a); }
      else { interp<GT, 2, false>(Ag, n0, n1, g); interp<uint8_t, 1, false>(A, n0, n1, &a); }
      const float gx = (float)g[0], gy = (float)g[1];
      G00 += gx * gx; G01 += gx * gy; G10 += gx * gy; G11 += gy * gy;
      cpt++;
      gs0[i] = gx; gs1[i] = gy; as[i] = (int)a;
    }
  }
  {  // lk.hh:75-81, min |eigenvalue| of G / cpt (symmetric 2x2 closed form)
    const float fc = (float)cpt;
    const float a = G00 / fc, b = G01 / fc, d = G11 / fc;
    const float hm = (a + d) * 0.5f, hd = (a - d) * 0.5f;
    const float s = sqrtf(hd * hd + b * b);
    const float e1 = fabsf(hm + s), e2 = fabsf(hm - s);
    float min_ev = 99999.f;
    if (e1 < min_ev) min_ev = e1;
    if (e2 < min_ev) min_ev = e2;
    if (min_ev < min_ev_th) return Match{-1.f, -1.f, FLT_MAX};
  }
  const float det = G00 * G11 - G10 * G01;  // lk.hh:83, Eigen 2x2 inverse
  const float invdet = 1.f / det;
  const float I00 = G11 * invdet, I10 = -G10 * invdet, I01 = -G01 * invdet, I11 = G00 * invdet;

  float v0 = p0 + tr0, v1 = p1 + tr1;
  float nk0 = 1.f, nk1 = 1.f;
  for (int k = 0; k <= max_it && sqrtf(nk0 * nk0 + nk1 * nk1) >= delta; k++) {  // lk.hh:116
    float bk0 = 0.f, bk1 = 0.f;
    const bool b_safe = window_inside(B, v0, v1, hws);
#pragma unroll(N <= 49 ? N : 1)
    for (int i = 0; i < N; i++) {
      const int r = i / WS - hws, c = i % WS - hws;
      const float n0 = p0 + (float)r, n1 = p1 + (float)c;
      if (A.has((int)n0, (int)n1)) {
        uint8_t b;
        if (b_safe) interp<uint8_t, 1, true>(B, v0 + (float)r, v1 + (float)c, &b);
        else interp<uint8_t, 1, false>(B, v0 + (float)r, v1 + (float)c, &b);
        const float dt = (float)as[i] - (float)b;  // lk.hh:130
        bk0 += gs0[i] * dt; bk1 += gs1[i] * dt;
      }
    }
    nk0 = I00 * bk0 + I01 * bk1;  // lk.hh:137
    nk1 = I10 * bk0 + I11 * bk1;
    v0 += nk0; v1 += nk1;
    if (!B.has((int)v0, (int)v1)) return Match{0.f, 0.f, FLT_MAX};  // lk.hh:145-146
  }
  float err = 0.f, stddev = 1.f;
  if (PYRLK) {  // lk.hh:151-159
    float avg = 0.f;
    stddev = 0.f;
#pragma unroll(N <= 49 ? N : 1)
    for (int i = 0; i < N; i++) avg += (float)as[i];
    avg /= N;
#pragma unroll(N <= 49 ? N : 1)
    for (int i = 0; i < N; i++) stddev += fabsf(avg - (float)as[i]);
    stddev /= N;
  }
  const bool b_safe = window_inside(B, v0, v1, hws);
#pragma unroll(N <= 49 ? N : 1)
  for (int i = 0; i < N; i++) {  // lk.hh:161-171 / lucas_kanade.hpp:116-126
    const int r = i / WS - hws, c = i % WS - hws;
    uint8_t b;
    if (b_safe) interp<uint8_t, 1, true>(B, v0 + (float)r, v1 + (float)c, &b);
    else interp<uint8_t, 1, false>(B, v0 + (float)r, v1 + (float)c, &b);
    err += fabsf((float)(as[i] - (int)b));
    cpt++;
  }
  if (PYRLK) return Match{v0 - p0, v1 - p1, err / (cpt * stddev)};  // lk.hh:173
  return Match{v0 - p0, v1 - p1, err / (cpt)};                        // lucas_kanade.hpp:128
}

// The iteration test `norm(nk) >= delta` (lk.hh:116, Eigen: sqrt of the squared norm) without the square root: sqrtf is
// correctly rounded and monotonic, so { x : sqrtf(x) >= delta } is [T, inf) for the smallest such float T, found once per
// launch by stepping from delta * delta over its neighbours (T = NaN-safe: a NaN delta makes every comparison false, as in the
// literal form; delta <= 0 gives T = 0).
__device__ __forceinline__ float norm_threshold(float delta) {
  if (!(delta > 0.f)) return delta <= 0.f ? 0.f : delta;
  float t = delta * delta;
  for (int k = 0; k < 8 && sqrtf(t) < delta && t < INFINITY; k++) t = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, t) + 1u);
  for (int k = 0; k < 8 && t > 0.f; k++) {
    const float p = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, t) - 1u);
    if (sqrtf(p) >= delta) t = p; else break;
  }
  return t;  // delta * delta is within two ulps of T: the eight steps each way always settle (checked over 2000 random and the extreme deltas)
}

// ---- LPK lanes per keypoint --------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in program order; this keeps the compiler from moving them across the hand-off
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// per group, after its two term arrays: R row + R column coordinate records of 3 dwords, R = 8 for windows up to 7 x 7 (8 x 8), 16 up to 15 x 15
__host__ __device__ constexpr int lk_rec_half(int ws) { return ws <= 8 ? 8 : 16; }
__host__ __device__ constexpr int lk_records(int ws) { return 2 * lk_rec_half(ws) * 3; }
// validity bits of a window's offsets (the group kernels serve windows up to 11 x 11 = 121 offsets)
struct OffsetMask {
  unsigned long long lo = 0, hi = 0;
  __device__ __forceinline__ void set(int i) { if (i < 64) lo |= 1ull << i; else hi |= 1ull << (i - 64); }
  __device__ __forceinline__ bool test(int i) const { return i < 64 ? (lo >> i) & 1ull : (hi >> (i - 64)) & 1ull; }
  template <int N> __device__ __forceinline__ bool all() const {
    if constexpr (N <= 64) return lo == (N == 64 ? ~0ull : (1ull << (N & 63)) - 1ull);
    else return lo == ~0ull && hi == (N == 128 ? ~0ull : (1ull << ((N - 64) & 63)) - 1ull);
  }
};
// LDS block of a group (round 3): the two term arrays of an offset sit in two planes — T0[i] at dword i, T1[i] at lk_plane(ns) + i — so that
// the ordered sums fetch FOUR consecutive terms of their chain with one ds_read_b128 (13 LDS instructions per 49-term sum instead of 25
// ds_read2_b32 on the interleaved layout; the kernel is issue bound and LDS instructions were 19 % of its issue cycles).  The reads are
// broadcasts inside a group, so the 64 / LPK groups x 2 planes must start on different banks: planes 4 dwords apart (mod 32), groups 8 apart
// (LPK >= 16: 4 groups x 2 planes = the eight 16-byte slots of the 32 banks) or 4 apart (LPK = 8: 16 chunks, two per slot at best).
__host__ __device__ constexpr int lk_plane(int ns) { return ns + 4; }
__host__ __device__ constexpr int lk_group_stride(int ns, int lpk, int ws) {
  int g = lk_plane(ns) + ns + lk_records(ws);
  const int want = lpk >= 16 ? 8 : 4;
  while (lpk < 64 && g % 32 != want) g++;
  return g;
}
// sum of t[0 .. N) in index order (the reference's left-to-right float chain), four terms per LDS read; t is 16-byte aligned
template <int N> __device__ __forceinline__ float ordered_sum(const float* t, float acc) {
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < N / 4; j++) { const f4 q = *(const f4*)(t + 4 * j); acc += q.x; acc += q.y; acc += q.z; acc += q.w; }
#pragma unroll
  for (int i = N / 4 * 4; i < N; i++) acc += t[i];
  return acc;
}
template <int N> __device__ __forceinline__ float ordered_dot(const float* a, const float* b, float acc) {
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < N / 4; j++) {
    const f4 x = *(const f4*)(a + 4 * j), y = *(const f4*)(b + 4 * j);
    acc += x.x * y.x; acc += x.y * y.y; acc += x.z * y.z; acc += x.w * y.w;
  }
#pragma unroll
  for (int i = N / 4 * 4; i < N; i++) acc += a[i] * b[i];
  return acc;
}
// lane k (0..3) of every quad, to all four lanes of the quad (v_mov_b32 dpp quad_perm)
template <int K> __device__ __forceinline__ float quad_bcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}

template <int WS, class GT, bool PYRLK, int LPK>
__device__ Match lk_match_group(  // WS <= 11 (OffsetMask: 128 offsets)
    float p0, float p1, float tr0, float tr1, const DImg& A_, const DImg& B_, const DImg& Ag_, float min_ev_th,
                                int max_it, float delta, float* lds, int gl, float norm_T) {
  constexpr int hws = WS / 2, N = WS * WS, PPL = (N + LPK - 1) / LPK, NS = PPL * LPK, TP = lk_plane(NS), RH = lk_rec_half(WS);
  static_assert(N <= 128, "OffsetMask holds 128 offsets");
  float* const lds1 = lds + TP;   // the second term plane (see lk_group_stride)
  // the level's descriptors by value: the callers index a kernel-argument array with the (runtime) level, and through the
  // references every use inside the iteration loop was a fresh scalar load + wait
  const DImg A = A_, B = B_, Ag = Ag_;
  const bool a_safe = window_inside(A, p0, p1, hws);
  float gs0[PPL], gs1[PPL];
  int as[PPL];
  OffsetMask mine;  // validity bits of this lane's offsets
  // this lane's window offsets as (row, column) of the window (a lane past the window repeats the window's last offset); kept as the
  // dword index of the tap's row / column record in the group's coordinate records (see the iteration), the float offsets of the
  // per-tap form are derived where that form is used (once per level, and on the rare unsafe paths)
  int rec_r[PPL], rec_c[PPL];
#pragma unroll
  for (int q = 0; q < PPL; q++) {
    const int i = gl + q * LPK, ii = i < N ? i : N - 1;
    rec_r[q] = 3 * (ii / WS); rec_c[q] = 3 * (RH + ii % WS);
  }
  auto off_r = [&](int q) { return (float)(rec_r[q] / 3 - hws); };
  auto off_c = [&](int q) { return (float)(rec_c[q] / 3 - RH - hws); };
  if (a_safe) {  // every tap lies in the bordered area: request all rounds before using any (one round trip, no branch around the loads)
    GT g[PPL][2]; uint8_t a[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) { interp<GT, 2, true>(Ag, p0 + off_r(k), p1 + off_c(k), g[k]); interp<uint8_t, 1, true>(A, p0 + off_r(k), p1 + off_c(k), &a[k]); }
#pragma unroll
    for (int k = 0; k < PPL; k++) {
      const int i = gl + k * LPK;
      const bool ok = i < N && A.has((int)(p0 + off_r(k)), (int)(p1 + off_c(k)));
      gs0[k] = ok ? (float)g[k][0] : 0.f; gs1[k] = ok ? (float)g[k][1] : 0.f; as[k] = ok ? (int)a[k] : 0;
      if (ok) mine.set(i);
      lds[i] = gs0[k]; lds1[i] = gs1[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < PPL; k++) {
      const int i = gl + k * LPK;
      gs0[k] = 0.f; gs1[k] = 0.f; as[k] = 0;
      if (i < N) {
        const float n0 = p0 + off_r(k), n1 = p1 + off_c(k);
        if (A.has((int)n0, (int)n1)) {
          GT g[2]; uint8_t a;
          interp<GT, 2, false>(Ag, n0, n1, g); interp<uint8_t, 1, false>(A, n0, n1, &a);
          gs0[k] = (float)g[0]; gs1[k] = (float)g[1]; as[k] = (int)a;
          mine.set(i);
        }
      }
      lds[i] = gs0[k]; lds1[i] = gs1[k];
    }
  }
  OffsetMask mask = mine;  // OR over the group's lanes (xor-butterfly stays inside aligned groups of LPK lanes)
#pragma unroll
  for (int d = 1; d < LPK; d <<= 1) {
    const unsigned lo = __shfl_xor((unsigned)mask.lo, d), hi = __shfl_xor((unsigned)(mask.lo >> 32), d);
    mask.lo |= ((unsigned long long)hi << 32) | lo;
    if constexpr (N > 64) {
      const unsigned lo2 = __shfl_xor((unsigned)mask.hi, d), hi2 = __shfl_xor((unsigned)(mask.hi >> 32), d);
      mask.hi |= ((unsigned long long)hi2 << 32) | lo2;
    }
  }
  const bool all_valid = mask.template all<N>();
  wave_lds_fence();
  // Every sum over the window is the reference's left-to-right float chain, so it cannot be split — but the independent chains can
  // sit on different lanes of the group (all lanes of a group would otherwise repeat all of them): lane gl & 3 = 0 / 1 / 2 accumulates
  // G00 / G01 (= G10, the same products in the same order) / G11, and the quad broadcasts them over DPP.
  float G00, G01, G10, G11;
  int cpt = 0;
  {
    const int ia = (gl & 3) == 2 ? 1 : 0, ib = (gl & 3) == 0 ? 0 : 1;   // (gx, gx), (gx, gy), (gy, gy), (gx, gy)
    float acc = 0.f;
    if (all_valid) {  // the common case, branch-free
      acc = ordered_dot<N>(lds + ia * TP, lds + ib * TP, acc);   // lk.hh:56-72 in offset order
      cpt = N;
    } else {
#pragma unroll 1
      for (int i = 0; i < N; i++) {
        if (mask.test(i)) { acc += lds[ia * TP + i] * lds[ib * TP + i]; cpt++; }
      }
    }
    G00 = quad_bcast<0>(acc); G01 = quad_bcast<1>(acc); G11 = quad_bcast<2>(acc); G10 = G01;
  }
  {
    const float fc = (float)cpt;
    const float a = G00 / fc, b = G01 / fc, d = G11 / fc;
    const float hm = (a + d) * 0.5f, hd = (a - d) * 0.5f;
    const float s = sqrtf(hd * hd + b * b);
    const float e1 = fabsf(hm + s), e2 = fabsf(hm - s);
    float min_ev = 99999.f;
    if (e1 < min_ev) min_ev = e1;
    if (e2 < min_ev) min_ev = e2;
    if (min_ev < min_ev_th) return Match{-1.f, -1.f, FLT_MAX};
  }
  const float det = G00 * G11 - G10 * G01;
  const float invdet = 1.f / det;
  const float I00 = G11 * invdet, I10 = -G10 * invdet, I01 = -G01 * invdet, I11 = G00 * invdet;

  float v0 = p0 + tr0, v1 = p1 + tr1;
  float nk0 = 1.f, nk1 = 1.f;
  for (int k = 0; k <= max_it && (nk0 * nk0 + nk1 * nk1) >= norm_T; k++) {  // lk.hh:116 (norm >= delta, see norm_threshold)
    const bool b_safe = window_inside(B, v0, v1, hws);
    wave_lds_fence();  // the previous pass' reads are done before its terms are overwritten
    if (b_safe && all_valid) {
      // The common case without a branch around the loads: all PPL rounds of taps are requested back to back and the lane pays
      // ONE memory round trip per iteration instead of PPL dependent ones (a lane past the window samples the window's last
      // offset and stages a term nobody reads).
      if constexpr (LPK >= 16) {
        // The coordinate part of linear_interpolate (imageNd.hpp:282-290) depends on the tap's row OR its column only: the WS row
        // records {a0, 1 - a0, byte offset of row x0} and the WS column records {a1, 1 - a1, x1} are evaluated once per group (one
        // record per lane, the same float operations on the same inputs as the per-tap form) and every tap reads its two records from
        // LDS: 1 address add per tap instead of 13 VALU operations (adds, conversions, fractions, the row multiply).
        float* xl = lds1 + NS;
#pragma unroll
        for (int t = 0; t < (LPK >= 2 * RH ? 1 : 2 * RH / LPK); t++) {
          const int rec = gl + t * LPK;
          if (rec < 2 * RH) {
            const bool isrow = rec < RH;
            const int k = (rec & (RH - 1)) < WS ? (rec & (RH - 1)) : WS - 1;
            const float nn = (isrow ? v0 : v1) + (float)(k - hws);
            const int x = (int)nn;
            const float a = nn - x;
            xl[3 * rec] = a; xl[3 * rec + 1] = 1 - a;
            ((int*)xl)[3 * rec + 2] = isrow ? __mul24(x + B.border, B.pitch) : x + B.border;
          }
        }
        wave_lds_fence();
        const uint8_t* bbase = B.p0 - ((ptrdiff_t)B.border * B.pitch + (ptrdiff_t)B.border);
        uint16_t t0[PPL], t1[PPL];
#pragma unroll
        for (int q = 0; q < PPL; q++) {   // all taps requested first: one memory round trip per iteration
          const uint32_t o00 = (uint32_t)(((const int*)xl)[rec_r[q] + 2] + ((const int*)xl)[rec_c[q] + 2]), o10 = o00 + (uint32_t)B.pitch;
          __builtin_memcpy(&t0[q], bbase + o00, 2);
          __builtin_memcpy(&t1[q], bbase + o10, 2);
        }
#pragma unroll
        for (int q = 0; q < PPL; q++) {
          const int i = gl + q * LPK;
          const float* rr = xl + rec_r[q];   // the fractions are read where they are used: LDS reads cost no VALU issue and keep 4 PPL registers free
          const float* cc = xl + rec_c[q];
          const float a0 = rr[0], m0 = rr[1], a1 = cc[0], m1 = cc[1];
          const float v = (m0 * m1) * (float)(uint8_t)(t0[q] & 255) + (a0 * m1) * (float)(uint8_t)(t1[q] & 255) + (m0 * a1) * (float)(uint8_t)(t0[q] >> 8) +
                          (a0 * a1) * (float)(uint8_t)(t1[q] >> 8);
          const uint8_t b = (uint8_t)v;
          const float dt = (float)as[q] - (float)b;  // lk.hh:130
          lds[i] = gs0[q] * dt; lds1[i] = gs1[q] * dt;
        }
      } else {
        // 7 taps per lane (LPK = 8): the wave's LDS pipe is as busy as its VALU with the 49 broadcast term reads alone, the record reads
        // on top made the 400 k-keypoint case slower (2.63 -> 2.98 ms): per-tap coordinates here
        uint8_t b[PPL];
#pragma unroll
        for (int q = 0; q < PPL; q++) interp<uint8_t, 1, true>(B, v0 + off_r(q), v1 + off_c(q), &b[q]);
#pragma unroll
        for (int q = 0; q < PPL; q++) {
          const int i = gl + q * LPK;
          const float dt = (float)as[q] - (float)b[q];  // lk.hh:130
          lds[i] = gs0[q] * dt; lds1[i] = gs1[q] * dt;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < PPL; q++) {
        const int i = gl + q * LPK;
        float t0 = 0.f, t1 = 0.f;
        if (i < N && (all_valid || mine.test(i))) {
          uint8_t b;
          if (b_safe) interp<uint8_t, 1, true>(B, v0 + off_r(q), v1 + off_c(q), &b);
          else interp<uint8_t, 1, false>(B, v0 + off_r(q), v1 + off_c(q), &b);
          const float dt = (float)as[q] - (float)b;  // lk.hh:130
          t0 = gs0[q] * dt; t1 = gs1[q] * dt;
        }
        lds[i] = t0; lds1[i] = t1;
      }
    }
    float bk0, bk1;
    {
      wave_lds_fence();
      float acc = 0.f;   // even lanes: bk[0], odd lanes: bk[1]
      if (all_valid) {
        acc = ordered_sum<N>(lds + (gl & 1) * TP, acc);
      } else {
#pragma unroll 1
        for (int i = 0; i < N; i++)
          if (mask.test(i)) acc += lds[(gl & 1) * TP + i];
      }
      bk0 = quad_bcast<0>(acc); bk1 = quad_bcast<1>(acc);
    }
    nk0 = I00 * bk0 + I01 * bk1;  // lk.hh:137
    nk1 = I10 * bk0 + I11 * bk1;
    v0 += nk0; v1 += nk1;
    if (!B.has((int)v0, (int)v1)) return Match{0.f, 0.f, FLT_MAX};  // lk.hh:145-146
  }
  // error: as[i] for every offset (unset entries are 0), |as[i] - B(v + offset)| for every offset (lk.hh:151-171)
  const bool b_safe = window_inside(B, v0, v1, hws);
  wave_lds_fence();
  if (b_safe) {
    uint8_t b[PPL];
#pragma unroll
    for (int q = 0; q < PPL; q++) interp<uint8_t, 1, true>(B, v0 + off_r(q), v1 + off_c(q), &b[q]);
#pragma unroll
    for (int q = 0; q < PPL; q++) {
      const int i = gl + q * LPK;
      lds[i] = (float)as[q]; lds1[i] = fabsf((float)(as[q] - (int)b[q]));
    }
  } else {
#pragma unroll
    for (int q = 0; q < PPL; q++) {
      const int i = gl + q * LPK;
      float e = 0.f;
      if (i < N) {
        uint8_t b;
        interp<uint8_t, 1, false>(B, v0 + off_r(q), v1 + off_c(q), &b);
        e = fabsf((float)(as[q] - (int)b));
      }
      lds[i] = (float)as[q]; lds1[i] = e;
    }
  }
  wave_lds_fence();
  float err, stddev = 1.f;
  {
    float acc = 0.f;   // even lanes: the sum of as[], odd lanes: the sum of |as - b|
    acc = ordered_sum<N>(lds + (gl & 1) * TP, acc);
    cpt += N;
    err = quad_bcast<1>(acc);
    if (PYRLK) {
      float avg = quad_bcast<0>(acc);
      stddev = 0.f;
      avg /= N;
#pragma unroll
      for (int i = 0; i < N; i++) stddev += fabsf(avg - lds[i]);
      stddev /= N;
    }
  }
  if (PYRLK) return Match{v0 - p0, v1 - p1, err / (cpt * stddev)};
  return Match{v0 - p0, v1 - p1, err / (cpt)};
}

// at least 4 waves per SIMD (<= 128 VGPRs): left alone, the 7-taps-per-lane instance took 138-163 registers for no gain in issue rate
template <int WS, int LPK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void pyrlk_match_group_kernel(Pyr P, Pyr G, Pyr Nx, int nlevels, vpp_keypoint_f32* __restrict__ kps, int n,
                                                               float min_ev, float max_err, int max_it, float delta, int min_scale,
                                                               float* __restrict__ out_dist) {
  constexpr int N = WS * WS, PPL = (N + LPK - 1) / LPK, NS = PPL * LPK;
  __shared__ __attribute__((aligned(16))) float smem[(64 / LPK) * lk_group_stride(NS, LPK, WS)];
  const int gl = threadIdx.x % LPK, grp = threadIdx.x / LPK;
  const int i = blockIdx.x * (64 / LPK) + grp;
  if (i >= n) return;
  float* lds = smem + grp * lk_group_stride(NS, LPK, WS);
  vpp_keypoint_f32 kp = kps[i];
  if (!(kp.age > 0)) { if (out_dist && gl == 0) out_dist[i] = 0.f; return; }
  float tr0 = 0.f, tr1 = 0.f, dist = 0.f;
  const float norm_T = norm_threshold(delta);
  for (int S = nlevels - 1; S >= min_scale; S--) {
    tr0 *= 2.f; tr1 *= 2.f;
    const float sc = (float)(1 << S);
    const Match m = lk_match_group<WS, float, true, LPK>(kp.pos_r / sc, kp.pos_c / sc, tr0, tr1, P.l[S], Nx.l[S], G.l[S], min_ev, max_it, delta, lds, gl, norm_T);
    if (m.err < max_err) { tr0 = m.f0; tr1 = m.f1; }
    dist = m.err;
  }
  if (gl != 0) return;
  const float q0 = kp.pos_r + tr0, q1 = kp.pos_c + tr1;
  if (out_dist) out_dist[i] = dist;
  if (dist > max_err || !P.l[0].has((int)q0, (int)q1)) kp.age = 0;
  else { kp.vel_r = q0 - kp.pos_r; kp.vel_c = q1 - kp.pos_c; kp.pos_r = q0; kp.pos_c = q1; kp.age++; }
  kps[i] = kp;
}

// ---- F frame pairs per launch (vpp_pyrlk_match_batch) -------------------------------------------------------------------------------------------------
// One match of 1 250 keypoints (a rank's slice of configs[3] on 8 GPUs) costs the same 71 us as one of 5 000: the launch is a single generation of waves whose
// length is the chain of a keypoint's levels x iterations, not its width.  Frame pairs are independent (pyrlk_match.hh:15-55 touches its own pyramids and
// container only), so F of them go out as ONE launch: the same group code, a keypoint group finds its frame from the launch's block table and takes its
// pyramids' first pixels from the frame table; the per-level geometry is the launch's (frames of one stream share it).  Results per frame are those of the
// single call — the same float chains in the same order.
constexpr int kLkBatchFrames = 16;   // frame pairs per launch
constexpr int kLkBatchSlots = 64;    // frames x levels per launch (16 x 4, 8 x 8): the table travels as kernel arguments (< 4 KB)
struct LkBatch {
  DImg gp[kMaxLevels], gg[kMaxLevels], gn[kMaxLevels];   // per-level geometry of every frame's prev / grad / next (p0 unused)
  uint8_t* p[3][kLkBatchSlots];                           // first pixels: [prev | grad | next][frame * nlevels + level]
  vpp_keypoint_f32* kps[kLkBatchFrames];
  float* dist[kLkBatchFrames];
  int n[kLkBatchFrames];
  int first_block[kLkBatchFrames + 1];                    // a frame's groups fill whole workgroups: workgroup b serves frame f with first_block[f] <= b < first_block[f + 1]
};
template <int WS, int LPK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void pyrlk_match_batch_kernel(LkBatch T, int nframes, int nlevels, float min_ev, float max_err, int max_it,
                                                                                                       float delta, int min_scale) {
  constexpr int N = WS * WS, PPL = (N + LPK - 1) / LPK, NS = PPL * LPK;
  __shared__ __attribute__((aligned(16))) float smem[(64 / LPK) * lk_group_stride(NS, LPK, WS)];
  const int gl = threadIdx.x % LPK, grp = threadIdx.x / LPK;
  int f = 0;
  while (f + 1 < nframes && (int)blockIdx.x >= T.first_block[f + 1]) f++;   // (wave-uniform: scalar loads of at most 16 words)
  const int i = ((int)blockIdx.x - T.first_block[f]) * (64 / LPK) + grp;
  if (i >= T.n[f]) return;
  float* lds = smem + grp * lk_group_stride(NS, LPK, WS);
  vpp_keypoint_f32* kps = T.kps[f];
  float* out_dist = T.dist[f];
  vpp_keypoint_f32 kp = kps[i];
  if (!(kp.age > 0)) { if (out_dist && gl == 0) out_dist[i] = 0.f; return; }
  float tr0 = 0.f, tr1 = 0.f, dist = 0.f;
  const float norm_T = norm_threshold(delta);
  for (int S = nlevels - 1; S >= min_scale; S--) {
    tr0 *= 2.f; tr1 *= 2.f;
    const float sc = (float)(1 << S);
    DImg A = T.gp[S], Ag = T.gg[S], B = T.gn[S];
    const int slot = f * nlevels + S;
    A.p0 = T.p[0][slot]; Ag.p0 = T.p[1][slot]; B.p0 = T.p[2][slot];
    const Match m = lk_match_group<WS, float, true, LPK>(kp.pos_r / sc, kp.pos_c / sc, tr0, tr1, A, B, Ag, min_ev, max_it, delta, lds, gl, norm_T);
    if (m.err < max_err) { tr0 = m.f0; tr1 = m.f1; }
    dist = m.err;
  }
  if (gl != 0) return;
  const float q0 = kp.pos_r + tr0, q1 = kp.pos_c + tr1;
  if (out_dist) out_dist[i] = dist;
  if (dist > max_err || !T.gp[0].has((int)q0, (int)q1)) kp.age = 0;
  else { kp.vel_r = q0 - kp.pos_r; kp.vel_c = q1 - kp.pos_c; kp.pos_r = q0; kp.pos_c = q1; kp.age++; }
  kps[i] = kp;
}

template <int WS, int LPK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void lucas_kanade_group_kernel(Pyr P, Pyr G, Pyr Nx, int nlevels, const float* __restrict__ pts,
                                                                const float* __restrict__ pred, int n, float min_ev, int niter, float delta,
                                                                float* __restrict__ out_flow, float* __restrict__ out_dist) {
  constexpr int N = WS * WS, PPL = (N + LPK - 1) / LPK, NS = PPL * LPK;
  __shared__ __attribute__((aligned(16))) float smem[(64 / LPK) * lk_group_stride(NS, LPK, WS)];
  const int gl = threadIdx.x % LPK, grp = threadIdx.x / LPK;
  const int i = blockIdx.x * (64 / LPK) + grp;
  if (i >= n) return;
  float* lds = smem + grp * lk_group_stride(NS, LPK, WS);
  const float k0 = pts[2 * i], k1 = pts[2 * i + 1];
  const float d = (float)(1 << nlevels);
  float tr0 = (pred ? pred[2 * i] : 0.f) / d, tr1 = (pred ? pred[2 * i + 1] : 0.f) / d;
  float dist = 0.f;
  const float norm_T = norm_threshold(delta);
  for (int S = nlevels - 1; S >= 0; S--) {
    tr0 *= 2.f; tr1 *= 2.f;
    const float sc = (float)(1 << S);
    const Match m = lk_match_group<WS, int32_t, false, LPK>(k0 / sc, k1 / sc, tr0, tr1, P.l[S], Nx.l[S], G.l[S], min_ev, niter, delta, lds, gl, norm_T);
    tr0 = m.f0; tr1 = m.f1; dist = m.err;
  }
  if (gl != 0) return;
  out_flow[2 * i] = tr0; out_flow[2 * i + 1] = tr1;
  if (out_dist) out_dist[i] = dist;
}

template <int WS>
__global__ __launch_bounds__(64) void pyrlk_match_kernel(Pyr P, Pyr G, Pyr Nx, int nlevels, vpp_keypoint_f32* __restrict__ kps, int n,
                                                         float min_ev, float max_err, int max_it, float delta, int min_scale,
                                                         float* __restrict__ out_dist) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  vpp_keypoint_f32 kp = kps[i];
  if (out_dist) out_dist[i] = 0.f;
  if (!(kp.age > 0)) return;  // kp.alive(), pyrlk_match.hh:27
  float tr0 = 0.f, tr1 = 0.f, dist = 0.f;
  for (int S = nlevels - 1; S >= min_scale; S--) {  // pyrlk_match.hh:31-42
    tr0 *= 2.f; tr1 *= 2.f;
    const float sc = (float)(1 << S);
    const Match m = lk_match<WS, float, true>(kp.pos_r / sc, kp.pos_c / sc, tr0, tr1, P.l[S], Nx.l[S], G.l[S], min_ev, max_it, delta);
    if (m.err < max_err) { tr0 = m.f0; tr1 = m.f1; }
    dist = m.err;
  }
  const float q0 = kp.pos_r + tr0, q1 = kp.pos_c + tr1;
  if (out_dist) out_dist[i] = dist;
  if (dist > max_err || !P.l[0].has((int)q0, (int)q1)) kp.age = 0;  // remove -> die(), pyrlk_match.hh:44-48
  else { kp.vel_r = q0 - kp.pos_r; kp.vel_c = q1 - kp.pos_c; kp.pos_r = q0; kp.pos_c = q1; kp.age++; }  // move
  kps[i] = kp;
}

template <int WS>
__global__ __launch_bounds__(64) void lucas_kanade_kernel(Pyr P, Pyr G, Pyr Nx, int nlevels, const float* __restrict__ pts,
                                                          const float* __restrict__ pred, int n, float min_ev, int niter, float delta,
                                                          float* __restrict__ out_flow, float* __restrict__ out_dist) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float k0 = pts[2 * i], k1 = pts[2 * i + 1];
  const float d = (float)(1 << nlevels);
  float tr0 = (pred ? pred[2 * i] : 0.f) / d, tr1 = (pred ? pred[2 * i + 1] : 0.f) / d;  // lucas_kanade.hpp:163
  float dist = 0.f;
  for (int S = nlevels - 1; S >= 0; S--) {  // :165-179
    tr0 *= 2.f; tr1 *= 2.f;
    const float sc = (float)(1 << S);  // kp / int(pow(2,S)) -> Eigen converts the int scalar to float
    const Match m = lk_match<WS, int32_t, false>(k0 / sc, k1 / sc, tr0, tr1, P.l[S], Nx.l[S], G.l[S], min_ev, niter, delta);
    tr0 = m.f0; tr1 = m.f1; dist = m.err;
  }
  out_flow[2 * i] = tr0; out_flow[2 * i + 1] = tr1;
  if (out_dist) out_dist[i] = dist;
}

int check_pyramids(const char* fn, const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                   int grad_dtype, Pyr& P, Pyr& G, Pyr& N) {
  VPP_REQUIRE(prev && grad && next && nlevels >= 1 && nlevels <= kMaxLevels, VPP_ERR_INVALID_ARG, "%s: need 1..%d pyramid levels", fn, kMaxLevels);
  for (int l = 0; l < nlevels; l++) {
    VPP_REQUIRE(valid_desc(prev + l) && valid_desc(grad + l) && valid_desc(next + l), VPP_ERR_INVALID_ARG, "%s: invalid descriptor at level %d", fn, l);
    VPP_REQUIRE(prev[l].dtype == VPP_U8 && prev[l].channels == 1 && next[l].dtype == VPP_U8 && next[l].channels == 1, VPP_ERR_UNSUPPORTED,
                "%s: frames must be u8 x1", fn);
    VPP_REQUIRE(grad[l].dtype == grad_dtype && grad[l].channels == 2, VPP_ERR_UNSUPPORTED, "%s: gradient pyramid has the wrong element type", fn);
    VPP_REQUIRE(same_domain(prev + l, grad + l) && same_domain(prev + l, next + l), VPP_ERR_INVALID_ARG, "%s: level %d domains differ", fn, l);
    VPP_REQUIRE(prev[l].border >= 1 && grad[l].border >= 1 && next[l].border >= 1, VPP_ERR_BORDER_TOO_SMALL, "%s: level %d needs border >= 1", fn, l);
    P.l[l] = dimg(prev + l); G.l[l] = dimg(grad + l); N.l[l] = dimg(next + l);
    // the two images read with one `safe` flag must agree on how far a window may reach
    G.l[l].border = P.l[l].border = (prev[l].border < grad[l].border ? prev[l].border : grad[l].border);
  }
  return VPP_OK;
}

}  // namespace

extern "C" {

int vpp_pyrlk_match(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                    vpp_keypoint_f32* kps, int n, int winsize, float min_ev, float max_err, int max_iterations,
                    float convergence_delta, int min_scale, float* out_dist, void* stream) {
  Pyr P, G, N;
  int rc = check_pyramids("vpp_pyrlk_match", prev, grad, next, nlevels, VPP_F32, P, G, N);
  if (rc != VPP_OK) return rc;
  VPP_REQUIRE(kps && n >= 0 && min_scale >= 0, VPP_ERR_INVALID_ARG, "vpp_pyrlk_match: invalid argument");
  if (n == 0) return VPP_OK;
  hipStream_t st = as_stream(stream);
  // lanes per keypoint: enough waves to cover the 1024 SIMDs a few times over, no more (total work grows with LPK)
  int lpk = tuning("pyrlk.lpk", 0);
  // measured: tools/lk_lpk_ab.py (round 6, 1080p, 3 levels, 7 x 7, us per call by lanes per keypoint 8 / 16 / 32 / 64): 2 000: 135 / 102 / 83 / 74 · 4 000: 112 / 91 / 85 / 91 ·
  // 8 000: 139 / 112 / 106 / 150 · 10 000: 134 / 109 / 118 / 169 · 16 000: 168 / 143 / 183 / 235 · 20 000: 171 / 201 / 213 / 287 · 40 000: 306 / 325 / 381 / 519.  16 lanes per keypoint are
  // n / 4 waves: past 16 384 keypoints they no longer fit the chip at once (4 per SIMD) and the second generation costs more than 8 lanes' longer chains (round 2's table
  // switched at 80 000).
  if (lpk == 0) lpk = n > 16384 ? 8 : (n >= 8000 ? 16 : (n >= 3500 ? 32 : 64));
  if (winsize > 11) lpk = 1;                  // the group kernels hold up to 128 window offsets (11 x 11 = 121: the window of the reference's own benchmark,
  else if (winsize > 7 && lpk == 8) lpk = 16;  // benchmarks/pyrlk_opencv_comparison.cc:47); 8 lanes per keypoint would hold 11-16 taps per lane in registers: 16 at least
  // 9 x 9 / 11 x 11 hold 2.5 x the taps: 32 lanes per keypoint stay ahead of 16 at every count measured (11 x 11, 4 levels, us: 10 k 334 vs 345, 20 k 534 vs 648, 40 k 952 vs 1 011)
  if (winsize > 7 && winsize <= 11 && tuning("pyrlk.lpk", 0) == 0 && lpk == 16) lpk = 32;
#define VPP_LK_LAUNCH(W)                                                                                                                           \
  if (lpk == 64) pyrlk_match_group_kernel<W, 64><<<n, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else if (lpk == 32) pyrlk_match_group_kernel<W, 32><<<(n + 1) / 2, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else if (lpk == 16) pyrlk_match_group_kernel<W, 16><<<(n + 3) / 4, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else if (lpk == 8) pyrlk_match_group_kernel<W, 8><<<(n + 7) / 8, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else pyrlk_match_kernel<W><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist);
#define VPP_LK_LAUNCH_WIDE(W)  /* 9 x 9 and 11 x 11: 16, 32 or 64 lanes per keypoint */                                                              \
  if (lpk == 64) pyrlk_match_group_kernel<W, 64><<<n, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else if (lpk == 32) pyrlk_match_group_kernel<W, 32><<<(n + 1) / 2, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else if (lpk == 16) pyrlk_match_group_kernel<W, 16><<<(n + 3) / 4, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); \
  else pyrlk_match_kernel<W><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist);
  switch (winsize) {
    case 3: VPP_LK_LAUNCH(3) break;
    case 5: VPP_LK_LAUNCH(5) break;
    case 7: VPP_LK_LAUNCH(7) break;
    case 9: VPP_LK_LAUNCH_WIDE(9) break;
    case 11: VPP_LK_LAUNCH_WIDE(11) break;
    case 15: pyrlk_match_kernel<15><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); break;
    case 21: pyrlk_match_kernel<21><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, kps, n, min_ev, max_err, max_iterations, convergence_delta, min_scale, out_dist); break;
    default: set_error("vpp_pyrlk_match: unsupported window size %d (3,5,7,9,11,15,21)", winsize); return VPP_ERR_UNSUPPORTED;
  }
#undef VPP_LK_LAUNCH
#undef VPP_LK_LAUNCH_WIDE
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_pyrlk_match_batch(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nframes, int nlevels, vpp_keypoint_f32* const* kps,
                          const int* n, int winsize, float min_ev, float max_err, int max_iterations, float convergence_delta, int min_scale, float* const* out_dist,
                          void* stream) {
  VPP_REQUIRE(nframes >= 0 && (nframes == 0 || (prev && grad && next && kps && n)) && min_scale >= 0, VPP_ERR_INVALID_ARG, "vpp_pyrlk_match_batch: invalid argument");
  VPP_REQUIRE(nlevels >= 1 && nlevels <= kMaxLevels, VPP_ERR_INVALID_ARG, "vpp_pyrlk_match_batch: need 1..%d pyramid levels", kMaxLevels);
  // every frame as the single call checks it; one launch serves frames of ONE geometry (per level: sizes, pitches, borders of the three pyramids)
  std::vector<Pyr> P(nframes), G(nframes), N(nframes);
  bool same = true;
  long long total = 0;
  for (int f = 0; f < nframes; f++) {
    VPP_REQUIRE(n[f] >= 0 && (n[f] == 0 || kps[f]), VPP_ERR_INVALID_ARG, "vpp_pyrlk_match_batch: frame %d: invalid keypoint array", f);
    const int rc = check_pyramids("vpp_pyrlk_match_batch", prev + (size_t)f * nlevels, grad + (size_t)f * nlevels, next + (size_t)f * nlevels, nlevels, VPP_F32, P[f], G[f], N[f]);
    if (rc != VPP_OK) return rc;
    auto geo = [](const DImg& a, const DImg& b) { return a.nr == b.nr && a.nc == b.nc && a.pitch == b.pitch && a.border == b.border; };
    for (int l = 0; l < nlevels; l++) same = same && geo(P[f].l[l], P[0].l[l]) && geo(G[f].l[l], G[0].l[l]) && geo(N[f].l[l], N[0].l[l]);
    total += n[f];
  }
  if (total == 0) return VPP_OK;
  int lpk = tuning("pyrlk.lpk", 0);
  if (lpk == 0) lpk = total > 16384 ? 8 : (total >= 8000 ? 16 : (total >= 3500 ? 32 : 64));   // the single call's table (vpp_pyrlk_match), on the launch's total
  if (winsize > 7 && lpk == 8) lpk = 16;
  if (winsize > 7 && winsize <= 11 && tuning("pyrlk.lpk", 0) == 0 && lpk == 16) lpk = 32;
  const bool grouped = (winsize == 3 || winsize == 5 || winsize == 7 || winsize == 9 || winsize == 11) && (lpk == 8 || lpk == 16 || lpk == 32 || lpk == 64);
  if (!same || !grouped || nframes == 1 || !tuning("pyrlk.batch", 1)) {   // what one launch does not serve goes out as the calls (whose results are the contract)
    for (int f = 0; f < nframes; f++) {
      const int rc = vpp_pyrlk_match(prev + (size_t)f * nlevels, grad + (size_t)f * nlevels, next + (size_t)f * nlevels, nlevels, kps[f], n[f], winsize, min_ev, max_err, max_iterations,
                                     convergence_delta, min_scale, out_dist ? out_dist[f] : nullptr, stream);
      if (rc != VPP_OK) return rc;
    }
    return VPP_OK;
  }
  hipStream_t st = as_stream(stream);
  const int per_launch = std::min(kLkBatchFrames, kLkBatchSlots / nlevels), gpb = 64 / lpk;   // frames per launch; keypoint groups per workgroup
  for (int f0 = 0; f0 < nframes; f0 += per_launch) {
    const int nf = std::min(per_launch, nframes - f0);
    LkBatch T{};
    for (int l = 0; l < nlevels; l++) { T.gp[l] = P[f0].l[l]; T.gg[l] = G[f0].l[l]; T.gn[l] = N[f0].l[l]; }
    int blocks = 0;
    for (int k = 0; k < nf; k++) {
      const int f = f0 + k;
      for (int l = 0; l < nlevels; l++) { T.p[0][k * nlevels + l] = P[f].l[l].p0; T.p[1][k * nlevels + l] = G[f].l[l].p0; T.p[2][k * nlevels + l] = N[f].l[l].p0; }
      T.kps[k] = kps[f]; T.dist[k] = out_dist ? out_dist[f] : nullptr; T.n[k] = n[f];
      T.first_block[k] = blocks;
      blocks += (n[f] + gpb - 1) / gpb;
    }
    for (int k = nf; k <= kLkBatchFrames; k++) T.first_block[k] = blocks;
    if (!blocks) continue;
#define VPP_LKB_LAUNCH(W, L) pyrlk_match_batch_kernel<W, L><<<blocks, 64, 0, st>>>(T, nf, nlevels, min_ev, max_err, max_iterations, convergence_delta, min_scale)
#define VPP_LKB_CASE(W, NARROW)                                \
    case W:                                                   \
      if (lpk == 64) VPP_LKB_LAUNCH(W, 64);                   \
      else if (lpk == 32) VPP_LKB_LAUNCH(W, 32);              \
      else if (lpk == 16) VPP_LKB_LAUNCH(W, 16);              \
      else NARROW;                                            \
      break;
    switch (winsize) {   // (9 x 9 and 11 x 11 have no 8-lane instance: lpk was raised above)
      VPP_LKB_CASE(3, VPP_LKB_LAUNCH(3, 8)) VPP_LKB_CASE(5, VPP_LKB_LAUNCH(5, 8)) VPP_LKB_CASE(7, VPP_LKB_LAUNCH(7, 8)) VPP_LKB_CASE(9, (void)0) VPP_LKB_CASE(11, (void)0)
    }
#undef VPP_LKB_CASE
#undef VPP_LKB_LAUNCH
    VPP_LAUNCH_CHECK();
  }
  return VPP_OK;
}

int vpp_lucas_kanade(const vpp_image_desc* prev, const vpp_image_desc* grad, const vpp_image_desc* next, int nlevels,
                     const float* pts, const float* prediction, int n, int winsize, int min_ev, int niterations, int delta,
                     float* out_flow, float* out_dist, void* stream) {
  Pyr P, G, N;
  int rc = check_pyramids("vpp_lucas_kanade", prev, grad, next, nlevels, VPP_I32, P, G, N);
  if (rc != VPP_OK) return rc;
  VPP_REQUIRE(pts && out_flow && n >= 0, VPP_ERR_INVALID_ARG, "vpp_lucas_kanade: invalid argument");
  if (n == 0) return VPP_OK;
  hipStream_t st = as_stream(stream);
  int lpk = tuning("pyrlk.lpk", 0);
  if (lpk == 0) lpk = n >= 80000 ? 8 : (n >= 8000 ? 16 : (n >= 3500 ? 32 : 64));  // measured: tools/tune_pyrlk.py (round 2: 16 lanes win up to 40 k keypoints, 8 from 100 k; 8 also beats 1 lane per keypoint at 400 k)
  if (winsize > 11) lpk = 1;
  else if (winsize > 7 && lpk == 8) lpk = 16;
  const float fev = (float)min_ev, fdelta = (float)delta;
#define VPP_LK_LAUNCH(W)                                                                                                                           \
  if (lpk == 64) lucas_kanade_group_kernel<W, 64><<<n, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else if (lpk == 32) lucas_kanade_group_kernel<W, 32><<<(n + 1) / 2, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else if (lpk == 16) lucas_kanade_group_kernel<W, 16><<<(n + 3) / 4, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else if (lpk == 8) lucas_kanade_group_kernel<W, 8><<<(n + 7) / 8, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else lucas_kanade_kernel<W><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist);
#define VPP_LK_LAUNCH_WIDE(W)                                                                                                                      \
  if (lpk == 64) lucas_kanade_group_kernel<W, 64><<<n, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else if (lpk == 32) lucas_kanade_group_kernel<W, 32><<<(n + 1) / 2, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else if (lpk == 16) lucas_kanade_group_kernel<W, 16><<<(n + 3) / 4, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); \
  else lucas_kanade_kernel<W><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist);
  switch (winsize) {
    case 3: VPP_LK_LAUNCH(3) break;
    case 5: VPP_LK_LAUNCH(5) break;
    case 7: VPP_LK_LAUNCH(7) break;
    case 9: VPP_LK_LAUNCH_WIDE(9) break;
    case 11: VPP_LK_LAUNCH_WIDE(11) break;
    case 15: lucas_kanade_kernel<15><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); break;
    case 21: lucas_kanade_kernel<21><<<(n + 63) / 64, 64, 0, st>>>(P, G, N, nlevels, pts, prediction, n, fev, niterations, fdelta, out_flow, out_dist); break;
    default: set_error("vpp_lucas_kanade: unsupported window size %d (3,5,7,9,11,15,21)", winsize); return VPP_ERR_UNSUPPORTED;
  }
#undef VPP_LK_LAUNCH
#undef VPP_LK_LAUNCH_WIDE
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

}  // extern "C"
