// sdof.hip — K11/K12: semi-dense optical flow (integer block matching + propagation).
// Reference: vpp/algorithms/optical_flow/semi_dense_optical_flow.hpp:48-214 (driver), :18-42 (sad_distance),
//            vpp/algorithms/optical_flow/gradient_descent.hh:10-89 (gradient_descent_match).
// Epipolar options (:62-66,79-90) are not supported (optional in the reference, UB as written — SURVEY Q11).
//
// Serial-order semantics (what the reference's tests run; its OpenMP build is racy) made parallel:
//   * cell claims (:114-143): the serial loop lets the LOWEST keypoint index that maps to a flow-map cell compute it.
//     Kernel 1 takes atomicMin(index) per cell, kernel 2 runs the descent for the winners only: same result, any order.
//   * propagation sweeps (:146-201) are in-place Gauss-Seidel scans in raster (Ki odd) / reverse raster (Ki even) order.
//     Cell (i,j) depends on the already-updated (i-1,j-1),(i-1,j),(i-1,j+1),(i,j-1).  The result of the serial scan is computed as
//     the fixed point of parallel Jacobi rounds over the cells whose inputs changed (sdof_classify_kernel + sdof_rounds_kernel, many
//     workgroups, a handful of rounds); the lock-step wavefront on one workgroup (sdof_propagate_kernel: all cells with 2i+j = t are
//     independent, one barrier per step) is kept as an independent on-device cross-check (tuning sdof.propagate = 1).
// Integer arithmetic throughout: bit-exact.  SAD windows are L2-resident; the path is integer-VALU / latency bound.
#include "common.hpp"
#include "sdof_tail.hpp"
#include <atomic>
#include "tracker_device.hpp"
#include <climits>
#include <vector>
#include <algorithm>
#include <type_traits>
#include <climits>
using namespace vpp_amd;

namespace vpp_amd { int launch_fill_border(const vpp_image_desc* img, int mode, const void* value, hipStream_t st); }

namespace {


struct Maps { DImg flow, mark, dist; };  // i32x2, u8, i32 per flow-map cell

// of_internals::sad_distance behind the domain test of the `distance` lambda (semi_dense_optical_flow.hpp:18-42,102-108).
// The reference sums |a-b| row by row and stops after the first row that pushes the sum above `th`; here every row of
// both windows is loaded up front as (unaligned) dwords — one memory round trip instead of WS dependent ones — the row
// sums come from v_sad_u8 (4 pixels per instruction), and the same early-out rule is applied to them, so the returned
// value is identical.  The dwords over-read at most 3 bytes past a window row: inside the 2*winsize border.
template <int WS>
__device__ __forceinline__ int sad_rows(const uint8_t* __restrict__ row1, const uint8_t* __restrict__ row2, int pitch1, int pitch2, int th) {
  constexpr int ND = (WS + 3) / 4;
  constexpr uint32_t tail_mask = (WS % 4) ? ((1u << (8 * (WS % 4))) - 1u) : 0xFFFFFFFFu;
  uint32_t a[WS][ND], b[WS][ND];
#pragma unroll
  for (int r = 0; r < WS; r++)
#pragma unroll
    for (int d = 0; d < ND; d++) {
      __builtin_memcpy(&a[r][d], row1 + (ptrdiff_t)r * pitch1 + 4 * d, 4);
      __builtin_memcpy(&b[r][d], row2 + (ptrdiff_t)r * pitch2 + 4 * d, 4);
    }
  int err = 0;
#pragma unroll
  for (int r = 0; r < WS; r++) {
    if (err <= th) {
      uint32_t err2 = 0;
#pragma unroll
      for (int d = 0; d < ND; d++) {
        const uint32_t m = d == ND - 1 ? tail_mask : 0xFFFFFFFFu;
        err2 = __builtin_amdgcn_sad_u8(a[r][d] & m, b[r][d] & m, err2);
      }
      err += (int)err2;
    }
  }
  return err;
}

// the same with the first window already in registers (callers that compare one fixed window against many candidates)
template <int WS> struct WindowRegs { uint32_t d[WS ? WS : 1][WS ? (WS + 3) / 4 : 1]; };
template <int WS>
__device__ __forceinline__ void load_window(WindowRegs<WS>& w, const uint8_t* __restrict__ row, int pitch) {
#pragma unroll
  for (int r = 0; r < WS; r++)
#pragma unroll
    for (int d = 0; d < (WS + 3) / 4; d++) __builtin_memcpy(&w.d[r][d], row + (ptrdiff_t)r * pitch + 4 * d, 4);
}
// The same window loaded ONCE per group of G lanes (G = 4: a DPP quad, G = 8: half a DPP row) instead of once per lane: lane j fetches the rows j, j + G, ...
// and every row goes round the group through DPP (quad_perm broadcast; for 8 lanes the quad broadcast and its row_half_mirror give the rows k and k + 4 together).
// The descent kernels are bound by the address rate of their gathers (LABNOTES round 6): with every lane loading all WS rows a wave of 16 four-lane groups spent
// 9 of its ~20 load instructions on 16 distinct windows; here it spends 3.  Every lane of a group must be active (the groups' control flow is group-uniform).
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false); }
template <int WS, int G>
__device__ __forceinline__ void load_window_group(WindowRegs<WS>& w, const uint8_t* __restrict__ row, int pitch, int j) {
  static_assert(G == 4 || G == 8, "a quad or two");
  constexpr int ND = (WS + 3) / 4, NV = (WS + G - 1) / G;
  uint32_t v[NV][ND];
#pragma unroll
  for (int q = 0; q < NV; q++) {
#pragma unroll
    for (int d = 0; d < ND; d++) __builtin_memcpy(&v[q][d], row + (ptrdiff_t)(q * G + G <= WS ? j + q * G : min(j + q * G, WS - 1)) * pitch + 4 * d, 4);   // (no branch: rows past the window repeat its last one)
  }
#pragma unroll
  for (int q = 0; q < NV; q++)
#pragma unroll
    for (int d = 0; d < ND; d++) {
      const uint32_t x = v[q][d];
      const uint32_t t0 = dpp_mov<0x00>(x), t1 = dpp_mov<0x55>(x), t2 = dpp_mov<0xAA>(x), t3 = dpp_mov<0xFF>(x);   // lane k of the lane's own quad
      const uint32_t t[4] = {t0, t1, t2, t3};
      if constexpr (G == 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (q * 4 + k < WS) w.d[q * 4 + k][d] = t[k];
      } else {
        const bool lowq = j < 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t u = dpp_mov<0x141>(t[k]);   // row_half_mirror: lane i of 8 reads lane 7 - i, i.e. the OTHER quad's broadcast
          if (q * 8 + k < WS) w.d[q * 8 + k][d] = lowq ? t[k] : u;
          if (q * 8 + k + 4 < WS) w.d[q * 8 + k + 4][d] = lowq ? u : t[k];
        }
      }
    }
}
template <int WS>
__device__ __forceinline__ int sad_rows_against(const WindowRegs<WS>& a, const uint8_t* __restrict__ row2, int pitch2, int th) {
  constexpr int ND = (WS + 3) / 4;
  constexpr uint32_t tail_mask = (WS % 4) ? ((1u << (8 * (WS % 4))) - 1u) : 0xFFFFFFFFu;
  uint32_t b[WS][ND];
#pragma unroll
  for (int r = 0; r < WS; r++)
#pragma unroll
    for (int d = 0; d < ND; d++) __builtin_memcpy(&b[r][d], row2 + (ptrdiff_t)r * pitch2 + 4 * d, 4);
  int err = 0;
#pragma unroll
  for (int r = 0; r < WS; r++) {
    if (err <= th) {
      uint32_t err2 = 0;
#pragma unroll
      for (int d = 0; d < ND; d++) {
        const uint32_t m = d == ND - 1 ? tail_mask : 0xFFFFFFFFu;
        err2 = __builtin_amdgcn_sad_u8(a.d[r][d] & m, b[r][d] & m, err2);
      }
      err += (int)err2;
    }
  }
  return err;
}

// WS: the window size when it is one of the register-window sizes (5, 7, 9, 11), else 0 = the generic loop over the runtime
// `ws`.  Every kernel of this file is instantiated per WS: with a runtime switch at each of the ~30 inlined call sites the
// out-of-line recomputation alone was 84 KB of code, more than the instruction cache, on the critical path of the ordered sweep.
template <int WS>
__device__ __forceinline__ int distance_fn(const DImg& i1, const DImg& i2, int a0, int a1, int b0, int b1, int ws, int th) {
  if (!(i1.has(a0, a1) && i2.has(b0, b1))) return INT_MAX;
  const uint8_t* row1 = i1.row<uint8_t>(a0 - ws / 2) + (a1 - ws / 2);
  const uint8_t* row2 = i2.row<uint8_t>(b0 - ws / 2) + (b1 - ws / 2);
  if constexpr (WS != 0) return sad_rows<WS>(row1, row2, i1.pitch, i2.pitch, th);
  int err = 0;
  for (int r = 0; r < ws && err <= th; r++) {
    int err2 = 0;
    for (int c = 0; c < ws; c++) err2 += abs((int)row1[c] - (int)row2[c]);
    err += err2;
    row1 += i1.pitch; row2 += i2.pitch;
  }
  return err;
}

struct Cell { int f0, f1, dist, mark; };  // mark: low byte = flow_map_mark value; bits 8.. see kTagShift
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Cell load_cell16(const Cell* p) { const v4i v = *(const v4i*)p; return Cell{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void store_cell16(Cell* p, const Cell& c) { *(v4i*)p = v4i{c.f0, c.f1, c.dist, c.mark}; }
// Write-through stores and L1-bypassing loads (relaxed agent-scope atomics lower to `sc1` accesses on gfx950): what one workgroup stores this way every other
// workgroup reads this way without an agent-scope release (an L2 write-back) on the one side and an acquire (an L1 invalidation) on the other —
// MI355X_MICROARCH.md, "inter-workgroup visibility": sc1 stores AND sc1 loads.  A record goes as two 8-byte halves; nobody reads it while it is written.
__device__ __forceinline__ Cell load_cell_sc1(const Cell* p) {
  unsigned long long* q = (unsigned long long*)p;
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return Cell{(int)(unsigned)lo, (int)(unsigned)(lo >> 32), (int)(unsigned)hi, (int)(unsigned)(hi >> 32)};
}
__device__ __forceinline__ void store_cell_sc1(Cell* p, const Cell& c) {
  unsigned long long* q = (unsigned long long*)p;
  __hip_atomic_store(q, ((unsigned long long)(unsigned)c.f1 << 32) | (unsigned)c.f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, ((unsigned long long)(unsigned)c.mark << 32) | (unsigned)c.dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t load_u32_sc1(const uint32_t* p) { return __hip_atomic_load((uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_u32_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The sweeps' cell records of one scale (fused sweeps, round 4): three arrays of one 16-byte record per cell of the sweep domain (row pitch nj) that
// every writer of the maps keeps equal to them between sweeps — `pre` and the two round buffers of sdof_sweep_kernel.  p[0] == nullptr: not kept.
struct Mirrors { Cell* p[3]; int nj; };

struct GdMatch { int f0, f1, distance; };

// gradient_descent_match (gradient_descent.hh:10-89), neighbour tables verbatim
template <int WS, class DF>
__device__ __forceinline__ GdMatch gradient_descent_impl(DF distance_of, int p0, int p1, int pr0, int pr1, int max_iteration) {
  int m0 = pr0, m1 = pr1;
  int match_distance = distance_of(pr0, pr1, INT_MAX);
  unsigned match_i = 8;
  // The reference's tables (gradient_descent.hh:28-38), packed into immediates: indexed by a per-lane value, the arrays would be
  // loads from constant memory in the dependency chain of every candidate.
  //   c8_it[9][2] = {{6,3},{0,3},{0,5},{2,5},{2,7},{4,7},{4,1},{6,1},{0,0}}  (first candidate, one-past-last; 3 bits each)
  //   c8[8][2]    = {{-1,1},{0,1},{1,1},{-1,0},{1,0},{-1,-1},{0,-1},{1,-1}}    (offset + 1; 2 bits each)
  constexpr unsigned kFirst = 0xd22406u, kEnd = 0x27fb5bu, kDr = 0x9224u, kDc = 0x16au;
#pragma nounroll
  for (int search = 0; search < max_iteration; search++) {
    unsigned i = (kFirst >> (3 * match_i)) & 7u;
    const unsigned end = (kEnd >> (3 * match_i)) & 7u;
#pragma nounroll
    do {  // the first candidate unconditionally, then up to `end` (the reference's peeled first step + for loop)
      const int n0 = pr0 + (int)((kDr >> (2 * i)) & 3u) - 1, n1 = pr1 + (int)((kDc >> (2 * i)) & 3u) - 1;
      const int d = distance_of(n0, n1, match_distance);
      if (d < match_distance) { m0 = n0; m1 = n1; match_i = i; match_distance = d; }
      i = (i + 1) & 7u;
    } while (i != end);
    if (pr0 == m0 && pr1 == m1) break;
    pr0 = m0; pr1 = m1;
  }
  return GdMatch{m0 - p0, m1 - p1, match_distance};
}
template <int WS>
__device__ __forceinline__ GdMatch gradient_descent_match(const DImg& i1, const DImg& i2, int ws, int p0, int p1, int pr0, int pr1, int max_iteration) {
  return gradient_descent_impl<WS>([&](int n0, int n1, int th) { return distance_fn<WS>(i1, i2, p0, p1, n0, n1, ws, th); }, p0, p1, pr0, pr1, max_iteration);
}

// cell_lo / cell_hi: the flow-map rows this launch owns (row-strip sharding of the per-keypoint phases; [0, INT_MAX) = all)
__global__ __launch_bounds__(256) void sdof_claim_kernel(const int32_t* __restrict__ kps, int n, int scale_div, int patch, DImg owner, int cell_lo, int cell_hi) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p0 = kps[2 * i] / scale_div, p1 = kps[2 * i + 1] / scale_div;
  const int pf0 = p0 / patch, pf1 = p1 / patch;
  if (!owner.has(pf0, pf1) || pf0 < cell_lo || pf0 >= cell_hi) return;
  atomicMin(owner.row<uint32_t>(pf0) + pf1, (uint32_t)i);
}

__global__ __launch_bounds__(256) void sdof_claim_all_kernel(const int32_t* __restrict__ kps, int n, int patch, ClaimAll c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k0 = kps[2 * i], k1 = kps[2 * i + 1];
  for (int s = c.first; s <= c.last; s++) {
    const int div = 1 << s;
    const int pf0 = (k0 / div) / patch, pf1 = (k1 / div) / patch;
    if (c.owner[s].has(pf0, pf1)) atomicMin(c.owner[s].row<uint32_t>(pf0) + pf1, (uint32_t)i);
  }
}

template <int WS>
__global__ __launch_bounds__(64) void sdof_descent_kernel(const int32_t* __restrict__ kps, int n, int scale_div, int patch, int ws, DImg owner,
                                                          DImg i1, DImg i2, Maps cur, Maps coarse, int has_coarse, int cell_lo, int cell_hi, int clean_owner) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const int p0 = kps[2 * i] / scale_div, p1 = kps[2 * i + 1] / scale_div;  // :116
  const int pf0 = p0 / patch, pf1 = p1 / patch;
  if (pf0 < cell_lo || pf0 >= cell_hi) return;
  if (!owner.has(pf0, pf1) || owner.row<uint32_t>(pf0)[pf1] != (uint32_t)i) return;  // :120 — first keypoint in index order claims
  if (clean_owner) owner.row<uint32_t>(pf0)[pf1] = 0xFFFFFFFFu;   // (see sdof_descent_group_kernel)
  int pr0 = p0, pr1 = p1;
  if (has_coarse) {  // multiscale prediction, :126-128
    const int pfm0 = p0 / (2 * patch), pfm1 = p1 / (2 * patch);
    if (coarse.mark.has(pfm0, pfm1) && coarse.mark.row<uint8_t>(pfm0)[pfm1]) {
      const int32_t* f = coarse.flow.row<int32_t>(pfm0) + 2 * pfm1;
      pr0 = p0 + f[0] * 2; pr1 = p1 + f[1] * 2;
    }
  }
  const GdMatch m = gradient_descent_match<WS>(i1, i2, ws, p0, p1, pr0, pr1, 5);  // :132-134
  int32_t* f = cur.flow.row<int32_t>(pf0) + 2 * pf1;
  f[0] = m.f0; f[1] = m.f1;                       // :137-139
  cur.dist.row<int32_t>(pf0)[pf1] = m.distance;  // :140
  cur.mark.row<uint8_t>(pf0)[pf1] = 2;           // :141
}

// gradient_descent_match (gradient_descent.hh:10-89) on a group of 8 lanes (j = lane in group): the candidates of one search step side by
// side, winner = minimum of (distance << 3 | position in the walk): the FIRST of equal minima, as the sequential strict-less rule keeps.  `start` = distance(p, prediction, INT_MAX).
template <class DIST>
__device__ __forceinline__ GdMatch group_descent(DIST dist, int p0, int p1, int pr0, int pr1, int start, int j) {
  constexpr unsigned kFirst = 0xd22406u, kEnd = 0x27fb5bu, kDr = 0x9224u, kDc = 0x16au;
  int m0 = pr0, m1 = pr1;
  int match_distance = start;
  unsigned match_i = 8;
#pragma nounroll
  for (int search = 0; search < 5; search++) {
    const unsigned first = (kFirst >> (3 * match_i)) & 7u, end = (kEnd >> (3 * match_i)) & 7u;
    const unsigned count = ((end - first - 1u) & 7u) + 1u;
    const unsigned ci = (first + (unsigned)j) & 7u;
    const int n0 = pr0 + (int)((kDr >> (2 * ci)) & 3u) - 1, n1 = pr1 + (int)((kDc >> (2 * ci)) & 3u) - 1;
    const int d = (unsigned)j < count ? dist(n0, n1, match_distance) : INT_MAX;
    unsigned long long key = ((unsigned long long)(unsigned)d << 3) | (unsigned)j;
#pragma unroll
    for (int x = 1; x < 8; x <<= 1) {
      const unsigned lo = __shfl_xor((unsigned)key, x), hi = __shfl_xor((unsigned)(key >> 32), x);
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      key = o < key ? o : key;
    }
    const int best = (int)(unsigned)(key >> 3);
    if (best < match_distance) {
      const unsigned wi = (first + (unsigned)(key & 7u)) & 7u;
      m0 = pr0 + (int)((kDr >> (2 * wi)) & 3u) - 1; m1 = pr1 + (int)((kDc >> (2 * wi)) & 3u) - 1;
      match_i = wi; match_distance = best;
    }
    if (pr0 == m0 && pr1 == m1) break;
    pr0 = m0; pr1 = m1;
  }
  return GdMatch{m0 - p0, m1 - p1, match_distance};
}

// LDS hand-off inside one wave (the 8 lanes of a group sit in one wave): LDS operations of a wave execute in program order, the fences keep
// the compiler from moving them across the hand-off.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int kUnionRows = 15;   // (WS + 2) rows of 16 bytes for WS <= 11, and the sweeps' 15-row regions (group_descent_region); 15 x 16 B = 60 dwords per group: the 8 groups of a wave start 28 banks apart

// group_descent with the candidates' windows staged through LDS.  The 3 / 5 / 8 candidates of a search step are the 8-neighbours of the
// step's centre: their WS x WS windows all lie inside one (WS + 2) x (WS + 2) patch of image 2.  With a window per lane (WS 12-byte row loads
// each) a step cost a group 8 x WS scattered row loads, and the phase was bound by the address rate of those gathers (~1 lane per clock
// and CU: 37-40 us per scale for 75 k keypoints at 4K, VALU issue at 26 %).  Here the group loads the patch ONCE per step — lane j takes
// rows j and j + 8, 16 bytes each — into its LDS slot, and every lane cuts its candidate's rows out of the staged rows (one 16-byte LDS
// read and three v_alignbyte per row).  Same windows, same row-wise early-out rule, same first-of-minima selection: bit-identical.
// A patch that is not entirely inside the bordered area (a walk along the frame's edge) takes the per-lane form `dist` for that step.
// have_start: `start` = distance(p, pr) is known (the sweeps pass d2); otherwise it is taken from the first staged patch's centre.
// G: lanes per group, 8 (a candidate per lane) or 4 (lane j takes the candidates j and j + 4 of a step one after the other: half the waves for the same walks)
template <int WS, int G = 8, class DIST>
__device__ __forceinline__ GdMatch group_descent_staged(const WindowRegs<WS>& wa, bool a_ok, const DImg& i2, int p0, int p1, int pr0, int pr1,
                                                        bool have_start, int start, int j, uint4* __restrict__ slot, DIST dist) {
  static_assert(WS >= 3 && WS + 2 <= kUnionRows, "the patch of a step is (WS + 2) rows of at most 16 bytes");
  static_assert(G == 8 || G == 4, "8 or 4 lanes per group");
  constexpr int H = WS / 2, UR = WS + 2, ND = (WS + 3) / 4;
  constexpr uint32_t tail_mask = (WS % 4) ? ((1u << (8 * (WS % 4))) - 1u) : 0xFFFFFFFFu;
  constexpr unsigned kFirst = 0xd22406u, kEnd = 0x27fb5bu, kDr = 0x9224u, kDc = 0x16au;
  int m0 = pr0, m1 = pr1;
  int match_distance = start;
  unsigned match_i = 8;
#pragma nounroll
  for (int search = 0; search < 5; search++) {
    const int ur0 = pr0 - H - 1, uc0 = pr1 - H - 1;   // the patch: rows [ur0, ur0 + UR), 16 bytes from column uc0
    const bool staged = a_ok && ur0 >= -i2.border && ur0 + UR <= i2.nr + i2.border && uc0 >= -i2.border && uc0 + 16 <= i2.nc + i2.border;
    if (staged) {
      wave_lds_fence();   // the previous step's reads are done
      const uint8_t* src = i2.row<uint8_t>(ur0 + j) + uc0;
      constexpr int NV = (UR + G - 1) / G;   // rows per lane: j, j + G, ...
      uint4 v[NV];
#pragma unroll
      for (int q = 0; q < NV; q++) if (j + q * G < UR) __builtin_memcpy(&v[q], src + (ptrdiff_t)(q * G) * i2.pitch, 16);
#pragma unroll
      for (int q = 0; q < NV; q++) if (j + q * G < UR) slot[j + q * G] = v[q];
      wave_lds_fence();
    }
    // sad_distance of the keypoint's window against the window centred (dr, dc) from the step's centre, cut out of the staged patch
    // The full sum, without sad_distance's row-wise early-out (:30-38): a candidate that the early-out would cut short has a partial sum above the
    // running best already, its full sum is larger still, and either way it is only ever compared `< best` — same decisions, and every distance that is
    // stored belongs to an accepted candidate, whose sum was never cut.  (The test per row was a compare, an exec-mask update and a branch that a wave of
    // 64 candidates practically never takes.)  A window of 4k + 1 bytes ends with a single byte that lies inside one dword of the row for every shift:
    // one bit-field extract instead of a byte-align and a mask.
    auto sad_at = [&](int dr, int dc, int) -> int {
      const uint32_t o = (uint32_t)(1 + dc);
      uint32_t err = 0;
#pragma unroll
      for (int r = 0; r < WS; r++) {
        const uint4 q = slot[1 + dr + r];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int d = 0; d < ND; d++) {
          if (d == ND - 1 && WS % 4 == 1) {
            err = __builtin_amdgcn_sad_u8(wa.d[r][d] & tail_mask, __builtin_amdgcn_ubfe(w[d], 8u * o, 8u), err);
          } else {
            const uint32_t m = d == ND - 1 ? tail_mask : 0xFFFFFFFFu;
            const uint32_t b = __builtin_amdgcn_alignbyte(w[d + 1], w[d], o);
            err = __builtin_amdgcn_sad_u8(wa.d[r][d] & m, b & m, err);
          }
        }
      }
      return (int)err;
    };
    if (!have_start) {
      match_distance = staged ? (i2.has(pr0, pr1) ? sad_at(0, 0, INT_MAX) : INT_MAX) : dist(pr0, pr1, INT_MAX);
      have_start = true;
    }
    const unsigned first = (kFirst >> (3 * match_i)) & 7u, end = (kEnd >> (3 * match_i)) & 7u;
    const unsigned count = ((end - first - 1u) & 7u) + 1u;
    unsigned long long key = ~0ull;
#pragma unroll
    for (int q = 0; q < 8 / G; q++) {   // this lane's candidates of the step, by position in the walk
      const unsigned jq = (unsigned)j + (unsigned)(q * G);
      const unsigned ci = (first + jq) & 7u;
      const int cdr = (int)((kDr >> (2 * ci)) & 3u) - 1, cdc = (int)((kDc >> (2 * ci)) & 3u) - 1;
      const int n0 = pr0 + cdr, n1 = pr1 + cdc;
      int d = INT_MAX;
      if (jq < count) d = staged ? (i2.has(n0, n1) ? sad_at(cdr, cdc, match_distance) : INT_MAX) : dist(n0, n1, match_distance);
      const unsigned long long kq = ((unsigned long long)(unsigned)d << 3) | jq;
      key = kq < key ? kq : key;
    }
    // minimum over the group by DPP (quad_perm for lanes ^ 1 and ^ 2, row_half_mirror to meet the other quad of an 8-lane group): a few cycles per exchange where
    // __shfl_xor's ds_bpermute was an LDS round trip, three of them in a row in every search step's chain
    {
      auto take_min = [&](unsigned lo, unsigned hi) { const unsigned long long ok = ((unsigned long long)hi << 32) | lo; key = ok < key ? ok : key; };
      take_min(dpp_mov<0xB1>((unsigned)key), dpp_mov<0xB1>((unsigned)(key >> 32)));
      take_min(dpp_mov<0x4E>((unsigned)key), dpp_mov<0x4E>((unsigned)(key >> 32)));
      if constexpr (G == 8) take_min(dpp_mov<0x141>((unsigned)key), dpp_mov<0x141>((unsigned)(key >> 32)));
    }
    const int best = (int)(unsigned)(key >> 3);
    if (best < match_distance) {
      const unsigned wi = (first + (unsigned)(key & 7u)) & 7u;
      m0 = pr0 + (int)((kDr >> (2 * wi)) & 3u) - 1; m1 = pr1 + (int)((kDc >> (2 * wi)) & 3u) - 1;
      match_i = wi; match_distance = best;
    }
    if (pr0 == m0 && pr1 == m1) break;
    pr0 = m0; pr1 = m1;
  }
  return GdMatch{m0 - p0, m1 - p1, match_distance};
}

// group_descent_staged for the sweeps' jobs (8 lanes, the start distance known) with a REGION instead of a patch per step (round 6): the group stages the rows within
// H + RH of the step's centre (RH = 3 for windows up to 9 x 9: 15 rows of 16 bytes, the two loads per lane a patch took) and keeps stepping out of it while the centre
// stays within RH - 1 of the region's — three search steps per memory round trip instead of one: a walk of 5 steps is 2 round trips.  A candidate's rows are cut out
// of the staged rows at a run-time byte offset (0 .. 2 RH: a dword select + v_alignbyte).  Same windows, same sums, same first-of-minima selection: bit-identical.
constexpr int kRegionRows = 15;
template <int WS, bool HAVE_START = true, class DIST>
__device__ __forceinline__ GdMatch group_descent_region(const WindowRegs<WS>& wa, bool a_ok, const DImg& i2, int p0, int p1, int pr0, int pr1, int start, int j,
                                                        uint4* __restrict__ slot, DIST dist) {
  constexpr int H = WS / 2, RH = ((kRegionRows - WS) / 2 < 3 ? (kRegionRows - WS) / 2 : 3), UR = WS + 2 * RH, ND = (WS + 3) / 4;
  static_assert(RH >= 1 && UR <= kRegionRows && 2 * RH + WS <= 16, "the region is UR rows of 16 bytes");
  constexpr uint32_t tail_mask = (WS % 4) ? ((1u << (8 * (WS % 4))) - 1u) : 0xFFFFFFFFu;
  constexpr unsigned kFirst = 0xd22406u, kEnd = 0x27fb5bu, kDr = 0x9224u, kDc = 0x16au;
  int m0 = pr0, m1 = pr1;
  int match_distance = start;
  unsigned match_i = 8;
  int cr = 0, cc = 0;        // the staged region's centre
  bool have_region = false;
#pragma nounroll
  for (int search = 0; search < 5; search++) {
    bool covered = have_region && abs(pr0 - cr) <= RH - 1 && abs(pr1 - cc) <= RH - 1;   // (the same on the group's 8 lanes)
    if (!covered) {
      const int ur0 = pr0 - H - RH, uc0 = pr1 - H - RH;   // rows [ur0, ur0 + UR), 16 bytes from column uc0
      have_region = a_ok && ur0 >= -i2.border && ur0 + UR <= i2.nr + i2.border && uc0 >= -i2.border && uc0 + 16 <= i2.nc + i2.border;
      if (have_region) {
        wave_lds_fence();   // the previous steps' reads are done
        const uint8_t* src = i2.row<uint8_t>(ur0 + j) + uc0;
        uint4 v0, v1 = make_uint4(0u, 0u, 0u, 0u);
        __builtin_memcpy(&v0, src, 16);
        if (j + 8 < UR) __builtin_memcpy(&v1, src + (ptrdiff_t)8 * i2.pitch, 16);
        slot[j] = v0;
        if (j + 8 < UR) slot[j + 8] = v1;
        wave_lds_fence();
        cr = pr0; cc = pr1; covered = true;
      }
    }
    auto sad_at = [&](int dr, int dc) -> int {   // the window centred (dr, dc) from the step's centre, cut out of the region (full sums: see group_descent_staged)
      const int row0 = pr0 + dr - cr + RH;
      const uint32_t o = (uint32_t)(pr1 + dc - cc + RH), sel = o >> 2, sh = o & 3u;
      uint32_t err = 0;
#pragma unroll
      for (int r = 0; r < WS; r++) {
        const uint4 q = slot[row0 + r];
        const uint32_t w[5] = {q.x, q.y, q.z, q.w, 0u};
        uint32_t x[ND + 1];
#pragma unroll
        for (int d = 0; d <= ND; d++) x[d] = sel ? w[d + 1 < 5 ? d + 1 : 4] : w[d < 5 ? d : 4];
#pragma unroll
        for (int d = 0; d < ND; d++) {
          const uint32_t m = d == ND - 1 ? tail_mask : 0xFFFFFFFFu;
          const uint32_t b = __builtin_amdgcn_alignbyte(x[d + 1], x[d], sh);
          err = __builtin_amdgcn_sad_u8(wa.d[r][d] & m, b & m, err);
        }
      }
      return (int)err;
    };
    if constexpr (!HAVE_START) {   // the distance at the prediction itself (the per-keypoint descents: the sweeps pass d2)
      if (search == 0) match_distance = covered ? (i2.has(pr0, pr1) ? sad_at(0, 0) : INT_MAX) : dist(pr0, pr1, INT_MAX);
    }
    const unsigned first = (kFirst >> (3 * match_i)) & 7u, end = (kEnd >> (3 * match_i)) & 7u;
    const unsigned count = ((end - first - 1u) & 7u) + 1u;
    const unsigned ci = (first + (unsigned)j) & 7u;
    const int cdr = (int)((kDr >> (2 * ci)) & 3u) - 1, cdc = (int)((kDc >> (2 * ci)) & 3u) - 1;
    const int n0 = pr0 + cdr, n1 = pr1 + cdc;
    int d = INT_MAX;
    if ((unsigned)j < count) d = covered ? (i2.has(n0, n1) ? sad_at(cdr, cdc) : INT_MAX) : dist(n0, n1, match_distance);
    unsigned long long key = ((unsigned long long)(unsigned)d << 3) | (unsigned)j;
    {
      auto take_min = [&](unsigned lo, unsigned hi) { const unsigned long long ok = ((unsigned long long)hi << 32) | lo; key = ok < key ? ok : key; };
      take_min(dpp_mov<0xB1>((unsigned)key), dpp_mov<0xB1>((unsigned)(key >> 32)));
      take_min(dpp_mov<0x4E>((unsigned)key), dpp_mov<0x4E>((unsigned)(key >> 32)));
      take_min(dpp_mov<0x141>((unsigned)key), dpp_mov<0x141>((unsigned)(key >> 32)));
    }
    const int best = (int)(unsigned)(key >> 3);
    if (best < match_distance) {
      const unsigned wi = (first + (unsigned)(key & 7u)) & 7u;
      m0 = pr0 + (int)((kDr >> (2 * wi)) & 3u) - 1; m1 = pr1 + (int)((kDc >> (2 * wi)) & 3u) - 1;
      match_i = wi; match_distance = best;
    }
    if (pr0 == m0 && pr1 == m1) break;
    pr0 = m0; pr1 = m1;
  }
  return GdMatch{m0 - p0, m1 - p1, match_distance};
}

// The same phase with 8 lanes per keypoint: the candidates of one search step (the 3, 5 or 8 neighbours the reference walks one after
// the other) are evaluated by the lanes of the group side by side and the winner is the minimum of (distance << 3 | position in the
// walk) — the sequential rule "replace on strictly smaller" keeps the FIRST of equal minima, and a candidate that the sequential walk
// would cut short (its early-out threshold is the running best there, the step's starting best here) is one that cannot win in either
// form: same flow, same distance.  One lane per keypoint left 1180 waves of 28 dependent window SADs each on a 4K frame.
// BYCELL: a group per flow-map cell instead of per keypoint — the same set of descents (a cell's owner is the keypoint that descends for it), launched
// where the cells are fewer than the keypoints (coarse scales: several keypoints share a cell and all but the owner would leave at once, yet their
// groups occupied 3/4 of every wave of the coarsest scale).
// G = 4 lanes per keypoint (window sizes with a register window only): 16 keypoints per wave, each lane two candidates of a step — the 10 k waves of a 4K scale's
// 82 k keypoints are two generations of resident waves at this kernel's 5 waves per SIMD, the 5 k waves of the 4-lane form are one.
// Register-window sizes (WS != 0, round 6): everything that depends on the keypoint alone — the owner entry, the coarser scale's mark and flow, the keypoint's window — is requested in ONE
// memory round trip (the loads used to follow each other behind the tests: keypoint -> owner -> coarse mark -> coarse flow -> window + first patch), and the window is
// loaded once per group (load_window_group) instead of once per lane.
template <int WS, bool BYCELL = false, int G = 8>
__global__ __launch_bounds__(64) void sdof_descent_group_kernel(const int32_t* __restrict__ kps, int n, int scale_div, int patch, int ws, DImg owner,
                                                                DImg i1, DImg i2, Maps cur, Maps coarse, int has_coarse, int cell_lo, int cell_hi, int clean_owner, Mirrors mir) {
  static_assert(G == 8 || (G == 4 && WS != 0), "4 lanes per keypoint: staged walks only");
  __shared__ uint4 s_union[64 / G][kUnionRows];   // per group: the (WS + 2)-row patch of the current search step
  const int grp = blockIdx.x * (64 / G) + (threadIdx.x / G), j = threadIdx.x & (G - 1);
  int i = grp;
  if constexpr (BYCELL) {
    if (grp >= owner.nr * owner.nc) return;
    const int c0 = grp / owner.nc, c1 = grp - c0 * owner.nc;
    if (c0 < cell_lo || c0 >= cell_hi) return;
    const uint32_t o = owner.row<uint32_t>(c0)[c1];
    if (o == 0xFFFFFFFFu) return;   // nobody claimed the cell
    i = (int)o;
  }
  if (i >= n) return;
  const int p0 = kps[2 * i] / scale_div, p1 = kps[2 * i + 1] / scale_div;  // :116
  const int pf0 = p0 / patch, pf1 = p1 / patch;
  int pr0 = p0, pr1 = p1;
  WindowRegs<WS> wa;
  const bool a_ok = i1.has(p0, p1);
  if constexpr (WS != 0) {
    // (unconditional loads at clamped coordinates: a load inside a branch is waited for at the branch's end, and the four would follow each other again)
    const bool mine = BYCELL || (pf0 >= cell_lo && pf0 < cell_hi && owner.has(pf0, pf1));
    uint32_t own = (uint32_t)i;
    if constexpr (!BYCELL) own = owner.row<uint32_t>(min(max(pf0, 0), owner.nr - 1))[min(max(pf1, 0), owner.nc - 1)];
    const int pfm0 = p0 / (2 * patch), pfm1 = p1 / (2 * patch);
    const bool cin = has_coarse && coarse.mark.has(pfm0, pfm1);
    const int cm0 = min(max(pfm0, 0), coarse.mark.nr - 1), cm1 = min(max(pfm1, 0), coarse.mark.nc - 1);
    const uint32_t cmk = coarse.mark.row<uint8_t>(cm0)[cm1];
    const int32_t* cfp = coarse.flow.row<int32_t>(cm0) + 2 * cm1;
    const int cf0 = cfp[0], cf1 = cfp[1];
    load_window_group<WS, G>(wa, i1.row<uint8_t>(min(max(p0, 0), i1.nr - 1) - ws / 2) + (min(max(p1, 0), i1.nc - 1) - ws / 2), i1.pitch, j);
    asm volatile("" :: "v"(cmk), "v"(cf0), "v"(cf1));   // (keeps the coarse loads above the branch below: the compiler sinks them behind it otherwise)
    if (!mine || (!BYCELL && own != (uint32_t)i)) return;  // :120 — first keypoint in index order claims
    if (clean_owner && j == 0) owner.row<uint32_t>(pf0)[pf1] = 0xFFFFFFFFu;
    if (cin && cmk) { pr0 = p0 + cf0 * 2; pr1 = p1 + cf1 * 2; }  // multiscale prediction, :126-128
  } else {
  if constexpr (!BYCELL) {
    if (pf0 < cell_lo || pf0 >= cell_hi) return;
    if (!owner.has(pf0, pf1) || owner.row<uint32_t>(pf0)[pf1] != (uint32_t)i) return;  // :120 — first keypoint in index order claims
  }
  // clean_owner: the cell's one owner hands the owner map back empty (every lane of the group has read the entry: one wave, program order; the cell's other
  // keypoints compare against their own index and leave on either value) — the next call finds the maps as a reset would leave them
  if (clean_owner && j == 0) owner.row<uint32_t>(pf0)[pf1] = 0xFFFFFFFFu;
  if (has_coarse) {  // multiscale prediction, :126-128
    const int pfm0 = p0 / (2 * patch), pfm1 = p1 / (2 * patch);
    if (coarse.mark.has(pfm0, pfm1) && coarse.mark.row<uint8_t>(pfm0)[pfm1]) {
      const int32_t* f = coarse.flow.row<int32_t>(pfm0) + 2 * pfm1;
      pr0 = p0 + f[0] * 2; pr1 = p1 + f[1] * 2;
    }
  }
  // the keypoint's own window (image 1 at p) is the same in every comparison of the walk: loaded once, the candidates cost one window each
  if (WS != 0 && a_ok) load_window<WS>(wa, i1.row<uint8_t>(p0 - ws / 2) + (p1 - ws / 2), i1.pitch);
  }
  auto dist = [&](int b0, int b1, int th) -> int {
    if constexpr (WS != 0) {
      if (!(a_ok && i2.has(b0, b1))) return INT_MAX;
      return sad_rows_against<WS>(wa, i2.row<uint8_t>(b0 - ws / 2) + (b1 - ws / 2), i2.pitch, th);
    } else return distance_fn<WS>(i1, i2, p0, p1, b0, b1, ws, th);
  };
  GdMatch g;
  // (the 8-lane kernels through group_descent_region<WS, false>: 4K pair 0.1633 -> 0.1625 ms, 1080p -0.7 us — within the noise, not taken)
  if constexpr (WS != 0) g = group_descent_staged<WS, G>(wa, a_ok, i2, p0, p1, pr0, pr1, false, INT_MAX, j, s_union[threadIdx.x / G], dist);
  else g = group_descent(dist, p0, p1, pr0, pr1, dist(pr0, pr1, INT_MAX), j);
  if (j != 0) return;
  int32_t* f = cur.flow.row<int32_t>(pf0) + 2 * pf1;
  f[0] = g.f0; f[1] = g.f1;                        // :137-139
  cur.dist.row<int32_t>(pf0)[pf1] = g.distance;    // :140
  cur.mark.row<uint8_t>(pf0)[pf1] = 2;               // :141
  if (mir.p[0]) {   // the sweeps' records of the cell (the reset launch left the records of unclaimed cells at mark 0)
    const Cell rec{g.f0, g.f1, g.distance, 2};
    const size_t at = (size_t)pf0 * mir.nj + pf1;
    store_cell16(mir.p[0] + at, rec); store_cell16(mir.p[1] + at, rec); store_cell16(mir.p[2] + at, rec);
  }
}

__device__ __forceinline__ int inorm(int a, int b) { return (int)sqrt((double)(a * a + b * b)); }  // Eigen norm() on vint2

// loop_body (:149-189) for cell (pf0, pf1) evaluated at image point (r, c)
template <int WS>
__device__ void propagate_cell(const DImg& i1, const DImg& i2, int ws, const Maps& m, int r, int c, int pf0, int pf1) {
  if (!m.mark.row<uint8_t>(pf0)[pf1]) return;
  int32_t* fpf = m.flow.row<int32_t>(pf0) + 2 * pf1;
  const int prev0 = fpf[0], prev1 = fpf[1];
  for (int dr = -1; dr <= 1; dr++)
    for (int dc = -1; dc <= 1; dc++) {
      if (!dr && !dc) continue;
      const int q0 = pf0 + dr, q1 = pf1 + dc;
      if (!(m.flow.has(q0, q1) && m.mark.row<uint8_t>(q0)[q1])) continue;
      const int32_t* fq = m.flow.row<int32_t>(q0) + 2 * q1;
      const int fq0 = fq[0], fq1 = fq[1];
      if (inorm(fpf[0] - fq0, fpf[1] - fq1) > 2 && inorm(prev0 - fq0, prev1 - fq1) > 2) {
        const int d1 = m.dist.row<int32_t>(pf0)[pf1];
        const int d2 = distance_fn<WS>(i1, i2, r, c, r + fq0, c + fq1, ws, INT_MAX);
        if (d2 < d1) {
          const GdMatch g = gradient_descent_match<WS>(i1, i2, ws, r, c, r + fq0, c + fq1, 5);
          if (g.distance < d1) {
            m.mark.row<uint8_t>(pf0)[pf1] = 1;
            fpf[0] = g.f0; fpf[1] = g.f1;
            m.dist.row<int32_t>(pf0)[pf1] = g.distance;
          }
        }
      }
    }
}

// All `niters` sweeps of one scale in one launch of ONE workgroup (skewed wavefront, barrier per step).
template <int WS>
__global__ __launch_bounds__(1024) void sdof_propagate_kernel(DImg i1, DImg i2, int ws, Maps m, int patch, int niters) {
  const int NI = (i1.nr - 1) / patch + 1, NJ = (i1.nc - 1) / patch + 1;  // cells visited by the loops at :192-200
  for (int Ki = 0; Ki < niters; Ki++) {
    const bool forward = (Ki % 2) != 0;  // :191
    const int tmax = 2 * (NI - 1) + (NJ - 1);
    for (int t = 0; t <= tmax; t++) {
      const int lo = max(0, (t - (NJ - 1) + 1) >> 1), hi = min(NI - 1, t >> 1);
      for (int iw = lo + (int)threadIdx.x; iw <= hi; iw += 1024) {
        const int jw = t - 2 * iw;
        int r, c;
        if (forward) { r = iw * patch; c = jw * patch; }
        else { r = i1.nr - 1 - iw * patch; c = i1.nc - 1 - jw * patch; }  // :198-200 start at nrows-1 / ncols-1
        propagate_cell<WS>(i1, i2, ws, m, r, c, r / patch, c / patch);
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ Cell load_map_cell(const Maps& m, int ci, int cj) {
  Cell c;
  const int32_t* f = m.flow.row<int32_t>(ci) + 2 * cj;
  c.f0 = f[0]; c.f1 = f[1]; c.dist = m.dist.row<int32_t>(ci)[cj]; c.mark = m.mark.row<uint8_t>(ci)[cj];
  return c;
}

// ---- propagation as Jacobi rounds to the fixed point (round 3) --------------------------------------------------------------
// A sweep (semi_dense_optical_flow.hpp:146-201) is an in-place raster scan: loop_body(c) reads the 4 neighbours visited earlier at their
// post-visit values R(n) and the 4 visited later (and c itself) at their pre-sweep values: R(c) = F(c; R|earlier(c)).  The lock-step
// wavefront of rounds 1-2 walked all 2 NI + NJ anti-diagonals (817 at the middle scale of a 4K pair, each waiting for the slowest cell of
// the step) on ONE workgroup.  But changes are sparse and their chains are short, so the same R is reached much faster as the fixed
// point of   S_0(c) = F(c; pre),   S_k(c) = F(c; S_{k-1}|earlier(c)):
//   by induction over the raster order, S_k(c) = R(c) as soon as every earlier neighbour is final, and a cell whose earlier neighbours
//   never differ from `pre` is final at k = 0 — the number of rounds is the longest chain of cells that change at all, not the length of
//   the wavefront (measured with tools/sdof_rounds_sim.cpp, which also checks the identity against the serial sweep: 2 rounds on the
//   4K bench scene, 5 with a keypoint in every cell, 18 on two unrelated noise frames).
// Round k only has to visit J_k = D_{k-1} + the later neighbours of D_{k-1} (D_k = cells whose value changed in round k); every other
// cell keeps its value.  Values are double-buffered by round parity (a round reads B[(k-1)&1] and writes B[k&1]; a cell of D_{k-1} is
// always in J_k, which carries its value over to the other buffer), so one pass per round needs no ordering between its jobs.
// All rounds of a sweep run in ONE launch of many workgroups: 8 lanes per job (the 8 neighbour SADs of loop_body side by side, then the
// 8 candidates of every descent step side by side), jobs pulled from a device queue, a grid barrier between rounds.  The barrier only
// waits for workgroups that have REGISTERED, and registration closes at the first arrival: a workgroup that was not yet resident when
// the others finished round 0 simply leaves, so the kernel cannot deadlock whatever else shares the GPU (other streams, other ranks).
// zero between sweeps (see the exits).  done / nchanged / flushn / ack: sdof_sweep_kernel only.
struct SweepCtl { unsigned count[2], head[2], reg, nreg, arrive, pad; unsigned long long gen; unsigned done, nchanged, flushn, ack, ncand, pad2; };   // ncand: sdof_sweep_kernel, the candidates all tiles found
constexpr unsigned kRegClosed = 0x80000000u;
constexpr int kTagShift = 8;            // Cell::mark bits 8..: round + 1 in which the value last changed (0: unchanged this sweep)
constexpr int kJobsPerGroup = 32;       // 256 threads / 8 lanes
constexpr unsigned kSpinLimit = 1u << 22;   // polls of a barrier wait: a poll is s_sleep(1) + an L2 load, ~1 us under load, so ~4 s — far beyond any legitimate wait; on a timeout the
                                            // workgroup raises kDevErrSweepBarrier in the sticky device error word (common.hpp) and leaves: the host's next vpp_sync reports VPP_ERR_HIP
// err: the sticky device error word (pinned host memory) or nullptr.  chg / cflag (sdof_sweep_kernel): the cells whose value changed in this sweep, listed once each.
// sub (sdof_sweep_kernel): the arrival counters below ctl->done, one per 128 bytes.
// skip (sdof_sweep_kernel): one word, written by EVERY fused sweep — id + 1 when the sweep found no candidate at all (the next sweep of the same scale then finds none either), else 0
struct RoundArrays { Cell* pre; Cell* B[2]; uint32_t* Q[2]; uint32_t* qflag[2]; SweepCtl* ctl; unsigned* err; uint32_t* chg; uint32_t* cflag; unsigned* sub; unsigned* skip; };
constexpr int kSubCounters = 64, kSubStride = 32;   // words
__device__ __forceinline__ void raise_barrier_timeout(const RoundArrays& a) {
  a.ctl->pad = 1;
  if (a.skip) __hip_atomic_store(a.skip, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // whatever an earlier sweep said about "no candidates" is void: no later sweep skips on it
  if (a.err) __hip_atomic_fetch_or(a.err, (unsigned)kDevErrSweepBarrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ unsigned g_round_stats[4];   // [0] rounds, [1] jobs, [2] jobs evaluated, [3] changes
// diagnostics (stats == 1 only, tools/sweep_log.py): a log of the fused sweeps — {code, count, 100 MHz wall clock} per entry.  code 255: a launch starts (count =
// workgroups); 253 / 252: a workgroup with `count` candidates has classified its tile / finished round 0 on them; 1..250: the size of that round is published;
// 254: `count` workgroups run the rounds; 251: workgroup `ticket` (count >> 8) has finished its share of round (count & 255)
__device__ unsigned long long g_sweep_log[4096];
__device__ unsigned g_sweep_log_n;
__device__ __forceinline__ void sweep_log(unsigned code, unsigned n) {
  const unsigned i = atomicAdd(&g_sweep_log_n, 1u);
  if (i < 4096) g_sweep_log[i] = ((unsigned long long)code << 56) | ((unsigned long long)(n & 0xFFFFFFu) << 32) | (unsigned long long)(unsigned)wall_clock64();
}


// Pass over all cells: copies the pre-sweep maps into the round buffers and lists the cells whose loop_body can do anything at all
// against the pre-sweep maps (a marked neighbour whose flow differs by more than 2 px, :164-165 with flow_map(pf) == prev_flow).
__global__ __launch_bounds__(256) void sdof_classify_kernel(Maps m, int NI, int NJ, RoundArrays a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) { a.ctl->reg = 0; a.ctl->nreg = 0; a.ctl->arrive = 0; a.ctl->gen = 0; }   // nobody touches these during this kernel
  bool cand = false;
  if (idx < NI * NJ) {
    const int ci = idx / NJ, cj = idx - ci * NJ;
    const Cell cur = load_map_cell(m, ci, cj);
    if (cur.mark) {
#pragma unroll
      for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++) {
          if (!dr && !dc) continue;
          const int q0 = ci + dr, q1 = cj + dc;
          if (q0 < 0 || q1 < 0 || q0 >= NI || q1 >= NJ) continue;
          if (!m.mark.row<uint8_t>(q0)[q1]) continue;
          const int32_t* f = m.flow.row<int32_t>(q0) + 2 * q1;
          const int a0 = cur.f0 - f[0], a1 = cur.f1 - f[1];
          if (a0 * a0 + a1 * a1 >= 9) cand = true;
        }
    }
    store_cell16(a.pre + idx, cur); store_cell16(a.B[0] + idx, cur); store_cell16(a.B[1] + idx, cur);
  }
  const unsigned long long b = __ballot(cand);
  if (b) {
    const int lane = __lane_id(), leader = __ffsll((long long)b) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&a.ctl->count[0], (unsigned)__popcll(b));
    base = __shfl(base, leader);
    if (cand) a.Q[0][base + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)idx;
  }
}

// Append the lanes' targets (-1: none) to a list, each at most once: the flag exchange keeps a cell from being listed twice, and the
// slots are taken with ONE atomic per wave (the counter is a single word: returning atomics on one address retire at ~11 ns each, and a
// changed cell lists up to five cells).  Called at a point every lane of the wave reaches.
__device__ __forceinline__ void append_unique(uint32_t* __restrict__ flags, unsigned* counter, uint32_t* __restrict__ list, int target) {
  bool push = false;
  if (target >= 0) push = __hip_atomic_exchange(&flags[target], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  const unsigned long long b = __ballot(push);
  if (b) {
    const int lane = __lane_id(), leader = __ffsll((long long)b) - 1;
    unsigned base = 0;
    if (lane == leader) base = __hip_atomic_fetch_add(counter, (unsigned)__popcll(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = __shfl(base, leader);
    if (push) store_u32_sc1(list + base + __popcll(b & ((1ull << lane) - 1ull)), (uint32_t)target);
  }
}
// The two appends that end a pass — the next round's queue and the list of changes — as ONE exchange and ONE slot atomic per wave (round 6): a lane carries at most one of
// the two (the caller puts a job's change on a lane that never carries a queue target), so both lists' exchanges are one instruction, and so are the two leaders' counter
// atomics: two memory round trips at the end of a pass instead of four.  Called at a point every lane of the wave reaches.
__device__ __forceinline__ void append_either(uint32_t* __restrict__ qflags, unsigned* qcounter, uint32_t* __restrict__ qlist, int target,
                                              uint32_t* __restrict__ cflags, unsigned* ccounter, uint32_t* __restrict__ clist, int chg) {
  const bool isq = target >= 0, isc = !isq && chg >= 0;
  bool push = false;
  if (isq || isc) push = __hip_atomic_exchange(isq ? &qflags[target] : &cflags[chg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  const unsigned long long bq = __ballot(push && isq), bc = __ballot(push && isc);
  if (!(bq | bc)) return;
  const int lane = __lane_id();
  const int lq = bq ? __ffsll((long long)bq) - 1 : -1, lc = bc ? __ffsll((long long)bc) - 1 : -2;
  unsigned base = 0;
  if (lane == lq || lane == lc) base = __hip_atomic_fetch_add(lane == lq ? qcounter : ccounter, (unsigned)__popcll(lane == lq ? bq : bc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned baseq = __shfl(base, max(lq, 0)), basec = __shfl(base, max(lc, 0));
  if (push) {
    if (isq) store_u32_sc1(qlist + baseq + __popcll(bq & ((1ull << lane) - 1ull)), (uint32_t)target);
    else store_u32_sc1(clist + basec + __popcll(bc & ((1ull << lane) - 1ull)), (uint32_t)chg);
  }
}
// the queue of round parity `q`
__device__ __forceinline__ void enqueue_targets(const RoundArrays& a, SweepCtl* ctl, int q, int target) { append_unique(a.qflag[q], &ctl->count[q], a.Q[q], target); }

// One job of round k: loop_body (:149-189) of `cell` on the 8 lanes of a group (j = lane in the group) — S_k(cell) = F(cell; pre on the later neighbours and
// the cell itself, S_{k-1} = Bprev on the earlier ones), written to Bcur and to the maps.  Returns the cell this lane wants in the next round's queue (-1: none);
// *changed_out: the cell's value differs from S_{k-1}(cell).  clear_flag: the cell came out of round k's queue (its flag is handed back).
// MAPS: the value also goes into the maps (sdof_rounds_kernel: every round writes them, ordered by the barriers' releases; sdof_sweep_kernel has no releases —
// two rounds' plain stores to one cell from two XCDs could reach memory in either order — and writes the maps once, at the end of the sweep).
// TILE (round 0 inside sdof_sweep_kernel): every record the job reads — the cell's and its neighbours', `pre` and the round buffer alike (equal between sweeps) — is
// in the workgroup's LDS copy of its tile + halo (tile_at: the cell's entry, row pitch kTilePitch, out-of-domain entries zero): no memory round trip before the windows.
constexpr int kTilePitch = 18;   // kSweepTile + 2
__device__ __forceinline__ Cell cell_of(const uint4& v) { return Cell{(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
template <int WS, bool MAPS, bool TILE = false, int TPITCH = kTilePitch>
__device__ __forceinline__ int round_job(const DImg& i1, const DImg& i2, int ws, const Maps& m, int patch, int forward, int NI, int NJ, const RoundArrays& a, int k,
                                         const Cell* __restrict__ Bprev, Cell* __restrict__ Bcur, int cell, int j, uint4* __restrict__ slot, int stats, bool clear_flag,
                                         bool* changed_out, const uint4* __restrict__ tile_at = nullptr) {
  const int par = k & 1;
  const int ci = cell / NJ, cj = cell - ci * NJ;
  if (clear_flag && j == 0) store_u32_sc1(a.qflag[par] + cell, 0u);   // set by whoever enqueued the cell for this round (nobody enqueues for round k during round k)
  Cell pre, old;
  if constexpr (TILE) { pre = cell_of(tile_at[0]); old = pre; }
  else {
    pre = load_cell16(a.pre + cell);   // (nobody writes `pre` during a sweep: plain loads)
    old = load_cell_sc1(Bprev + cell);
  }
  // lane j holds neighbour j in loop_body's order: (-1,-1) (-1,0) (-1,1) (0,-1) (0,1) (1,-1) (1,0) (1,1)
  const int jj = j + (j >= 4 ? 1 : 0);
  const int dr = jj / 3 - 1, dc = jj % 3 - 1;
  const int q0 = ci + dr, q1 = cj + dc;
  const bool in = q0 >= 0 && q1 >= 0 && q0 < NI && q1 < NJ;
  const bool earlier = forward ? j < 4 : j >= 4;   // raster / reverse raster visiting order
  // Whether the neighbour is marked at all is read from its `pre` record (a sweep only ever turns a 2 into a 1, :179): the round buffers' records are written for
  // marked cells only (descents, the end of a sweep) and never reset — an unmarked cell's entry there holds anything and is not looked at.
  Cell nb{0, 0, 0, 0};
  if constexpr (TILE) nb = cell_of(tile_at[dr * TPITCH + dc]);
  else if (in) {
    nb = load_cell16(a.pre + (size_t)q0 * NJ + q1);
    if (earlier) { const Cell bq = load_cell_sc1(Bprev + (size_t)q0 * NJ + q1); if ((nb.mark & 0xFF) != 0) nb = bq; }
  }
  const bool nbm = (nb.mark & 0xFF) != 0;
  // does any earlier neighbour hold a value that changed in round k - 1?  (round 0: everything is evaluated once)
  const unsigned long long grp = 0xFFull << (__lane_id() & ~7);
  const bool need = k == 0 || (__ballot(in && earlier && (nb.mark >> kTagShift) == k) & grp) != 0;
  Cell cur = Cell{old.f0, old.f1, old.dist, old.mark & 0xFF};
  const int r = forward ? ci * patch : i1.nr - 1 - (NI - 1 - ci) * patch, c = forward ? cj * patch : i1.nc - 1 - (NJ - 1 - cj) * patch;
  WindowRegs<WS> wa;
  bool a_ok = false, okb = false, need_walk = false;
  int d2 = INT_MAX;
  auto dist = [&](int b0, int b1, int th) -> int {
    if constexpr (WS != 0) {
      if (!(a_ok && i2.has(b0, b1))) return INT_MAX;
      return sad_rows_against<WS>(wa, i2.row<uint8_t>(b0 - ws / 2) + (b1 - ws / 2), i2.pitch, th);
    } else return distance_fn<WS>(i1, i2, r, c, b0, b1, ws, th);
  };
  if (need) {
    a_ok = i1.has(r, c);
    if constexpr (WS != 0) {
      if (a_ok) load_window_group<WS, 8>(wa, i1.row<uint8_t>(r - ws / 2) + (c - ws / 2), i1.pitch, j);   // (`need` and a_ok are the same on the 8 lanes of a job)
    }
    // static part of the test at :164-165: the neighbour is marked and differs from prev_flow; its d2 (:169) depends on nothing else
    const int b0 = pre.f0 - nb.f0, b1 = pre.f1 - nb.f1;
    okb = nbm && b0 * b0 + b1 * b1 >= 9;
    d2 = okb ? dist(r + nb.f0, c + nb.f1, INT_MAX) : INT_MAX;
    cur = Cell{pre.f0, pre.f1, pre.dist, pre.mark & 0xFF};
    need_walk = true;
  }
  // loop_body's walk over the neighbours (:160-187).  Each group keeps its OWN cursor: it skips ahead to its next neighbour that passes the
  // tests at :164-170 against its running best, and the groups of the wave then run their descents TOGETHER, whatever neighbour each of them
  // is at.  (With one loop over kk for the whole wave, a descent was executed once per kk at which ANY of the 8 groups needed one — up to 8
  // descents of up to 5 dependent search steps per wave, most lanes idle: a round lasted as long as that chain, ~17 us on the 4K bench scene.)
  // Per group the sequence of tests, descents and updates is exactly the sequential one.
  int kk = 0;
  for (;;) {
    bool found = false;
    int n0 = 0, n1 = 0, d2k = 0;
    while (need_walk && kk < 8) {
      const int ok = __shfl((int)okb, kk, 8);
      n0 = __shfl(nb.f0, kk, 8); n1 = __shfl(nb.f1, kk, 8); d2k = __shfl(d2, kk, 8);
      kk++;
      const int a0 = cur.f0 - n0, a1 = cur.f1 - n1;
      if (ok && a0 * a0 + a1 * a1 >= 9 && d2k < cur.dist) { found = true; break; }
    }
    if (!__ballot(found)) break;   // no group of this wave has a descent left
    if (found) {
      GdMatch g;   // :173-175; its first distance is d2 itself
      if constexpr (WS != 0) g = group_descent_region<WS>(wa, a_ok, i2, r, c, r + n0, c + n1, d2k, j, slot, dist);
      else g = group_descent(dist, r, c, r + n0, c + n1, d2k, j);
      if (g.distance < cur.dist) { cur.mark = 1; cur.f0 = g.f0; cur.f1 = g.f1; cur.dist = g.distance; }   // :179-184
    }
  }
  const bool changed = cur.f0 != old.f0 || cur.f1 != old.f1 || cur.dist != old.dist || cur.mark != (old.mark & 0xFF);
  if (j == 0) {
    store_cell_sc1(Bcur + cell, Cell{cur.f0, cur.f1, cur.dist, cur.mark | (changed ? (k + 1) << kTagShift : 0)});
    if constexpr (MAPS) {
      int32_t* f = m.flow.row<int32_t>(ci) + 2 * cj;
      f[0] = cur.f0; f[1] = cur.f1; m.dist.row<int32_t>(ci)[cj] = cur.dist; m.mark.row<uint8_t>(ci)[cj] = (uint8_t)cur.mark;
    }
    if (stats == 1) { atomicAdd(&g_round_stats[1], 1u); if (need) atomicAdd(&g_round_stats[2], 1u); if (changed) atomicAdd(&g_round_stats[3], 1u); }
  }
  *changed_out = changed;
  int target = -1;
  if (changed) {   // next round: this cell (its value must reach the other buffer) and the marked cells that read it as an earlier neighbour
    if (j == (forward ? 0 : 7)) target = cell;                       // one of the earlier-neighbour lanes speaks for the cell itself
    else if (!earlier && in && nbm) target = q0 * NJ + q1;
  }
  return target;
}

template <int WS>
__global__ __launch_bounds__(256) void sdof_rounds_kernel(DImg i1, DImg i2, int ws, Maps m, int patch, int forward, int NI, int NJ, RoundArrays a, int stats) {
  __shared__ unsigned s_val, s_nreg;
  __shared__ unsigned long long s_gen;
  __shared__ uint4 s_union[kJobsPerGroup][kUnionRows];   // per 8-lane group: the candidate patch of a descent step (group_descent_staged)
  SweepCtl* const ctl = a.ctl;
  const int tid = threadIdx.x, j = tid & 7;
  unsigned n = ctl->count[0];   // complete: written by the classify kernel
  if (n == 0) return;
  const unsigned wanted = n <= (unsigned)kJobsPerGroup ? 1u : max(4u, (n + kJobsPerGroup - 1) / kJobsPerGroup);
  if (blockIdx.x >= wanted) return;
  if (tid == 0) s_val = __hip_atomic_fetch_add(&ctl->reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_val & kRegClosed) return;   // the others have already finished round 0: every job of it has been handed out
  const unsigned ticket = s_val;   // this workgroup's registration number
  __syncthreads();
  unsigned N = 0;   // registered workgroups (thread 0, known at the first barrier)
  unsigned nall = 0;   // the same for every thread, from the first barrier on
  const bool static_batches = stats != 2;   // (stats == 2: every round pops its batches from the counter, as round 0 does — kept for A/B)
  for (int k = 0;; k++) {
    const int par = k & 1;
    const Cell* __restrict__ Bprev = a.B[par ^ 1];
    Cell* __restrict__ Bcur = a.B[par];
    const uint32_t* __restrict__ Qcur = a.Q[par];
    unsigned batch = ticket;
    for (;;) {
      // Round 0 hands its batches out from a counter (a workgroup that is not resident yet must not own jobs: it may never register).  From round 1 on
      // the registered workgroups are known and all resident: workgroup `ticket` takes the batches ticket, ticket + N, ... — no atomic round trip in front
      // of a round's first loads and none behind its last batch.
      unsigned base;
      if (k == 0 || !static_batches) {
        if (tid == 0) s_val = __hip_atomic_fetch_add(&ctl->head[par], (unsigned)kJobsPerGroup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        base = s_val;
        __syncthreads();
      } else { base = batch * (unsigned)kJobsPerGroup; batch += nall; }
      if (base >= n) break;
      const unsigned job = base + (unsigned)(tid >> 3);
      int target = -1;   // the cell this lane wants in the next round's queue
      if (job < n) {
        bool changed;
        target = round_job<WS, true>(i1, i2, ws, m, patch, forward, NI, NJ, a, k, Bprev, Bcur, (int)load_u32_sc1(Qcur + job), j, s_union[tid >> 3], stats, true, &changed);
      }
      enqueue_targets(a, ctl, par ^ 1, target);
    }
    // ---- grid barrier over the registered workgroups; the last arriver publishes the size of the next round with the generation
    __syncthreads();
    if (tid == 0) {
      if (N == 0) {
        const unsigned o = __hip_atomic_fetch_or(&ctl->reg, kRegClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(o & kRegClosed)) { N = o; __hip_atomic_store(&ctl->nreg, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else for (unsigned spin = 0; (N = __hip_atomic_load(&ctl->nreg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && spin < kSpinLimit; spin++) __builtin_amdgcn_s_sleep(1);
        if (N == 0) { N = 1; raise_barrier_timeout(a); }   // cannot happen (the closer stores nreg right after closing); never hang the GPU on a bug
        s_nreg = N;
      }
      if (N > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned arrived = __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long g;
      if (arrived == N - 1) {
        const unsigned nn = __hip_atomic_load(&ctl->count[par ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->count[par], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->head[par], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g = ((unsigned long long)(k + 1) << 32) | nn;
        __hip_atomic_store(&ctl->gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stats == 1) atomicAdd(&g_round_stats[0], 1u);
      } else {
        unsigned spin = 0;
        while (((g = __hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (unsigned long long)(k + 1) && ++spin < kSpinLimit) __builtin_amdgcn_s_sleep(1);
        if (spin >= kSpinLimit) { g = 0; raise_barrier_timeout(a); }   // every registered workgroup is resident and arrives: a timeout is a bug — leave instead of hanging
      }
      if (N > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      s_gen = g;
    }
    __syncthreads();
    n = (unsigned)s_gen;
    if (k == 0) nall = s_nreg;
    __syncthreads();
    if (n == 0) break;
    // A round that fits one batch (the tail of a sweep: a handful of cells whose neighbours changed) is left to ONE workgroup, the first that
    // registered; what the others wrote was released and acquired at the barrier above.  Alone, its barriers are workgroup-scoped: no L2 write-back,
    // no L2 invalidation in front of the next round's loads (which made a round of 5 jobs cost as much as one of 1 500).
    if (n <= (unsigned)kJobsPerGroup && N != 1) {
      if (ticket != 0) return;
      N = 1; nall = 1;
    }
  }
}

// ---- a whole sweep in ONE launch (round 4) -------------------------------------------------------------------------------------------------------------
// sdof_classify_kernel + sdof_rounds_kernel are two dependent launches per sweep, 12 of the 18 of a 3-scale pair, and a dependent launch costs ~4.5 us
// whatever it does.  Fusing them behind an in-kernel grid barrier was measured slower (LABNOTES.md section 3, K12): on this chip the barrier costs more than
// the boundary.  This kernel has NO barrier on its common path:
//   * the cells' records (`pre`, B[0], B[1] of the scale: Mirrors) are kept equal to the maps BETWEEN sweeps by everyone who writes the maps — the reset
//     launch (mark 0), the descents, and the end of every sweep (the changed cells are written back) — so there is no copy pass, and the pre-sweep values
//     stay readable in `pre` while round 0 already writes B[0] and the maps;
//   * every workgroup classifies its own 256 cells against `pre` and runs ROUND 0 for its own candidates at once: S_0(c) = F(c; pre) reads pre-sweep
//     values only, so round 0 needs nothing from any other workgroup;
//   * a workgroup that is done arrives on a counter and LEAVES; the last one to arrive (it alone knows that round 0 is complete) runs the remaining
//     rounds by itself — on the scenes measured a sweep's later rounds are a handful of cells — with workgroup-scoped synchronisation, writes the
//     changed cells' final values back to the three record arrays and hands the control block back zeroed.
// Long tails (unrelated frames: thousands of changes per round, tens of rounds) must not run on one workgroup: a workgroup that finds more than
// kStayThreshold jobs queued for round 1 when it is done STAYS (at most kMaxStay do: the others, and with them the last arriver, always find a free
// slot — the wait cannot deadlock), and the rounds then run on the stayers + the last arriver behind the grid barrier of sdof_rounds_kernel.
// Every registered workgroup stays until the sweep is over (parked once a round fits one batch), takes its share of the write-back and acknowledges;
// ticket 0 zeroes the control block after the last acknowledgement.
[[maybe_unused]] constexpr unsigned kStayThreshold = kJobsPerGroup;   // (a round that fits one batch is run by one workgroup anyway)
constexpr unsigned kMaxStay = 48;
constexpr int kSweepTile = 16;   // a workgroup's cells: a 16 x 16 tile of the sweep domain (a motion boundary along a row of cells would hand one workgroup of 256 consecutive
                                 // cells 256 candidates, 8 passes of round 0 one after the other: measured 90 us for the middle scale of the 4K bench scene)
constexpr unsigned kGenFinal = 0xFFFFFFFFu;   // round field of the message that ends the sweep for parked workgroups

// The read-back of ONE keypoint (semi_dense_optical_flow.hpp:205-212; sdof_readback_kernel, and the fused sweep's read-back workgroups).  SC1: the maps are read
// with L1-bypassing loads (what the same launch's flush wrote write-through).  LINK (the tracker's step): the keypoint's match is also threaded onto the merge step's
// per-cell list (merge_link_one: the first pass of video_extruder.hpp:60-84 needs exactly what this thread holds — old and new position and whether it matched).
template <bool LINK, bool SC1>
__device__ __forceinline__ void readback_one(int i, const int32_t* __restrict__ kps, int div, int ms, const Maps& m, int32_t* __restrict__ out_pos, int32_t* __restrict__ out_dist,
                                             uint8_t* __restrict__ out_valid, const MergeLinkArgs& link) {
  const int k0 = kps[2 * i], k1 = kps[2 * i + 1];
  const int q0 = k0 / div, q1 = k1 / div;  // :207-208
  int o0 = k0, o1 = k1, d = 0; uint8_t v = 0;
  // (mark, flow and distance in one round trip: unconditional loads at clamped coordinates)
  const int c0 = min(max(q0, 0), m.mark.nr - 1), c1 = min(max(q1, 0), m.mark.nc - 1);
  uint8_t mk; int f0, f1, dd;
  if constexpr (SC1) {
    mk = __hip_atomic_load(m.mark.row<uint8_t>(c0) + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long ff = __hip_atomic_load((unsigned long long*)(m.flow.row<int32_t>(c0) + 2 * c1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f0 = (int)(unsigned)ff; f1 = (int)(unsigned)(ff >> 32);
    dd = (int)__hip_atomic_load((uint32_t*)m.dist.row<int32_t>(c0) + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    mk = m.mark.row<uint8_t>(c0)[c1];
    const int32_t* f = m.flow.row<int32_t>(c0) + 2 * c1;
    f0 = f[0]; f1 = f[1]; dd = m.dist.row<int32_t>(c0)[c1];
  }
  if (m.mark.has(q0, q1) && mk) { o0 = k0 + f0 * ms; o1 = k1 + f1 * ms; d = dd; v = 1; }  // :210-211
  out_pos[2 * i] = o0; out_pos[2 * i + 1] = o1; out_dist[i] = d; out_valid[i] = v;
  if constexpr (LINK) merge_link_one(link, i, o0, o1, k0, k1, v != 0);
}
// The read-back riding in the flow's LAST sweep launch (round 6): `blocks` extra workgroups behind the sweep's tiles.  They leave at once to their keypoints when the
// sweep is one that skips (the common case: the same word every tile tests), otherwise when the sweep's last workgroup has said so in `go` (the sweep's id; after
// the flush's write-through stores and every stayer's acknowledgement).  One dependent launch less per pair; the maps they read were written by earlier launches
// or by this launch's flush — write-through, and no workgroup of this launch reads a map before (no stale line in any XCD's L2).
struct ReadbackTail { const int32_t* kps; int n, div, ms; int32_t* out_pos; int32_t* out_dist; uint8_t* out_valid; MergeLinkArgs link; int blocks; unsigned* go; };

// NT threads per workgroup = NT / 8 jobs per pass (round 6).  512 where two such workgroups per CU hold every tile at once (at most 512 tiles; the kernel's ~106 VGPRs
// leave a SIMD 4 waves): round 0 of a tile with 33 ... 64 candidates is ONE pass instead of two one after the other — the passes are latency chains, not work, and on the
// 4K bench scene the middle scale's tiles that set round 0's length have 40-66 candidates (round 0 ends 15.7 instead of 21 us after the classification).  256 where the
// tiles are more (the finest 4K scale: 1 296).  1 024 threads (one workgroup per CU, a second generation of tiles) measured 0.197 against 0.180 ms per pair.
// RB: 0 = the sweep alone; 1 / 2 = with the read-back's workgroups behind its tiles (2: the tracker's LINK form)
// TS: the tile's edge in cells, 16 or 32 (round 6: 32 where 16 x 16 tiles would be more than 512 workgroups — the finest 4K scale, whose cells are a quarter marked:
// 336 workgroups of 512 threads instead of 1 296 of 256 to dispatch, classify and count in)
template <int WS, int NT = 256, int RB = 0, int TS = kSweepTile>
__global__ __launch_bounds__(NT) void sdof_sweep_kernel(DImg i1, DImg i2, int ws, Maps m, int patch, int forward, int NI, int NJ, RoundArrays a, int stats, int nsub, unsigned stay_above, unsigned sweep_id, int may_skip,
                                                        ReadbackTail rb) {
  const unsigned ntiles = RB ? gridDim.x - (unsigned)rb.blocks : gridDim.x;   // the sweep's own workgroups
  if constexpr (RB != 0) {
    if (blockIdx.x >= ntiles) {
      __shared__ unsigned s_go;
      if (threadIdx.x == 0) {
        const unsigned sw = may_skip ? load_u32_sc1(a.skip) : 0u;
        unsigned ok = may_skip && (sw == sweep_id || sw == sweep_id + 1u) ? 1u : 0u;   // a sweep that skips has nothing to wait for
        // (a few hundred workgroups poll one word while the sweep runs: a fifth of a microsecond apart, so that they do not crowd the sweep's own hand-offs out of that L2 channel)
        for (unsigned spin = 0; !ok && spin < kSpinLimit; spin++) { ok = load_u32_sc1(rb.go) == sweep_id ? 1u : 0u; if (!ok) __builtin_amdgcn_s_sleep(8); }
        if (!ok) raise_barrier_timeout(a);
        s_go = ok;
      }
      __syncthreads();
      if (!s_go) return;
      const int i = (int)(blockIdx.x - ntiles) * NT + (int)threadIdx.x;
      if (i < rb.n) readback_one<RB == 2, true>(i, rb.kps, rb.div, rb.ms, m, rb.out_pos, rb.out_dist, rb.out_valid, rb.link);
      return;
    }
  }
  __shared__ unsigned s_val, s_nreg, s_flags, s_ncand, s_flushn, s_giveup;
  __shared__ unsigned long long s_gen;
  constexpr int TP = TS + 2, CPT = (TS * TS + NT - 1) / NT, NHALO = 4 * TS + 4;   // tile pitch with the halo, cells per thread, halo entries
  static_assert(NHALO <= NT, "one halo entry per thread");
  __shared__ uint32_t s_cand[TS * TS];
  constexpr int JOBS = NT / 8;
  __shared__ uint4 s_union[JOBS][kUnionRows];   // per 8-lane group: the candidate patch of a descent step (group_descent_staged)
  __shared__ uint4 s_tile[TP * TP];       // the `pre` records of the workgroup's tile + halo
  SweepCtl* const ctl = a.ctl;
  const int tid = threadIdx.x, j = tid & 7;
  if (stats == 1 && blockIdx.x == 0 && tid == 0) sweep_log(255u, ntiles);
  if (tid == 0) { s_ncand = 0; s_giveup = 0; }
  // (round 5) A sweep that found no candidate at all changed nothing, so the next sweep of the same scale (same test, same maps) finds none either: the earlier
  // sweep says so in a.skip and this one returns after one load instead of classifying, arriving and handing the control block back (9 us -> the launch's own cost).
  const unsigned skip_word = may_skip ? load_u32_sc1(a.skip) : 0u;   // (every lane: one address, one broadcast)
  // This workgroup's TS x TS cells and their 1-cell halo, `pre` records, into LDS: a thread's own cell(s) + (threads 0 .. 4 TS + 3) one halo cell.  Requested BEFORE the skip
  // word is looked at (one round trip for both), read by the classification below and by round 0's jobs (round 6: the classification used to load a marked cell's 8
  // neighbours behind the cell's own record, and every job of round 0 its records again: two and three dependent round trips).
  const int tiles_x = (NJ + TS - 1) / TS;
  const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
  int hi_ = 0, hj_ = 0;   // the halo entry this thread fills
  if (tid < TP) { hi_ = 0; hj_ = tid; } else if (tid < 2 * TP) { hi_ = TP - 1; hj_ = tid - TP; }
  else if (tid < 2 * TP + TS) { hi_ = 1 + tid - 2 * TP; hj_ = 0; } else { hi_ = 1 + tid - 2 * TP - TS; hj_ = TP - 1; }
  const int gi = ty * TS - 1 + hi_, gj = tx * TS - 1 + hj_;
  const bool halo_in = tid < NHALO && gi >= 0 && gj >= 0 && gi < NI && gj < NJ;
  uint4 own_rec[CPT];   // (cells past the tile / the domain: a clamped address, dropped)
#pragma unroll
  for (int q = 0; q < CPT; q++) {
    const int cell_l = tid + q * NT, li = cell_l / TS, lj = cell_l - li * TS;
    own_rec[q] = *(const uint4*)(a.pre + (size_t)min(ty * TS + li, NI - 1) * NJ + min(tx * TS + lj, NJ - 1));
  }
  uint4 halo_rec = *(const uint4*)(a.pre + (size_t)min(max(gi, 0), NI - 1) * NJ + min(max(gj, 0), NJ - 1));
  asm volatile("" ::: "memory");   // (the record loads stay above the skip test)
  __syncthreads();
  if (may_skip && (skip_word == sweep_id || skip_word == sweep_id + 1u)) {   // (+ 1: what workgroup 0 of this very launch writes below — every workgroup decides alike)
    if (blockIdx.x == 0 && tid == 0) { store_u32_sc1(a.skip, sweep_id + 1u); if (stats == 1) sweep_log(250u, ntiles); }   // still nothing: the next one may skip as well
    return;
  }
#pragma unroll
  for (int q = 0; q < CPT; q++) {
    const int cell_l = tid + q * NT, li = cell_l / TS, lj = cell_l - li * TS;
    if (!(cell_l < TS * TS && ty * TS + li < NI && tx * TS + lj < NJ)) own_rec[q] = make_uint4(0u, 0u, 0u, 0u);
    if (cell_l < TS * TS) s_tile[(li + 1) * TP + lj + 1] = own_rec[q];
  }
  if (!halo_in) halo_rec = make_uint4(0u, 0u, 0u, 0u);
  if (tid < NHALO) s_tile[hi_ * TP + hj_] = halo_rec;
  __syncthreads();
  // ---- this workgroup's cells: which of them can loop_body change at all (sdof_classify_kernel's test, on the records)
#pragma unroll
  for (int q = 0; q < CPT; q++) {
    const int cell_l = tid + q * NT, li = min(cell_l / TS, TS - 1), lj = cell_l - (cell_l / TS) * TS;
    const int idx = (ty * TS + li) * NJ + tx * TS + lj;
    bool cand = false;
    const Cell cur = cell_of(own_rec[q]);
    if (cur.mark & 0xFF) {   // (an out-of-domain / out-of-tile entry is unmarked)
#pragma unroll
      for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++) {
          if (!dr && !dc) continue;
          const Cell nb = cell_of(s_tile[(li + 1 + dr) * TP + lj + 1 + dc]);
          const int a0 = cur.f0 - nb.f0, a1 = cur.f1 - nb.f1;
          if ((nb.mark & 0xFF) && a0 * a0 + a1 * a1 >= 9) cand = true;
        }
    }
    const unsigned long long b = __ballot(cand);
    if (b) {
      const int lane = __lane_id(), leader = __ffsll((long long)b) - 1;
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(&s_ncand, (unsigned)__popcll(b));
      base = __shfl(base, leader);
      if (cand) s_cand[base + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)idx;
    }
  }
  __syncthreads();
  const unsigned n0 = s_ncand;
  if (tid == 0 && n0) (void)__hip_atomic_fetch_add(&ctl->ncand, n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (waited for with this workgroup's other accesses before it arrives)
  if (stats == 1 && tid == 0 && n0) sweep_log(253u, n0);
  // ---- round 0 on this workgroup's candidates: reads `pre` (and B[1], equal to it), writes B[0], round 1's queue and the list of changes
  for (unsigned base = 0; base < n0; base += (unsigned)JOBS) {
    const unsigned job = base + (unsigned)(tid >> 3);
    int target = -1, chg = -1;
    if (job < n0) {
      const int cell = (int)s_cand[job];
      bool changed;
      const int qi = cell / NJ, qj = cell - qi * NJ;
      target = round_job<WS, false, true, TP>(i1, i2, ws, m, patch, forward, NI, NJ, a, 0, a.B[1], a.B[0], cell, j, s_union[tid >> 3], stats, false, &changed,
                                              s_tile + (qi - ty * TS + 1) * TP + (qj - tx * TS + 1));
      if (changed && j == (forward ? 1 : 6)) chg = cell;   // (a lane that never carries a queue target: round_job's are the cell's own lane 0 / 7 and the LATER neighbours' lanes)
    }
    append_either(a.qflag[1], &ctl->count[1], a.Q[1], target, a.cflag, &ctl->nchanged, a.chg, chg);
  }
  // ---- arrive; everybody but the last arriver and the stayers leaves
  // (the queue's length — only whether it is worth staying — is requested with the pass's last stores, not behind them: a round trip less on the path of the tile that
  // finishes last; it may miss what this workgroup's other waves are still appending, which the decision can live with)
  unsigned queued = 0;
  if (tid == 0 && n0) queued = __hip_atomic_load(&ctl->count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (stats == 1 && tid == 0 && n0) sweep_log(252u, n0);
  if (tid == 0) {
    unsigned flags = 0, ticket = 0;
    // (what this workgroup wrote for the others — records, queue, lists — went out as write-through stores and every wave has waited for its own: no release)
    // (only a workgroup that had candidates itself looks at the queue: the others — nearly all of them at the finest scale — arrive one memory round trip earlier)
    if (n0 && queued > stay_above) {
      const unsigned t = __hip_atomic_fetch_add(&ctl->reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // before the arrival: the last arriver reads the final count
      // The registration must be PERFORMED before this workgroup's arrival below is: the two atomics go to different cache lines (different channels), and nothing
      // else orders them — an arrival that overtook the registration would let the last arriver read a short `reg` and publish a wrong nreg.  A returning atomic
      // has been performed when its value is back: wait for it (stayers only; the asm form, because the compiler drops waits it believes redundant).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t < kMaxStay) { ticket = t; flags |= 1u; }
    }
    // Two-level arrival: returning atomics on ONE word retire at ~11 ns each (1 300 workgroups at the finest 4K scale: 14 us, measured 30 us per empty sweep); the
    // workgroups arrive on nsub counters, 128 bytes apart, and the last one of each (which also hands its counter back zeroed) on ctl->done.
    const unsigned sc = blockIdx.x % (unsigned)nsub, expect = ntiles / (unsigned)nsub + (sc < ntiles % (unsigned)nsub ? 1u : 0u);
    if (__hip_atomic_fetch_add(&a.sub[sc * kSubStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1) {
      __hip_atomic_store(&a.sub[sc * kSubStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsub - 1) flags |= 2u;
    }
    s_flags = flags; s_val = ticket;
  }
  __syncthreads();
  const unsigned flags = s_flags;
  if (!flags) return;
  if (tid == 0) {
    unsigned ticket = s_val;
    unsigned long long g;
    if (flags & 2u) {   // round 0 is complete everywhere (what the others wrote is read with L1-bypassing loads: no acquire)
      const unsigned n1 = __hip_atomic_load(&ctl->count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      store_u32_sc1(a.skip, __hip_atomic_load(&ctl->ncand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u ? sweep_id + 1u : 0u);   // for the next sweep of this scale
      const unsigned R = min(__hip_atomic_load(&ctl->reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), kMaxStay);
      unsigned N = R;
      if (!(flags & 1u)) { ticket = R; N = R + 1; }
      __hip_atomic_store(&ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      g = (1ull << 32) | n1;
      if (N > 1) {
        __hip_atomic_store(&ctl->nreg, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->flushn, __hip_atomic_load(&ctl->nchanged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&ctl->gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_nreg = N;
      if (stats == 1) { atomicAdd(&g_round_stats[0], 1u); sweep_log(1u, n1); sweep_log(254u, N); }
    } else {   // a stayer: the last arriver publishes the size of round 1
      unsigned spin = 0;
      while (((g = __hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != 1ull && (g >> 32) != (unsigned long long)kGenFinal && ++spin < kSpinLimit)
        __builtin_amdgcn_s_sleep(1);
      if (spin >= kSpinLimit) { raise_barrier_timeout(a); s_giveup = 1; }
      s_nreg = __hip_atomic_load(&ctl->nreg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_gen = g; s_val = ticket;
  }
  __syncthreads();
  if (s_giveup) return;   // gave up waiting (a bug or a dead GPU): leave without touching anything
  const unsigned N0 = s_nreg, ticket = s_val;
  unsigned n = (unsigned)s_gen;
  bool solo = N0 == 1, parked = (s_gen >> 32) == (unsigned long long)kGenFinal;   // (a message this workgroup was too slow to read was followed by the sweep's last one)
  if (parked) n = 0;
  __syncthreads();
  for (int k = 1; n != 0; k++) {
    // A round that fits one batch is left to ticket 0 alone (workgroup-scoped synchronisation: no L2 write-back, no invalidation in front of the next
    // round's loads); the others wait for the end of the sweep.  Once alone, always alone: longer lists are walked in passes.
    if (!solo && n <= min(stay_above, (unsigned)JOBS)) {
      if (ticket != 0) { parked = true; break; }
      solo = true;
    }
    const int par = k & 1;
    const Cell* __restrict__ Bprev = a.B[par ^ 1];
    Cell* __restrict__ Bcur = a.B[par];
    const uint32_t* __restrict__ Qcur = a.Q[par];
    const unsigned stride = solo ? 1u : N0;
    for (unsigned batch = solo ? 0u : ticket; batch * (unsigned)JOBS < n; batch += stride) {
      const unsigned job = batch * (unsigned)JOBS + (unsigned)(tid >> 3);
      int target = -1, chg = -1;
      if (job < n) {
        const int cell = (int)load_u32_sc1(Qcur + job);
        bool changed;
        target = round_job<WS, false>(i1, i2, ws, m, patch, forward, NI, NJ, a, k, Bprev, Bcur, cell, j, s_union[tid >> 3], stats, true, &changed);
        if (changed && j == (forward ? 1 : 6)) chg = cell;
      }
      append_either(a.qflag[par ^ 1], &ctl->count[par ^ 1], a.Q[par ^ 1], target, a.cflag, &ctl->nchanged, a.chg, chg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stats == 1 && tid == 0) sweep_log(251u, (unsigned)k | (ticket << 8));
    if (tid == 0) {
      unsigned long long g;
      if (solo) {   // this workgroup's own stores and atomics only: in order, through its own L1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const unsigned nn = __hip_atomic_load(&ctl->count[par ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->count[par], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        g = nn;
        if (stats == 1) { atomicAdd(&g_round_stats[0], 1u); sweep_log((unsigned)(k + 1), nn); }
      } else {   // the grid barrier of sdof_rounds_kernel over the N0 registered workgroups, without its fences (write-through stores, L1-bypassing loads)
        const unsigned arrived = __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == N0 - 1) {
          const unsigned nn = __hip_atomic_load(&ctl->count[par ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ctl->count[par], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ctl->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ctl->flushn, __hip_atomic_load(&ctl->nchanged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          g = ((unsigned long long)(k + 1) << 32) | nn;
          __hip_atomic_store(&ctl->gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (stats == 1) { atomicAdd(&g_round_stats[0], 1u); sweep_log((unsigned)(k + 1), nn); }
        } else {
          unsigned spin = 0;
          // (the sweep's last message instead: ticket 0, alone since this barrier, was done before this workgroup read the barrier's own message)
          while (((g = __hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (unsigned long long)(k + 1) && (g >> 32) != (unsigned long long)kGenFinal &&
                 ++spin < kSpinLimit)
            __builtin_amdgcn_s_sleep(1);
          if (spin >= kSpinLimit) { g = 0; raise_barrier_timeout(a); s_giveup = 1; }
        }
      }
      s_gen = g;
    }
    __syncthreads();
    n = (unsigned)s_gen;
    const bool final_seen = (s_gen >> 32) == (unsigned long long)kGenFinal, gave_up = s_giveup != 0;
    __syncthreads();
    if (gave_up) return;
    if (final_seen) { parked = true; break; }
  }
  // ---- the sweep is over (or this workgroup is parked until it is): the changed cells' final values go back into the three record arrays, tags cleared
  if (tid == 0) {
    unsigned fn;
    if (parked) {
      unsigned long long g;
      unsigned spin = 0;
      while (((g = __hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (unsigned long long)kGenFinal && ++spin < kSpinLimit) __builtin_amdgcn_s_sleep(1);
      if (spin >= kSpinLimit) { raise_barrier_timeout(a); s_giveup = 1; }
      fn = __hip_atomic_load(&ctl->flushn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (solo) {
      fn = __hip_atomic_load(&ctl->nchanged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (N0 > 1) {   // wake the parked workgroups (every wave of this workgroup has waited for its write-through stores at the round's end)
        __hip_atomic_store(&ctl->flushn, fn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&ctl->gen, (unsigned long long)kGenFinal << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else fn = __hip_atomic_load(&ctl->flushn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // published with the barrier's last message (n == 0)
    s_flushn = fn;
  }
  __syncthreads();
  if (s_giveup) return;
  const unsigned fn = s_flushn;
  for (unsigned i = ticket * (unsigned)NT + (unsigned)tid; i < fn; i += N0 * (unsigned)NT) {
    const uint32_t cell = load_u32_sc1(a.chg + i);
    Cell c = load_cell_sc1(a.B[0] + cell);   // a changed cell is carried into the round after its change: both round buffers hold its final value
    c.mark &= 0xFF;
    store_cell16(a.pre + cell, c); store_cell16(a.B[0] + cell, c); store_cell16(a.B[1] + cell, c);   // (read by later launches only)
    const int ci = (int)cell / NJ, cj = (int)cell - ci * NJ;
    int32_t* f = m.flow.row<int32_t>(ci) + 2 * cj;   // the maps: written here only, one writer per cell
    if constexpr (RB != 0) {   // (write-through: this launch's read-back workgroups read them)
      __hip_atomic_store((unsigned long long*)f, ((unsigned long long)(unsigned)c.f1 << 32) | (unsigned)c.f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((uint32_t*)m.dist.row<int32_t>(ci) + cj, (uint32_t)c.dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(m.mark.row<uint8_t>(ci) + cj, (uint8_t)c.mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { f[0] = c.f0; f[1] = c.f1; m.dist.row<int32_t>(ci)[cj] = c.dist; m.mark.row<uint8_t>(ci)[cj] = (uint8_t)c.mark; }
    a.cflag[cell] = 0;
  }
  if (ticket != 0) {   // registered and done: ticket 0 may hand the control block back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&ctl->ack, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (tid == 0 && N0 > 1) {
    unsigned spin = 0;
    while (__hip_atomic_load(&ctl->ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != N0 - 1 && ++spin < kSpinLimit) __builtin_amdgcn_s_sleep(1);
    if (spin >= kSpinLimit) raise_barrier_timeout(a);
  }
  if constexpr (RB != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's own share of the flush has been performed
  __syncthreads();
  if (RB != 0 && tid == 0) store_u32_sc1(rb.go, sweep_id);   // the sweep is over, every changed cell is in the maps: the read-back workgroups may go
  if (tid < (int)(sizeof(SweepCtl) / 4)) ((unsigned*)ctl)[tid] = 0u;   // zero between sweeps
}

template <bool LINK>
__global__ __launch_bounds__(256) void sdof_readback_kernel(const int32_t* __restrict__ kps, int n, int div, int ms, Maps m,
                                                            int32_t* __restrict__ out_pos, int32_t* __restrict__ out_dist, uint8_t* __restrict__ out_valid, MergeLinkArgs link) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) readback_one<LINK, false>(i, kps, div, ms, m, out_pos, out_dist, out_valid, link);
}

thread_local Scratch g_scratch;

struct Carver {
  uint8_t* base; size_t off = 0;
  vpp_image_desc image(int nr, int nc, int dtype, int ch, int border) {
    int32_t pitch; size_t bytes, first;
    vpp_image_layout(nr, nc, dtype_size(dtype) * ch, border, 32, &pitch, &bytes, &first);
    vpp_image_desc d{base ? base + off + first : nullptr, nr, nc, pitch, border, dtype, ch};
    last_off = off; last_bytes = (bytes + 255) / 256 * 256;
    off += last_bytes;
    return d;
  }
  size_t last_off = 0, last_bytes = 0;   // the block of the image carved last (whole allocation, a multiple of 256 bytes)
};

// (ResetArgs, ClaimAll, reset_claim_block: sdof_tail.hpp — the pyramid pair launch can carry these blocks)
__global__ __launch_bounds__(256) void sdof_reset_kernel(ResetArgs a) {
  int sgm = 0;
  while (sgm + 1 < a.nseg && blockIdx.x >= a.first_block[sgm + 1]) sgm++;
  const uint32_t u = (blockIdx.x - a.first_block[sgm]) * 256 + threadIdx.x;
  if (u < a.units[sgm]) { const uint32_t v = a.value[sgm]; a.p[sgm][u] = make_uint4(v, v, v, v); }
}
// The mark maps' reset and the claims of every scale in ONE launch: the first blocks fill the mark blocks, the others claim — the two touch different maps.
// The owner maps need no reset here: every descent hands its cell back empty (clean_owner), so a call finds them as the previous one left them.
__global__ __launch_bounds__(256) void sdof_reset_claim_kernel(ResetArgs a, const int32_t* __restrict__ kps, int n, int patch, ClaimAll c) {
  reset_claim_block(a, kps, n, patch, c, blockIdx.x);
}

}  // namespace

// Row-strip sharding of the per-keypoint phases (SURVEY 8e bullet 2), in the shape a multi-GPU run has: strip k > 0 owns the flow-map
// rows [fr k / S, fr (k + 1) / S) of every scale, has PRIVATE flow / mark / distance / owner maps and its own stream, and runs claim +
// descent for the keypoints that fall into its rows against the shared (replicated) image pyramids and its own copy of the coarser
// scale's final maps.  Exchanges, here device copies on the owner's stream (RCCL gather / broadcast of the same rows across GPUs): after
// the descent the strips' rows are gathered into the owner's (strip 0's) maps, the owner runs the Jacobi pre-passes and the ordered sweeps
// (their critical path is the wavefront over the whole map whatever the split), and the swept maps are broadcast back as the next finer
// scale's prediction.  Every cell is written by exactly one strip from the same inputs as the single-strip run, so the result is identical.
namespace {
struct StripStreams {
  std::vector<hipStream_t> st; std::vector<hipEvent_t> done; hipEvent_t start = nullptr;
  int ensure(int n) {
    while ((int)st.size() < n) {
      hipStream_t s; hipEvent_t e;
      VPP_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      VPP_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      st.push_back(s); done.push_back(e);
    }
    if (!start) VPP_HIP_TRY(hipEventCreateWithFlags(&start, hipEventDisableTiming));
    return VPP_OK;
  }
};
thread_local StripStreams g_strips;
// rows [lo, hi) of a map image, full pitch-wide rows, from one set of maps to another of the same layout
int copy_rows(const vpp_image_desc* dst, const vpp_image_desc* src, int lo, int hi, hipStream_t st) {
  if (hi <= lo) return VPP_OK;
  const ptrdiff_t off = (ptrdiff_t)lo * src->pitch - (ptrdiff_t)src->border * elem_bytes(src);
  VPP_HIP_TRY(hipMemcpyAsync((uint8_t*)dst->first_pixel + off, (const uint8_t*)src->first_pixel + off, (size_t)(hi - lo) * src->pitch, hipMemcpyDeviceToDevice, st));
  return VPP_OK;
}
}  // namespace

extern "C" int vpp_pyramid_build_pair(const vpp_image_desc* levels_a, const vpp_image_desc* src_a, const vpp_image_desc* levels_b, const vpp_image_desc* src_b, int nlevels, void* stream);   // pyramid_fused.hip

namespace {
// comm != nullptr: one process per GPU (vpp_semi_dense_optical_flow_sharded).  Rank g owns the flow-map rows [g per, (g + 1) per) of every
// scale (per = ceil(rows / ranks)): it claims and descends only the keypoints of its rows, the ranks' rows of the three maps are then
// completed on every rank by ONE grouped in-place RCCL all-gather per scale (the maps are carved with per * ranks rows of memory, so that a
// rank's rows are one contiguous chunk at rank * chunk), and every rank runs the propagation rounds on the complete maps — they are
// deterministic and cost a few tens of microseconds, less than broadcasting their result.  Frames and pyramids are replicated.
int flow_impl(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n, int winsize,
              int nscales, int min_scale, int propagation, int patchsize, int nstrips, vpp_comm* comm, int32_t* out_pos,
              int32_t* out_dist, uint8_t* out_valid, void* stream, const vpp_image_desc* pre1 = nullptr, const vpp_image_desc* pre2 = nullptr,
              const MergeLinkArgs* link = nullptr, size_t link_head_units = 0, const vpp_image_desc* build2_levels = nullptr, const vpp_image_desc* build2_src = nullptr) {
  // link != nullptr (the tracker): the merge lists' heads are reset with the maps and the read-back threads every match onto its cell's list
  // build2_src != nullptr (the tracker's one-call-per-frame form, with pre1 / pre2): pre2's levels are not built yet — this call builds them from the frame `build2_src`
  // (u8 x1, or x3 / x4 through the ingest) into `build2_levels` (pre2's levels with the border to fill), in its first launch where that launch can carry them
  // pre1 / pre2 != nullptr (vpp_semi_dense_optical_flow_pyramids): the caller's pyramids of the two frames are used as they are, i1 / i2 are their levels 0
  VPP_REQUIRE(valid_desc(i1) && valid_desc(i2) && same_domain(i1, i2), VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow: invalid frames");
  VPP_REQUIRE(i1->dtype == VPP_U8 && i1->channels == 1 && i2->dtype == VPP_U8 && i2->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_semi_dense_optical_flow: u8 x1 frames only");
  VPP_REQUIRE(n >= 0 && (n == 0 || (kps && out_pos && out_dist && out_valid)), VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow: null buffer");
  VPP_REQUIRE(winsize > 0 && patchsize > 0 && nscales >= 1 && nscales <= kMaxScales && min_scale >= 0 && min_scale < nscales && propagation >= 0,
              VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow: bad parameters");
  VPP_REQUIRE(nstrips >= 1 && nstrips <= 16, VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow_strips: 1 to 16 strips");
  if (n == 0) return VPP_OK;
  hipStream_t st = as_stream(stream);
  int rank = 0, world = 1;
  if (comm) { int rc = comm_info(comm, &rank, &world); if (rc) return rc; }
  // carve the scratch: two image pyramids (border 2*winsize, :72-73), flow / mark / distance pyramids (border nscales, :70-74) and owners per strip
  vpp_image_desc P1[kMaxScales], P2[kMaxScales];
  std::vector<vpp_image_desc> FLs((size_t)nstrips * kMaxScales), MKs(FLs.size()), DMs(FLs.size()), OWs(FLs.size());
  auto FL = [&](int k, int s_) -> vpp_image_desc& { return FLs[(size_t)k * kMaxScales + s_]; };
  auto MK = [&](int k, int s_) -> vpp_image_desc& { return MKs[(size_t)k * kMaxScales + s_]; };
  auto DM = [&](int k, int s_) -> vpp_image_desc& { return DMs[(size_t)k * kMaxScales + s_]; };
  auto OW = [&](int k, int s_) -> vpp_image_desc& { return OWs[(size_t)k * kMaxScales + s_]; };
  RoundArrays ra{};   // the fixed-point rounds' queues and flags, sized for the finest scale; the cells' records per scale (rec_*)
  Cell* rec[kMaxScales][3] = {};   // pre, B[0], B[1] of a scale: one block; `pre` is zeroed by the reset launch when the sweeps are fused (Mirrors)
  size_t rec_off[kMaxScales] = {}, rec_bytes[kMaxScales] = {};
  size_t ra_flags_off = 0, ra_flags_bytes = 0;
  size_t mk_off[kMaxScales] = {}, mk_bytes[kMaxScales] = {}, ow_off[kMaxScales] = {}, ow_bytes[kMaxScales] = {};   // strip 0's mark / owner blocks
  for (int pass = 0; pass < 2; pass++) {
    Carver cv{pass ? (uint8_t*)g_scratch.p : nullptr};
    int fr = i1->nrows / patchsize, fc = i1->ncols / patchsize, ir = i1->nrows, ic = i1->ncols;
    VPP_REQUIRE(fr > 0 && fc > 0, VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow: image smaller than one patch");
    for (int s_ = 0; s_ < nscales; s_++) {
      if (!pre1) { P1[s_] = cv.image(ir, ic, VPP_U8, 1, 2 * winsize); P2[s_] = cv.image(ir, ic, VPP_U8, 1, 2 * winsize); }
      else P1[s_] = P2[s_] = vpp_image_desc{nullptr, ir, ic, 0, 0, VPP_U8, 1};   // the level's size, checked against the caller's below
      const int fr_mem = (fr + world - 1) / world * world;   // rows of memory: whole chunks per rank (the descriptors keep fr rows)
      for (int k = 0; k < nstrips; k++) {
        FL(k, s_) = cv.image(fr_mem, fc, VPP_I32, 2, nscales); FL(k, s_).nrows = fr;
        MK(k, s_) = cv.image(fr_mem, fc, VPP_U8, 1, nscales); MK(k, s_).nrows = fr; if (k == 0) { mk_off[s_] = cv.last_off; mk_bytes[s_] = cv.last_bytes; }
        DM(k, s_) = cv.image(fr_mem, fc, VPP_I32, 1, nscales); DM(k, s_).nrows = fr;
        OW(k, s_) = cv.image(fr, fc, VPP_U32, 1, 0); if (k == 0) { ow_off[s_] = cv.last_off; ow_bytes[s_] = cv.last_bytes; }
      }
      {  // the scale's cell records: one entry per cell of its sweep domain (:192-200)
        const size_t cells = (size_t)((ir - 1) / patchsize + 1) * ((ic - 1) / patchsize + 1), each = (cells * sizeof(Cell) + 255) / 256 * 256;
        rec_off[s_] = cv.off; rec_bytes[s_] = 3 * each;
        for (int q = 0; q < 3; q++) { rec[s_][q] = cv.base ? (Cell*)(cv.base + cv.off) : nullptr; cv.off += each; }
      }
      fr = 1 + fr / 2; fc = 1 + fc / 2; ir = 1 + ir / 2; ic = 1 + ic / 2;  // pyramid.hh:154
    }
    {  // the rounds' buffers: one entry per cell of the finest scale's sweep domain
      const size_t cells = (size_t)((i1->nrows - 1) / patchsize + 1) * ((i1->ncols - 1) / patchsize + 1);
      auto take = [&](size_t bytes) { uint8_t* q = cv.base ? cv.base + cv.off : nullptr; cv.off += (bytes + 255) / 256 * 256; return q; };
      ra.Q[0] = (uint32_t*)take(cells * 4); ra.Q[1] = (uint32_t*)take(cells * 4); ra.chg = (uint32_t*)take(cells * 4);
      ra_flags_off = cv.off;   // zero between sweeps: queue / change flags (self-cleaning) and the control block
      ra.qflag[0] = (uint32_t*)take(cells * 4); ra.qflag[1] = (uint32_t*)take(cells * 4); ra.cflag = (uint32_t*)take(cells * 4); ra.sub = (unsigned*)take((size_t)kSubCounters * kSubStride * 4); ra.ctl = (SweepCtl*)take(sizeof(SweepCtl)); ra.skip = (unsigned*)take(8);   // [0] the skip word, [1] the read-back tail's go word
      ra_flags_bytes = cv.off - ra_flags_off;
      ra.err = device_error_word();
    }
    if (!pass) { int rc = g_scratch.ensure(cv.off, st); if (rc != VPP_OK) return rc; }
  }
  // a device-side fault nobody has collected yet (a caller that synchronises outside this ABI never does): whatever the faulting kernel left in ANY scratch
  // region is unknown — stop believing the notes (the bits stay up for the next vpp_sync to report)
  if (peek_device_error()) invalidate_scratch_notes();
  if (nstrips > 1) { int rc = g_strips.ensure(nstrips - 1); if (rc != VPP_OK) return rc; }
  // single strip: the mark and owner maps of all scales are reset here in one launch (nothing writes a scale's maps before its own phase)
  const bool reset_up_front = nstrips == 1 && 3 * (nscales - min_scale) + 2 <= kResetSegs && tuning("sdof.reset_up_front", 1);
  unsigned* const skip_go_words = ra.skip;   // [0] the sweeps' skip word, [1] the read-back tail's go word: zeroed with the maps at the start of every pair
  const bool claim_up_front = reset_up_front && world == 1 && tuning("sdof.claim_up_front", 1);
  // Self-cleaning owner maps (single strip, single rank): every descent hands its cell back empty, so the owner maps are not part of the reset and the
  // mark reset shares ONE launch with the claims.  The slot's note says whether the maps of this layout were left clean by a call that ran to its end;
  // anything else (first use, another layout, another code path, a call that returned early) gets a plain 0xFF fill first.
  const bool self_cleaning = claim_up_front && tuning("sdof.self_cleaning_owner", 1);
  // Fused sweeps (sdof_sweep_kernel): single strip, single rank (the strips' and ranks' rows of the maps are completed by copies / all-gathers that know
  // nothing of the records), 8 lanes per keypoint in the descent (the kernel that writes the records)
  const bool fused_sweeps = reset_up_front && world == 1 && propagation > 0 && tuning("sdof.descent_lanes", 8) == 8 && tuning("sdof.propagate", 0) != 1 &&
                            tuning("sdof.fused_sweep", 1);
  {  // the queue flags clean themselves up and the control block is left zeroed by every sweep: zeroed here once per (buffer, layout).  A call that is
     // being recorded into a launch graph, and every call on a buffer that a graph has been recorded on, carries the reset itself (Scratch::Slot::note):
     // a replay runs between arbitrary other calls, whose layouts may have left anything in this region.
    // (Reset by a kernel of this file, not by hipMemsetAsync: recorded into a launch graph the runtime's memset nodes were not ordered with the kernel
    // nodes around them on ROCm 7.2 — replays read a half-zeroed control block and faulted; tests/test_gpu_sdof.py::test_sdof_graph_replays_between_...)
    Scratch::Slot& sl = *g_scratch.cur;
    // (per kind of sweep: the two-launch sweeps leave reg / nreg / gen of the control block behind — their classify pass resets them — the fused sweep expects zeros)
    const unsigned long long sig = ((unsigned long long)ra_flags_off << 24) ^ (unsigned long long)ra_flags_bytes ^ 1ull ^ (fused_sweeps ? 1ull << 62 : 0ull);
    if (!sl.note(0, sig)) {
      ResetArgs z; z.nseg = 1; z.p[0] = (uint4*)((uint8_t*)g_scratch.p + ra_flags_off); z.units[0] = (uint32_t)(ra_flags_bytes / 16); z.value[0] = 0u;
      z.first_block[0] = 0; z.first_block[1] = (z.units[0] + 255) / 256;
      sdof_reset_kernel<<<z.first_block[1], 256, 0, st>>>(z);
      VPP_LAUNCH_CHECK();
      sl.set_note(0, sig);
    }
  }
  // the image pyramids: built once here; across GPUs every rank builds them from the broadcast frames (pyramid::update, pyramid.hh:194-198)
  // The levels are laid out with the reference's border of 2 * winsize (:72-73), but only winsize / 2 pixels beyond a level's domain are ever read into a SAD
  // (window centres must lie inside the domain, :102-108): the mirror fill is limited to that (>= 2: the per-level low-pass of other pyramid depths reads 2) —
  // with the full 18 pixels three rings of the one-launch pyramid kernel's tiles took its slow edge path (16.4 vs 13.5 us per 4K pyramid).
  int rc = VPP_OK;
  bool pyr_pending = false;
  vpp_image_desc B1[kMaxScales], B2[kMaxScales];
  if (pre1) {
    for (int s_ = 0; s_ < nscales; s_++) {   // the levels must be the ones pyramid.hh:154 halves to, with the pixels a SAD can reach (see above) in their border
      for (const vpp_image_desc* L : {&pre1[s_], &pre2[s_]})
        VPP_REQUIRE(valid_desc(L) && L->dtype == VPP_U8 && L->channels == 1 && L->nrows == P1[s_].nrows && L->ncols == P1[s_].ncols && L->border >= winsize / 2,
                    VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow_pyramids: level %d: u8 x1, %d x %d with a filled border of %d at least expected", s_,
                    P1[s_].nrows, P1[s_].ncols, winsize / 2);
      P1[s_] = pre1[s_]; P2[s_] = pre2[s_];
    }
    if (build2_src) {
      VPP_REQUIRE(build2_levels && valid_desc(build2_src), VPP_ERR_INVALID_ARG, "semi-dense flow: the second pyramid's levels / frame are missing");
      pyr_pending = self_cleaning && nscales == 3 && tuning("sdof.tail_in_pyramid", 1);
      if (!pyr_pending) {
        rc = build2_src->channels == 1 ? vpp_pyramid_build(build2_levels, nscales, build2_src, stream) : vpp_rgb_pyramid_build(build2_levels, nscales, build2_src, stream);
        if (rc) return rc;
      }
    }
  } else {
    for (int s_ = 0; s_ < nscales; s_++) {
      B1[s_] = P1[s_]; B2[s_] = P2[s_];
      B1[s_].border = B2[s_].border = std::min(P1[s_].border, std::max(winsize / 2, 2));
    }
    // (round 5) the mark reset + claims of all scales are independent of the pyramids: where they are one launch of their own (self_cleaning) they ride in the
    // pyramids' launch as extra workgroups — queued below, where their arguments are known
    pyr_pending = self_cleaning && nscales == 3 && tuning("sdof.tail_in_pyramid", 1);
    if (!pyr_pending) { rc = vpp_pyramid_build_pair(B1, i1, B2, i2, nscales, stream); if (rc) return rc; }   // one launch for both when the packed kernel takes them
  }
  auto maps = [&](int k, int s_) { return Maps{dimg(&FL(k, s_)), dimg(&MK(k, s_)), dimg(&DM(k, s_))}; };
  Scratch::Slot& slot = *g_scratch.cur;
  unsigned long long owner_sig = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(nscales * 16 + min_scale);
  for (int s_ = min_scale; s_ < nscales; s_++) owner_sig = (owner_sig * 1099511628211ull) ^ ((unsigned long long)ow_off[s_] << 20) ^ (unsigned long long)ow_bytes[s_];
  const bool owner_known_clean = self_cleaning && slot.note(1, owner_sig);   // never while recording / on a buffer a graph was recorded on
  slot.set_note(1, 0);   // until this call has queued every descent
  auto build_plain = [&]() {
    if (!build2_src) return vpp_pyramid_build_pair(B1, i1, B2, i2, nscales, stream);
    return build2_src->channels == 1 ? vpp_pyramid_build(build2_levels, nscales, build2_src, stream) : vpp_rgb_pyramid_build(build2_levels, nscales, build2_src, stream);
  };
  if (pyr_pending && !(reset_up_front && self_cleaning)) { pyr_pending = false; rc = build_plain(); if (rc) return rc; }   // (cannot happen: self_cleaning implies both)
  if (reset_up_front) {
    ResetArgs ra; ra.nseg = 0; uint32_t blocks = 0;
    for (int s_ = min_scale; s_ < nscales; s_++) {
      for (int w = 0; w < (self_cleaning ? 1 : 2); w++) {
        const int q = ra.nseg++;
        ra.p[q] = (uint4*)((uint8_t*)g_scratch.p + (w ? ow_off[s_] : mk_off[s_]));
        ra.units[q] = (uint32_t)((w ? ow_bytes[s_] : mk_bytes[s_]) / 16); ra.value[q] = w ? 0xFFFFFFFFu : 0u;
        ra.first_block[q] = blocks; blocks += (ra.units[q] + 255) / 256;
      }
      if (fused_sweeps) {   // the cells' `pre` records: mark 0 wherever no descent writes (the round buffers' records of unmarked cells are never looked at)
        const int q = ra.nseg++;
        ra.p[q] = (uint4*)((uint8_t*)g_scratch.p + rec_off[s_]); ra.units[q] = (uint32_t)(rec_bytes[s_] / 3 / 16); ra.value[q] = 0u;
        ra.first_block[q] = blocks; blocks += (ra.units[q] + 255) / 256;
      }
    }
    if (link) {   // the merge lists' heads (-1), beside the mark maps
      const int q = ra.nseg++;
      ra.p[q] = (uint4*)link->head; ra.units[q] = (uint32_t)link_head_units; ra.value[q] = 0xFFFFFFFFu;
      ra.first_block[q] = blocks; blocks += (ra.units[q] + 255) / 256;
    }
    if (fused_sweeps) {   // the skip / go words: a replayed launch graph carries the same sweep ids again — the words of its previous replay must not pass for this one's
      const int q = ra.nseg++;
      ra.p[q] = (uint4*)skip_go_words; ra.units[q] = 1u; ra.value[q] = 0u;
      ra.first_block[q] = blocks; blocks += 1;
    }
    ra.first_block[ra.nseg] = blocks;
    if (self_cleaning) {
      if (!owner_known_clean) {   // one launch for the owner maps of all scales (before the launch that claims into them)
        ResetArgs z; z.nseg = 0; uint32_t zb = 0;
        for (int s_ = min_scale; s_ < nscales; s_++) {
          const int q = z.nseg++;
          z.p[q] = (uint4*)((uint8_t*)g_scratch.p + ow_off[s_]); z.units[q] = (uint32_t)(ow_bytes[s_] / 16); z.value[q] = 0xFFFFFFFFu;
          z.first_block[q] = zb; zb += (z.units[q] + 255) / 256;
        }
        z.first_block[z.nseg] = zb;
        sdof_reset_kernel<<<zb, 256, 0, st>>>(z);
      }
      ClaimAll ca; ca.first = min_scale; ca.last = nscales - 1;
      for (int s_ = min_scale; s_ < nscales; s_++) ca.owner[s_] = dimg(&OW(0, s_));
      bool fused = false;
      if (pyr_pending) {
        pyr_pending = false;
        ResetClaimTail tail{ra, kps, n, patchsize, ca};
        rc = build2_src ? pyramid_one_with_tail(build2_levels, build2_src, nscales, tail, blocks + (unsigned)((n + 255) / 256), st, &fused)
                        : pyramid_pair_with_tail(B1, i1, B2, i2, nscales, tail, blocks + (unsigned)((n + 255) / 256), st, &fused);
        if (rc) return rc;
        if (!fused) { rc = build_plain(); if (rc) return rc; }
      }
      if (!fused) sdof_reset_claim_kernel<<<blocks + (unsigned)((n + 255) / 256), 256, 0, st>>>(ra, kps, n, patchsize, ca);
    } else sdof_reset_kernel<<<blocks, 256, 0, st>>>(ra);
    VPP_LAUNCH_CHECK();
  }
  if (claim_up_front && !self_cleaning) {
    ClaimAll ca; ca.first = min_scale; ca.last = nscales - 1;
    for (int s_ = min_scale; s_ < nscales; s_++) ca.owner[s_] = dimg(&OW(0, s_));
    sdof_claim_all_kernel<<<(n + 255) / 256, 256, 0, st>>>(kps, n, patchsize, ca);
    VPP_LAUNCH_CHECK();
  }
  bool readback_done = false;   // the last fused sweep's launch carried the read-back (ReadbackTail)
  for (int scale = nscales - 1; scale >= min_scale; scale--) {  // :92
    const int scale_div = 1 << scale;
    const uint8_t zero = 0;
    const bool has_coarse = scale < nscales - 1;
    const int fr = FL(0, scale).nrows;
    // every kernel that evaluates SAD windows exists once per register-window size (WS = 0: any other size, generic loop)
    auto launch_scale = [&](auto WSc) -> int {
      constexpr int WS = decltype(WSc)::value;
      // claim + descent of strip k on stream sk, rows [lo, hi) of the flow map, into strip k's own maps
      auto strip_phase = [&](int k, hipStream_t sk) -> int {
        int lo = nstrips == 1 ? 0 : (int)((long long)fr * k / nstrips), hi = nstrips == 1 ? INT_MAX : (int)((long long)fr * (k + 1) / nstrips);
        if (world > 1) { const int per = (fr + world - 1) / world; lo = rank * per; hi = rank + 1 == world ? INT_MAX : (rank + 1) * per; }
        if (!reset_up_front) {
          int r2 = vpp_fill(&MK(k, scale), &zero, 1, (void*)sk); if (r2) return r2;  // fill_with_border(flow_map_mark, 0), :111
          { const int rf = device_fill(OW(k, scale).first_pixel, 0xFF, (size_t)OW(k, scale).pitch * OW(k, scale).nrows, sk); if (rf != VPP_OK) return rf; }
        }
        if (!claim_up_front) sdof_claim_kernel<<<(n + 255) / 256, 256, 0, sk>>>(kps, n, scale_div, patchsize, dimg(&OW(k, scale)), lo, hi);
        const long long cells = (long long)OW(k, scale).nrows * OW(k, scale).ncols;
        Mirrors mir{{nullptr, nullptr, nullptr}, 0};
        if (fused_sweeps) mir = Mirrors{{rec[scale][0], rec[scale][1], rec[scale][2]}, (P1[scale].ncols - 1) / patchsize + 1};
        // 4 lanes per keypoint / cell instead of 8 where 8 would be more waves than the chip holds at once (6 144 at this kernel's 6 waves per SIMD): 4K, 82 k keypoints
        // per scale 10 219 -> 5 110 waves, pair 0.208 -> 0.203 ms, with a keypoint every 5 px 0.481 -> 0.459 ms; below that 8 lanes keep the walks' steps shorter
        // (1080p, 20 k keypoints: 0.164 against 0.167 ms with 4).  sdof.descent_group = 8 / 4 forces one.
        const long long groups = (cells * 2 <= n && tuning("sdof.descent_bycell", 1)) ? cells : (long long)n;
        const int want_g = tuning("sdof.descent_group", 0);
        const bool four = WS != 0 && tuning("sdof.descent_lanes", 8) == 8 && (want_g == 4 || (want_g == 0 && groups > 6144 * 8));
        const bool bycell = cells * 2 <= n && tuning("sdof.descent_bycell", 1);
        const bool grouped = tuning("sdof.descent_lanes", 8) == 8;
        auto go = [&](auto Bc, auto Gc) {
          constexpr bool BC = decltype(Bc)::value; constexpr int GG = decltype(Gc)::value;
          if constexpr (WS != 0 || GG == 8) {
            const long long ng = BC ? cells : (long long)n;
            sdof_descent_group_kernel<WS, BC, GG><<<(unsigned)((ng + 64 / GG - 1) / (64 / GG)), 64, 0, sk>>>(kps, n, scale_div, patchsize, winsize, dimg(&OW(k, scale)), dimg(&P1[scale]), dimg(&P2[scale]),
                                                                                                   maps(k, scale), maps(k, has_coarse ? scale + 1 : scale), has_coarse ? 1 : 0, lo, hi, self_cleaning ? 1 : 0, mir);
          }
        };
        using T = std::true_type; using F = std::false_type; using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
        if (grouped) {
          if (four && bycell) go(T{}, I4{}); else if (four) go(F{}, I4{}); else if (bycell) go(T{}, I8{}); else go(F{}, I8{});
        }
        else
          sdof_descent_kernel<WS><<<(n + 63) / 64, 64, 0, sk>>>(kps, n, scale_div, patchsize, winsize, dimg(&OW(k, scale)), dimg(&P1[scale]), dimg(&P2[scale]),
                                                               maps(k, scale), maps(k, has_coarse ? scale + 1 : scale), has_coarse ? 1 : 0, lo, hi, self_cleaning ? 1 : 0);
        return VPP_OK;
      };
      if (nstrips > 1) VPP_HIP_TRY(hipEventRecord(g_strips.start, st));   // the pyramids / the previous scale's broadcast are behind this point
      for (int k = 1; k < nstrips; k++) {
        hipStream_t sk = g_strips.st[k - 1];
        VPP_HIP_TRY(hipStreamWaitEvent(sk, g_strips.start, 0));
        int r2 = strip_phase(k, sk); if (r2) return r2;
        VPP_HIP_TRY(hipEventRecord(g_strips.done[k - 1], sk));
      }
      { int r2 = strip_phase(0, st); if (r2) return r2; }
      for (int k = 1; k < nstrips; k++) {   // gather: strip k's rows of the three maps into the owner's
        VPP_HIP_TRY(hipStreamWaitEvent(st, g_strips.done[k - 1], 0));
        const int lo = (int)((long long)fr * k / nstrips), hi = (int)((long long)fr * (k + 1) / nstrips);
        int r2 = copy_rows(&FL(0, scale), &FL(k, scale), lo, hi, st); if (r2) return r2;
        r2 = copy_rows(&MK(0, scale), &MK(k, scale), lo, hi, st); if (r2) return r2;
        r2 = copy_rows(&DM(0, scale), &DM(k, scale), lo, hi, st); if (r2) return r2;
      }
      if (world > 1) {   // the ranks' rows of the three maps -> every rank, in place, one RCCL launch
        const int per = (fr + world - 1) / world;
        auto rows0 = [](const vpp_image_desc& d) { return (uint8_t*)d.first_pixel - (ptrdiff_t)d.border * elem_bytes(&d); };   // byte 0 of row 0 incl. its left border
        int r2 = comm_group_begin(); if (r2) return r2;
        r2 = comm_allgather_inplace(comm, rows0(FL(0, scale)), (size_t)per * FL(0, scale).pitch, st); if (r2) return r2;
        r2 = comm_allgather_inplace(comm, rows0(MK(0, scale)), (size_t)per * MK(0, scale).pitch, st); if (r2) return r2;
        r2 = comm_allgather_inplace(comm, rows0(DM(0, scale)), (size_t)per * DM(0, scale).pitch, st); if (r2) return r2;
        r2 = comm_group_end(); if (r2) return r2;
      }
      if (propagation > 0) {
        const int NI = (P1[scale].nrows - 1) / patchsize + 1;
        const int mode = tuning("sdof.propagate", 0);  // 0: Jacobi rounds to the fixed point; 1: the lock-step wavefront on one workgroup (cross-check)
        const int NJ = (P1[scale].ncols - 1) / patchsize + 1;
        if (mode != 1) {
          const int cells = NI * NJ;
          RoundArrays rs = ra;
          rs.pre = rec[scale][0]; rs.B[0] = rec[scale][1]; rs.B[1] = rec[scale][2];
          // most sweeps have nothing or a few hundred cells to do: a small grid keeps the empty launch cheap; long lists are walked in passes
          const int grid = std::min(tuning("sdof.rounds_grid", 256), (cells + kJobsPerGroup - 1) / kJobsPerGroup);
          // consecutive ids for this scale's sweeps (a.skip): sweep Ki + 1 skips only on the word sweep Ki wrote
          static std::atomic<unsigned> g_sweep_seq{1};
          unsigned sweep_base = g_sweep_seq.fetch_add((unsigned)propagation + 1u, std::memory_order_relaxed);
          if (sweep_base + (unsigned)propagation + 1u < sweep_base || sweep_base == 0u)   // the counter wrapped: 0 means "nothing to skip on" in the word, never an id
            sweep_base = g_sweep_seq.fetch_add((unsigned)propagation + 1u, std::memory_order_relaxed);
          for (int Ki = 0; Ki < propagation; Ki++) {
            if (fused_sweeps) {   // one launch per sweep: every workgroup classifies its tile of cells and runs round 0 on them, the last one to finish runs the rest
              const int tiles16 = ((NI + kSweepTile - 1) / kSweepTile) * ((NJ + kSweepTile - 1) / kSweepTile);
              const int want_ts = tuning("sdof.sweep_tile", 0), ts = want_ts ? want_ts : (tiles16 > 512 ? 32 : 16);   // 32 x 32 cells per workgroup where 16 x 16 would be more than 512 workgroups
              const int tiles = ((NI + ts - 1) / ts) * ((NJ + ts - 1) / ts);
              const int nsub = std::max(1, std::min(kSubCounters, tiles / 16));
              const int want_nt = tuning("sdof.sweep_threads", 0), nt = want_nt ? want_nt : (tiles <= 512 ? 512 : 256);
              // the pair's LAST sweep carries the read-back as extra workgroups (ReadbackTail): one dependent launch less
              const bool tail = scale == min_scale && Ki == propagation - 1 && tuning("sdof.readback_tail", 1);
              auto sweep_ts = [&](auto NTc, auto RBc, auto TSc) {
                constexpr int NT = decltype(NTc)::value, RB = decltype(RBc)::value, TS = decltype(TSc)::value;
                ReadbackTail rb{};
                if constexpr (RB != 0) {
                  const int ms_ = 1 << min_scale;
                  rb.kps = kps; rb.n = n; rb.div = patchsize * ms_; rb.ms = ms_; rb.out_pos = out_pos; rb.out_dist = out_dist; rb.out_valid = out_valid;
                  if (link) rb.link = *link;
                  rb.blocks = (n + NT - 1) / NT; rb.go = rs.skip + 1;
                }
                sdof_sweep_kernel<WS, NT, RB, TS><<<tiles + (RB ? (n + NT - 1) / NT : 0), NT, 0, st>>>(dimg(&P1[scale]), dimg(&P2[scale]), winsize, maps(0, scale), patchsize, Ki % 2, NI, NJ, rs,
                                                                tuning("sdof.stats", 0), nsub, (unsigned)tuning("sdof.sweep_stay", NT / 8), sweep_base + (unsigned)Ki,
                                                                Ki > 0 && tuning("sdof.skip_empty", 1) ? 1 : 0, rb);
              };
              auto sweep = [&](auto NTc, auto RBc) { if (ts == 32) sweep_ts(NTc, RBc, std::integral_constant<int, 32>()); else sweep_ts(NTc, RBc, std::integral_constant<int, 16>()); };
              using N512 = std::integral_constant<int, 512>; using N256 = std::integral_constant<int, 256>;
              using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
              if (tail) {
                readback_done = true;
                if (nt == 512) { if (link) sweep(N512(), R2()); else sweep(N512(), R1()); } else { if (link) sweep(N256(), R2()); else sweep(N256(), R1()); }
              } else if (nt == 512) sweep(N512(), R0()); else sweep(N256(), R0());
              continue;
            }
            sdof_classify_kernel<<<(cells + 255) / 256, 256, 0, st>>>(maps(0, scale), NI, NJ, rs);
            sdof_rounds_kernel<WS><<<grid, 256, 0, st>>>(dimg(&P1[scale]), dimg(&P2[scale]), winsize, maps(0, scale), patchsize, Ki % 2, NI, NJ, rs, tuning("sdof.stats", 0));
          }
        } else
          sdof_propagate_kernel<WS><<<1, 1024, 0, st>>>(dimg(&P1[scale]), dimg(&P2[scale]), winsize, maps(0, scale), patchsize, propagation);
      }
      if (scale > min_scale)   // broadcast: the swept maps are the next finer scale's prediction in every strip
        for (int k = 1; k < nstrips; k++) {
          const int b = FL(0, scale).border;
          int r2 = copy_rows(&FL(k, scale), &FL(0, scale), -b, fr + b, st); if (r2) return r2;
          r2 = copy_rows(&MK(k, scale), &MK(0, scale), -b, fr + b, st); if (r2) return r2;
          r2 = copy_rows(&DM(k, scale), &DM(0, scale), -b, fr + b, st); if (r2) return r2;
        }
      return VPP_OK;
    };
    switch (winsize) {
      case 5: rc = launch_scale(std::integral_constant<int, 5>()); break;
      case 7: rc = launch_scale(std::integral_constant<int, 7>()); break;
      case 9: rc = launch_scale(std::integral_constant<int, 9>()); break;
      case 11: rc = launch_scale(std::integral_constant<int, 11>()); break;
      default: rc = launch_scale(std::integral_constant<int, 0>()); break;
    }
    if (rc) return rc;
    VPP_LAUNCH_CHECK();
  }
  const int ms = 1 << min_scale;
  if (readback_done) { /* queued with the last sweep */ }
  else if (link) {
    if (!reset_up_front) {   // no up-front reset launch on this path: the heads get their own
      ResetArgs rh; rh.nseg = 1; rh.p[0] = (uint4*)link->head; rh.units[0] = (uint32_t)link_head_units; rh.value[0] = 0xFFFFFFFFu; rh.first_block[0] = 0;
      rh.first_block[1] = (rh.units[0] + 255) / 256;
      sdof_reset_kernel<<<rh.first_block[1], 256, 0, st>>>(rh);
    }
    sdof_readback_kernel<true><<<(n + 255) / 256, 256, 0, st>>>(kps, n, patchsize * ms, ms, maps(0, min_scale), out_pos, out_dist, out_valid, *link);
  } else sdof_readback_kernel<false><<<(n + 255) / 256, 256, 0, st>>>(kps, n, patchsize * ms, ms, maps(0, min_scale), out_pos, out_dist, out_valid, MergeLinkArgs{});
  VPP_LAUNCH_CHECK();
  if (self_cleaning) slot.set_note(1, owner_sig);   // every descent of this call is queued: the owner maps of this layout end up empty
  return VPP_OK;
}

}  // namespace

extern "C" int vpp_semi_dense_optical_flow_strips(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n, int winsize,
                                                  int nscales, int min_scale, int propagation, int patchsize, int nstrips, int32_t* out_pos,
                                                  int32_t* out_dist, uint8_t* out_valid, void* stream) {
  return flow_impl(i1, i2, kps, n, winsize, nscales, min_scale, propagation, patchsize, nstrips, nullptr, out_pos, out_dist, out_valid, stream);
}

extern "C" int vpp_semi_dense_optical_flow(const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n, int winsize,
                                           int nscales, int min_scale, int propagation, int patchsize, int32_t* out_pos,
                                           int32_t* out_dist, uint8_t* out_valid, void* stream) {
  return flow_impl(i1, i2, kps, n, winsize, nscales, min_scale, propagation, patchsize, 1, nullptr, out_pos, out_dist, out_valid, stream);
}

extern "C" int vpp_semi_dense_optical_flow_pyramids(const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, int nscales, const int32_t* kps, int n, int winsize,
                                                     int min_scale, int propagation, int patchsize, int32_t* out_pos, int32_t* out_dist,
                                                     uint8_t* out_valid, void* stream) {
  VPP_REQUIRE(pyr1 && pyr2 && nscales >= 1, VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow_pyramids: null pyramid");
  return flow_impl(&pyr1[0], &pyr2[0], kps, n, winsize, nscales, min_scale, propagation, patchsize, 1, nullptr, out_pos, out_dist, out_valid, stream, pyr1, pyr2);
}

// The tracker's flow (extruder.hip): over the frames (pyr1 == nullptr) or over its own pyramids, with the merge step's first pass folded in.
namespace vpp_amd {
int sdof_flow_linked(const vpp_image_desc* i1, const vpp_image_desc* i2, const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, const int32_t* kps, int n, int winsize, int nscales,
                     int propagation, int patchsize, int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid, const MergeLinkArgs* link, size_t link_head_units, void* stream,
                     const vpp_image_desc* build2_levels, const vpp_image_desc* build2_src) {
  return flow_impl(pyr1 ? &pyr1[0] : i1, pyr1 ? &pyr2[0] : i2, kps, n, winsize, nscales, 0, propagation, patchsize, 1, nullptr, out_pos, out_dist, out_valid, stream, pyr1, pyr2, link,
                   link_head_units, pyr1 ? build2_levels : nullptr, pyr1 ? build2_src : nullptr);
}
}  // namespace vpp_amd

extern "C" int vpp_semi_dense_optical_flow_sharded(vpp_comm* comm, const vpp_image_desc* i1, const vpp_image_desc* i2, const int32_t* kps, int n, int winsize,
                                                   int nscales, int min_scale, int propagation, int patchsize, int32_t* out_pos,
                                                   int32_t* out_dist, uint8_t* out_valid, void* stream) {
  VPP_REQUIRE(comm, VPP_ERR_INVALID_ARG, "vpp_semi_dense_optical_flow_sharded: null communicator");
  return flow_impl(i1, i2, kps, n, winsize, nscales, min_scale, propagation, patchsize, 1, comm, out_pos, out_dist, out_valid, stream);
}

// diagnostics (not part of include/vpp_amd.h): counters of the propagation rounds, enabled by tuning "sdof.stats"
// diagnostics (not part of include/vpp_amd.h): the log of the fused sweeps since the last reset (tools/sweep_log.py); 4096 entries, the count through *n
extern "C" int vpp_debug_sdof_sweep_log(unsigned long long* out4096, unsigned* n, int reset) {
  VPP_HIP_TRY(hipDeviceSynchronize());
  VPP_HIP_TRY(hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_sweep_log), 4096 * sizeof(unsigned long long)));
  VPP_HIP_TRY(hipMemcpyFromSymbol(n, HIP_SYMBOL(g_sweep_log_n), sizeof(unsigned)));
  if (reset) { unsigned z = 0; VPP_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_log_n), &z, sizeof z)); }
  return VPP_OK;
}
extern "C" int vpp_debug_sdof_round_stats(unsigned* out4, int reset) {
  VPP_HIP_TRY(hipDeviceSynchronize());
  VPP_HIP_TRY(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_round_stats), 4 * sizeof(unsigned)));
  if (reset) { unsigned z[4] = {0, 0, 0, 0}; VPP_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_stats), z, sizeof z)); }
  return VPP_OK;
}
