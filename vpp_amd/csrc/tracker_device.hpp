// tracker_device.hpp — device functions shared by the kernels of the tracker step (video.hip, fast9.hip, extruder.hip); not part of the C ABI.
#pragma once
#include "common.hpp"

namespace vpp_amd {

// fast9_score at one pixel (vpp/algorithms/fast_detector/fast.hpp:38-77): the true Bresenham ring of radius 3 (fast.hpp:52-74), the sum of the
// differences beyond the threshold on the brighter and on the darker side, the larger of the two.
__device__ __forceinline__ int fast9_score_at(const DImg& A, int r, int c, int th) {
  constexpr int dr[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  constexpr int dc[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  const int v = A.row<uint8_t>(r)[c];
  int sum_inf = 0, sum_sup = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int a = A.row<uint8_t>(r + dr[i])[c + dc[i]];
    const int diff = v - a;
    if (diff < -th) sum_inf -= diff;
    else if (diff > th) sum_sup += diff;
  }
  return max(sum_sup, sum_inf);
}

// The per-cell keypoint lists of the merge step (video_extruder.hpp:60-84), built by merge_link_kernel (video.hip): head[cell] -> next[...] chains of
// keypoint indices, age_now = the age after the match callback, cell_of = the cell a keypoint landed in.
struct MergeLists { const int32_t *head, *next, *age_now, *cell_of; };

// Is keypoint i removed by the merge?  The serial loop keeps one champion per cell — the running maximum of the ages in container order — so with
// E = max age of the EARLIER keypoints of the cell and L = max age of the LATER ones: first of its cell or age > E -> it becomes champion, removed iff
// L > age; age == E -> kept; age < E -> removed.  The unordered list gives the ordered answer by comparing indices.
__device__ __forceinline__ bool merge_removes(const MergeLists& m, int i) {
  const int a = m.age_now[i];
  int E = -1, L = -1;  // ages are >= 0
  bool earlier = false;
  for (int j = m.head[m.cell_of[i]]; j >= 0; j = m.next[j]) {
    if (j < i) { earlier = true; E = max(E, m.age_now[j]); }
    else if (j > i) L = max(L, m.age_now[j]);
  }
  return (earlier && a <= E) ? (a < E) : (L > a);
}

// Pass 1 of the merge for ONE keypoint (merge_link_kernel, video.hip; also run by the flow's read-back kernel for the tracker, sdof.hip): the keypoint's age
// after the match callback (keypoint_container::move :51 / remove :52) and its cell, then threaded onto the cell's list by an atomic exchange of the head.
struct MergeLinkArgs { const int32_t* age_prev; int nr, nc, spacing, gr, gc; int32_t *head, *next, *age_now, *cell_of; };
__device__ __forceinline__ void merge_link_one(const MergeLinkArgs& a, int i, int moved_r, int moved_c, int prev_r, int prev_c, bool matched) {
  int r = moved_r, c = moved_c, age = a.age_prev[i];
  if (matched) {
    if (r >= 0 && c >= 0 && r < a.nr && c < a.nc) age++;   // keypoint_container::move (:51)
    else { age = 0; r = prev_r; c = prev_c; }               // remove (:52): dies where it was
  } else { r = prev_r; c = prev_c; }
  const int cell = min(max(r / a.spacing, 0), a.gr - 1) * a.gc + min(max(c / a.spacing, 0), a.gc - 1);
  a.age_now[i] = age; a.cell_of[i] = cell;
  a.next[i] = atomicExch(&a.head[cell], i);
}
// the lists' storage for n keypoints (no launch): head[] must be reset to -1 (head_units16 16-byte units from head) before the first merge_link_one
int keypoint_merge_prepare(const int32_t* age_prev, int n, int nrows, int ncols, int spacing, MergeLinkArgs* args, size_t* head_units16, hipStream_t st);

// memset of the heads + merge_link_kernel on `st`; the lists stay valid until the next call on the same host thread
int keypoint_merge_link(const int32_t* rc_moved, const int32_t* rc_prev, const uint8_t* matched, const int32_t* age_prev, int n, int nrows, int ncols, int spacing,
                        MergeLists* lists, hipStream_t st);

}  // namespace vpp_amd
