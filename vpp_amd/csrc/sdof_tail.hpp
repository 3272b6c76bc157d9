// sdof_tail.hpp — the flow's up-front map reset + claims (sdof.hip: sdof_reset_claim_kernel) as device code that ANOTHER kernel's extra workgroups can run:
// the two frames' pyramids and the reset / claims are independent work at the start of vpp_semi_dense_optical_flow, and a dependent launch costs ~4.5 us whatever
// it does — pyramid_fused.hip appends these blocks to the pair launch (round 5: 12 -> 11 dependent launches per 4K pair).
// Reference: semi_dense_optical_flow.hpp:111 (fill_with_border(flow_map_mark, 0)) and :116-121 (the first keypoint in index order claims its flow-map cell).
#pragma once
#include "common.hpp"

namespace vpp_amd {

constexpr int kMaxScales = 8;
// One launch resets the maps of every scale that the per-scale phases used to reset one by one: the segments are whole carved blocks, written as 16-byte units.
constexpr int kResetSegs = 28;
struct ResetArgs { uint4* p[kResetSegs]; uint32_t first_block[kResetSegs + 1]; uint32_t units[kResetSegs]; uint32_t value[kResetSegs]; int nseg; };
// the claims of every scale in one launch (single strip, single rank): a claim depends on the keypoint list and the scale only, not on any flow
struct ClaimAll { DImg owner[kMaxScales]; int first, last; };
struct ResetClaimTail { ResetArgs a; const int32_t* kps; int n, patch; ClaimAll c; };

// block `b` of the reset + claim grid (a.first_block[a.nseg] reset blocks, then one block per 256 keypoints), 256 threads
__device__ __forceinline__ void reset_claim_block(const ResetArgs& a, const int32_t* __restrict__ kps, int n, int patch, const ClaimAll& c, unsigned b) {
  const uint32_t reset_blocks = a.first_block[a.nseg];
  if (b < reset_blocks) {
    int sgm = 0;
    while (sgm + 1 < a.nseg && b >= a.first_block[sgm + 1]) sgm++;
    const uint32_t u = (b - a.first_block[sgm]) * 256 + threadIdx.x;
    if (u < a.units[sgm]) { const uint32_t v = a.value[sgm]; a.p[sgm][u] = make_uint4(v, v, v, v); }
    return;
  }
  const int i = (int)(b - reset_blocks) * 256 + (int)threadIdx.x;
  if (i >= n) return;
  const int k0 = kps[2 * i], k1 = kps[2 * i + 1];
  for (int s = c.first; s <= c.last; s++) {
    const int div = 1 << s;
    const int pf0 = (k0 / div) / patch, pf1 = (k1 / div) / patch;
    if (c.owner[s].has(pf0, pf1)) atomicMin(c.owner[s].row<uint32_t>(pf0) + pf1, (uint32_t)i);
  }
}

// pyramid_fused.hip: the two frames' 3-level u8 pyramids (vpp_pyramid_build_pair's one-launch case) with `tail_blocks` blocks of reset_claim_block in the SAME
// launch.  *fused = false (nothing launched) when the pair does not take the packed kernel: the caller then launches the two things itself.
int pyramid_pair_with_tail(const vpp_image_desc* levels_a, const vpp_image_desc* src_a, const vpp_image_desc* levels_b, const vpp_image_desc* src_b, int nlevels,
                           const ResetClaimTail& tail, unsigned tail_blocks, hipStream_t st, bool* fused);
// the same for ONE pyramid (u8 x1 source, or x3 / x4 through the ingest)
int pyramid_one_with_tail(const vpp_image_desc* levels, const vpp_image_desc* src, int nlevels, const ResetClaimTail& tail, unsigned tail_blocks, hipStream_t st, bool* fused);

}  // namespace vpp_amd
