// shard_plan.hh — how N keypoint records are split over W ranks and put back together (SURVEY 8e bullet 1; no reference counterpart:
// pyrlk_match.hh:24-51 iterates independent keypoints in one process).  Rank g owns the contiguous slice [g N / W, (g + 1) N / W);
// the exchange is ONE fixed-size all-gather, so every rank contributes `per_rank` = the largest slice, padded with dead records
// (age 0: vpp_pyrlk_match skips them), and the gathered array of W * per_rank records is compacted back into index order.
#pragma once
#include <cstring>
#include <vector>

namespace vpp_shard {
struct plan {
  int n, world, per_rank;
  plan(int n_, int world_) : n(n_), world(world_), per_rank(0) { for (int g = 0; g < world; g++) per_rank = std::max(per_rank, hi(g) - lo(g)); }
  int lo(int g) const { return int((long long)g * n / world); }
  int hi(int g) const { return int((long long)(g + 1) * n / world); }
  int count(int g) const { return hi(g) - lo(g); }
  // the padded shard of rank g out of the full array (records of `bytes` bytes; padding = zero bytes, i.e. age 0)
  template <class R> std::vector<R> shard_of(const std::vector<R>& all, int g) const {
    std::vector<R> s(per_rank);
    std::memset(s.data(), 0, s.size() * sizeof(R));
    std::memcpy(s.data(), all.data() + lo(g), size_t(count(g)) * sizeof(R));
    return s;
  }
  // gathered (world * per_rank records, rank-major) -> the n records in index order
  template <class R> std::vector<R> unpad(const std::vector<R>& gathered) const {
    std::vector<R> out(n);
    for (int g = 0; g < world; g++) std::memcpy(out.data() + lo(g), gathered.data() + size_t(g) * per_rank, size_t(count(g)) * sizeof(R));
    return out;
  }
};
}  // namespace vpp_shard
