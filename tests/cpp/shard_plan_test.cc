// The sharding / padding / unpadding logic of the C++ multi-process harness (benchmarks/shard_plan.hh), simulated for every rank in one
// process: world sizes 1-8 incl. 2 and 3, even and ragged keypoint counts, more ranks than keypoints.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../benchmarks/shard_plan.hh"
struct rec { float a, b, c, d; int age; };
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)
int main() {
  for (int world = 1; world <= 8; world++)
    for (int n : {0, 1, 2, 3, 7, 150, 151, 10000, 10001}) {
      std::vector<rec> all(n);
      for (int i = 0; i < n; i++) all[i] = rec{float(i), float(2 * i), 0, 0, 1 + i % 5};
      vpp_shard::plan p(n, world);
      int covered = 0;
      std::vector<rec> gathered;
      for (int g = 0; g < world; g++) {
        CHECK(p.lo(g) == covered && p.hi(g) >= p.lo(g) && p.count(g) <= p.per_rank);
        covered = p.hi(g);
        auto s = p.shard_of(all, g);
        CHECK(int(s.size()) == p.per_rank);
        for (int k = p.count(g); k < p.per_rank; k++) CHECK(s[k].age == 0);   // padding is dead: the kernel skips it
        gathered.insert(gathered.end(), s.begin(), s.end());                // what the fixed-size all-gather delivers, rank-major
      }
      CHECK(covered == n);
      auto back = p.unpad(gathered);
      CHECK(int(back.size()) == n);
      for (int i = 0; i < n; i++) CHECK(back[i].a == all[i].a && back[i].age == all[i].age);
    }
  std::puts("shard_plan_test ok");
  return 0;
}
