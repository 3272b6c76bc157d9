// gradient_descent.hh — integer hill climbing over the 8-neighbourhood with a caller-supplied distance (reference:
// vpp/algorithms/optical_flow/gradient_descent.hh:10-89).  The distance is an opaque callable, so this generic form runs on the
// host like any pixel_wise kernel; semi_dense_optical_flow does not come through here — its descent, with the SAD distance, is the
// device kernel behind vpp_semi_dense_optical_flow (csrc/sdof.hip: gradient_descent_impl, same visiting order).
#pragma once
#include <climits>
#include <vpp/core/vector.hh>

namespace vpp {
struct gradient_descent_result { vint2 flow; int distance; };  // the reference returns iod::D(_flow = ..., _distance = ...): same member names

// distance(p, candidate, best_so_far) -> int.  Each round visits the neighbours of the current position that the previous
// round has not already covered (the arc opposite to the last move), keeps the first strict improvement in visiting order as
// it goes, and stops when a round leaves the position unchanged or after max_iteration rounds.
template <class D> gradient_descent_result gradient_descent_match(const vint2 p, vint2 prediction, D distance, int max_iteration = 10) {
  // neighbour k of the ring   0 1 2      first neighbour to visit / one past the last, by the index of the last move (8 = none yet)
  //                           7 . 3
  //                           6 5 4
  static const signed char ring[8][2] = {{-1, 1}, {0, 1}, {1, 1}, {-1, 0}, {1, 0}, {-1, -1}, {0, -1}, {1, -1}};
  static const unsigned char first[9] = {6, 0, 0, 2, 2, 4, 4, 6, 0}, past[9] = {3, 3, 5, 5, 7, 7, 1, 1, 0};
  vint2 match = prediction;
  int best = distance(p, prediction, INT_MAX);
  unsigned last_move = 8;
  for (int round = 0; round < max_iteration; round++) {
    unsigned k = first[last_move];
    const unsigned stop = past[last_move];
    do {
      const vint2 n = prediction + vint2(ring[k][0], ring[k][1]);
      const int d = distance(p, n, best);
      if (d < best) { match = n; last_move = k; best = d; }
      k = (k + 1) & 7;
    } while (k != stop);
    if (prediction == match) break;
    prediction = match;
  }
  return gradient_descent_result{match - p, best};
}
}  // namespace vpp
