// lbp_distance.hh — Hamming distance of two LBP codes (reference: vpp/algorithms/lbp/lbp_distance.hh).  Host scalar helper.
#pragma once
namespace vpp {
inline int lbp_hamming_distance(unsigned char a, unsigned b) {
  unsigned char val = a ^ b;
  int dist = 0;
  while (val) { ++dist; val &= val - 1; }
  return dist;
}
}  // namespace vpp
