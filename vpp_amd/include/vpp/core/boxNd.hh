// boxNd.hh — inclusive integer boxes (reference: vpp/core/boxNd.hh:11-150, boxNd_iterator.hh).
#pragma once
#include <cassert>
#include <vpp/core/vector.hh>

namespace vpp {

template <unsigned N, class C = int> class boxNd;

// raster-order iterator: the last coordinate varies fastest (tests/boxNd_iterator.cc:8-48)
template <unsigned N, class C = int> class boxNd_iterator {
 public:
  typedef vector<C, N> coord_type;
  boxNd_iterator(const coord_type& p, const boxNd<N, C>& b) : p_(p), box_(&b) {}
  const coord_type& operator*() const { return p_; }
  operator coord_type() const { return p_; }
  boxNd_iterator& next() {  // row-major successor; one past the last point is (p2[0] + 1, p1[1], ...) = end() (boxNd_iterator.hh:23-24)
    for (int d = int(N) - 1; d >= 0; d--) {
      if (d == 0 || p_[d] < box_->p2()[d]) { p_[d]++; break; }
      p_[d] = box_->p1()[d];
    }
    return *this;
  }
  boxNd_iterator& operator++() { return next(); }
  bool operator==(const boxNd_iterator& o) const { return p_ == o.p_; }
  bool operator!=(const boxNd_iterator& o) const { return !(p_ == o.p_); }
 private:
  coord_type p_;
  const boxNd<N, C>* box_;
};

template <unsigned N, class C> class boxNd {
 public:
  typedef vector<C, N> coord_type;
  typedef boxNd_iterator<N, C> iterator;
  boxNd() : p1_(coord_type::Zero()), p2_(coord_type::Zero()) { for (unsigned i = 0; i < N; i++) size_[i] = 0; }
  boxNd(coord_type p1, coord_type p2) : p1_(p1), p2_(p2) { for (unsigned i = 0; i < N; i++) size_[i] = p2_[i] - p1_[i] + 1; }
  bool has(const coord_type& p) const { for (unsigned i = 0; i < N; i++) if (p[i] < p1_[i] || p[i] > p2_[i]) return false; return true; }
  iterator begin() const { return iterator(p1_, *this); }
  iterator end() const { coord_type e = p1_; e[0] = p2_[0] + 1; return iterator(e, *this); }
  const coord_type& first_point_coordinates() const { return p1_; }
  const coord_type& last_point_coordinates() const { return p2_; }
  const coord_type& p1() const { return p1_; }
  const coord_type& p2() const { return p2_; }
  const int& size(int d) const { assert(d >= 0 && d < int(N)); return size_[d]; }
  const int& ncols() const { return size_[N - 1]; }
  const int& nrows() const { return size_[N - 2]; }
 private:
  coord_type size_, p1_, p2_;
};
template <unsigned N, class C> bool operator==(const boxNd<N, C>& a, const boxNd<N, C>& b) { return a.p1() == b.p1() && a.p2() == b.p2(); }
template <unsigned N, class C> bool operator!=(const boxNd<N, C>& a, const boxNd<N, C>& b) { return !(a == b); }

typedef boxNd<1> box1d; typedef boxNd<2> box2d; typedef boxNd<3> box3d; typedef boxNd<4> box4d;
inline box1d make_box1d(int nc) { vint1 a, b; a[0] = 0; b[0] = nc - 1; return box1d(a, b); }
inline box2d make_box2d(int nr, int nc) { return box2d(vint2(0, 0), vint2(nr - 1, nc - 1)); }
inline box3d make_box3d(int ns, int nr, int nc) { return box3d(vint3(0, 0, 0), vint3(ns - 1, nr - 1, nc - 1)); }

class border {
 public:
  border(int n) : size_(n) {}
  int size() const { return size_; }
 private:
  int size_;
};
template <unsigned N, class C> boxNd<N, C> operator-(const boxNd<N, C>& b, const border& bd) {
  auto p1 = b.p1(); auto p2 = b.p2();
  for (unsigned n = 0; n < N; n++) { p1[n] += bd.size(); p2[n] -= bd.size(); }
  return boxNd<N, C>(p1, p2);
}
template <unsigned N, class C> boxNd<N, C> operator+(const boxNd<N, C>& b, const border& bd) {
  auto p1 = b.p1(); auto p2 = b.p2();
  for (unsigned n = 0; n < N; n++) { p1[n] -= bd.size(); p2[n] += bd.size(); }
  return boxNd<N, C>(p1, p2);
}
template <unsigned N> boxNd<N> operator|(const boxNd<N>&, const boxNd<N>& b) { return b; }

}  // namespace vpp
