// keypoint_container.hh — keypoints + features with a 2-D index (reference: vpp/core/keypoint_container.hh:13-90,
// keypoint_container.hpp:11-167) and trajectories (vpp/core/keypoint_trajectory.hh:11-72).  Host-side bookkeeping.
// The index image is refilled only over the cells that were set (the reference memsets the whole frame-sized image on
// every prepare_matching: 8-33 MB per frame); observable behaviour is the same.
#pragma once
#include <cassert>
#include <deque>
#include <vector>
#include <vpp/core/fill.hh>
#include <vpp/core/image2d.hh>

namespace vpp {

template <class C> struct keypoint {
  keypoint() : age(0) {}
  keypoint(vector<C, 2> pos) : position(pos), velocity(0, 0), age(1) {}
  vector<C, 2> position, velocity;
  int age;
  void die() { age = 0; }
  bool alive() const { return age > 0; }
};

template <class P, class F> struct keypoint_container {
  typedef P keypoint_type;
  typedef F feature_type;
  typedef std::vector<P> keypoint_vector_type;
  typedef std::vector<F> feature_vector_type;

  keypoint_container(const box2d& d) : index2d_(d, _border = 10), compact_has_run_(false) {
    fill_with_border(index2d_, -1);
    keypoint_vector_.reserve((d.nrows() * d.ncols()) / 10);
    feature_vector_.reserve((d.nrows() * d.ncols()) / 10);
  }
  void compact() {
    compact_has_run_ = true;
    matches_.assign(keypoint_vector_.size(), -1);
    size_t w = 0;
    for (size_t i = 0; i < keypoint_vector_.size(); i++)
      if (keypoint_vector_[i].alive()) {
        keypoint_vector_[w] = keypoint_vector_[i]; feature_vector_[w] = feature_vector_[i];
        set_index(cast<vint2>(keypoint_vector_[w].position), int(w));
        matches_[i] = int(w);
        w++;
      }
    keypoint_vector_.resize(w); feature_vector_.resize(w);
  }
  void prepare_matching() {
    compact_has_run_ = false;
    if (!idx_base_ && !touched_.empty()) { idx_base_ = (char*)&index2d_(0, 0); idx_pitch_ = index2d_.pitch(); }
    for (const vint2& p : touched_) *(int*)(idx_base_ + (ptrdiff_t)p[0] * idx_pitch_ + (ptrdiff_t)p[1] * (ptrdiff_t)sizeof(int)) = -1;
    touched_.clear();
    std::fill(matches_.begin(), matches_.end(), -1);
  }
  struct no_op { template <class T> void operator()(T&) {} };
  template <class T, class D = no_op> void sync_attributes(T& v, typename T::value_type new_value = typename T::value_type(), D die_fun = D()) const {
    const size_t nparts = keypoint_vector_.size();
    if (compact_has_run_) {
      // compact() maps survivors to increasing new indices ni <= i, so the attributes are compacted in place with swaps
      // (the reference builds a second vector of nparts copies of new_value and moves into it — for trajectories that is
      // two heap allocations per std::deque; the resulting vector is the same)
      using std::swap;
      const size_t n_old = v.size() < matches_.size() ? v.size() : matches_.size();
      size_t survivors = 0;
      for (size_t i = 0; i < n_old; i++) {
        const int ni = matches_[i];
        if (ni >= 0) { assert(size_t(ni) == survivors && size_t(ni) <= i); if (size_t(ni) != i) swap(v[ni], v[i]); survivors++; }
        else die_fun(v[i]);
      }
      v.resize(survivors);            // drop the dead tail ...
      v.resize(nparts, new_value);    // ... and give every keypoint added since the last sync a fresh attribute
    } else v.resize(nparts, new_value);
  }
  template <class T, class U> void sync_attributes(T& c, typename T::value_type new_value, std::vector<U>& dead) const {
    sync_attributes(c, new_value, [&dead](typename T::value_type& x) { dead.push_back(std::move(x)); });
  }
  void add(const keypoint_type& p, const feature_type& f = feature_type()) {
    set_index(cast<vint2>(p.position), int(keypoint_vector_.size()));
    keypoint_vector_.push_back(p); feature_vector_.push_back(f);
  }
  void add(const vfloat2& p) { add(keypoint_type(cast<decltype(keypoint_type().position)>(p))); }
  void remove(int i) {
    assert(i < size());
    keypoint_vector_[i].die();
    int& index = index2d_(cast<vint2>(keypoint_vector_[i].position));
    if (index == i) index = -1;
  }
  void remove(vint2 pos) { assert(has(pos)); remove(index2d_(pos)); }
  template <class T> void move(int i, T position) {
    assert(i >= 0 && i < size());
    auto& kp = keypoint_vector_[i];
    kp.velocity = position - kp.position;
    kp.position = position;
    kp.age++;
    set_index(cast<vint2>(kp.position), i);
  }
  void update(unsigned i, const keypoint_type& p, const feature_type& f) { keypoint_vector_[i] = p; feature_vector_[i] = f; set_index(cast<vint2>(p.position), int(i)); }
  void update_index(unsigned i, const vint2& p) { set_index(p, int(i)); }

  keypoint_vector_type& keypoints() { return keypoint_vector_; }
  const keypoint_vector_type& keypoints() const { return keypoint_vector_; }
  image2d<int>& index2d() { return index2d_; }
  const image2d<int>& index2d() const { return index2d_; }
  int index_of(vint2& p) const { return index2d_(p); }
  keypoint_type& operator[](unsigned i) { return keypoint_vector_[i]; }
  const keypoint_type& operator[](unsigned i) const { return keypoint_vector_[i]; }
  keypoint_type& operator()(vint2 p) { return keypoint_vector_[index2d_(p)]; }
  const keypoint_type& operator()(vint2 p) const { return keypoint_vector_[index2d_(p)]; }
  int size() const { return int(keypoint_vector_.size()); }
  bool has(vint2 p) const { return index2d_(p) >= 0; }

 private:
  // index2d_ lives on the host only: its cells are addressed through a cached base pointer / pitch (the generic accessor re-checks
  // the device-mirror state on every call, which shows at ~100 k updates per frame)
  void set_index(const vint2& p, int i) {
    if (!idx_base_) { idx_base_ = (char*)&index2d_(0, 0); idx_pitch_ = index2d_.pitch(); }
    *(int*)(idx_base_ + (ptrdiff_t)p[0] * idx_pitch_ + (ptrdiff_t)p[1] * (ptrdiff_t)sizeof(int)) = i;
    touched_.push_back(p);
  }
  char* idx_base_ = nullptr; ptrdiff_t idx_pitch_ = 0;
  std::vector<int> matches_;
  std::vector<vint2> touched_;
  image2d<int> index2d_;
  keypoint_vector_type keypoint_vector_;
  feature_vector_type feature_vector_;
  bool compact_has_run_;
};

struct keypoint_trajectory {
  keypoint_trajectory() : start_frame_(0), alive_(true) {}
  keypoint_trajectory(int frame_cpt) : start_frame_(frame_cpt), alive_(true) {}
  void die() { alive_ = false; }
  bool alive() const { return alive_; }
  vfloat2 position() const { assert(size() > 0); return history_.front(); }
  int size() const { return int(history_.size()); }
  vfloat2 position_at_frame(int frame_cpt) const { return history_[history_.size() - 1 - (frame_cpt - start_frame_)]; }
  void move_to(vfloat2 p) { history_.push_front(p); }
  void pop_oldest_position() { history_.pop_back(); }
  vfloat2 operator[](unsigned i) const { return history_[i]; }
  const std::deque<vfloat2>& positions() const { return history_; }
  int start_frame() const { return start_frame_; }
  int end_frame() const { return start_frame_ + int(history_.size()) - 1; }
  void swap(keypoint_trajectory& o) { std::swap(start_frame_, o.start_frame_); std::swap(alive_, o.alive_); history_.swap(o.history_); }
  friend void swap(keypoint_trajectory& a, keypoint_trajectory& b) { a.swap(b); }
 private:
  int start_frame_;
  bool alive_;
  std::deque<vfloat2> history_;
};

}  // namespace vpp
