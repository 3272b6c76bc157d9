// keypoint_container.hh — keypoints + features with a 2-D index (reference: vpp/core/keypoint_container.hh:13-90,
// keypoint_container.hpp:11-167) and trajectories (vpp/core/keypoint_trajectory.hh:11-72).  Host-side bookkeeping.
// The index image is refilled only over the cells that were set (the reference memsets the whole frame-sized image on
// every prepare_matching: 8-33 MB per frame), and it is written LAZILY: move / add / remove / compact append
// (cell, value) records to a log that is replayed, in order, the first time the index is looked at (has, operator()(pos),
// index2d(), index_of, remove(pos)).  A tracker that only moves and culls keypoints never pays the ~100 k scattered
// stores per frame into a frame-sized image (1.3 ms of a 4K video_extruder update); observable behaviour is the same.
#pragma once
#include <atomic>
#include <cassert>
#include <deque>
#include <mutex>
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
    log_.clear();               // whatever was pending is overwritten by the fill
    reset_pending_ = true;
    lazy_.dirty.store(true, std::memory_order_release);
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
    log_op(cast<vint2>(keypoint_vector_[i].position), ~i);  // "if (index == i) index = -1"
  }
  void remove(vint2 pos) { assert(has(pos)); remove(cell(pos)); }
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
  void resize_features(size_t n) { feature_vector_.resize(n); }   // after keypoints() was assigned wholesale (the device tracker's host view)

  keypoint_vector_type& keypoints() { return keypoint_vector_; }
  const keypoint_vector_type& keypoints() const { return keypoint_vector_; }
  // a caller holding the mutable image may write cells this container does not know about: the next prepare_matching refills it whole
  image2d<int>& index2d() { materialise(); refill_whole_ = true; return index2d_; }
  const image2d<int>& index2d() const { materialise(); return index2d_; }
  int index_of(vint2& p) const { return cell(p); }
  keypoint_type& operator[](unsigned i) { return keypoint_vector_[i]; }
  const keypoint_type& operator[](unsigned i) const { return keypoint_vector_[i]; }
  keypoint_type& operator()(vint2 p) { return keypoint_vector_[cell(p)]; }
  const keypoint_type& operator()(vint2 p) const { return keypoint_vector_[cell(p)]; }
  int size() const { return int(keypoint_vector_.size()); }
  bool has(vint2 p) const { return cell(p) >= 0; }

 private:
  // One pending write to the index image: i >= 0 sets the cell to i, i < 0 clears it if it holds ~i (remove(int)).
  struct index_op { int r, c, i; };
  // const look-ups may come from several threads at once (the reference's were plain reads): the first one replays the log under a mutex
  struct lazy_state {
    std::atomic<bool> dirty{false}; std::mutex m;
    lazy_state() {}
    lazy_state(const lazy_state& o) : dirty(o.dirty.load()) {}
    lazy_state& operator=(const lazy_state& o) { dirty.store(o.dirty.load()); return *this; }
  };
  void set_index(const vint2& p, int i) { log_op(p, i); }
  void log_op(const vint2& p, int i) {
    log_.push_back(index_op{p[0], p[1], i});
    lazy_.dirty.store(true, std::memory_order_release);
    if (log_.size() > (size_t(1) << 22)) materialise();  // bound the log of a caller that never looks at the index
  }
  int& cell_ref(int r, int c) const { return *(int*)(idx_base_ + (ptrdiff_t)r * idx_pitch_ + (ptrdiff_t)c * (ptrdiff_t)sizeof(int)); }
  int cell(const vint2& p) const { materialise(); return cell_ref(p[0], p[1]); }
  void materialise() const {
    if (!lazy_.dirty.load(std::memory_order_acquire) && idx_base_) return;
    std::lock_guard<std::mutex> lock(lazy_.m);
    // index2d_ lives on the host only: its cells are addressed through a cached base pointer / pitch (the generic accessor
    // re-checks the device-mirror state on every call)
    if (!idx_base_) { idx_base_ = (char*)&const_cast<image2d<int>&>(index2d_)(0, 0); idx_pitch_ = index2d_.pitch(); }
    if (!lazy_.dirty.load(std::memory_order_relaxed)) return;
    if (reset_pending_) {
      if (refill_whole_) { fill_with_border(const_cast<image2d<int>&>(index2d_), -1); refill_whole_ = false; }
      else for (const vint2& p : touched_) cell_ref(p[0], p[1]) = -1;
      touched_.clear();
      reset_pending_ = false;
    }
    for (const index_op& o : log_) {
      int& c = cell_ref(o.r, o.c);
      if (o.i >= 0) { c = o.i; touched_.push_back(vint2(o.r, o.c)); }
      else if (c == ~o.i) c = -1;
    }
    log_.clear();
    lazy_.dirty.store(false, std::memory_order_release);
  }
  mutable char* idx_base_ = nullptr; mutable ptrdiff_t idx_pitch_ = 0;
  std::vector<int> matches_;
  mutable std::vector<index_op> log_;
  mutable std::vector<vint2> touched_;
  mutable bool reset_pending_ = false, refill_whole_ = false;
  mutable lazy_state lazy_;
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
