// video_extruder.hh — semi-dense keypoint tracker (reference: vpp/algorithms/video_extruder.hh:10-44,
// video_extruder/video_extruder.hpp:24-135).  The whole update runs on the device and the tracker's state — the keypoint
// container (positions, velocities, ages; dead entries kept in place until the next compaction, exactly as keypoint_container
// does) and the trajectories — lives in HBM behind the C ABI (vpp_video_extruder_*, include/vpp_amd.h).  `ctx.keypoints` and
// `ctx.trajectories` are host VIEWS of that state with the reference's types: they are filled from HBM the first time they are
// looked at after an update, so a loop that only tracks never moves a keypoint across PCIe, and they are uploaded again before
// the next update only if the caller really changed them: a non-const access (which includes the implicit conversion to T& of
// `draw::draw_trajectories(display, ctx.trajectories, 200)`, examples/video_extruder.cc:57) marks them as possibly edited, and the next
// update compares the host copies with what was downloaded — equal copies leave the device state untouched (rings, heads and all).
//
// One documented difference: in the view, `keypoints.index2d()` / `has(p)` index every ALIVE keypoint at its position.  The
// reference's index image is rebuilt by side effect (prepare_matching clears it, move / add / compact set cells), so between two
// compactions it misses alive keypoints that the flow did not match in the last update (keypoint_container.hpp:57-63,136-150).
#pragma once
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>
#include <vpp/algorithms/fast_detector/fast.hh>
#include <vpp/algorithms/optical_flow.hh>
#include <vpp/core/keypoint_container.hh>

namespace vpp {
namespace ve_internals {
// optional wall-clock breakdown of video_extruder_update (benchmarks/video_extruder_bench.cc defines VPP_AMD_TIMING)
struct timing_t { double step = 0, upload = 0, view = 0; };
inline timing_t& timing() { static timing_t t; return t; }
#ifdef VPP_AMD_TIMING
struct stopwatch { double& acc; std::chrono::steady_clock::time_point t0; explicit stopwatch(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                   ~stopwatch() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
#else
struct stopwatch { explicit stopwatch(double&) {} };
#endif

typedef keypoint_container<keypoint<int>, int> container_type;
typedef std::vector<keypoint_trajectory> trajectories_type;

// the device tracker + the host copies of its state; shared by copies of the ctx (video_extruder_init returns it by value)
struct state {
  box2d domain;
  vpp_video_extruder* h = nullptr;
  int slots = 0;
  container_type keypoints;
  trajectories_type trajectories;
  bool host_stale = false;    // the device ran an update the host copies have not seen
  bool host_edited = false;   // the caller may have changed the host copies since the last upload (a non-const access was handed out)
  bool force_upload = false;  // the device tracker was rebuilt (longer rings): it is EMPTY, whatever the host copies compare equal to
  int frame_id = -1;
  // the state in the flat form the C ABI moves (positions, velocities, ages, trajectory lengths / start frames / alive flags, ring points newest
  // first from slot 0); `seen` = what the last download produced, compared against the host copies before an upload
  struct flat {
    std::vector<vint2> pos, vel; std::vector<int> age, len, start; std::vector<unsigned char> alive; std::vector<float> ring;
    bool operator==(const flat& o) const {
      auto same2 = [](const std::vector<vint2>& a, const std::vector<vint2>& b) { return a.size() == b.size() && (a.empty() || !std::memcmp(a.data(), b.data(), a.size() * sizeof(vint2))); };
      return same2(pos, o.pos) && same2(vel, o.vel) && age == o.age && len == o.len && start == o.start && alive == o.alive && ring == o.ring;
    }
  };
  flat seen;
  bool seen_valid = false;
  explicit state(box2d d) : domain(d), keypoints(d) {}
  ~state() { if (h) vpp_video_extruder_destroy(h); }
  state(const state&) = delete;
  state& operator=(const state&) = delete;

  void ensure_device(int max_trajectory_length) {
    if (h && max_trajectory_length < slots) return;
    // longer trajectories than the rings hold: rebuild around the host copy.  The new tracker starts empty, so the upload must happen even though the
    // host copies equal what was last downloaded (the host_differs() shortcut of begin_update is about an UNCHANGED device state).
    if (h) { materialise(); host_edited = true; force_upload = true; vpp_video_extruder_destroy(h); h = nullptr; }
    const int capacity = std::max(max_trajectory_length, 15);
    device::check(vpp_video_extruder_create(&h, domain.nrows(), domain.ncols(), capacity), "vpp_video_extruder_create");
    device::check(vpp_video_extruder_trajectory_slots(h, &slots), "vpp_video_extruder_trajectory_slots");
  }
  // HBM -> host copies (keypoint_container + std::vector<keypoint_trajectory>)
  void materialise() {
    if (!host_stale) return;
    stopwatch sw(timing().view);
    host_stale = false;
    int n = 0;
    device::check(vpp_video_extruder_count(h, &n, &frame_id), "vpp_video_extruder_count");
    std::vector<vint2> pos(n), vel(n); std::vector<int> age(n), len(n), start(n), head(n); std::vector<unsigned char> alive(n);
    std::vector<float> ring(size_t(n) * slots * 2);
    device::check(vpp_video_extruder_keypoints(h, (int32_t*)pos.data(), (int32_t*)vel.data(), age.data(), n, device::stream()), "vpp_video_extruder_keypoints");
    device::check(vpp_video_extruder_trajectories(h, len.data(), start.data(), alive.data(), head.data(), ring.data(), n, device::stream()), "vpp_video_extruder_trajectories");
    keypoints.prepare_matching();
    auto& kv = keypoints.keypoints();
    kv.resize(n);
    for (int i = 0; i < n; i++) { kv[i].position = pos[i]; kv[i].velocity = vel[i]; kv[i].age = age[i]; if (age[i] > 0) keypoints.update_index(i, pos[i]); }
    keypoints.resize_features(n);
    trajectories.assign(n, keypoint_trajectory());
    for (int i = 0; i < n; i++) {
      keypoint_trajectory t(start[i]);
      for (int k = len[i] - 1; k >= 0; k--) { const float* p = &ring[(size_t(i) * slots + (head[i] + k) % slots) * 2]; t.move_to(vfloat2(p[0], p[1])); }  // oldest first: move_to pushes to the front
      if (!alive[i]) t.die();
      trajectories[i].swap(t);
    }
    seen = pack();   // what an unedited host copy packs to
    seen_valid = true;
  }
  // the host copies in the flat form of the C ABI
  flat pack() const {
    flat f;
    const int n = keypoints.size();
    f.pos.resize(n); f.vel.resize(n); f.age.resize(n); f.len.resize(n); f.start.resize(n); f.alive.resize(n); f.ring.assign(size_t(n) * slots * 2, 0.f);
    for (int i = 0; i < n; i++) {
      f.pos[i] = keypoints[i].position; f.vel[i] = keypoints[i].velocity; f.age[i] = keypoints[i].age;
      const keypoint_trajectory& t = trajectories[i];
      f.len[i] = std::min(t.size(), slots - 1); f.start[i] = t.start_frame(); f.alive[i] = t.alive();
      for (int k = 0; k < f.len[i]; k++) { f.ring[(size_t(i) * slots + k) * 2] = t[k][0]; f.ring[(size_t(i) * slots + k) * 2 + 1] = t[k][1]; }
    }
    return f;
  }
  // a non-const access was handed out since the download: did anything actually change?  (an unchanged copy needs no upload — the device state,
  // ring heads included, stays as it is)
  bool host_differs() {
    if (!seen_valid || (int)trajectories.size() != keypoints.size()) return true;
    return !(pack() == seen);
  }
  // host copies -> HBM (the caller edited them)
  void upload() {
    stopwatch sw(timing().upload);
    const int n = keypoints.size();
    trajectories.resize(n, keypoint_trajectory(frame_id));
    const flat f = pack();
    const std::vector<int> head(n, 0);
    device::check(vpp_video_extruder_upload(h, n, frame_id, (const int32_t*)f.pos.data(), (const int32_t*)f.vel.data(), f.age.data(), f.len.data(), f.start.data(), f.alive.data(), head.data(),
                                            f.ring.data(), device::stream()), "vpp_video_extruder_upload");
    host_edited = false;
    force_upload = false;
    seen_valid = false;
  }
};

// a member of the ctx that reads like the reference's member (same type behind it) and keeps it coherent with HBM
template <class T, T state::*M> class view {
 public:
  explicit view(std::shared_ptr<state> s) : s_(std::move(s)) {}
  const T& get() const { s_->materialise(); return (*s_).*M; }
  T& edit() { s_->materialise(); s_->host_edited = true; return (*s_).*M; }
  operator const T&() const { return get(); }
  operator T&() { return edit(); }
  T* operator->() { return &edit(); }
  const T* operator->() const { return &get(); }
  auto size() const { return get().size(); }
  decltype(auto) operator[](size_t i) { return edit()[i]; }
  decltype(auto) operator[](size_t i) const { return get()[i]; }
  auto begin() { return edit().begin(); }
  auto end() { return edit().end(); }
  auto begin() const { return get().begin(); }
  auto end() const { return get().end(); }
 private:
  std::shared_ptr<state> s_;
};
}  // namespace ve_internals

struct video_extruder_ctx {
  video_extruder_ctx(box2d domain) : state_(std::make_shared<ve_internals::state>(domain)), keypoints(state_), trajectories(state_), frame_id(0) {}
 private:
  std::shared_ptr<ve_internals::state> state_;   // declared first: the views below share it
 public:
  ve_internals::view<ve_internals::container_type, &ve_internals::state::keypoints> keypoints;            // keypoint_container<keypoint<int>, int>  (video_extruder.hh:14-16)
  ve_internals::view<ve_internals::trajectories_type, &ve_internals::state::trajectories> trajectories;   // std::vector<keypoint_trajectory>        (:18-20)
  int frame_id;
  ve_internals::state& internal_state() { return *state_; }
};
inline video_extruder_ctx video_extruder_init(box2d domain) { video_extruder_ctx res(domain); res.frame_id = -1; return res; }

namespace ve_internals {
// the options of video_extruder.hpp:35-41 and the host-view bookkeeping every update starts with
template <class... OPTS> vpp_video_extruder_params begin_update(video_extruder_ctx& ctx, OPTS... options) {
  auto opts = opt::make(options...);
  vpp_video_extruder_params p;
  p.detector_th = opts.get(_detector_th, 10); p.keypoint_spacing = opts.get(_keypoint_spacing, 10); p.detector_period = opts.get(_detector_period, 5);
  p.max_trajectory_length = opts.get(_max_trajectory_length, 15); p.nscales = opts.get(_nscales, 3); p.winsize = opts.get(_winsize, 9);
  p.propagation = opts.get(_propagation, 2);
  state& s = ctx.internal_state();
  s.ensure_device(p.max_trajectory_length);
  if (s.host_edited && !s.force_upload && s.frame_id == ctx.frame_id) {   // a non-const view was handed out: upload only if the copies really changed
    s.materialise();
    if (!s.host_differs()) s.host_edited = false;
  }
  if (s.host_edited || s.force_upload || s.frame_id != ctx.frame_id) {  // the caller edited the views (or ctx.frame_id): the device continues from the host's copy
    s.materialise();
    s.frame_id = ctx.frame_id;
    s.upload();
  }
  return p;
}
inline void end_update(video_extruder_ctx& ctx, bool wait = true) {
  state& s = ctx.internal_state();
  if (wait) device::check(vpp_sync(device::stream()), "vpp_sync");   // synchronous like the reference call
  ctx.frame_id++;
  s.frame_id = ctx.frame_id;
  s.host_stale = true;
}
}  // namespace ve_internals

template <class... OPTS>
void video_extruder_update(video_extruder_ctx& ctx, const image2d<unsigned char>& frame1, const image2d<unsigned char>& frame2, OPTS... options) {
  const vpp_video_extruder_params p = ve_internals::begin_update(ctx, options...);
  ve_internals::state& s = ctx.internal_state();
  ve_internals::stopwatch sw(ve_internals::timing().step);
  const vpp_image_desc d1 = frame1.device_desc(false), d2 = frame2.device_desc(false);
  device::check(vpp_video_extruder_step(s.h, &d1, &d2, &p, device::stream()), "vpp_video_extruder_step");
  ve_internals::end_update(ctx);
}

// Extension (no reference counterpart): the video loop of examples/video_extruder.cc:43-58 with `prev` kept by the tracker.  One call per frame —
// gray (unsigned char) or colour (vuchar3 / vuchar4: the loop's rgb_to_graylevel happens on the device, fused with the pyramid) — replaces
// clone(_border = 3) + fill_border_mirror + rgb_to_graylevel + video_extruder_update + copy(frame_gl, prev_frame).  The first frame only becomes
// `prev` (returns false, like the example's `first`); later calls run one update (return true) whose results are those of video_extruder_update on
// the mirror-bordered gray frames.  The frame's own border is not read.
// (A call that changes _nscales / _winsize, or asks for longer trajectories than the tracker's rings hold, starts over: that frame only becomes `prev`.)
// A frame whose newest pixels are on the host (a decoder wrote them) is handed over as a host frame: the tracker stages it on a copy stream while the
// previous update still computes (vpp_video_extruder_push_host_frame).  The call returns when the frame has been consumed — the image may be written
// again — and the update itself completes asynchronously: looking at ctx.keypoints / ctx.trajectories (or any synchronous call) waits for it.
template <class V, class... OPTS>
bool video_extruder_push_frame(video_extruder_ctx& ctx, const image2d<V>& frame, OPTS... options) {
  static_assert(std::is_same<V, unsigned char>::value || std::is_same<V, vuchar3>::value || std::is_same<V, vuchar4>::value, "gray, rgb or rgba 8-bit frames");
  const vpp_video_extruder_params p = ve_internals::begin_update(ctx, options...);
  ve_internals::state& s = ctx.internal_state();
  ve_internals::stopwatch sw(ve_internals::timing().step);
  int before = 0, after = 0;
  device::check(vpp_video_extruder_count(s.h, nullptr, &before), "vpp_video_extruder_count");   // (the frame id alone: asking for the size could wait for a re-detection)
  if (frame.device_current()) {
    const vpp_image_desc d = frame.device_desc(false);
    device::check(vpp_video_extruder_push_frame(s.h, &d, &p, device::stream()), "vpp_video_extruder_push_frame");
  } else {
    const vpp_image_desc d = frame.host_desc();
    device::check(vpp_video_extruder_push_host_frame(s.h, &d, &p, device::stream()), "vpp_video_extruder_push_host_frame");
  }
  device::check(vpp_video_extruder_count(s.h, nullptr, &after), "vpp_video_extruder_count");
  if (after == before) return false;
  ve_internals::end_update(ctx, false);
  return true;
}
}  // namespace vpp
