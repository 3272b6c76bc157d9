// video_extruder.hh — semi-dense keypoint tracker (reference: vpp/algorithms/video_extruder.hh:10-44,
// video_extruder/video_extruder.hpp:24-135).  Flow, FAST scores and FAST re-detection run on the device; the keypoint
// merge (which particles converged to the same cell) is decided on the device; applying it, the cull and the trajectories stay on
// the host as in the reference.
#pragma once
#include <chrono>
#include <vector>
#include <vpp/algorithms/fast_detector/fast.hh>
#include <vpp/algorithms/optical_flow.hh>
#include <vpp/core/keypoint_container.hh>

namespace vpp {
struct video_extruder_ctx {
  video_extruder_ctx(box2d domain) : keypoints(domain), frame_id(0) {
    // a std::vector<keypoint_trajectory> that reallocates COPIES every std::deque (libstdc++'s deque move is not noexcept): 24 ms at 75 k keypoints
    trajectories.reserve(size_t(domain.nrows()) * domain.ncols() / 50);
  }
  keypoint_container<keypoint<int>, int> keypoints;
  std::vector<keypoint_trajectory> trajectories;
  int frame_id;
};
inline video_extruder_ctx video_extruder_init(box2d domain) { video_extruder_ctx res(domain); res.frame_id = -1; return res; }

namespace ve_internals {
// optional wall-clock breakdown of video_extruder_update (benchmarks/video_extruder_bench.cc defines VPP_AMD_TIMING)
struct timing_t { double flow = 0, merge = 0, scores = 0, redetect = 0, traj = 0, redetect_mask = 0, redetect_fast9 = 0, redetect_add = 0, redetect_compact = 0, redetect_sync = 0; };
inline timing_t& timing() { static timing_t t; return t; }
#ifdef VPP_AMD_TIMING
struct stopwatch { double& acc; std::chrono::steady_clock::time_point t0; explicit stopwatch(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                   ~stopwatch() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
#else
struct stopwatch { explicit stopwatch(double&) {} };
#endif
}
namespace ve_internals { struct position_view { const video_extruder_ctx* c; int size() const { return c->keypoints.size(); } vint2 operator[](int i) const { return c->keypoints[i].position; } }; }

template <class... OPTS>
void video_extruder_update(video_extruder_ctx& ctx, const image2d<unsigned char>& frame1, const image2d<unsigned char>& frame2, OPTS... options) {
  ctx.frame_id++;
  auto opts = opt::make(options...);
  const int detector_th = opts.get(_detector_th, 10), keypoint_spacing = opts.get(_keypoint_spacing, 10), detector_period = opts.get(_detector_period, 5);
  const int max_trajectory_length = opts.get(_max_trajectory_length, 15), nscales = opts.get(_nscales, 3), winsize = opts.get(_winsize, 9);
  const int regularisation_niters = opts.get(_propagation, 2);

  // One device submission per update: the flow, then — on the keypoints as the match callback is about to leave them — the
  // merge of particles that converged to the same cell (:60-84) and the FAST scores (:87-91); one wait.  The host then
  // applies move / remove per keypoint in one pass (the three loops of the reference only interact through `age`, which the
  // merge kernel reconstructs; the index log replays to the same image whether a removal follows its own move or all moves).
  const int n = ctx.keypoints.size();
  ctx.keypoints.prepare_matching();
  if (n) {
    device::hbuf<int> scores(n), ages(n);
    device::hbuf<unsigned char> merged(n);
    of_internals::flow_buffers b(n);
    {
      ve_internals::stopwatch sw(ve_internals::timing().flow);
      for (int i = 0; i < n; i++) ages[i] = ctx.keypoints[i].age;
      const of_internals::flow_params fp{winsize, nscales, 0, regularisation_niters, 5};
      of_internals::run(ve_internals::position_view{&ctx}, frame1, frame2, fp, b, [&] {
        const vpp_image_desc d2 = frame2.device_desc(false);
        device::check(vpp_keypoint_merge((const int32_t*)b.pos.data(), (const int32_t*)b.dk.p, b.valid.data(), ages.data(), n, frame2.nrows(), frame2.ncols(),
                                         keypoint_spacing, merged.data(), device::stream()), "vpp_keypoint_merge");
        device::check(vpp_fast9_scores_moved(&d2, detector_th, (const int32_t*)b.pos.data(), (const int32_t*)b.dk.p, n, scores.data(), device::stream()),
                      "vpp_fast9_scores_moved");
      });
    }
    ve_internals::stopwatch sw(ve_internals::timing().merge);
    for (int i = 0; i < n; i++) {
      if (b.valid[i]) { if (frame1.has(b.pos[i])) ctx.keypoints.move(i, b.pos[i]); else ctx.keypoints.remove(i); }  // :50-53
      if (merged[i] || scores[i] < 3) ctx.keypoints.remove(i);                                                        // :60-84, :87-91
    }
  }
  if (!(ctx.frame_id % detector_period)) {  // re-detect away from the live keypoints (:94-119)
    ve_internals::stopwatch sw(ve_internals::timing().redetect);
    image2d<unsigned char> mask(frame2.domain().nrows(), frame2.domain().ncols(), _border = keypoint_spacing);
    {  ve_internals::stopwatch sw2(ve_internals::timing().redetect_mask);
       // fill_with_border(mask, 1) + the 2s x 2s zero square of every container entry (:101-110), built in HBM
      const int n = ctx.keypoints.size();
      std::vector<vint2> pts(n);
      for (int i = 0; i < n; i++) pts[i] = ctx.keypoints[i].position;
      device::dbuf rc(size_t(n) * 8);
      rc.upload(pts.data(), size_t(n) * 8);
      const vpp_image_desc dm = mask.device_desc(true, true);
      device::check(vpp_keypoint_mask(&dm, (const int32_t*)rc.p, n, keypoint_spacing, device::stream()), "vpp_keypoint_mask");
    }
    std::vector<vint2> kps;
    { ve_internals::stopwatch sw2(ve_internals::timing().redetect_fast9); kps = fast9(frame2, detector_th, _blockwise, _block_size = keypoint_spacing, _mask = mask); }
    { ve_internals::stopwatch sw2(ve_internals::timing().redetect_add); for (auto kp : kps) ctx.keypoints.add(keypoint<int>(kp)); }
    { ve_internals::stopwatch sw2(ve_internals::timing().redetect_compact); ctx.keypoints.compact(); }
    { ve_internals::stopwatch sw2(ve_internals::timing().redetect_sync); ctx.keypoints.sync_attributes(ctx.trajectories, keypoint_trajectory(ctx.frame_id)); }
  }
  ve_internals::stopwatch sw(ve_internals::timing().traj);
  for (int i = 0; i < ctx.keypoints.size(); i++) {  // trajectories (:123-133)
    if (ctx.keypoints[i].alive()) {
      ctx.trajectories[i].move_to(ctx.keypoints[i].position.template cast<float>());
      if (ctx.trajectories[i].size() > max_trajectory_length) ctx.trajectories[i].pop_oldest_position();
    } else ctx.trajectories[i].die();
  }
}
}  // namespace vpp
