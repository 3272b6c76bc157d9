// ref_video_extruder.cpp — the reference's video_extruder (vpp/algorithms/video_extruder.hh, video_extruder/video_extruder.hpp:12-135)
// run over a frame sequence.  TEST INFRASTRUCTURE ONLY.  Built into its own library with -DNDEBUG, the flag of the
// reference's example / benchmark builds (examples/CMakeLists.txt, benchmarks/CMakeLists.txt:10): with assertions on,
// keypoint_container::sync_attributes aborts on "age != 1" (keypoint_container.hpp:82) as soon as a keypoint of the
// first detection was neither moved nor removed by the flow step — the tracker is only ever run without assertions upstream.
#include <vpp/vpp.hh>
#include <vpp/algorithms/video_extruder.hh>

#include "../../include/vpp_amd.h"

using namespace vpp;

namespace {
template <class V> image2d<V> wrap(const vpp_image_desc* d) {
  return image2d<V>(make_box2d(d->nrows, d->ncols), _data = (V*)d->first_pixel, _pitch = (int)d->pitch, _border = (int)d->border);
}
static int dump(video_extruder_ctx& ctx, int32_t* out, int32_t* traj_len, int capacity, int* count, int* frame_id) {
  *count = ctx.keypoints.size();
  *frame_id = ctx.frame_id;
  for (int i = 0; i < ctx.keypoints.size() && i < capacity; i++) {
    const auto& k = ctx.keypoints[i];
    out[5 * i] = k.position[0]; out[5 * i + 1] = k.position[1]; out[5 * i + 2] = k.velocity[0]; out[5 * i + 3] = k.velocity[1]; out[5 * i + 4] = k.age;
    traj_len[i] = i < (int)ctx.trajectories.size() ? ctx.trajectories[i].size() : -1;
  }
  return ctx.keypoints.size() > capacity ? VPP_ERR_CAPACITY : 0;
}
}  // namespace

#pragma GCC visibility push(default)
extern "C" {

// video_extruder_init + video_extruder_update over a frame sequence (video_extruder.hpp:12-135).  out: 5 ints per keypoint
// (row, col, velocity row, velocity col, age) after the last frame; traj_len: trajectory length per keypoint.
int ref_video_extruder_run(const vpp_image_desc* frames, int nframes, int detector_th, int keypoint_spacing, int detector_period,
                           int max_trajectory_length, int nscales, int winsize, int propagation, int32_t* out, int32_t* traj_len, int capacity,
                           int* count, int* frame_id) {
  video_extruder_ctx ctx = video_extruder_init(make_box2d(frames[0].nrows, frames[0].ncols));
  for (int t = 1; t < nframes; t++) {
    auto f1 = wrap<unsigned char>(&frames[t - 1]); auto f2 = wrap<unsigned char>(&frames[t]);
    video_extruder_update(ctx, f1, f2, _detector_th = detector_th, _keypoint_spacing = keypoint_spacing, _detector_period = detector_period,
                          _max_trajectory_length = max_trajectory_length, _nscales = nscales, _winsize = winsize, _propagation = propagation);
  }
  return dump(ctx, out, traj_len, capacity, count, frame_id);
}
// The same with a per-update _max_trajectory_length (max_len[t - 1] for the update frames[t - 1] -> frames[t]): the option is an argument of every
// call in the reference (video_extruder.hpp:40), a caller may change it mid-sequence (trajectories then grow beyond / stay above the old bound).
int ref_video_extruder_run_schedule(const vpp_image_desc* frames, int nframes, int detector_th, int keypoint_spacing, int detector_period,
                                    const int* max_len, int nscales, int winsize, int propagation, int32_t* out, int32_t* traj_len, int capacity,
                                    int* count, int* frame_id) {
  video_extruder_ctx ctx = video_extruder_init(make_box2d(frames[0].nrows, frames[0].ncols));
  for (int t = 1; t < nframes; t++) {
    auto f1 = wrap<unsigned char>(&frames[t - 1]); auto f2 = wrap<unsigned char>(&frames[t]);
    video_extruder_update(ctx, f1, f2, _detector_th = detector_th, _keypoint_spacing = keypoint_spacing, _detector_period = detector_period,
                          _max_trajectory_length = max_len[t - 1], _nscales = nscales, _winsize = winsize, _propagation = propagation);
  }
  return dump(ctx, out, traj_len, capacity, count, frame_id);
}

}  // extern "C"
#pragma GCC visibility pop
