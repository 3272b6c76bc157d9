// video_extruder_bench.cc — frames/s of video_extruder_update on 4K frames (defaults of
// video_extruder.hpp:35-41 — th 10, spacing 10, period 5, nscales 3, winsize 9, propagation 2), written against the
// drop-in <vpp/...> surface exactly like the reference's examples/video_extruder.cc:44-58 loop.
// usage: video_extruder_bench [nrows ncols nframes]   -> one JSON line on stdout
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define VPP_AMD_TIMING 1
#include <vpp/vpp.hh>
#include <vpp/algorithms/video_extruder.hh>

using namespace vpp;
typedef std::chrono::steady_clock clk;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char** argv) {
  const int nr = argc > 1 ? std::atoi(argv[1]) : 2160, nc = argc > 2 ? std::atoi(argv[2]) : 3840, T = argc > 3 ? std::atoi(argv[3]) : 12;
  // smooth random texture (3 box blurs of white noise) + high-contrast rectangles, translated by (1, 2) px per frame
  const int W = nc + 2 * T + 32, H = nr + T + 32;
  std::mt19937 rng(6);
  std::vector<float> a(size_t(W) * H), b(a.size());
  for (auto& x : a) x = float(rng() & 0xFFFF);
  for (int pass = 0; pass < 3; pass++) {
    for (int r = 0; r < H; r++) for (int c = 2; c < W - 2; c++) b[size_t(r) * W + c] = (a[size_t(r) * W + c - 2] + a[size_t(r) * W + c - 1] + a[size_t(r) * W + c] + a[size_t(r) * W + c + 1] + a[size_t(r) * W + c + 2]) / 5;
    for (int r = 2; r < H - 2; r++) for (int c = 0; c < W; c++) a[size_t(r) * W + c] = (b[size_t(r - 2) * W + c] + b[size_t(r - 1) * W + c] + b[size_t(r) * W + c] + b[size_t(r + 1) * W + c] + b[size_t(r + 2) * W + c]) / 5;
  }
  float lo = 1e30f, hi = -1e30f;
  for (int r = 8; r < H - 8; r++) for (int c = 8; c < W - 8; c++) { lo = std::min(lo, a[size_t(r) * W + c]); hi = std::max(hi, a[size_t(r) * W + c]); }
  std::vector<unsigned char> scene(a.size());
  for (size_t i = 0; i < a.size(); i++) scene[i] = (unsigned char)std::min(255.f, std::max(0.f, (a[i] - lo) / (hi - lo) * 255.f));
  for (int k = 0; k < (nr / 40) * (nc / 40); k++) {
    const int r = 8 + rng() % (H - 40), c = 8 + rng() % (W - 40), h = 6 + rng() % 18, w = 6 + rng() % 18, v = rng() & 255;
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) scene[size_t(r + i) * W + c + j] = (unsigned char)v;
  }
  std::vector<image2d<unsigned char>> frames;
  for (int t = 0; t < T; t++) {
    image2d<unsigned char> f(nr, nc, _border = 3);
    for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) f(r, c) = scene[size_t(r + 8 + (T - t)) * W + c + 8 + 2 * (T - t)];  // content moves by (+1, +2) per frame
    fill_border_mirror(f);
    frames.push_back(f);
  }
  video_extruder_ctx ctx = video_extruder_init(make_box2d(nr, nc));
  std::vector<double> per;
  std::vector<int> nk;
  for (int t = 1; t < T; t++) {
    const auto t0 = clk::now();
    video_extruder_update(ctx, frames[t - 1], frames[t], _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15,
                          _nscales = 3, _winsize = 9, _propagation = 2);
    per.push_back(ms(t0, clk::now()));
    { int cnt = 0, fid = 0; vpp_video_extruder_count(ctx.internal_state().h, &cnt, &fid); nk.push_back(cnt); }   // the container size without materialising the host view
    if (t == 1) ve_internals::timing() = ve_internals::timing_t();  // the first update only detects (no keypoints yet): excluded from the breakdown
  }
  double sum = 0; for (size_t i = 1; i < per.size(); i++) sum += per[i];
  const double mean = sum / double(per.size() - 1);
  std::vector<double> steady(per.begin() + 2, per.end());   // without the detecting update and the one after it (first use of the flow's scratch: allocations)
  std::sort(steady.begin(), steady.end());
  const double median = steady[steady.size() / 2];
  const auto tm = ve_internals::timing();   // before the host views are looked at below
  // The loop's own shape, one call per frame (video_extruder_push_frame: the tracker keeps `prev` and its pyramid; the call returns when the frame has been
  // consumed, the update completes asynchronously).  Timed as a whole: frames 3 .. T-1 including the final wait, per frame — detection frames included.
  // Colour frames: the chain a caller of the reference's API writes (rgb_to_graylevel_mirror + video_extruder_update) beside the one call; frames resident
  // in HBM, and frames in pinned host memory as a decoder leaves them (the one call stages frame t + 1 on a copy stream while update t computes).
  const int first = 3;
  auto sync = [] { device::check(vpp_sync(device::stream()), "vpp_sync"); };
  auto per_frame = [&](auto&& body) {   // body(t) for every frame; returns ms per frame over [first, T)
    clk::time_point t0;
    for (int t = 0; t < T; t++) { if (t == first) { sync(); t0 = clk::now(); } body(t); }
    sync();
    return ms(t0, clk::now()) / double(T - first);
  };
  std::vector<void*> pinned;
  std::vector<image2d<vuchar3>> colour, colour_host;
  std::vector<image2d<unsigned char>> gray_host;
  for (int t = 0; t < T; t++) {
    const image2d<unsigned char>& gray = frames[t];   // (a const access: the frame's HBM mirror stays valid)
    void *h3 = nullptr, *h1 = nullptr;
    device::check(vpp_malloc_host(size_t(nr) * nc * 3, &h3), "vpp_malloc_host"); device::check(vpp_malloc_host(size_t(nr) * nc, &h1), "vpp_malloc_host");
    pinned.push_back(h3); pinned.push_back(h1);
    image2d<vuchar3> c3(nr, nc, _border = 0), p3(nr, nc, _data = (vuchar3*)h3, _pitch = nc * 3);
    image2d<unsigned char> p1(nr, nc, _data = (unsigned char*)h1, _pitch = nc);
    for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) { const unsigned char g = gray(r, c); c3(r, c) = vuchar3(g, g, g); p3(r, c) = vuchar3(g, g, g); p1(r, c) = g; }
    (void)c3.device_desc(false); (void)gray.device_desc(false);   // resident in HBM before the timed loops
    colour.push_back(c3); colour_host.push_back(p3); gray_host.push_back(p1);
  }
  auto push_leg = [&](auto& seq, int* entries) {
    video_extruder_ctx c = video_extruder_init(make_box2d(nr, nc));
    const double v = per_frame([&](int t) { video_extruder_push_frame(c, seq[t], _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15, _nscales = 3, _winsize = 9, _propagation = 2); });
    int fid = 0; vpp_video_extruder_count(c.internal_state().h, entries, &fid);
    return v;
  };
  auto chain_leg = [&](auto& seq) {
    video_extruder_ctx c = video_extruder_init(make_box2d(nr, nc));
    image2d<unsigned char> prev, cur;
    return per_frame([&](int t) {
      cur = rgb_to_graylevel_mirror(seq[t], 3);   // clone(_border = 3) + fill_border_mirror + rgb_to_graylevel of the example's loop, one device pass (uploads a host frame first)
      if (t > 0) video_extruder_update(c, prev, cur, _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15, _nscales = 3, _winsize = 9, _propagation = 2);
      prev.swap(cur);
    });
  };
  int push_entries[4] = {0, 0, 0, 0};
  const double update_gray = [&] {   // the reference's two-frame call on resident gray frames, timed the same way
    video_extruder_ctx c = video_extruder_init(make_box2d(nr, nc));
    return per_frame([&](int t) { if (t > 0) video_extruder_update(c, frames[t - 1], frames[t], _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15, _nscales = 3, _winsize = 9, _propagation = 2); });
  }();
  const double push_rgb_host = push_leg(colour_host, &push_entries[2]);     // host legs first: they leave the images' mirrors untouched ...
  const double push_gray_host = push_leg(gray_host, &push_entries[3]);
  auto nowait_leg = [&](auto& seq) {   // the C ABI's two-buffer protocol: wait for the frame before last, push without waiting
    vpp_video_extruder* h = nullptr;
    device::check(vpp_video_extruder_create(&h, nr, nc, 15), "vpp_video_extruder_create");
    const vpp_video_extruder_params p{10, 10, 5, 15, 3, 9, 2};
    const double v = per_frame([&](int t) {
      const vpp_image_desc d = seq[t].host_desc();
      device::check(vpp_video_extruder_wait_host_frame(h, 1), "vpp_video_extruder_wait_host_frame");
      device::check(vpp_video_extruder_push_host_frame_nowait(h, &d, &p, device::stream()), "vpp_video_extruder_push_host_frame_nowait");
    });
    vpp_video_extruder_destroy(h);
    return v;
  };
  const double nowait_gray_host = nowait_leg(gray_host), nowait_rgb_host = nowait_leg(colour_host);
  const double chain_rgb_host = chain_leg(colour_host);                      // ... this one uploads them (as a caller of the plain API would, frame by frame)
  const double push_gray = push_leg(frames, &push_entries[0]);
  const double push_rgb = push_leg(colour, &push_entries[1]);
  const double chain_rgb = chain_leg(colour);
  const double n = double(per.size() - 1);
  const auto tv0 = clk::now();
  int alive = 0, good = 0;
  for (int i = 0; i < ctx.keypoints.size(); i++) if (ctx.keypoints[i].alive()) { alive++; good += ctx.keypoints[i].velocity == vint2(1, 2); }
  size_t traj_points = 0;
  for (const auto& t : ctx.trajectories) traj_points += size_t(t.size());
  const double view_ms = ms(tv0, clk::now());
  std::printf("{\"workload\": \"video_extruder_update %dx%d uchar, defaults (th 10, spacing 10, period 5, 3 scales, winsize 9, 2 sweeps), %d updates after the detecting one\", "
              "\"ms_per_update_incl_the_sequence_start\": %.3f, \"ms_per_update_median_steady\": %.3f, \"ms_per_update\": %.3f, \"frames_per_s\": %.2f, \"keypoints\": %d, \"alive\": %d, \"velocity_ok\": %d, \"trajectory_points\": %zu, "
              "\"breakdown_ms\": {\"device_step_incl_wait\": %.3f, \"host_upload\": %.3f, \"host_view_during_updates\": %.3f}, "
              "\"host_view_once_after_the_run_ms\": %.3f, \"ms_per_frame_frames_3_to_end_incl_detection_frames\": {\"frames_in_hbm\": {\"video_extruder_update_gray\": %.3f, \"push_frame_gray\": %.3f, \"push_frame_rgb\": %.3f, \"rgb_to_graylevel_mirror_then_update\": %.3f}, \"frames_in_pinned_host_memory\": {\"push_frame_gray\": %.3f, \"push_frame_rgb\": %.3f, \"two_host_buffers_nowait_gray\": %.3f, \"two_host_buffers_nowait_rgb\": %.3f, \"rgb_to_graylevel_mirror_then_update\": %.3f}, \"entries\": [%d, %d, %d, %d]}, \"state\": \"keypoints and trajectories resident in HBM (vpp_video_extruder_*)\", \"per_update_ms\": [",
              nr, nc, int(per.size() - 1), mean, median, update_gray, 1000.0 / update_gray /* ONE rate: video_extruder_update on resident gray frames, frames 3 .. end incl. the detection frames and the final wait */, nk.back(), alive, good, traj_points, tm.step / n, tm.upload / n, tm.view / n, view_ms, update_gray, push_gray, push_rgb, chain_rgb, push_gray_host, push_rgb_host, nowait_gray_host, nowait_rgb_host, chain_rgb_host, push_entries[0], push_entries[1], push_entries[2], push_entries[3]);
  for (size_t i = 0; i < per.size(); i++) std::printf("%s%.2f", i ? ", " : "", per[i]);
  std::printf("]}\n");
  colour_host.clear(); gray_host.clear();
  for (void* h : pinned) vpp_free_host(h);
  return alive > 0 && good > alive / 2 ? 0 : 1;
}
