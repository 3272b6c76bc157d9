// video_extruder end to end against the REAL reference (oracle/_ref/libvpp_ref_ve.so = matt-42/vpp's own video_extruder headers,
// serial build): the same synthetic frame sequence through the reference on the host and through the drop-in front-end on the GPU;
// every keypoint (position, velocity, age) and every trajectory length must be identical after the last frame.
//   usage: video_extruder_parity <nrows> <ncols> <nframes>      (BASELINE configs[4]: 2160 3840 10, defaults of video_extruder.hpp:35-41)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <vpp/vpp.hh>
#include <vpp/algorithms/video_extruder.hh>

using namespace vpp;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

extern "C" int ref_video_extruder_run(const vpp_image_desc* frames, int nframes, int detector_th, int keypoint_spacing, int detector_period,
                                      int max_trajectory_length, int nscales, int winsize, int propagation, int32_t* out, int32_t* traj_len,
                                      int capacity, int* count, int* frame_id);

extern "C" int ref_video_extruder_run_schedule(const vpp_image_desc* frames, int nframes, int detector_th, int keypoint_spacing, int detector_period,
                                               const int* max_len, int nscales, int winsize, int propagation, int32_t* out, int32_t* traj_len,
                                               int capacity, int* count, int* frame_id);

static vpp_image_desc host_desc(const image2d<unsigned char>& i) { return vpp_image_desc{(void*)&i(0, 0), i.nrows(), i.ncols(), i.pitch(), i.border(), VPP_U8, 1}; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// blurred-noise texture with piecewise translations (four quadrants, |flow| <= 6 px per frame) plus moving
// high-contrast rectangles so that FAST finds corners everywhere
static std::vector<image2d<unsigned char>> make_frames(int nr, int nc, int T) {
  std::mt19937 rng(6);
  const int W = nc + 128, H = nr + 128;
  std::vector<float> n(size_t(W) * H), base(size_t(W) * H, 0.f);
  for (auto& x : n) x = float(rng() & 0xFFFF) / 65535.f;
  for (int pass = 0; pass < 2; pass++) {  // separable 5-tap box blurs
    for (int r = 0; r < H; r++) for (int c = 2; c < W - 2; c++) base[size_t(r) * W + c] = (n[size_t(r) * W + c - 2] + n[size_t(r) * W + c - 1] + n[size_t(r) * W + c] + n[size_t(r) * W + c + 1] + n[size_t(r) * W + c + 2]) / 5;
    for (int r = 2; r < H - 2; r++) for (int c = 0; c < W; c++) n[size_t(r) * W + c] = (base[size_t(r - 2) * W + c] + base[size_t(r - 1) * W + c] + base[size_t(r) * W + c] + base[size_t(r + 1) * W + c] + base[size_t(r + 2) * W + c]) / 5;
  }
  float lo = 1e9f, hi = -1e9f;
  for (int r = 8; r < H - 8; r++) for (int c = 8; c < W - 8; c++) { lo = std::min(lo, n[size_t(r) * W + c]); hi = std::max(hi, n[size_t(r) * W + c]); }
  for (auto& x : n) x = std::min(255.f, std::max(0.f, (x - lo) / (hi - lo) * 255.f));
  const float shifts[4][2] = {{2.0f, -3.0f}, {-4.0f, 1.0f}, {5.0f, 4.0f}, {0.0f, -6.0f}};
  std::vector<image2d<unsigned char>> frames;
  for (int t = 0; t < T; t++) {
    image2d<unsigned char> f(nr, nc, _border = 3, _aligned = 32);
    for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) {
      const int q = (r >= nr / 2) * 2 + (c >= nc / 2);
      const float y = r + 64 - shifts[q][0] * t * 0.5f, x = c + 64 - shifts[q][1] * t * 0.5f;   // half the shift per frame keeps the content inside the margin for 10 frames
      const int y0 = int(y), x0 = int(x); const float a = y - y0, b = x - x0;
      const float v = (1 - a) * (1 - b) * n[size_t(y0) * W + x0] + a * (1 - b) * n[size_t(y0 + 1) * W + x0] + (1 - a) * b * n[size_t(y0) * W + x0 + 1] + a * b * n[size_t(y0 + 1) * W + x0 + 1];
      f(r, c) = (unsigned char)(v + 0.5f);
    }
    const int nrect = std::max(25, nr * nc / 3000);
    std::mt19937 rr(99);
    for (int k = 0; k < nrect; k++) {
      const int r0 = 8 + int(rr() % unsigned(nr - 40)) + t, c0 = 8 + int(rr() % unsigned(nc - 48)) - (k & 1 ? 2 : -1) * t, v = int(rr() & 255);
      for (int i = 0; i < 14; i++) for (int j = 0; j < 18; j++) { const int y = r0 + i, x = c0 + j; if (y >= 0 && y < nr && x >= 0 && x < nc) f(y, x) = (unsigned char)v; }
    }
    fill_border_mirror(f);
    frames.push_back(f);
  }
  return frames;
}

static bool same_as_reference(const char* what, video_extruder_ctx& ctx, const std::vector<int32_t>& want, const std::vector<int32_t>& wlen, int wn, int wfid, int nr, int nc,
                              int* alive_out, int* moved_out) {
  CHECK(ctx.frame_id == wfid);
  CHECK(ctx.keypoints.size() == wn);
  CHECK(wn > nr * nc / 400);
  int alive = 0, moved = 0;
  for (int i = 0; i < wn; i++) {
    const auto& k = ctx.keypoints[i];
    if (!(k.position[0] == want[5 * i] && k.position[1] == want[5 * i + 1] && k.velocity[0] == want[5 * i + 2] && k.velocity[1] == want[5 * i + 3] && k.age == want[5 * i + 4])) {
      std::fprintf(stderr, "%s: keypoint %d: got pos (%d,%d) vel (%d,%d) age %d, reference pos (%d,%d) vel (%d,%d) age %d\n", what, i, k.position[0], k.position[1], k.velocity[0], k.velocity[1], k.age,
                   want[5 * i], want[5 * i + 1], want[5 * i + 2], want[5 * i + 3], want[5 * i + 4]);
      return false;
    }
    if (ctx.trajectories[i].size() != wlen[i]) { std::fprintf(stderr, "%s: trajectory %d: length %d, reference %d\n", what, i, ctx.trajectories[i].size(), wlen[i]); return false; }
    alive += k.age > 0;
    moved += k.velocity[0] != 0 || k.velocity[1] != 0;
  }
  *alive_out = alive; *moved_out = moved;
  return true;
}

int main(int argc, char** argv) {
  const int nr = argc > 1 ? atoi(argv[1]) : 240, nc = argc > 2 ? atoi(argv[2]) : 320, T = argc > 3 ? atoi(argv[3]) : 9;
  CHECK(vpp_init(0) == 0);
  auto frames = make_frames(nr, nc, T);
  std::vector<vpp_image_desc> descs;
  for (auto& f : frames) descs.push_back(host_desc(f));
  const int cap = nr * nc / 20 + 1000;
  std::vector<int32_t> want(size_t(5) * cap), wlen(cap); int wn = 0, wfid = 0;
  std::printf("video_extruder parity %dx%d x %d frames — checker: the reference's own headers (oracle/_ref/libvpp_ref_ve.so)\n", nr, nc, T);
  double t0 = now();
  CHECK(ref_video_extruder_run(descs.data(), T, 10, 10, 5, 15, 3, 9, 2, want.data(), wlen.data(), cap, &wn, &wfid) == 0);
  std::printf("  reference: %d container entries after frame %d, %.1f ms per update (1 thread)\n", wn, wfid, (now() - t0) * 1e3 / (T - 1));
  video_extruder_ctx ctx = video_extruder_init(make_box2d(nr, nc));
  t0 = now();
  for (int t = 1; t < T; t++)
    video_extruder_update(ctx, frames[t - 1], frames[t], _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15,
                          _nscales = 3, _winsize = 9, _propagation = 2);
  std::printf("  drop-in:   %d container entries after frame %d, %.2f ms per update (incl. first-touch uploads)\n", ctx.keypoints.size(), ctx.frame_id, (now() - t0) * 1e3 / (T - 1));
  int alive = 0, moved = 0;
  if (!same_as_reference("drop-in", ctx, want, wlen, wn, wfid, nr, nc, &alive, &moved)) return 1;
  CHECK(alive > wn / 2 && moved > alive / 4);
  std::printf("  identical: %d entries (%d alive, %d moving), positions / velocities / ages / trajectory lengths\n", wn, alive, moved);

  // the one-frame-per-call loop (video_extruder_push_frame, the tracker keeps `prev` and its pyramid): gray frames WITHOUT a border, then colour
  // frames whose rgb_to_graylevel is the gray sequence — both must land on the reference's state too
  {
    video_extruder_ctx c2 = video_extruder_init(make_box2d(nr, nc));
    t0 = now();
    for (int t = 0; t < T; t++) {
      image2d<unsigned char> bare(nr, nc, _border = 0);
      for (int r = 0; r < nr; r++) std::memcpy(&bare(r, 0), &frames[t](r, 0), nc);
      const bool ran = video_extruder_push_frame(c2, bare, _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15, _nscales = 3, _winsize = 9, _propagation = 2);
      CHECK(ran == (t > 0));
    }
    std::printf("  push_frame (gray):  %d entries, %.2f ms per frame (incl. the frame's upload)\n", c2.keypoints.size(), (now() - t0) * 1e3 / T);
    if (!same_as_reference("push_frame(gray)", c2, want, wlen, wn, wfid, nr, nc, &alive, &moved)) return 1;
  }
  for (int resident = 0; resident < 2; resident++) {   // colour frames in host memory (staged by the tracker's copy stream), then already in HBM
    video_extruder_ctx c3 = video_extruder_init(make_box2d(nr, nc));
    std::mt19937 rng(17);
    for (int t = 0; t < T; t++) {
      image2d<vuchar3> rgb(nr, nc, _border = 0);
      for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) {   // (g + d, g, g - d + k), k in {0, 1, 2}: the integer mean is g
        const int g = frames[t](r, c), d = std::min(std::min(g, 255 - g), int(rng() & 31)), k = g - d + 2 <= 255 ? int(rng() % 3) : 0;
        rgb(r, c) = vuchar3(g + d, g, g - d + k);
      }
      if (resident) (void)rgb.device_desc(false);
      CHECK(rgb.device_current() == (resident != 0));
      const bool ran = video_extruder_push_frame(c3, rgb, _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = 15, _nscales = 3, _winsize = 9, _propagation = 2);
      CHECK(ran == (t > 0));
    }   // (every rgb image dies while its update may still be queued: the tracker has its own copy, or the stream orders the mirror's reuse)
    std::printf("  push_frame (rgb, %s):   %d entries\n", resident ? "frames in HBM" : "host frames", c3.keypoints.size());
    if (!same_as_reference(resident ? "push_frame(rgb, device)" : "push_frame(rgb, host)", c3, want, wlen, wn, wfid, nr, nc, &alive, &moved)) return 1;
  }
  std::printf("  push_frame: gray and colour sequences identical to the reference as well\n");
  // _max_trajectory_length changed mid-sequence (an argument of every call, video_extruder.hpp:40): raised beyond what the tracker's rings hold (the
  // device tracker is rebuilt around the host copy and must continue from it), then lowered (one pop per update: lengths stay above the new bound)
  if (T >= 7) {
    std::vector<int> sched(T - 1);
    for (int t = 0; t < T - 1; t++) sched[t] = t < 2 ? 3 : t < T - 3 ? 40 : 5;
    CHECK(ref_video_extruder_run_schedule(descs.data(), T, 10, 10, 5, sched.data(), 3, 9, 2, want.data(), wlen.data(), cap, &wn, &wfid) == 0);
    video_extruder_ctx c4 = video_extruder_init(make_box2d(nr, nc));
    for (int t = 1; t < T; t++)
      video_extruder_update(c4, frames[t - 1], frames[t], _detector_th = 10, _keypoint_spacing = 10, _detector_period = 5, _max_trajectory_length = sched[t - 1],
                            _nscales = 3, _winsize = 9, _propagation = 2);
    if (!same_as_reference("changing max_trajectory_length", c4, want, wlen, wn, wfid, nr, nc, &alive, &moved)) return 1;
    int longest = 0;
    for (int i = 0; i < wn; i++) longest = std::max(longest, wlen[i]);
    CHECK(longest > 5);   // the schedule really left trajectories above the final bound (one pop per update: a trajectory of 6 or 7 points stays that long)
    std::printf("  _max_trajectory_length 3 -> 40 -> 5 mid-sequence: identical (%d entries, longest trajectory %d)\n", wn, longest);
  }
  std::printf("video_extruder_parity ok\n");
  return 0;
}
