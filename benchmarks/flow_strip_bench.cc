// flow_strip_bench.cc — the tile-sharded frame step of BASELINE configs[4] (semi-dense flow on 3840x2160 frame pairs, row strips over
// the GPUs of a node, RCCL exchanges over xGMI) as one process per GPU, written against the C ABI only.
//   usage: flow_strip_bench <rank> <world> <uid_file> [steps] [nrows ncols]
// What a rank owns: rows [g NR / G, (g + 1) NR / G) of every frame (what a sharded capture / decode front-end leaves on each GPU).
// One step =
//   (1) image-row exchange: vpp_allgather_rows on both frames (a flow match may land anywhere, so every rank needs the whole pair);
//   (2) vpp_semi_dense_optical_flow_sharded: claim + descent for the keypoints of the rank's flow-map rows, one grouped RCCL all-gather of
//       the maps per scale, propagation on every rank;
// and, separately timed, the detector of the re-detection frames on strips:
//   (3) vpp_halo_exchange of the 4 rows either side of the rank's strip of the new frame + vpp_fast9_detect on the strip.
// Parity, checked on every rank and summed: the gathered frames == the full frames, the sharded flow == vpp_semi_dense_optical_flow on the
// same rank, the strip detections (raw, local maxima, blockwise) == the rows [r0, r1) of the full-frame detection.
// Rank 0 prints one JSON line; times are the maximum over ranks.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <vpp_amd.h>

#define CK(x) do { const int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #x, rc_, vpp_last_error()); std::exit(2); } } while (0)
static int g_rank = 0;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct dev_image { vpp_image_desc d; void* base = nullptr; size_t bytes = 0; };
static dev_image alloc_image(int nr, int nc, int dtype, int ch, int es, int border) {
  dev_image im; int32_t pitch; size_t bytes, first;
  CK(vpp_image_layout(nr, nc, es * ch, border, 32, &pitch, &bytes, &first));
  CK(vpp_malloc(bytes + 64, &im.base));
  CK(vpp_memset(im.base, 0, bytes + 64, nullptr));
  im.d = vpp_image_desc{(char*)im.base + first, nr, nc, pitch, border, dtype, ch};
  im.bytes = bytes;
  return im;
}
static void upload_rows(const dev_image& im, const std::vector<unsigned char>& f, int NC, int src_r0, int dst_r0, int nrows) {
  for (int r = 0; r < nrows; r++) CK(vpp_memcpy_h2d((char*)im.d.first_pixel + size_t(dst_r0 + r) * im.d.pitch, &f[size_t(src_r0 + r) * NC], NC, nullptr));
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <rank> <world> <uid_file> [steps] [nrows ncols]\n", argv[0]); return 1; }
  const int rank = g_rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string uid_file = argv[3];
  const int steps = argc > 4 ? std::atoi(argv[4]) : 50;
  const int NR = argc > 6 ? std::atoi(argv[5]) : 2160, NC = argc > 6 ? std::atoi(argv[6]) : 3840;
  const int WS = 9, NSCALES = 3, PROP = 2, PATCH = 5, SPACING = 10, TH = 10;   // video_extruder.hpp:35-41
  if (NR % world) { std::fprintf(stderr, "%d rows do not split over %d ranks\n", NR, world); return 1; }
  int ndev = 1;
  CK(vpp_device_count(&ndev));
  CK(vpp_init(rank % ndev));

  // ---- the communicator (rank 0 publishes the RCCL id through a file)
  char id[128];
  if (rank == 0) {
    CK(vpp_comm_unique_id(id));
    { std::ofstream f(uid_file + ".tmp", std::ios::binary); f.write(id, 128); }
    std::rename((uid_file + ".tmp").c_str(), uid_file.c_str());
  } else {
    const double t0 = now();
    for (;;) {
      std::ifstream f(uid_file, std::ios::binary);
      if (f && f.read(id, 128)) break;
      if (now() - t0 > 60) { std::fprintf(stderr, "[rank %d] no unique id in %s after 60 s\n", rank, uid_file.c_str()); return 3; }
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  vpp_comm* comm = nullptr;
  CK(vpp_comm_init(&comm, world, id, rank));

  // ---- the frame pair (the same on every rank's host; only the rank's rows go to its GPU): a smooth random texture, frame 2 = the four
  //      quadrants translated by (2,-3) (-4,1) (5,4) (0,-6) px
  std::mt19937 rng(6);
  const int W = NC + 32, H = NR + 32;
  std::vector<float> a(size_t(W) * H), b(a.size());
  for (auto& x : a) x = float(rng() & 0xFFFF);
  for (int pass = 0; pass < 2; pass++) {
    for (int r = 0; r < H; r++) for (int c = 1; c < W - 1; c++) b[size_t(r) * W + c] = (a[size_t(r) * W + c - 1] + a[size_t(r) * W + c] + a[size_t(r) * W + c + 1]) / 3;
    for (int r = 1; r < H - 1; r++) for (int c = 0; c < W; c++) a[size_t(r) * W + c] = (b[size_t(r - 1) * W + c] + b[size_t(r) * W + c] + b[size_t(r + 1) * W + c]) / 3;
  }
  float lo = 1e30f, hi = -1e30f;
  for (int r = 8; r < H - 8; r++) for (int c = 8; c < W - 8; c++) { lo = std::min(lo, a[size_t(r) * W + c]); hi = std::max(hi, a[size_t(r) * W + c]); }
  auto px = [&](int y, int x) {
    y = std::min(H - 1, std::max(0, y)); x = std::min(W - 1, std::max(0, x));
    return (unsigned char)std::min(255.f, std::max(0.f, (a[size_t(y) * W + x] - lo) / (hi - lo) * 255.f + 0.5f));
  };
  std::vector<unsigned char> f1(size_t(NR) * NC), f2(f1.size());
  const int sh[4][2] = {{2, -3}, {-4, 1}, {5, 4}, {0, -6}};
  for (int r = 0; r < NR; r++)
    for (int c = 0; c < NC; c++) {
      const int q = (r >= NR / 2 ? 2 : 0) + (c >= NC / 2 ? 1 : 0);
      f1[size_t(r) * NC + c] = px(r + 16, c + 16);
      f2[size_t(r) * NC + c] = px(r + 16 - sh[q][0], c + 16 - sh[q][1]);
    }
  std::vector<int32_t> kps;
  for (int r = SPACING; r < NR - SPACING; r += SPACING) for (int c = SPACING; c < NC - SPACING; c += SPACING) { kps.push_back(r); kps.push_back(c); }
  const int NK = int(kps.size() / 2);

  const int per = NR / world, r0 = rank * per;
  dev_image g1 = alloc_image(NR, NC, VPP_U8, 1, 1, 0), g2 = alloc_image(NR, NC, VPP_U8, 1, 1, 0);       // assembled by the row exchange
  dev_image full1 = alloc_image(NR, NC, VPP_U8, 1, 1, 0), full2 = alloc_image(NR, NC, VPP_U8, 1, 1, 0); // uploaded whole: the single-rank check
  upload_rows(full1, f1, NC, 0, 0, NR); upload_rows(full2, f2, NC, 0, 0, NR);
  // a rank's rows of the "new" frames, kept aside: every step starts from them (the exchange writes the other ranks' rows)
  void *d_k = nullptr, *o_pos = nullptr, *o_dist = nullptr, *o_val = nullptr, *w_pos = nullptr, *w_dist = nullptr, *w_val = nullptr;
  CK(vpp_malloc(size_t(NK) * 8, &d_k)); CK(vpp_malloc(size_t(NK) * 8, &o_pos)); CK(vpp_malloc(size_t(NK) * 4, &o_dist)); CK(vpp_malloc(NK, &o_val));
  CK(vpp_malloc(size_t(NK) * 8, &w_pos)); CK(vpp_malloc(size_t(NK) * 4, &w_dist)); CK(vpp_malloc(NK, &w_val));
  CK(vpp_memcpy_h2d(d_k, kps.data(), size_t(NK) * 8, nullptr));
  upload_rows(g1, f1, NC, r0, r0, per); upload_rows(g2, f2, NC, r0, r0, per);
  CK(vpp_sync(nullptr));

  auto step = [&](void* st) {
    CK(vpp_allgather_rows(comm, &g1.d, st));
    CK(vpp_allgather_rows(comm, &g2.d, st));
    CK(vpp_semi_dense_optical_flow_sharded(comm, &g1.d, &g2.d, (const int32_t*)d_k, NK, WS, NSCALES, 0, PROP, PATCH, (int32_t*)o_pos, (int32_t*)o_dist, (uint8_t*)o_val, st));
  };
  for (int i = 0; i < 3; i++) step(nullptr);
  CK(vpp_sync(nullptr));
  if (std::getenv("VPP_STRIP_SHARE_ONLY")) {   // tools/flow_replicated_share.py: nothing but the sharded step in the process (no single-rank calls, no detector leg)
    for (int i = 0; i < steps; i++) step(nullptr);
    CK(vpp_sync(nullptr));
    if (rank == 0) { std::printf("{\"share_only\": true, \"ranks\": %d, \"steps\": %d}\n", world, steps + 3); std::remove(uid_file.c_str()); }
    CK(vpp_comm_destroy(comm));
    return 0;
  }

  // ---- parity of the row exchange and of the sharded flow (every rank, against its own single-rank call on the full frames)
  long bad_rows = 0, bad_flow = 0;
  {
    std::vector<unsigned char> back(size_t(NR) * g1.d.pitch);
    for (int which = 0; which < 2; which++) {
      const dev_image& g = which ? g2 : g1; const std::vector<unsigned char>& f = which ? f2 : f1;
      CK(vpp_memcpy_d2h(back.data(), g.d.first_pixel, back.size(), nullptr)); CK(vpp_sync(nullptr));
      for (int r = 0; r < NR; r++) bad_rows += std::memcmp(&back[size_t(r) * g.d.pitch], &f[size_t(r) * NC], NC) != 0;
    }
    CK(vpp_semi_dense_optical_flow(&full1.d, &full2.d, (const int32_t*)d_k, NK, WS, NSCALES, 0, PROP, PATCH, (int32_t*)w_pos, (int32_t*)w_dist, (uint8_t*)w_val, nullptr));
    std::vector<int32_t> gp(size_t(NK) * 2), gd(NK), wp(size_t(NK) * 2), wd(NK); std::vector<uint8_t> gv(NK), wv(NK);
    CK(vpp_memcpy_d2h(gp.data(), o_pos, gp.size() * 4, nullptr)); CK(vpp_memcpy_d2h(gd.data(), o_dist, gd.size() * 4, nullptr)); CK(vpp_memcpy_d2h(gv.data(), o_val, NK, nullptr));
    CK(vpp_memcpy_d2h(wp.data(), w_pos, wp.size() * 4, nullptr)); CK(vpp_memcpy_d2h(wd.data(), w_dist, wd.size() * 4, nullptr)); CK(vpp_memcpy_d2h(wv.data(), w_val, NK, nullptr));
    CK(vpp_sync(nullptr));
    long valid = 0, moved = 0;
    for (int i = 0; i < NK; i++) {
      bad_flow += gv[i] != wv[i] || gp[2 * i] != wp[2 * i] || gp[2 * i + 1] != wp[2 * i + 1] || gd[i] != wd[i];
      valid += wv[i]; moved += wp[2 * i] != kps[2 * i] || wp[2 * i + 1] != kps[2 * i + 1];
    }
    if (valid < NK * 9 / 10 || moved < NK / 2) { std::fprintf(stderr, "[rank %d] degenerate scene: %ld valid, %ld moved of %d\n", rank, valid, moved, NK); bad_flow += 1; }
  }

  // ---- timing: `steps` steps (exchange + sharded flow) in one launch graph per rank; eager if the collectives cannot be recorded
  const char* mode = "eager launches";
  vpp_graph* graph = nullptr;
  void* side = nullptr;
  CK(vpp_stream_create(&side));
  step(side); CK(vpp_sync(side));
  if (!std::getenv("VPP_SHARD_EAGER") && vpp_graph_begin(side) == 0) {
    for (int i = 0; i < steps; i++) step(side);
    if (vpp_graph_end(side, 0, &graph) == 0) mode = "vpp_graph (row exchange + sharded flow recorded per rank)";
    else { graph = nullptr; std::fprintf(stderr, "[rank %d] the step could not be recorded (%s): eager launches\n", rank, vpp_last_error()); }
  }
  step(side); CK(vpp_sync(side));   // the collectives double as the barrier in front of the timed region
  double t0 = now();
  if (graph) CK(vpp_graph_launch(graph, side)); else for (int i = 0; i < steps; i++) step(side);
  CK(vpp_sync(side));
  const double ms_step = (now() - t0) * 1e3 / steps;
  // the same pair through the single-rank call on this rank alone (what a replica does), eager, same stream
  for (int i = 0; i < 3; i++) CK(vpp_semi_dense_optical_flow(&full1.d, &full2.d, (const int32_t*)d_k, NK, WS, NSCALES, 0, PROP, PATCH, (int32_t*)w_pos, (int32_t*)w_dist, (uint8_t*)w_val, side));
  CK(vpp_sync(side));
  t0 = now();
  for (int i = 0; i < steps; i++) CK(vpp_semi_dense_optical_flow(&full1.d, &full2.d, (const int32_t*)d_k, NK, WS, NSCALES, 0, PROP, PATCH, (int32_t*)w_pos, (int32_t*)w_dist, (uint8_t*)w_val, side));
  CK(vpp_sync(side));
  const double ms_single = (now() - t0) * 1e3 / steps;

  // ---- the detector on strips: 4 halo rows either side (FAST-9 reads 3 rows around a pixel; a local maximum also compares with the scores
  //      of the row above / below, so the detection runs on the view [r0 - 1, r1 + 1) of the strip — border 3 inside the 4 exchanged rows —
  //      and keypoints of the two extra rows are dropped)
  long bad_fast = 0; double ms_fast = 0; long nkp_strip = 0;
  {
    const int B = 4;
    dev_image strip = alloc_image(per, NC, VPP_U8, 1, 1, B), whole = alloc_image(NR, NC, VPP_U8, 1, 1, 3);
    upload_rows(strip, f2, NC, r0, 0, per); upload_rows(whole, f2, NC, 0, 0, NR);
    CK(vpp_fill_border(&whole.d, 0 /* mirror */, nullptr, nullptr));
    const int cap = 4000000;
    void *rc_s = nullptr, *sc_s = nullptr, *rc_w = nullptr, *sc_w = nullptr;
    CK(vpp_malloc(size_t(cap) * 8, &rc_s)); CK(vpp_malloc(size_t(cap) * 4, &sc_s)); CK(vpp_malloc(size_t(cap) * 8, &rc_w)); CK(vpp_malloc(size_t(cap) * 4, &sc_w));
    auto exchange = [&](void* st) {
      CK(vpp_fill_border(&strip.d, 0, nullptr, st));          // every border mirrored first (right at the frame's edges) ...
      CK(vpp_halo_exchange(comm, &strip.d, B, st));           // ... then the inner edges receive the neighbour's rows
    };
    const int up = rank > 0 ? 1 : 0, down = rank + 1 < world ? 1 : 0;
    vpp_image_desc view = strip.d;                             // rows [r0 - up, r1 + down) with border 3: all inside the exchanged halo
    view.first_pixel = (char*)strip.d.first_pixel - size_t(up) * strip.d.pitch; view.nrows = per + up + down; view.border = 3;
    for (int m = 0; m < 3; m++) {
      exchange(nullptr);
      int ns = 0, nw = 0;
      const vpp_image_desc& det = m == 1 ? view : strip.d;    // raw / blockwise need no extra row (block size 10 divides the strip bounds)
      const int dr = m == 1 ? r0 - up : r0;
      CK(vpp_fast9_detect(&det, TH, nullptr, m, 10, 0, (int32_t*)rc_s, (int32_t*)sc_s, cap, &ns, nullptr));
      CK(vpp_fast9_detect(&whole.d, TH, nullptr, m, 10, 0, (int32_t*)rc_w, (int32_t*)sc_w, cap, &nw, nullptr));
      std::vector<int32_t> a_rc(size_t(ns) * 2), a_sc(ns), w_rc(size_t(nw) * 2), w_sc(nw);
      CK(vpp_memcpy_d2h(a_rc.data(), rc_s, a_rc.size() * 4, nullptr)); CK(vpp_memcpy_d2h(a_sc.data(), sc_s, a_sc.size() * 4, nullptr));
      CK(vpp_memcpy_d2h(w_rc.data(), rc_w, w_rc.size() * 4, nullptr)); CK(vpp_memcpy_d2h(w_sc.data(), sc_w, w_sc.size() * 4, nullptr));
      CK(vpp_sync(nullptr));
      std::vector<int32_t> got, want;   // (row, col, score) triples of the rank's own rows, in output order
      for (int i = 0; i < ns; i++) { const int r = a_rc[2 * i] + dr; if (r >= r0 && r < r0 + per) { got.push_back(r); got.push_back(a_rc[2 * i + 1]); got.push_back(a_sc[i]); } }
      for (int i = 0; i < nw; i++) { const int r = w_rc[2 * i]; if (r >= r0 && r < r0 + per) { want.push_back(r); want.push_back(w_rc[2 * i + 1]); want.push_back(w_sc[i]); } }
      if (got != want) { bad_fast += 1 + std::labs(long(got.size()) - long(want.size())); std::fprintf(stderr, "[rank %d] FAST mode %d: strip %zu vs frame %zu entries differ\n", rank, m, got.size() / 3, want.size() / 3); }
      if (want.size() < 30) { std::fprintf(stderr, "[rank %d] FAST mode %d: only %zu keypoints in the strip\n", rank, m, want.size() / 3); bad_fast += 1; }
      if (m == 2) nkp_strip = long(want.size() / 3);
    }
    // time: halo exchange + blockwise detection on the strip (the re-detection of video_extruder_update, :93-112)
    int ns = 0;
    for (int i = 0; i < 3; i++) { exchange(side); CK(vpp_fast9_detect(&strip.d, TH, nullptr, 2, 10, 0, (int32_t*)rc_s, (int32_t*)sc_s, cap, &ns, side)); }
    t0 = now();
    for (int i = 0; i < steps; i++) { exchange(side); CK(vpp_fast9_detect(&strip.d, TH, nullptr, 2, 10, 0, (int32_t*)rc_s, (int32_t*)sc_s, cap, &ns, side)); }
    CK(vpp_sync(side));
    ms_fast = (now() - t0) * 1e3 / steps;
  }

  // ---- max / sum over ranks through the keypoint all-gather (one 20-byte record per rank)
  std::vector<vpp_keypoint_f32> tmine(1, vpp_keypoint_f32{float(ms_step), float(ms_fast), float(bad_rows + bad_flow), float(bad_fast), 1}), tall(world);
  void *d_t = nullptr, *d_tall = nullptr;
  CK(vpp_malloc(sizeof(vpp_keypoint_f32), &d_t)); CK(vpp_malloc(sizeof(vpp_keypoint_f32) * world, &d_tall));
  CK(vpp_memcpy_h2d(d_t, tmine.data(), sizeof(vpp_keypoint_f32), nullptr));
  CK(vpp_allgather_tracks(comm, (const vpp_keypoint_f32*)d_t, 1, (vpp_keypoint_f32*)d_tall, nullptr));
  CK(vpp_memcpy_d2h(tall.data(), d_tall, sizeof(vpp_keypoint_f32) * world, nullptr)); CK(vpp_sync(nullptr));
  double worst = 0, worst_fast = 0, bad_a = 0, bad_b = 0;
  for (auto& t : tall) { worst = std::max(worst, double(t.pos_r)); worst_fast = std::max(worst_fast, double(t.pos_c)); bad_a += t.vel_r; bad_b += t.vel_c; }
  if (rank == 0) {
    std::printf("{\"workload\": \"semi-dense flow %dx%d, %d keypoints, winsize %d, %d scales, %d sweeps: rows of both frames all-gathered, claim + descent sharded by flow-map "
                "row strips over %d ranks, one grouped RCCL all-gather of the maps per scale (C++ harness, one process per GPU)\", \"frame_pairs_per_s\": %.1f, "
                "\"ms_per_frame_pair\": %.4f, \"ms_per_frame_pair_single_rank_call\": %.4f, \"launch\": \"%s\", \"steps\": %d, "
                "\"fast9_strip\": {\"workload\": \"vpp_halo_exchange (4 rows, grouped RCCL send/recv) + blockwise FAST-9 on the rank's strip of %d rows\", \"ms\": %.4f, \"keypoints_in_rank0_strip\": %ld}, "
                "\"mismatched_rows_or_flow_records\": %.0f, \"mismatched_fast9_strips\": %.0f}\n",
                NR, NC, NK, WS, NSCALES, PROP, world, 1e3 / worst, worst, ms_single, mode, steps, per, worst_fast, nkp_strip, bad_a, bad_b);
    std::remove(uid_file.c_str());
  }
  CK(vpp_comm_destroy(comm));
  return (bad_a > 0 || bad_b > 0) ? 4 : 0;
}
