// pyrlk_shard_bench.cc — the keypoint-sharded pyrLK step of BASELINE configs[3] as one process per GPU, written against the C ABI only
// (no Python on the step): vpp_pyrlk_match on this rank's slice + vpp_allgather_tracks (RCCL over xGMI) on one stream, recorded once
// into a launch graph (vpp_graph_*) and replayed; falls back to eager launches if the collective cannot be recorded.
//   usage: pyrlk_shard_bench <rank> <world> <uid_file> [steps] [keypoints] [frame_pairs]
// frame_pairs = F > 1 (round 6): the step covers F DISTINCT frame pairs — this rank's slice of each pair's keypoints in ONE vpp_pyrlk_match_batch launch and ONE
// all-gather of the F x slice records.  A rank's 1 250 keypoints of one pair cost the latency of a single keypoint's chain (71 us); F pairs per launch leave that
// floor, which is what lets the keypoint-sharded job scale (tracks/s = F x keypoints / step time).
// Rank 0 creates the RCCL unique id and publishes it through <uid_file> (written to a temporary name, then renamed); the other
// ranks wait for the file.  Every rank builds the same synthetic frame pair (1920x1080, a smooth random texture translated by
// (1.5, -2.25) px) and both pyramids; rank 0 prints one JSON line.  Times are the maximum over ranks (gathered with the same collective).
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
#include <unistd.h>

#include <vpp_amd.h>
#include "shard_plan.hh"

#define CK(x) do { const int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #x, rc_, vpp_last_error()); std::exit(2); } } while (0)
static int g_rank = 0;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct dev_image { vpp_image_desc d; void* base = nullptr; };
static dev_image alloc_image(int nr, int nc, int dtype, int ch, int es, int border) {
  dev_image im; int32_t pitch; size_t bytes, first;
  CK(vpp_image_layout(nr, nc, es * ch, border, 32, &pitch, &bytes, &first));
  CK(vpp_malloc(bytes + 64, &im.base));
  CK(vpp_memset(im.base, 0, bytes + 64, nullptr));
  im.d = vpp_image_desc{(char*)im.base + first, nr, nc, pitch, border, dtype, ch};
  return im;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <rank> <world> <uid_file> [steps] [keypoints] [frame_pairs]\n", argv[0]); return 1; }
  const int rank = g_rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string uid_file = argv[3];
  const int steps = argc > 4 ? std::atoi(argv[4]) : 200, NK = argc > 5 ? std::atoi(argv[5]) : 10000, F = argc > 6 ? std::max(1, std::atoi(argv[6])) : 1;
  const int NR = 1080, NC = 1920, L = 3, B = 3, WS = 7;
  int ndev = 1;
  CK(vpp_device_count(&ndev));
  CK(vpp_init(rank % ndev));

  // ---- the communicator
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

  // ---- frames (the same on every rank) and pyramids
  std::mt19937 rng(5);
  const int W = NC + 32, H = NR + 32;
  std::vector<float> a(size_t(W) * H), b(a.size());
  for (auto& x : a) x = float(rng() & 0xFFFF);
  for (int pass = 0; pass < 3; pass++) {
    for (int r = 0; r < H; r++) for (int c = 2; c < W - 2; c++) b[size_t(r) * W + c] = (a[size_t(r) * W + c - 2] + a[size_t(r) * W + c - 1] + a[size_t(r) * W + c] + a[size_t(r) * W + c + 1] + a[size_t(r) * W + c + 2]) / 5;
    for (int r = 2; r < H - 2; r++) for (int c = 0; c < W; c++) a[size_t(r) * W + c] = (b[size_t(r - 2) * W + c] + b[size_t(r - 1) * W + c] + b[size_t(r) * W + c] + b[size_t(r + 1) * W + c] + b[size_t(r + 2) * W + c]) / 5;
  }
  float lo = 1e30f, hi = -1e30f;
  for (int r = 8; r < H - 8; r++) for (int c = 8; c < W - 8; c++) { lo = std::min(lo, a[size_t(r) * W + c]); hi = std::max(hi, a[size_t(r) * W + c]); }
  auto sample = [&](float y, float x) {
    const int y0 = int(y), x0 = int(x); const float fy = y - y0, fx = x - x0;
    const float v = (1 - fy) * (1 - fx) * a[size_t(y0) * W + x0] + fy * (1 - fx) * a[size_t(y0 + 1) * W + x0] + (1 - fy) * fx * a[size_t(y0) * W + x0 + 1] + fy * fx * a[size_t(y0 + 1) * W + x0 + 1];
    return (unsigned char)std::min(255.f, std::max(0.f, (v - lo) / (hi - lo) * 255.f + 0.5f));
  };
  // pair f: the texture seen 5 f px further along its diagonal, translated by (1.5 - 0.1 f, -2.25 + 0.2 f) px between its two frames (pair 0 = configs[3]'s scene)
  std::vector<vpp_image_desc> P1(size_t(F) * L), P2(size_t(F) * L), G1(size_t(F) * L);
  {
    std::vector<unsigned char> f1(size_t(NR) * NC), f2(f1.size());
    dev_image s1 = alloc_image(NR, NC, VPP_U8, 1, 1, 0), s2 = alloc_image(NR, NC, VPP_U8, 1, 1, 0);
    for (int f = 0; f < F; f++) {
      const float o = 10.f + 0.3f * f, tr = 1.5f - 0.1f * f, tc = -2.25f + 0.2f * f;   // (the 16-px margin of the texture holds every offset for F <= 16)
      for (int r = 0; r < NR; r++) for (int c = 0; c < NC; c++) { f1[size_t(r) * NC + c] = sample(r + (f ? o : 16.f), c + (f ? o : 16.f)); f2[size_t(r) * NC + c] = sample(r + (f ? o : 16.f) - tr, c + (f ? o : 16.f) - tc); }
      for (int r = 0; r < NR; r++) {
        CK(vpp_memcpy_h2d((char*)s1.d.first_pixel + size_t(r) * s1.d.pitch, &f1[size_t(r) * NC], NC, nullptr));
        CK(vpp_memcpy_h2d((char*)s2.d.first_pixel + size_t(r) * s2.d.pitch, &f2[size_t(r) * NC], NC, nullptr));
      }
      for (int l = 0, nr = NR, nc = NC; l < L; l++, nr = 1 + nr / 2, nc = 1 + nc / 2) {
        P1[size_t(f) * L + l] = alloc_image(nr, nc, VPP_U8, 1, 1, B).d; P2[size_t(f) * L + l] = alloc_image(nr, nc, VPP_U8, 1, 1, B).d; G1[size_t(f) * L + l] = alloc_image(nr, nc, VPP_F32, 2, 4, B).d;
      }
      CK(vpp_pyramid_build(&P1[size_t(f) * L], L, &s1.d, nullptr));
      CK(vpp_pyramid_build(&P2[size_t(f) * L], L, &s2.d, nullptr));
      CK(vpp_scharr_pyramid_build(&G1[size_t(f) * L], L, &P1[size_t(f) * L], nullptr));
      CK(vpp_sync(nullptr));   // (the staging frames are reused by the next pair)
    }
  }

  // ---- keypoints: jittered grid >= 32 px from every edge, this rank's padded slice
  std::vector<vpp_keypoint_f32> all(NK);
  {
    const int gr = int(std::ceil(std::sqrt(double(NK) * NR / NC))), gc = (NK + gr - 1) / gr;
    std::mt19937 kr(7);
    for (int i = 0; i < NK; i++) {
      const float r = 34.f + (NR - 70.f) * float(i / gc) / float(std::max(1, gr - 1)) + float(kr() % 100) / 100.f;
      const float c = 34.f + (NC - 70.f) * float(i % gc) / float(std::max(1, gc - 1)) + float(kr() % 100) / 100.f;
      all[i] = vpp_keypoint_f32{r, c, 0.f, 0.f, 1};
    }
  }
  const vpp_shard::plan plan(NK, world);
  const std::vector<vpp_keypoint_f32> mine = plan.shard_of(all, rank);
  // this rank's records of the F pairs, pair after pair: [f][i] (every pair tracks the same grid of keypoints on its own frames)
  void *d_src = nullptr, *d_shard = nullptr, *d_all = nullptr;
  const size_t shard_bytes = size_t(plan.per_rank) * sizeof(vpp_keypoint_f32) * F;
  CK(vpp_malloc(shard_bytes, &d_src)); CK(vpp_malloc(shard_bytes, &d_shard)); CK(vpp_malloc(shard_bytes * world, &d_all));
  for (int f = 0; f < F; f++) CK(vpp_memcpy_h2d((char*)d_src + size_t(f) * plan.per_rank * sizeof(vpp_keypoint_f32), mine.data(), size_t(plan.per_rank) * sizeof(vpp_keypoint_f32), nullptr));
  CK(vpp_sync(nullptr));
  std::vector<vpp_keypoint_f32*> kp_of(F); std::vector<int> n_of(F, plan.per_rank);
  for (int f = 0; f < F; f++) kp_of[f] = (vpp_keypoint_f32*)d_shard + size_t(f) * plan.per_rank;

  auto step = [&](void* st) {
    CK(vpp_memcpy_d2d(d_shard, d_src, shard_bytes, st));   // restore the tracks: pyrlk_match moves them in place
    if (F == 1) CK(vpp_pyrlk_match(P1.data(), G1.data(), P2.data(), L, (vpp_keypoint_f32*)d_shard, plan.per_rank, WS, 1e-4f, 500.f, 30, 0.01f, 0, nullptr, st));
    else CK(vpp_pyrlk_match_batch(P1.data(), G1.data(), P2.data(), F, L, kp_of.data(), n_of.data(), WS, 1e-4f, 500.f, 30, 0.01f, 0, nullptr, st));
    CK(vpp_allgather_tracks(comm, (const vpp_keypoint_f32*)d_shard, plan.per_rank * F, (vpp_keypoint_f32*)d_all, st));
  };
  for (int i = 0; i < 5; i++) step(nullptr);
  CK(vpp_sync(nullptr));

  // ---- one launch graph of `steps` steps (match + all-gather) on a stream of this process; eager if the collective cannot be recorded
  const char* mode = "eager launches";
  vpp_graph* graph = nullptr;
  void* side = nullptr;
  CK(vpp_stream_create(&side));
  if (!std::getenv("VPP_SHARD_EAGER") && vpp_graph_begin(side) == 0) {
    for (int i = 0; i < steps; i++) step(side);
    if (vpp_graph_end(side, 0, &graph) == 0) mode = "vpp_graph (match + RCCL all-gather recorded per rank)";
    else { graph = nullptr; std::fprintf(stderr, "[rank %d] the step could not be recorded (%s): eager launches\n", rank, vpp_last_error()); }
  }
  step(side); CK(vpp_sync(side));   // the gather doubles as the barrier in front of the timed region
  const double t0 = now();
  if (graph) CK(vpp_graph_launch(graph, side));
  else for (int i = 0; i < steps; i++) step(side);
  CK(vpp_sync(side));
  const double ms_step = (now() - t0) * 1e3 / steps;

  // ---- parity of the exchange: the gathered, unpadded records of every pair == a single-rank run over all of that pair's keypoints (single calls, same inputs)
  std::vector<vpp_keypoint_f32> gathered(size_t(plan.per_rank) * world * F);
  CK(vpp_memcpy_d2h(gathered.data(), d_all, shard_bytes * world, nullptr)); CK(vpp_sync(nullptr));
  int mismatched = -1;
  if (rank == 0) {
    void* d_full = nullptr;
    CK(vpp_malloc(size_t(NK) * sizeof(vpp_keypoint_f32), &d_full));
    mismatched = 0;
    for (int f = 0; f < F; f++) {
      std::vector<vpp_keypoint_f32> of_pair(size_t(plan.per_rank) * world);   // rank g's block holds its F slices one after the other
      for (int g = 0; g < world; g++)
        std::memcpy(&of_pair[size_t(g) * plan.per_rank], &gathered[(size_t(g) * F + f) * plan.per_rank], size_t(plan.per_rank) * sizeof(vpp_keypoint_f32));
      const std::vector<vpp_keypoint_f32> got = plan.unpad(of_pair);
      CK(vpp_memcpy_h2d(d_full, all.data(), size_t(NK) * sizeof(vpp_keypoint_f32), nullptr));
      CK(vpp_pyrlk_match(&P1[size_t(f) * L], &G1[size_t(f) * L], &P2[size_t(f) * L], L, (vpp_keypoint_f32*)d_full, NK, WS, 1e-4f, 500.f, 30, 0.01f, 0, nullptr, nullptr));
      std::vector<vpp_keypoint_f32> want(NK);
      CK(vpp_memcpy_d2h(want.data(), d_full, size_t(NK) * sizeof(vpp_keypoint_f32), nullptr)); CK(vpp_sync(nullptr));
      for (int i = 0; i < NK; i++) mismatched += std::memcmp(&want[i], &got[i], sizeof(vpp_keypoint_f32)) != 0;
    }
  }

  // ---- max over ranks of the step time, through the same collective (one record per rank carrying the time)
  std::vector<vpp_keypoint_f32> tmine(1, vpp_keypoint_f32{float(ms_step), 0, 0, 0, 1}), tall(world);
  void *d_t = nullptr, *d_tall = nullptr;
  CK(vpp_malloc(sizeof(vpp_keypoint_f32), &d_t)); CK(vpp_malloc(sizeof(vpp_keypoint_f32) * world, &d_tall));
  CK(vpp_memcpy_h2d(d_t, tmine.data(), sizeof(vpp_keypoint_f32), nullptr));
  CK(vpp_allgather_tracks(comm, (const vpp_keypoint_f32*)d_t, 1, (vpp_keypoint_f32*)d_tall, nullptr));
  CK(vpp_memcpy_d2h(tall.data(), d_tall, sizeof(vpp_keypoint_f32) * world, nullptr)); CK(vpp_sync(nullptr));
  double worst = 0; for (auto& t : tall) worst = std::max(worst, double(t.pos_r));
  if (rank == 0) {
    std::printf("{\"workload\": \"pyrlk_match 1920x1080, 3 levels, %d keypoints, 7x7, sharded over %d ranks (C++ harness, one process per GPU), %d frame pair%s per step\", \"tracks_per_s\": %.1f, "
                "\"ms_per_step\": %.5f, \"ms_per_frame\": %.5f, \"frame_pairs_per_step\": %d, \"keypoints_per_rank\": %d, \"exchange\": \"rccl all_gather of %d 20-byte records per rank (vpp_allgather_tracks)\", "
                "\"launch\": \"%s\", \"steps\": %d, \"mismatched_vs_single_rank\": %d}\n",
                NK, world, F, F == 1 ? "" : "s", double(F) * NK / (worst * 1e-3), worst, worst / F, F, plan.per_rank, plan.per_rank * F, mode, steps, mismatched);
    std::remove(uid_file.c_str());
  }
  CK(vpp_comm_destroy(comm));
  return mismatched > 0 ? 4 : 0;
}
