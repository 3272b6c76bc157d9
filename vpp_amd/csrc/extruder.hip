// extruder.hip — the video_extruder tracker with its state resident in HBM (SURVEY 8f row 2, 8b `video_extruder_step`).
// Reference: vpp/algorithms/video_extruder/video_extruder.hpp:24-135 (video_extruder_update), vpp/core/keypoint_container.hpp
// (move :136-150, remove :118-126, add :96-115, compact :22-55, sync_attributes :67-102), vpp/core/keypoint_trajectory.hh:11-72.
//
// One vpp_video_extruder_step = one video_extruder_update, queued on one stream:
//   flow (K11/K12; its read-back launch also threads every match onto the merge lists, K16) -> ve_finish_kernel: merge verdict, FAST score of the
//   moved keypoints, apply move / remove per keypoint, and on ordinary frames the trajectory update
//   -> every detector_period-th frame: re-detection mask (K14), FAST-9 blockwise (K7-9), ve_rebuild_kernel: compaction of the container and of
//      the trajectories + append of the new keypoints + trajectory update.
// The container keeps the reference's layout semantics: dead entries (age 0) stay in place until the next compaction and take
// part in every step exactly as they do there (the flow still moves them, which revives them: move() increments the age).
// Trajectories are fixed-capacity rings in HBM, newest position first from `head`.  Nothing in an update waits for the device: a
// re-detection leaves the container's new size in HBM for the kernels behind it and the host reads its copy when it next needs it
// (ve_resolve).  vpp_video_extruder_push_frame / _push_host_frame are the video loop's shape: one frame per call, `prev` and its pyramid kept here.
#include <chrono>
#include "common.hpp"
#include "tracker_device.hpp"
#include <algorithm>
#include <vector>
using namespace vpp_amd;

struct vpp_video_extruder {
  int nrows = 0, ncols = 0, ring = 0;   // ring = slots per trajectory
  int n = 0, cap = 0, frame_id = -1, device = 0;
  // container, double buffered for the out-of-place compaction: pos / vel = (row, col) int32 pairs
  int32_t *pos[2] = {nullptr, nullptr}, *vel[2] = {nullptr, nullptr}, *age[2] = {nullptr, nullptr};
  // trajectories: ring[cap][ring] float2, head / len / start frame / alive per trajectory
  float *tring[2] = {nullptr, nullptr};
  int32_t *thead[2] = {nullptr, nullptr}, *tlen[2] = {nullptr, nullptr}, *tstart[2] = {nullptr, nullptr};
  uint8_t* talive[2] = {nullptr, nullptr};
  int cur = 0;
  // per-step scratch
  int32_t *fpos = nullptr, *fdist = nullptr, *scores = nullptr, *det = nullptr, *newidx = nullptr, *blocksum = nullptr;
  uint8_t *fvalid = nullptr, *merged = nullptr, *mask = nullptr;
  int mask_spacing = -1, mask_pitch = 0, det_cap = 0;
  size_t mask_bytes = 0;
  int32_t* host_count = nullptr;  // pinned: [0] alive count of the compaction, [1] keypoints found by the re-detection (copies of dcount);
                                  // bytes 8 .. 15: the same two counts + the update's frame id in ONE 64-bit word (stamp << 42 | found << 21 | alive; stamp = 2^21 | frame id mod 2^21), stored by the
                                  // rebuild launch's FIRST block — the host polls it instead of waiting for the launch's end (ve_resolve)
  unsigned pending_stamp = 0;
  int32_t* dcount = nullptr;      // the same two words in HBM, read by the kernels queued behind the re-detection
  hipEvent_t count_ready = nullptr;
  bool pending = false;           // a re-detection left the container size on the device: n is an upper bound until ve_resolve
  int pending_cap = 0;
  // vpp_video_extruder_push_frame: the pyramids of the previous and of the incoming frame, kept here so that every frame's pyramid is built once
  uint8_t* pyr_mem[2] = {nullptr, nullptr};
  vpp_image_desc pyr[2][8];        // levels with the flow's layout (border 2 * winsize of memory)
  int pyr_scales = 0, pyr_winsize = 0, pyr_fill = 0, pyr_prev = 0;
  bool have_prev = false;
  // vpp_video_extruder_push_host_frame: two staging frames filled by the copy engine on a stream of their own, while the previous update computes
  uint8_t* stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  int stage_k = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t staged[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};   // staged[k]: the upload into stage[k] is in HBM; consumed[k]: the pyramid built from it is done
  int last_k = 0;                 // the staging frame of the push made last
  bool consumed_set[2] = {false, false};
  // Updates are queued, never waited for: `queued` is recorded on the caller's stream behind the last kernel of every update, and whatever hands one of the
  // tracker's blocks back to the pool (destroy, larger buffers, other pyramid / mask / detection sizes) waits for it first — vpp_free does not synchronise,
  // and another stream or thread could be handed memory the queued kernels still use.  (An event outlives the stream it was recorded on.)
  hipEvent_t queued = nullptr;
  bool queued_set = false;
};
// nothing of the tracker's queued work may still touch its blocks
static void ve_quiesce(vpp_video_extruder* ve) {
  if (ve->queued && ve->queued_set) { if (hipEventSynchronize(ve->queued) != hipSuccess) (void)hipGetLastError(); }   // (an event recorded into a launch graph cannot be waited for: the graph's owner orders that)
}

namespace vpp_amd {
int keypoint_mask_squares(const vpp_image_desc* mask, const int32_t* rc, int n, int spacing, hipStream_t st);
int sdof_flow_linked(const vpp_image_desc* i1, const vpp_image_desc* i2, const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, const int32_t* kps, int n, int winsize, int nscales,
                     int propagation, int patchsize, int32_t* out_pos, int32_t* out_dist, uint8_t* out_valid, const MergeLinkArgs* link, size_t link_head_units, void* stream,
                     const vpp_image_desc* build2_levels = nullptr, const vpp_image_desc* build2_src = nullptr);   // sdof.hip
}

namespace {


// The rest of an update after the flow and the merge lists, per keypoint and in ONE launch (round 3): the merge's verdict (video_extruder.hpp:60-84),
// fast9_score at the keypoint's new position (:87-91), the match callback + merge + score cull applied to the container (:48-53), and — on the
// frames without re-detection, where nothing moves entries afterwards — the trajectory update (:122-133).  They were four launches (+ a fifth for the
// trajectories) of ~5 us each for a few instructions per keypoint.
template <bool TRAJ>
__global__ __launch_bounds__(256) void ve_finish_kernel(int n, int32_t* __restrict__ pos, int32_t* __restrict__ vel, int32_t* __restrict__ age, const int32_t* __restrict__ fpos,
                                                        const uint8_t* __restrict__ fvalid, MergeLists lists, DImg frame2, int th, float* __restrict__ ring,
                                                        int32_t* __restrict__ head, int32_t* __restrict__ len, uint8_t* __restrict__ alive, int slots, int max_len,
                                                        int32_t* __restrict__ blocksum) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int a = 0;
  if (i < n) {
  const int nr = frame2.nr, nc = frame2.nc;
  // (round 6) Three memory round trips instead of seven: everything that depends on i alone is requested first, in one go (the loads used to follow each other behind
  // the merge's list walk, the score's branch, the match's branch and the trajectory's branch — each waited for where its branch ended); then the cell's list head with
  // the score's 17 taps (at the position the match selects: no branch around them); then the walk.
  const int a_now = lists.age_now[i], cell = lists.cell_of[i];
  const int p0 = pos[2 * i], p1 = pos[2 * i + 1];
  const int r = fpos[2 * i], c = fpos[2 * i + 1];
  a = age[i];
  const bool matched = fvalid[i] != 0;
  int hd = 0, ln = 0;
  if (TRAJ) { hd = head[i]; ln = len[i]; }
  asm volatile("" :: "v"(a_now), "v"(cell), "v"(p0), "v"(p1), "v"(r), "v"(c), "v"(a), "v"(hd), "v"(ln));
  const bool inside = r >= 0 && c >= 0 && r < nr && c < nc;
  int j = lists.head[cell];
  const int score = fast9_score_at(frame2, inside ? r : p0, inside ? c : p1, th);   // an out-of-frame match removes the keypoint where it was (:50-53)
  bool merged;
  {  // merge_removes(lists, i) with its first two loads already here
    int E = -1, L = -1;
    bool earlier = false;
    for (; j >= 0; j = lists.next[j]) {
      if (j < i) { earlier = true; E = max(E, lists.age_now[j]); }
      else if (j > i) L = max(L, lists.age_now[j]);
    }
    merged = (earlier && a_now <= E) ? (a_now < E) : (L > a_now);
  }
  int q0 = p0, q1 = p1;
  if (matched) {  // the match callback (video_extruder.hpp:48-53)
    if (inside) {   // keypoint_container::move (keypoint_container.hpp:136-150)
      vel[2 * i] = r - p0; vel[2 * i + 1] = c - p1;
      pos[2 * i] = r; pos[2 * i + 1] = c;
      q0 = r; q1 = c;
      a++;
    } else a = 0;   // remove (:118-126)
  }
  if (merged || score < 3) a = 0;   // merge (:60-84) and score cull (:87-91)
  age[i] = a;
  if (TRAJ) {
    if (a > 0) {  // move_to + pop_oldest_position (video_extruder.hpp:125-130)
      const int h = (hd + slots - 1) % slots;
      float* p = ring + ((size_t)i * slots + h) * 2;
      p[0] = (float)q0; p[1] = (float)q1;
      head[i] = h;
      int l = ln + 1;
      if (l > max_len) l--;
      len[i] = l;
    } else alive[i] = 0;  // die() (:132)
  }
  }
  // a re-detection frame's compaction starts from the number of entries this workgroup leaves alive (round 6: the count pass that re-read every age is gone)
  if (!TRAJ && blocksum) {
    __shared__ int wcount[4];
    const unsigned long long b = __ballot(a > 0);
    if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) blocksum[blockIdx.x] = wcount[0] + wcount[1] + wcount[2] + wcount[3];
  }
}

// ---- compaction: alive entries keep their order (keypoint_container.hpp:22-55) -----------------------------------------
constexpr int kScanBlock = 1024;
__device__ __forceinline__ void ve_index_block(const unsigned bid, const unsigned nblocks, int n, const int32_t* __restrict__ age, const int32_t* __restrict__ blocksum, int32_t* __restrict__ newidx, int32_t* __restrict__ total) {
  // new index of every alive entry: block offset + rank inside the block (waves in order, lanes in order).  The block offset = the sum of the COUNTS of the blocks
  // before this one, summed here (a few hundred words out of L2, left per 256 entries by ve_finish_kernel; round 6: the count pass and the single-workgroup scan launch in front of this pass are gone); the last block leaves the total.
  __shared__ int wsum[4];
  __shared__ int psum[4];
  int run;
  {
    int s = 0;
    for (int j = threadIdx.x; j < (int)bid * (kScanBlock / 256); j += 256) s += blocksum[j];   // (counts per 256 entries: ve_finish_kernel's workgroups)
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) psum[threadIdx.x >> 6] = s;
    __syncthreads();
    run = psum[0] + psum[1] + psum[2] + psum[3];
  }
  for (int k = 0; k < kScanBlock / 256; k++) {
    const int i = (int)bid * kScanBlock + k * 256 + threadIdx.x;
    const bool a = i < n && age[i] > 0;
    const unsigned long long b = __ballot(a);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int off = run;
    for (int j = 0; j < w; j++) off += wsum[j];
    if (i < n) newidx[i] = a ? off + __popcll(b & ((1ull << lane) - 1)) : -1;
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (bid == nblocks - 1 && threadIdx.x == 0) *total = run;
}

__global__ __launch_bounds__(256) void ve_index_kernel(int n, const int32_t* __restrict__ age, const int32_t* __restrict__ blocksum, int32_t* __restrict__ newidx, int32_t* __restrict__ total) {
  ve_index_block(blockIdx.x, gridDim.x, n, age, blocksum, newidx, total);
}
// the mask's fill with the index pass as the launch's first blocks: two independent things behind ve_finish_kernel in one launch (round 6: a launch of a few blocks costs ~4.8 us by itself)
__global__ __launch_bounds__(256) void ve_fill16_index_kernel(uint4* __restrict__ p, size_t units, uint32_t v, unsigned index_blocks, int n, const int32_t* __restrict__ age,
                                                              const int32_t* __restrict__ blocksum, int32_t* __restrict__ newidx, int32_t* __restrict__ total) {
  if (blockIdx.x < index_blocks) { ve_index_block(blockIdx.x, index_blocks, n, age, blocksum, newidx, total); return; }
  const size_t u = (size_t)(blockIdx.x - index_blocks) * 256 + threadIdx.x;
  if (u < units) p[u] = make_uint4(v, v, v, v);
}
__global__ __launch_bounds__(256) void ve_fill16_kernel(uint4* __restrict__ p, size_t units, uint32_t v) {
  const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (u < units) p[u] = make_uint4(v, v, v, v);
}

// A re-detection frame's rebuild of the container in ONE launch: compaction of the surviving entries with their trajectory rings (keypoint_container::compact
// :22-55, sync_attributes :67-102), append of the new keypoints (:96-115) and the frame's trajectory update of both (video_extruder.hpp:122-133) — they were three
// launches (+ a device-to-host copy of the two counts, now written to the pinned words by this kernel).  The first `compact_blocks` blocks take the old
// entries, 16 lanes each (lane j copies ring slots j, j + 16, ...; the lane that owns the slot of the new head writes the new position there instead of
// the copied one), the others the new keypoints.
struct RebuildArgs {
  const int32_t *newidx, *pos, *vel, *age, *head, *len, *start; const float* ring; const uint8_t* alive;   // the old container
  int32_t *pos2, *vel2, *age2, *head2, *len2, *start2; float* ring2; uint8_t* alive2;                        // the new one
  const int32_t *dn, *det; int32_t* host_counts;
  int n, dcap, slots, max_len, frame_id, compact_blocks;
};
__global__ __launch_bounds__(256) void ve_rebuild_kernel(RebuildArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // (both counts are final before this launch starts: the host need not wait for its end to learn them)
    const unsigned long long w = ((unsigned long long)(0x200000u | ((unsigned)a.frame_id & 0x1FFFFFu)) << 42) | ((unsigned long long)(unsigned)min(a.dn[1], (1 << 21) - 1) << 21) | (unsigned long long)(unsigned)a.dn[0];
    __hip_atomic_store((unsigned long long*)(a.host_counts + 2), w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if ((int)blockIdx.x < a.compact_blocks) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, j = threadIdx.x & 15;
    if (i >= a.n) return;
    const int d = a.newidx[i];
    if (d < 0) return;   // (dead entries are dropped: age 0)
    const int h = (a.head[i] + a.slots - 1) % a.slots;   // compacted entries are alive: move_to + pop_oldest_position (:125-130)
    const int p0 = a.pos[2 * i], p1 = a.pos[2 * i + 1];
    if (j == 0) {
      a.pos2[2 * d] = p0; a.pos2[2 * d + 1] = p1; a.vel2[2 * d] = a.vel[2 * i]; a.vel2[2 * d + 1] = a.vel[2 * i + 1]; a.age2[d] = a.age[i];
      int l = a.len[i] + 1;
      if (l > a.max_len) l--;
      a.head2[d] = h; a.len2[d] = l; a.start2[d] = a.start[i]; a.alive2[d] = a.alive[i];
    }
    const float2* src = (const float2*)a.ring + (size_t)i * a.slots;
    float2* dst = (float2*)a.ring2 + (size_t)d * a.slots;
    for (int s = j; s < a.slots; s += 16) dst[s] = s == h ? make_float2((float)p0, (float)p1) : src[s];
    return;
  }
  const int k = ((int)blockIdx.x - a.compact_blocks) * 256 + threadIdx.x;
  const int m = a.dn[0], count = min(a.dn[1], a.dcap);
  if (k == 0) { a.host_counts[0] = m; a.host_counts[1] = a.dn[1]; }   // pinned words: the host reads them behind the event that follows this launch
  if (k >= count) return;
  const int d = m + k, h = a.slots - 1;   // keypoint<int>(kp) (keypoint_container.hh:16-18), keypoint_trajectory(frame_id), then its first move_to
  const int p0 = a.det[2 * k], p1 = a.det[2 * k + 1];
  a.pos2[2 * d] = p0; a.pos2[2 * d + 1] = p1; a.vel2[2 * d] = 0; a.vel2[2 * d + 1] = 0; a.age2[d] = 1;
  float* q = a.ring2 + ((size_t)d * a.slots + h) * 2;
  q[0] = (float)p0; q[1] = (float)p1;
  a.head2[d] = h; a.len2[d] = 1 > a.max_len ? 0 : 1; a.start2[d] = a.frame_id; a.alive2[d] = 1;
}

template <class T> int dalloc(T** p, size_t count) { void* v = nullptr; const int rc = vpp_malloc(count * sizeof(T), &v); *p = (T*)v; return rc; }
template <class T> void dfree(T*& p) { if (p) { vpp_free(p); p = nullptr; } }

// Grows the state's buffers.  Everything is allocated into a copy of the state first and *ve is only replaced once every allocation (and the
// carry-over of the live entries) has succeeded: on failure the new blocks are released and *ve still describes the old, intact buffers.
int ve_reserve(vpp_video_extruder* ve, int want, hipStream_t st) {
  if (want <= ve->cap) return VPP_OK;
  const int ncap = std::max(want + want / 4, 4096);
  vpp_video_extruder nw = *ve;
  for (int b = 0; b < 2; b++) { nw.pos[b] = nullptr; nw.vel[b] = nullptr; nw.age[b] = nullptr; nw.tring[b] = nullptr; nw.thead[b] = nullptr; nw.tlen[b] = nullptr; nw.tstart[b] = nullptr; nw.talive[b] = nullptr; }
  nw.fpos = nullptr; nw.fdist = nullptr; nw.scores = nullptr; nw.newidx = nullptr; nw.blocksum = nullptr; nw.fvalid = nullptr; nw.merged = nullptr;
  auto release = [](vpp_video_extruder& x) {
    for (int b = 0; b < 2; b++) { dfree(x.pos[b]); dfree(x.vel[b]); dfree(x.age[b]); dfree(x.tring[b]); dfree(x.thead[b]); dfree(x.tlen[b]); dfree(x.tstart[b]); dfree(x.talive[b]); }
    dfree(x.fpos); dfree(x.fdist); dfree(x.scores); dfree(x.newidx); dfree(x.blocksum); dfree(x.fvalid); dfree(x.merged);
  };
  int rc = VPP_OK;
  for (int b = 0; b < 2 && rc == VPP_OK; b++) {
    rc = dalloc(&nw.pos[b], (size_t)ncap * 2); if (rc) break;
    rc = dalloc(&nw.vel[b], (size_t)ncap * 2); if (rc) break;
    rc = dalloc(&nw.age[b], ncap); if (rc) break;
    rc = dalloc(&nw.tring[b], (size_t)ncap * ve->ring * 2); if (rc) break;
    rc = dalloc(&nw.thead[b], ncap); if (rc) break;
    rc = dalloc(&nw.tlen[b], ncap); if (rc) break;
    rc = dalloc(&nw.tstart[b], ncap); if (rc) break;
    rc = dalloc(&nw.talive[b], ncap);
  }
  if (rc == VPP_OK) rc = dalloc(&nw.fpos, (size_t)ncap * 2);
  if (rc == VPP_OK) rc = dalloc(&nw.fdist, ncap);
  if (rc == VPP_OK) rc = dalloc(&nw.scores, ncap);
  if (rc == VPP_OK) rc = dalloc(&nw.newidx, ncap);
  if (rc == VPP_OK) rc = dalloc(&nw.blocksum, (size_t)ncap / 256 + 2);   // alive counts per 256 entries (ve_finish_kernel's workgroups)
  if (rc == VPP_OK) rc = dalloc(&nw.fvalid, ncap);
  if (rc == VPP_OK) rc = dalloc(&nw.merged, ncap);
  if (rc != VPP_OK) { release(nw); return rc; }
  const int c = ve->cur;
  if (ve->n > 0) {  // carry the live state over (the other buffer of each pair is scratch)
    const size_t n = (size_t)ve->n;
    hipError_t e = hipMemcpyAsync(nw.pos[c], ve->pos[c], n * 8, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.vel[c], ve->vel[c], n * 8, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.age[c], ve->age[c], n * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.tring[c], ve->tring[c], n * ve->ring * 8, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.thead[c], ve->thead[c], n * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.tlen[c], ve->tlen[c], n * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.tstart[c], ve->tstart[c], n * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.talive[c], ve->talive[c], n, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nw.newidx, ve->newidx, n * 4, hipMemcpyDeviceToDevice, st);   // a compaction may be in progress
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(st);   // nothing may still be writing into the blocks that are about to be released
      release(nw);
      set_error("video_extruder: carrying the state over to larger buffers failed: %s", hipGetErrorString(e));
      return VPP_ERR_HIP;
    }
  }
  ve_quiesce(ve);   // n == 0 skipped the synchronising copy above: kernels of the last update may still be queued on the old blocks
  vpp_video_extruder old = *ve;
  *ve = nw;
  ve->cap = ncap;
  release(old);
  return VPP_OK;
}

}  // namespace

extern "C" {

int vpp_video_extruder_create(vpp_video_extruder** out, int nrows, int ncols, int trajectory_capacity) {
  VPP_REQUIRE(out && nrows > 0 && ncols > 0 && trajectory_capacity > 0, VPP_ERR_INVALID_ARG, "vpp_video_extruder_create: invalid argument");
  vpp_video_extruder* ve = new vpp_video_extruder();
  ve->nrows = nrows; ve->ncols = ncols; ve->ring = trajectory_capacity + 1;
  (void)hipGetDevice(&ve->device);
  void* h = nullptr;
  if (vpp_malloc_host(64, &h) != VPP_OK) { delete ve; return VPP_ERR_HIP; }
  ve->host_count = (int32_t*)h;
  ve->host_count[0] = ve->host_count[1] = 0; ve->host_count[2] = ve->host_count[3] = 0;
  if (dalloc(&ve->dcount, 2) != VPP_OK || hipEventCreateWithFlags(&ve->count_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ve->queued, hipEventDisableTiming) != hipSuccess) {
    if (ve->count_ready) (void)hipEventDestroy(ve->count_ready);
    dfree(ve->dcount); vpp_free_host(h); delete ve; return VPP_ERR_HIP;
  }
  // a first capacity that a blockwise detection with the default spacing cannot exceed: one keypoint per 10 x 10 block, twice over
  const int rc = ve_reserve(ve, std::max(4096, (nrows / 10 + 1) * (ncols / 10 + 1) * 2), nullptr);
  if (rc != VPP_OK) { (void)hipEventDestroy(ve->count_ready); (void)hipEventDestroy(ve->queued); dfree(ve->dcount); vpp_free_host(h); delete ve; return rc; }
  *out = ve;
  return VPP_OK;
}

int vpp_video_extruder_destroy(vpp_video_extruder* ve) {
  if (!ve) return VPP_OK;
  ve_quiesce(ve);   // the last update may still be queued on the caller's stream
  if (ve->queued) (void)hipEventDestroy(ve->queued);
  for (int b = 0; b < 2; b++) { dfree(ve->pos[b]); dfree(ve->vel[b]); dfree(ve->age[b]); dfree(ve->tring[b]); dfree(ve->thead[b]); dfree(ve->tlen[b]); dfree(ve->tstart[b]); dfree(ve->talive[b]); }
  dfree(ve->fpos); dfree(ve->fdist); dfree(ve->scores); dfree(ve->newidx); dfree(ve->blocksum); dfree(ve->fvalid); dfree(ve->merged); dfree(ve->det); dfree(ve->mask); dfree(ve->pyr_mem[0]); dfree(ve->pyr_mem[1]);
  if (ve->copy_stream) { (void)hipStreamSynchronize(ve->copy_stream); (void)hipStreamDestroy(ve->copy_stream); }
  for (int k = 0; k < 2; k++) { if (ve->staged[k]) (void)hipEventDestroy(ve->staged[k]); if (ve->consumed[k]) (void)hipEventDestroy(ve->consumed[k]); dfree(ve->stage[k]); }
  if (ve->count_ready) { (void)hipEventSynchronize(ve->count_ready); (void)hipEventDestroy(ve->count_ready); }
  dfree(ve->dcount);
  if (ve->host_count) vpp_free_host(ve->host_count);
  delete ve;
  return VPP_OK;
}

// A re-detection leaves the container's new size on the device (alive entries + keypoints found) and queues everything behind it against an upper bound;
// the host learns the size here, the next time it needs it — by then the two words have usually landed — instead of waiting inside the update.
static int ve_resolve(const vpp_video_extruder* cve) {
  vpp_video_extruder* ve = const_cast<vpp_video_extruder*>(cve);
  if (!ve->pending) return VPP_OK;
  // (round 6) The counts are known when the rebuild launch STARTS, and its first block stores them — stamped with the update's frame id — into a pinned word: polling that
  // word returns ~13 us (the rebuild) + the runtime's completion path earlier than the event behind the launch; everything queued next is ordered behind it by the stream.
  // Bounded: after 2 ms (a fault, a stream that has not started) the event decides.
  if (tuning("ve.poll_counts", 1) && ve->det_cap < (1 << 21) && ve->n < (1 << 21)) {
    const volatile unsigned long long* w = (const volatile unsigned long long*)(ve->host_count + 2);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
      const unsigned long long v = __atomic_load_n((const unsigned long long*)w, __ATOMIC_ACQUIRE);
      if ((unsigned)(v >> 42) == ve->pending_stamp) {
        ve->n = (int)(v & 0x1FFFFFu) + std::min((int)((v >> 21) & 0x1FFFFFu), ve->pending_cap);
        ve->pending = false;
        return VPP_OK;
      }
      if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  VPP_HIP_TRY(hipEventSynchronize(ve->count_ready));
  ve->n = ve->host_count[0] + std::min(ve->host_count[1], ve->pending_cap);
  ve->pending = false;
  return VPP_OK;
}

// pyr1 / pyr2 != nullptr: the frames' pyramids are the tracker's own (vpp_video_extruder_push_frame) and the flow takes them as they are
// build2_src != nullptr (with them): pyr2 is not built yet — the flow builds it from that frame into build2_levels in its first launch (or this function does, when there is no flow)
static int step_body(vpp_video_extruder* ve, const vpp_image_desc* frame1, const vpp_image_desc* frame2, const vpp_video_extruder_params* p, void* stream,
                     const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, const vpp_image_desc* build2_levels, const vpp_image_desc* build2_src);
static int step_impl(vpp_video_extruder* ve, const vpp_image_desc* frame1, const vpp_image_desc* frame2, const vpp_video_extruder_params* p, void* stream,
                     const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, const vpp_image_desc* build2_levels = nullptr, const vpp_image_desc* build2_src = nullptr) {
  const int frame_id = ve ? ve->frame_id : 0;
  const int rc = step_body(ve, frame1, frame2, p, stream, pyr1, pyr2, build2_levels, build2_src);
  // A failed update must not advance the tracker's clock: the caller's frame counter has not advanced either, and frame_id % detector_period decides which
  // frames re-detect for the rest of the sequence.  (What the failed call had already queued is the caller's to discard: re-upload the state.)
  if (rc != VPP_OK && ve) ve->frame_id = frame_id;
  if (ve && ve->queued && hipEventRecord(ve->queued, as_stream(stream)) == hipSuccess) ve->queued_set = true; else (void)hipGetLastError();
  return rc;
}
static int step_body(vpp_video_extruder* ve, const vpp_image_desc* frame1, const vpp_image_desc* frame2, const vpp_video_extruder_params* p, void* stream,
                     const vpp_image_desc* pyr1, const vpp_image_desc* pyr2, const vpp_image_desc* build2_levels, const vpp_image_desc* build2_src) {
  VPP_REQUIRE(ve && p && valid_desc(frame1) && valid_desc(frame2), VPP_ERR_INVALID_ARG, "vpp_video_extruder_step: invalid argument");
  // everything that can be refused is refused before anything is queued or counted (the reference throws here too: fast.hpp:937-938)
  VPP_REQUIRE(frame2->dtype == VPP_U8 && frame2->channels == 1 && frame1->dtype == VPP_U8 && frame1->channels == 1, VPP_ERR_UNSUPPORTED, "vpp_video_extruder_step: u8 x1 frames only");
  VPP_REQUIRE(frame2->border >= 3, VPP_ERR_BORDER_TOO_SMALL, "Image need a border of 3px at least for the FAST detector");
  VPP_REQUIRE(frame1->nrows == ve->nrows && frame1->ncols == ve->ncols && same_domain(frame1, frame2), VPP_ERR_INVALID_ARG, "vpp_video_extruder_step: frames do not match the tracker's domain");
  VPP_REQUIRE(p->keypoint_spacing > 0 && p->detector_period > 0 && p->max_trajectory_length > 0 && p->max_trajectory_length < ve->ring, VPP_ERR_INVALID_ARG,
              "vpp_video_extruder_step: max_trajectory_length %d exceeds the tracker's trajectory capacity %d", p->max_trajectory_length, ve->ring - 1);
  hipStream_t st = as_stream(stream);
  int rc = ve_resolve(ve);
  if (rc != VPP_OK) return rc;
  ve->frame_id++;
  const bool detect = ve->frame_id % p->detector_period == 0;   // re-detection frame: the compaction moves entries, the trajectories are updated after it
  const int s_ = p->keypoint_spacing;
  const int det_cap = (ve->nrows / s_ + 1) * (ve->ncols / s_ + 1);  // blockwise: at most one keypoint per s x s block
  if (detect) {   // room for every entry + every block's keypoint, before anything of this update is queued on the buffers
    rc = ve_reserve(ve, ve->n + det_cap, st);
    if (rc != VPP_OK) return rc;
  }
  const int c = ve->cur, n = ve->n;
  if (n > 0) {
    // the flow, with the merge's first pass (video_extruder.hpp:60-84: every keypoint threaded onto its cell's list) folded into its read-back launch
    // and the lists' heads reset beside the flow maps: two launches fewer per update than flow + heads reset + link
    MergeLinkArgs link; size_t head_units = 0;
    rc = keypoint_merge_prepare(ve->age[c], n, ve->nrows, ve->ncols, p->keypoint_spacing, &link, &head_units, st);
    if (rc != VPP_OK) return rc;
    rc = sdof_flow_linked(frame1, frame2, pyr1, pyr2, ve->pos[c], n, p->winsize, p->nscales, p->propagation, 5, ve->fpos, ve->fdist, ve->fvalid, &link, head_units, stream,
                          build2_levels, build2_src);
    if (rc != VPP_OK) return rc;
    const MergeLists lists{link.head, link.next, link.age_now, link.cell_of};
    if (detect) ve_finish_kernel<false><<<(n + 255) / 256, 256, 0, st>>>(n, ve->pos[c], ve->vel[c], ve->age[c], ve->fpos, ve->fvalid, lists, dimg(frame2), p->detector_th, ve->tring[c],
                                                                           ve->thead[c], ve->tlen[c], ve->talive[c], ve->ring, p->max_trajectory_length, ve->blocksum);
    else ve_finish_kernel<true><<<(n + 255) / 256, 256, 0, st>>>(n, ve->pos[c], ve->vel[c], ve->age[c], ve->fpos, ve->fvalid, lists, dimg(frame2), p->detector_th, ve->tring[c],
                                                                  ve->thead[c], ve->tlen[c], ve->talive[c], ve->ring, p->max_trajectory_length, nullptr);
  }
  if (n <= 0 && build2_src) {   // no flow to carry the new frame's pyramid: built here (the detection below reads its level 0, the next update all of it)
    rc = build2_src->channels == 1 ? vpp_pyramid_build(build2_levels, p->nscales, build2_src, stream) : vpp_rgb_pyramid_build(build2_levels, p->nscales, build2_src, stream);
    if (rc != VPP_OK) return rc;
  }
  if (detect) {  // re-detection away from every container entry (:94-119)
    const int s = p->keypoint_spacing;
    if (ve->mask_spacing != s) {
      ve_quiesce(ve);
      dfree(ve->mask);
      ve->mask_pitch = (ve->ncols + 2 * s + 31) / 32 * 32;
      ve->mask_bytes = (size_t)(ve->nrows + 2 * s) * ve->mask_pitch;
      rc = dalloc(&ve->mask, ve->mask_bytes);
      if (rc != VPP_OK) return rc;
      ve->mask_spacing = s;
    }
    vpp_image_desc md{ve->mask + (size_t)s * ve->mask_pitch + s, ve->nrows, ve->ncols, ve->mask_pitch, s, VPP_U8, 1};
    // fill_with_border(mask, 1) (:101): the mask block is the tracker's own, rows and padding: one memset instead of a pass over bordered rows
    // the alive count and the number of keypoints found stay on the device: the kernels behind them read the two words, the host reads its copy
    // when it next needs the container's size (ve_resolve)
    const int nblocks = (n + kScanBlock - 1) / kScanBlock;
    const unsigned fill_blocks = (unsigned)((ve->mask_bytes / 16 + 255) / 256);
    if (n > 0 && tuning("ve.index_in_fill", 1))
      ve_fill16_index_kernel<<<fill_blocks + (unsigned)nblocks, 256, 0, st>>>((uint4*)ve->mask, ve->mask_bytes / 16, 0x01010101u, (unsigned)nblocks, n, ve->age[c], ve->blocksum, ve->newidx, ve->dcount);
    else {
      ve_fill16_kernel<<<fill_blocks, 256, 0, st>>>((uint4*)ve->mask, ve->mask_bytes / 16, 0x01010101u);   // (the runtime's memset is two dispatches)
      if (n > 0) ve_index_kernel<<<nblocks, 256, 0, st>>>(n, ve->age[c], ve->blocksum, ve->newidx, ve->dcount);
      else { rc = device_fill(ve->dcount, 0, 4, st); if (rc != VPP_OK) return rc; }
    }
    rc = keypoint_mask_squares(&md, ve->pos[c], n, s, st);
    if (rc != VPP_OK) return rc;
    if (det_cap > ve->det_cap) {
      ve_quiesce(ve);
      dfree(ve->det);
      rc = dalloc(&ve->det, (size_t)det_cap * 2);
      if (rc != VPP_OK) return rc;
      ve->det_cap = det_cap;
    }
    rc = vpp_fast9_detect_async(frame2, p->detector_th, &md, VPP_FAST9_BLOCKWISE, s, VPP_FAST9_REFERENCE, ve->det, nullptr, det_cap, (uint32_t*)(ve->dcount + 1), stream);
    if (rc != VPP_OK) return rc;
    const int d = 1 - c;
    RebuildArgs ra;
    ra.newidx = ve->newidx; ra.pos = ve->pos[c]; ra.vel = ve->vel[c]; ra.age = ve->age[c]; ra.head = ve->thead[c]; ra.len = ve->tlen[c]; ra.start = ve->tstart[c];
    ra.ring = ve->tring[c]; ra.alive = ve->talive[c];
    ra.pos2 = ve->pos[d]; ra.vel2 = ve->vel[d]; ra.age2 = ve->age[d]; ra.head2 = ve->thead[d]; ra.len2 = ve->tlen[d]; ra.start2 = ve->tstart[d]; ra.ring2 = ve->tring[d];
    ra.alive2 = ve->talive[d];
    ra.dn = ve->dcount; ra.det = ve->det; ra.host_counts = ve->host_count;
    ra.n = n; ra.dcap = det_cap; ra.slots = ve->ring; ra.max_len = p->max_trajectory_length; ra.frame_id = ve->frame_id;
    ra.compact_blocks = (int)(((size_t)n * 16 + 255) / 256);
    __atomic_store_n((unsigned long long*)(ve->host_count + 2), 0ull, __ATOMIC_RELEASE);   // (a tracker whose state was re-uploaded sees the same frame ids again)
    ve_rebuild_kernel<<<(unsigned)(ra.compact_blocks + (det_cap + 255) / 256), 256, 0, st>>>(ra);
    VPP_HIP_TRY(hipEventRecord(ve->count_ready, st));
    ve->cur = d;
    ve->n = n + det_cap;   // upper bound until ve_resolve
    ve->pending = true; ve->pending_cap = det_cap; ve->pending_stamp = 0x200000u | ((unsigned)ve->frame_id & 0x1FFFFFu);   // (never 0: the word is zeroed before the launch)
  }
  VPP_LAUNCH_CHECK();
  return VPP_OK;
}

int vpp_video_extruder_step(vpp_video_extruder* ve, const vpp_image_desc* frame1, const vpp_image_desc* frame2, const vpp_video_extruder_params* p, void* stream) {
  return step_impl(ve, frame1, frame2, p, stream, nullptr, nullptr);
}

// (Re)carves the two pyramid sets for (nscales, winsize): levels of pyramid.hh:154's sizes, laid out like the flow's own (border 2 * winsize of memory, :72-73).
static int ve_pyramids(vpp_video_extruder* ve, int nscales, int winsize) {
  if (ve->pyr_mem[0] && ve->pyr_scales == nscales && ve->pyr_winsize == winsize) return VPP_OK;
  const int border = std::max(2 * winsize, 3);
  uint8_t* mem[2] = {nullptr, nullptr};
  vpp_image_desc lv[2][8];
  for (int pass = 0; pass < 2; pass++) {
    size_t off = 0;
    int nr = ve->nrows, nc = ve->ncols;
    for (int l = 0; l < nscales; l++) {
      int32_t pitch; size_t bytes, first;
      vpp_image_layout(nr, nc, 1, border, 32, &pitch, &bytes, &first);
      if (pass) for (int b = 0; b < 2; b++) lv[b][l] = vpp_image_desc{mem[b] + off + first, nr, nc, pitch, border, VPP_U8, 1};
      off += (bytes + 255) / 256 * 256;
      nr = 1 + nr / 2; nc = 1 + nc / 2;
    }
    if (!pass) {
      int rc = dalloc(&mem[0], off);
      if (rc == VPP_OK) rc = dalloc(&mem[1], off);
      if (rc != VPP_OK) { dfree(mem[0]); dfree(mem[1]); return rc; }
    }
  }
  ve_quiesce(ve);   // the last update's flow may still be reading the old pyramids
  dfree(ve->pyr_mem[0]); dfree(ve->pyr_mem[1]);
  for (int b = 0; b < 2; b++) { ve->pyr_mem[b] = mem[b]; for (int l = 0; l < nscales; l++) ve->pyr[b][l] = lv[b][l]; }
  ve->pyr_scales = nscales; ve->pyr_winsize = winsize;
  // only the pixels a SAD (winsize / 2) or the FAST ring (3) can reach beyond a level are mirror-filled (see flow_impl, sdof.hip)
  ve->pyr_fill = std::min(border, std::max(winsize / 2, 3));
  ve->have_prev = false;
  return VPP_OK;
}

// One frame in, one update out (examples/video_extruder.cc:44-58 keeps `prev` and calls video_extruder_update(ctx, prev, frame) per frame): the tracker
// keeps the previous frame's pyramid, so each frame's pyramid is built once (the two-frame entry builds both every update), and an RGB frame goes
// through the fused ingest + pyramid launch without a gray frame in between.  The first frame after create (or after nscales / winsize change)
// only becomes `prev` — no update, frame_id unchanged — exactly as the example's first iteration.
// ingested != nullptr: recorded on the stream right behind the pyramid launch, the last reader of `frame`
static int push_impl(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* p, void* stream, hipEvent_t ingested, bool* recorded) {
  VPP_REQUIRE(ve && p && valid_desc(frame), VPP_ERR_INVALID_ARG, "vpp_video_extruder_push_frame: invalid argument");
  VPP_REQUIRE(frame->nrows == ve->nrows && frame->ncols == ve->ncols, VPP_ERR_INVALID_ARG, "vpp_video_extruder_push_frame: the frame does not match the tracker's domain");
  VPP_REQUIRE(frame->dtype == VPP_U8 && (frame->channels == 1 || frame->channels == 3 || frame->channels == 4), VPP_ERR_UNSUPPORTED,
              "vpp_video_extruder_push_frame: u8 x1 (gray), x3 or x4 (rgb / rgba) frames only");
  VPP_REQUIRE(p->nscales >= 1 && p->nscales <= 8 && p->winsize > 0, VPP_ERR_INVALID_ARG, "vpp_video_extruder_push_frame: bad parameters");
  int rc = ve_pyramids(ve, p->nscales, p->winsize);
  if (rc != VPP_OK) return rc;
  const int in = ve->have_prev ? 1 - ve->pyr_prev : ve->pyr_prev;
  vpp_image_desc fill[8];
  for (int l = 0; l < p->nscales; l++) { fill[l] = ve->pyr[in][l]; fill[l].border = ve->pyr_fill; }
  // (round 6) With a previous frame there is an update behind this pyramid, and the update's flow starts with a launch of independent work (map reset + claims): the pyramid's
  // tiles ride in that launch (step_body hands frame and levels to the flow) instead of being one of their own — 9 + 6.6 us of launches become one of ~12 at 4K
  // (not for frames coming through the host staging buffers: `ingested` would be recorded behind the whole update instead of behind the frame's last reader)
  // (nor behind a re-detection: the update first waits for the host to learn the container's new size — a pyramid launch queued ahead of that wait runs during it)
  const bool in_flow = ve->have_prev && !ingested && !ve->pending && tuning("ve.pyramid_in_flow", 1);
  if (!in_flow) {
    rc = frame->channels == 1 ? vpp_pyramid_build(fill, p->nscales, frame, stream) : vpp_rgb_pyramid_build(fill, p->nscales, frame, stream);
    if (rc != VPP_OK) return rc;
    if (ingested) { VPP_HIP_TRY(hipEventRecord(ingested, as_stream(stream))); *recorded = true; }
  }
  if (!ve->have_prev) { ve->have_prev = true; return VPP_OK; }
  vpp_image_desc f1 = ve->pyr[ve->pyr_prev][0], f2 = fill[0];
  f1.border = ve->pyr_fill;
  rc = step_impl(ve, &f1, &f2, p, stream, ve->pyr[ve->pyr_prev], ve->pyr[in], in_flow ? fill : nullptr, in_flow ? frame : nullptr);
  if (rc != VPP_OK) { ve->have_prev = false; return rc; }   // the state of a failed update is the caller's to re-upload; the next frame starts over as `prev`
  ve->pyr_prev = in;
  return VPP_OK;
}

int vpp_video_extruder_push_frame(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* p, void* stream) {
  return push_impl(ve, frame, p, stream, nullptr, nullptr);
}

// The same call for a frame in HOST memory (a decoder's output; pinned memory — vpp_malloc_host — for the copy engine's full rate): the frame is
// copied into one of two staging frames on a stream of the tracker's own, so the upload of frame t + 1 runs while the update of frame t computes
// (a 4K rgb frame is 0.45 ms of PCIe, an update 0.2 ms of GPU: back to back they would add up).  Returns when the host buffer has been read —
// the caller may decode the next frame into it — with the update queued on `stream`.  Not recordable into a launch graph (it waits for the copy).
static int push_host_impl(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* p, void* stream, bool wait) {
  VPP_REQUIRE(ve && p && frame && frame->first_pixel && frame->nrows == ve->nrows && frame->ncols == ve->ncols, VPP_ERR_INVALID_ARG,
              "vpp_video_extruder_push_host_frame: the frame does not match the tracker's domain");
  VPP_REQUIRE(frame->dtype == VPP_U8 && (frame->channels == 1 || frame->channels == 3 || frame->channels == 4), VPP_ERR_UNSUPPORTED,
              "vpp_video_extruder_push_host_frame: u8 x1 (gray), x3 or x4 (rgb / rgba) frames only");
  const size_t row_bytes = (size_t)frame->ncols * frame->channels;
  VPP_REQUIRE(frame->pitch >= (int64_t)row_bytes, VPP_ERR_INVALID_ARG, "vpp_video_extruder_push_host_frame: pitch smaller than a row");
  const size_t dpitch = (row_bytes + 255) / 256 * 256, bytes = dpitch * frame->nrows;
  hipStream_t st = as_stream(stream);
  if (!ve->copy_stream) {
    VPP_HIP_TRY(hipStreamCreateWithFlags(&ve->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) { VPP_HIP_TRY(hipEventCreateWithFlags(&ve->staged[k], hipEventDisableTiming)); VPP_HIP_TRY(hipEventCreateWithFlags(&ve->consumed[k], hipEventDisableTiming)); }
  }
  if (ve->stage_bytes < bytes) {   // first frame, or wider pixels than before: nothing may still read the old staging frames
    VPP_HIP_TRY(hipStreamSynchronize(ve->copy_stream));
    VPP_HIP_TRY(hipStreamSynchronize(st));
    uint8_t* nw[2] = {nullptr, nullptr};
    int rc = dalloc(&nw[0], bytes);
    if (rc == VPP_OK) rc = dalloc(&nw[1], bytes);
    if (rc != VPP_OK) { dfree(nw[0]); dfree(nw[1]); return rc; }
    for (int k = 0; k < 2; k++) { dfree(ve->stage[k]); ve->stage[k] = nw[k]; ve->consumed_set[k] = false; }
    ve->stage_bytes = bytes;
  }
  const int k = ve->stage_k;
  ve->stage_k ^= 1; ve->last_k = k;
  if (ve->consumed_set[k]) VPP_HIP_TRY(hipStreamWaitEvent(ve->copy_stream, ve->consumed[k], 0));   // the pyramid built from this staging frame two frames ago
  if ((size_t)frame->pitch == row_bytes && dpitch == row_bytes)   // tight rows on both sides (3840-wide frames are): one linear copy for the copy engine
    VPP_HIP_TRY(hipMemcpyAsync(ve->stage[k], frame->first_pixel, bytes, hipMemcpyHostToDevice, ve->copy_stream));
  else
    VPP_HIP_TRY(hipMemcpy2DAsync(ve->stage[k], dpitch, frame->first_pixel, (size_t)frame->pitch, row_bytes, (size_t)frame->nrows, hipMemcpyHostToDevice, ve->copy_stream));
  VPP_HIP_TRY(hipEventRecord(ve->staged[k], ve->copy_stream));
  VPP_HIP_TRY(hipStreamWaitEvent(st, ve->staged[k], 0));
  const vpp_image_desc d{ve->stage[k], frame->nrows, frame->ncols, (int32_t)dpitch, 0, VPP_U8, frame->channels};
  bool recorded = false;
  const int rc = push_impl(ve, &d, p, stream, ve->consumed[k], &recorded);   // queued while the copy is still in flight
  ve->consumed_set[k] = recorded;
  if (wait) VPP_HIP_TRY(hipEventSynchronize(ve->staged[k]));   // the host buffer is the caller's again
  return rc;
}

int vpp_video_extruder_push_host_frame(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* p, void* stream) {
  return push_host_impl(ve, frame, p, stream, true);
}
// For a caller that rotates two host buffers (decode frame t + 1 into one while frame t uploads from the other): returns with the copy still in flight.
// vpp_video_extruder_wait_host_frame(ve, back) returns once the frame pushed last (back = 0) or the one before it (back = 1) has been read.
int vpp_video_extruder_push_host_frame_nowait(vpp_video_extruder* ve, const vpp_image_desc* frame, const vpp_video_extruder_params* p, void* stream) {
  return push_host_impl(ve, frame, p, stream, false);
}
int vpp_video_extruder_wait_host_frame(vpp_video_extruder* ve, int back) {
  VPP_REQUIRE(ve && (back == 0 || back == 1), VPP_ERR_INVALID_ARG, "vpp_video_extruder_wait_host_frame: back must be 0 or 1");
  const int k = back ? 1 - ve->last_k : ve->last_k;
  if (ve->staged[k]) VPP_HIP_TRY(hipEventSynchronize(ve->staged[k]));   // (never recorded: returns at once)
  return VPP_OK;
}

int vpp_video_extruder_count(const vpp_video_extruder* ve, int* n, int* frame_id) {
  VPP_REQUIRE(ve, VPP_ERR_INVALID_ARG, "vpp_video_extruder_count: null");
  if (n) { const int rc = ve_resolve(ve); if (rc != VPP_OK) return rc; *n = ve->n; }   // (frame_id alone never waits)
  if (frame_id) *frame_id = ve->frame_id;
  return VPP_OK;
}

int vpp_video_extruder_keypoints(const vpp_video_extruder* ve, int32_t* pos_rc, int32_t* vel_rc, int32_t* age, int capacity, void* stream) {
  if (ve) { const int rc = ve_resolve(ve); if (rc != VPP_OK) return rc; }
  VPP_REQUIRE(ve && capacity >= ve->n, VPP_ERR_INVALID_ARG, "vpp_video_extruder_keypoints: capacity %d < %d entries", capacity, ve ? ve->n : 0);
  hipStream_t st = as_stream(stream);
  const int k = ve->cur; const size_t n = ve->n;
  if (n && pos_rc) VPP_HIP_TRY(hipMemcpyAsync(pos_rc, ve->pos[k], n * 8, hipMemcpyDeviceToHost, st));
  if (n && vel_rc) VPP_HIP_TRY(hipMemcpyAsync(vel_rc, ve->vel[k], n * 8, hipMemcpyDeviceToHost, st));
  if (n && age) VPP_HIP_TRY(hipMemcpyAsync(age, ve->age[k], n * 4, hipMemcpyDeviceToHost, st));
  VPP_HIP_TRY(hipStreamSynchronize(st));
  return VPP_OK;
}

int vpp_video_extruder_trajectories(const vpp_video_extruder* ve, int32_t* len, int32_t* start_frame, uint8_t* alive, int32_t* head, float* ring_rc, int capacity, void* stream) {
  if (ve) { const int rc = ve_resolve(ve); if (rc != VPP_OK) return rc; }
  VPP_REQUIRE(ve && capacity >= ve->n, VPP_ERR_INVALID_ARG, "vpp_video_extruder_trajectories: capacity %d < %d entries", capacity, ve ? ve->n : 0);
  hipStream_t st = as_stream(stream);
  const int k = ve->cur; const size_t n = ve->n;
  if (n && len) VPP_HIP_TRY(hipMemcpyAsync(len, ve->tlen[k], n * 4, hipMemcpyDeviceToHost, st));
  if (n && start_frame) VPP_HIP_TRY(hipMemcpyAsync(start_frame, ve->tstart[k], n * 4, hipMemcpyDeviceToHost, st));
  if (n && alive) VPP_HIP_TRY(hipMemcpyAsync(alive, ve->talive[k], n, hipMemcpyDeviceToHost, st));
  if (n && head) VPP_HIP_TRY(hipMemcpyAsync(head, ve->thead[k], n * 4, hipMemcpyDeviceToHost, st));
  if (n && ring_rc) VPP_HIP_TRY(hipMemcpyAsync(ring_rc, ve->tring[k], n * ve->ring * 8, hipMemcpyDeviceToHost, st));
  VPP_HIP_TRY(hipStreamSynchronize(st));
  return VPP_OK;
}

int vpp_video_extruder_trajectory_slots(const vpp_video_extruder* ve, int* slots) {
  VPP_REQUIRE(ve && slots, VPP_ERR_INVALID_ARG, "vpp_video_extruder_trajectory_slots: null");
  *slots = ve->ring;
  return VPP_OK;
}

// Replace the tracker's state with host data (a caller that edited the container between two updates): n entries,
// ring_rc = n x slots x (row, col) floats laid out like vpp_video_extruder_trajectories returns them.
int vpp_video_extruder_upload(vpp_video_extruder* ve, int n, int frame_id, const int32_t* pos_rc, const int32_t* vel_rc, const int32_t* age, const int32_t* len,
                              const int32_t* start_frame, const uint8_t* alive, const int32_t* head, const float* ring_rc, void* stream) {
  VPP_REQUIRE(ve && n >= 0, VPP_ERR_INVALID_ARG, "vpp_video_extruder_upload: invalid argument");
  hipStream_t st = as_stream(stream);
  { const int rc0 = ve_resolve(ve); if (rc0 != VPP_OK) return rc0; }
  ve->n = 0;  // nothing to carry over
  int rc = ve_reserve(ve, n, st);
  if (rc != VPP_OK) return rc;
  const int k = ve->cur; const size_t m = n;
  if (m) {
    VPP_HIP_TRY(hipMemcpyAsync(ve->pos[k], pos_rc, m * 8, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->vel[k], vel_rc, m * 8, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->age[k], age, m * 4, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->tlen[k], len, m * 4, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->tstart[k], start_frame, m * 4, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->talive[k], alive, m, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->thead[k], head, m * 4, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipMemcpyAsync(ve->tring[k], ring_rc, m * ve->ring * 8, hipMemcpyHostToDevice, st));
    VPP_HIP_TRY(hipStreamSynchronize(st));
  }
  ve->n = n; ve->frame_id = frame_id;
  return VPP_OK;
}

}  // extern "C"
