// comm.hip — the one exchange step of the sharded path: keypoints are independent (pyrlk_match.hh:24-51), rank g
// tracks its slice against pyramids every rank holds, and one RCCL all-gather of the 20-byte keypoint records rebuilds the
// full set on every rank.  This is the C-ABI form for hosts written in the reference's language (C++): the Python harness
// (bench.py) does the same collective through torch.distributed, whose "nccl" backend is this RCCL.
// RCCL is resolved at first use with dlopen (reusing an already loaded librccl — e.g. the copy inside a PyTorch process —
// instead of pinning a second one at link time); a build without any RCCL fails loudly at vpp_comm_init, nowhere else.
#include "common.hpp"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
using namespace vpp_amd;

namespace {
// the slice of the RCCL C API used here (rccl.h: ncclUniqueId is 128 bytes, results are ints with 0 = success)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*CommDestroy_t)(void*);
typedef int (*AllGather_t)(const void*, void*, size_t, int /*ncclDataType_t*/, void*, hipStream_t);
typedef const char* (*GetErrorString_t)(int);
typedef int (*SendRecv_t)(void*, size_t, int /*ncclDataType_t*/, int /*peer*/, void*, hipStream_t);   // ncclSend (const void*) / ncclRecv
typedef int (*Group_t)(void);
struct Rccl {
  void* h = nullptr; GetUniqueId_t get_id = nullptr; CommInitRank_t init = nullptr; CommDestroy_t destroy = nullptr; AllGather_t all_gather = nullptr;
  GetErrorString_t err = nullptr; SendRecv_t send = nullptr, recv = nullptr; Group_t group_start = nullptr, group_end = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return;
    r.get_id = (GetUniqueId_t)dlsym(r.h, "ncclGetUniqueId"); r.init = (CommInitRank_t)dlsym(r.h, "ncclCommInitRank");
    r.destroy = (CommDestroy_t)dlsym(r.h, "ncclCommDestroy"); r.all_gather = (AllGather_t)dlsym(r.h, "ncclAllGather");
    r.err = (GetErrorString_t)dlsym(r.h, "ncclGetErrorString");
    r.send = (SendRecv_t)dlsym(r.h, "ncclSend"); r.recv = (SendRecv_t)dlsym(r.h, "ncclRecv");
    r.group_start = (Group_t)dlsym(r.h, "ncclGroupStart"); r.group_end = (Group_t)dlsym(r.h, "ncclGroupEnd");
    if (!(r.get_id && r.init && r.destroy && r.all_gather)) r.h = nullptr;
  });
  return r.h ? &r : nullptr;
}
#define VPP_RCCL_TRY(expr)                                                                                          \
  do {                                                                                                              \
    const int e_ = (expr);                                                                                          \
    if (e_ != 0) { set_error("%s failed: %s", #expr, R->err ? R->err(e_) : "rccl error"); return VPP_ERR_HIP; }     \
  } while (0)
}  // namespace

struct vpp_comm { void* nccl; int nranks, rank; };

namespace vpp_amd {
int comm_info(const vpp_comm* comm, int* rank, int* nranks) {
  VPP_REQUIRE(comm, VPP_ERR_INVALID_ARG, "null communicator");
  *rank = comm->rank; *nranks = comm->nranks;
  return VPP_OK;
}
int comm_allgather_inplace(vpp_comm* comm, void* base, size_t bytes_per_rank, hipStream_t st) {
  VPP_REQUIRE(comm && base, VPP_ERR_INVALID_ARG, "comm_allgather_inplace: invalid argument");
  if (comm->nranks == 1 || bytes_per_rank == 0) return VPP_OK;
  Rccl* R = rccl();
  VPP_REQUIRE(R, VPP_ERR_UNSUPPORTED, "no RCCL library");
  VPP_RCCL_TRY(R->all_gather((const char*)base + (size_t)comm->rank * bytes_per_rank, base, bytes_per_rank, 0 /* ncclInt8 */, comm->nccl, st));
  return VPP_OK;
}
int comm_group_begin() { Rccl* R = rccl(); VPP_REQUIRE(R && R->group_start, VPP_ERR_UNSUPPORTED, "no RCCL library"); VPP_RCCL_TRY(R->group_start()); return VPP_OK; }
int comm_group_end() { Rccl* R = rccl(); VPP_REQUIRE(R && R->group_end, VPP_ERR_UNSUPPORTED, "no RCCL library"); VPP_RCCL_TRY(R->group_end()); return VPP_OK; }
}  // namespace vpp_amd

extern "C" {

int vpp_comm_unique_id(void* id128) {
  VPP_REQUIRE(id128, VPP_ERR_INVALID_ARG, "vpp_comm_unique_id: null");
  Rccl* R = rccl();
  VPP_REQUIRE(R, VPP_ERR_UNSUPPORTED, "vpp_comm_unique_id: no RCCL library could be loaded (librccl.so.1)");
  VPP_RCCL_TRY(R->get_id((UniqueId*)id128));
  return VPP_OK;
}

int vpp_comm_init(vpp_comm** comm, int nranks, const void* id128, int rank) {
  VPP_REQUIRE(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, VPP_ERR_INVALID_ARG, "vpp_comm_init: invalid argument");
  Rccl* R = rccl();
  VPP_REQUIRE(R, VPP_ERR_UNSUPPORTED, "vpp_comm_init: no RCCL library could be loaded (librccl.so.1)");
  UniqueId id; memcpy(&id, id128, sizeof id);
  void* c = nullptr;
  VPP_RCCL_TRY(R->init(&c, nranks, id, rank));
  *comm = new vpp_comm{c, nranks, rank};
  return VPP_OK;
}

int vpp_comm_destroy(vpp_comm* comm) {
  if (!comm) return VPP_OK;
  Rccl* R = rccl();
  if (R && comm->nccl) VPP_RCCL_TRY(R->destroy(comm->nccl));
  delete comm;
  return VPP_OK;
}

int vpp_allgather_tracks(vpp_comm* comm, const vpp_keypoint_f32* shard, int n_per_rank, vpp_keypoint_f32* all, void* stream) {
  VPP_REQUIRE(comm && n_per_rank >= 0 && (n_per_rank == 0 || (shard && all)), VPP_ERR_INVALID_ARG, "vpp_allgather_tracks: invalid argument");
  if (n_per_rank == 0) return VPP_OK;
  Rccl* R = rccl();
  VPP_REQUIRE(R, VPP_ERR_UNSUPPORTED, "vpp_allgather_tracks: no RCCL library");
  static_assert(sizeof(vpp_keypoint_f32) == 20, "keypoint record is 20 bytes");
  VPP_RCCL_TRY(R->all_gather(shard, all, (size_t)n_per_rank * sizeof(vpp_keypoint_f32), 0 /* ncclInt8 / ncclChar */, comm->nccl, as_stream(stream)));
  return VPP_OK;
}

// ---- row-strip sharding of the image-space phases (SURVEY 8e bullet 2): halo rows --------------------------------------------
// A strip is an ordinary bordered image holding rows [r0, r1) of a frame; its border rows are the HALO: at a true frame edge they
// are filled like any border (fill_border_mirror), at an inner edge they must hold the neighbouring strip's first / last rows —
// whole rows of the pitch, column borders included, so that a stencil kernel run on the strip reads exactly what it would read
// on the full frame.  halo <= border rows are exchanged.
namespace {
inline uint8_t* strip_row(const vpp_image_desc* d, int r) { return (uint8_t*)d->first_pixel + (ptrdiff_t)r * d->pitch - (ptrdiff_t)d->border * elem_bytes(d); }
inline size_t strip_row_bytes(const vpp_image_desc* d) { return (size_t)(d->ncols + 2 * d->border) * elem_bytes(d); }
}  // namespace

// two strips of ONE process (two streams of one GPU, or two GPUs with peer access): upper's last rows -> lower's top halo and
// lower's first rows -> upper's bottom halo
int vpp_halo_copy(const vpp_image_desc* upper, const vpp_image_desc* lower, int halo, void* stream) {
  VPP_REQUIRE(valid_desc(upper) && valid_desc(lower) && same_type(upper, lower) && upper->ncols == lower->ncols && upper->border == lower->border, VPP_ERR_INVALID_ARG,
              "vpp_halo_copy: the strips must have the same width, border and element type");
  VPP_REQUIRE(halo >= 0 && halo <= upper->border && halo <= upper->nrows && halo <= lower->nrows, VPP_ERR_INVALID_ARG, "vpp_halo_copy: halo %d exceeds the border / the strips", halo);
  if (halo == 0) return VPP_OK;
  const size_t rb = strip_row_bytes(upper);
  VPP_HIP_TRY(hipMemcpy2DAsync(strip_row(lower, -halo), lower->pitch, strip_row(upper, upper->nrows - halo), upper->pitch, rb, halo, hipMemcpyDeviceToDevice, as_stream(stream)));
  VPP_HIP_TRY(hipMemcpy2DAsync(strip_row(upper, upper->nrows), upper->pitch, strip_row(lower, 0), lower->pitch, rb, halo, hipMemcpyDeviceToDevice, as_stream(stream)));
  return VPP_OK;
}

// one strip per rank, rank order = top to bottom: grouped RCCL send / recv with rank - 1 and rank + 1 over xGMI (rows are packed
// through staging buffers so that each direction is one message of halo * row_bytes)
int vpp_halo_exchange(vpp_comm* comm, const vpp_image_desc* strip, int halo, void* stream) {
  VPP_REQUIRE(comm && valid_desc(strip), VPP_ERR_INVALID_ARG, "vpp_halo_exchange: invalid argument");
  VPP_REQUIRE(halo >= 0 && halo <= strip->border && halo <= strip->nrows, VPP_ERR_INVALID_ARG, "vpp_halo_exchange: halo %d exceeds the border / the strip", halo);
  if (halo == 0 || comm->nranks == 1) return VPP_OK;
  Rccl* R = rccl();
  VPP_REQUIRE(R && R->send && R->recv && R->group_start && R->group_end, VPP_ERR_UNSUPPORTED, "vpp_halo_exchange: no RCCL send / recv");
  hipStream_t st = as_stream(stream);
  const size_t rb = strip_row_bytes(strip), msg = rb * halo;
  static thread_local Scratch scratch;   // 4 messages: send up, send down, recv from up, recv from down
  int rc = scratch.ensure(4 * msg, st);
  if (rc != VPP_OK) return rc;
  uint8_t* b = (uint8_t*)scratch.p;
  const bool up = comm->rank > 0, down = comm->rank + 1 < comm->nranks;
  if (up) VPP_HIP_TRY(hipMemcpy2DAsync(b, rb, strip_row(strip, 0), strip->pitch, rb, halo, hipMemcpyDeviceToDevice, st));
  if (down) VPP_HIP_TRY(hipMemcpy2DAsync(b + msg, rb, strip_row(strip, strip->nrows - halo), strip->pitch, rb, halo, hipMemcpyDeviceToDevice, st));
  VPP_RCCL_TRY(R->group_start());
  if (up) { VPP_RCCL_TRY(R->send(b, msg, 0, comm->rank - 1, comm->nccl, st)); VPP_RCCL_TRY(R->recv(b + 2 * msg, msg, 0, comm->rank - 1, comm->nccl, st)); }
  if (down) { VPP_RCCL_TRY(R->send(b + msg, msg, 0, comm->rank + 1, comm->nccl, st)); VPP_RCCL_TRY(R->recv(b + 3 * msg, msg, 0, comm->rank + 1, comm->nccl, st)); }
  VPP_RCCL_TRY(R->group_end());
  if (up) VPP_HIP_TRY(hipMemcpy2DAsync(strip_row(strip, -halo), strip->pitch, b + 2 * msg, rb, rb, halo, hipMemcpyDeviceToDevice, st));
  if (down) VPP_HIP_TRY(hipMemcpy2DAsync(strip_row(strip, strip->nrows), strip->pitch, b + 3 * msg, rb, rb, halo, hipMemcpyDeviceToDevice, st));
  return VPP_OK;
}

// Row-sharded frames: rank g holds rows [g * nrows / G, (g + 1) * nrows / G) of a frame (what a sharded decoder / capture front-end
// leaves on each GPU); one in-place RCCL all-gather of whole pitch rows completes the frame on every rank — the image-row exchange of
// the sharded semi-dense flow, whose matches may land anywhere in the frame (a fixed halo would not be exact).  nrows must divide evenly.
int vpp_allgather_rows(vpp_comm* comm, const vpp_image_desc* img, void* stream) {
  VPP_REQUIRE(comm && valid_desc(img), VPP_ERR_INVALID_ARG, "vpp_allgather_rows: invalid argument");
  VPP_REQUIRE(img->nrows % comm->nranks == 0, VPP_ERR_UNSUPPORTED, "vpp_allgather_rows: %d rows do not split evenly over %d ranks", img->nrows, comm->nranks);
  const int per = img->nrows / comm->nranks;
  return comm_allgather_inplace(comm, strip_row(img, 0), (size_t)per * img->pitch, as_stream(stream));
}

}  // extern "C"
