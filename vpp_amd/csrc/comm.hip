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
struct Rccl {
  void* h = nullptr; GetUniqueId_t get_id = nullptr; CommInitRank_t init = nullptr; CommDestroy_t destroy = nullptr; AllGather_t all_gather = nullptr;
  GetErrorString_t err = nullptr;
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

}  // extern "C"
