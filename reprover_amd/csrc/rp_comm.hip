// libreprover_hip - the collective of the sharded retrieve step behind the C ABI (rp_comm_init /
// rp_allgather_topk; §8(e): ONE ncclAllGather of the per-rank lists, then the per-query merge).
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: the library keeps loading on a box that has no
// librccl.so for callers that never shard, and a process that already carries an RCCL (torch ships its own copy) gets
// THAT copy instead of a second one.  A communicator is one (rank, device) pair; every call is enqueued on the caller's
// stream, nothing synchronises, no allocation after rp_comm_init.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include "rp_util.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;  // why binding failed (kept for every later call)
};

Rccl* rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* env = getenv("RP_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // a copy the process already holds wins (RTLD_NOLOAD), then the usual search path, then the ROCm tree
    for (int pass = 0; pass < 2 && !x.handle; ++pass)
      for (const char* n : names)
        if (n && !x.handle) x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!x.handle) {
      const char* why = dlerror();  // (one call: dlerror() clears the message it returns)
      x.error = std::string("librccl.so not found (set RP_RCCL_LIB): ") + (why ? why : "");
      return x;
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(x.handle, name);
      if (!p && x.error.empty()) x.error = std::string("librccl has no symbol ") + name;
      return p;
    };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return &r;
}

#define RP_RCCL_BOUND(r)                                                              \
  do {                                                                                \
    if (!(r)->error.empty()) return rp::fail(RP_E_COMM, "%s", (r)->error.c_str());    \
  } while (0)

#define RP_RCCL(r, expr)                                                                                       \
  do {                                                                                                         \
    ncclResult_t _e = (expr);                                                                                  \
    if (_e != ncclSuccess)                                                                                     \
      return rp::fail(RP_E_COMM, "%s failed: %s (%s:%d)", #expr, (r)->GetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

}  // namespace

struct RpComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

static_assert(RP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the ABI's id size is RCCL's");

extern "C" RpStatus rp_comm_unique_id(void* id_out) {
  RP_REQUIRE(id_out, "rp_comm_unique_id: null output");
  Rccl* r = rccl();
  RP_RCCL_BOUND(r);
  ncclUniqueId id;
  RP_RCCL(r, r->GetUniqueId(&id));
  memcpy(id_out, id.internal, RP_COMM_ID_BYTES);
  return RP_OK;
}

extern "C" RpStatus rp_comm_init(const void* unique_id, int32_t rank, int32_t world, RpComm** out) {
  RP_REQUIRE(unique_id && out, "rp_comm_init: null argument");
  RP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rp_comm_init: rank %d of %d", rank, world);
  Rccl* r = rccl();
  RP_RCCL_BOUND(r);
  ncclUniqueId id;
  memcpy(id.internal, unique_id, RP_COMM_ID_BYTES);
  RpComm* c = new RpComm;
  c->rank = rank;
  c->world = world;
  hipError_t he = hipGetDevice(&c->device);  // the communicator belongs to the calling thread's current device
  if (he != hipSuccess) {
    delete c;
    return rp::fail(RP_E_HIP, "hipGetDevice failed: %s", hipGetErrorString(he));
  }
  ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
  if (e != ncclSuccess) {
    delete c;
    return rp::fail(RP_E_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r->GetErrorString(e));
  }
  *out = c;
  return RP_OK;
}

extern "C" RpStatus rp_comm_destroy(RpComm* c) {
  if (!c) return RP_OK;
  Rccl* r = rccl();
  ncclResult_t e = (r->error.empty() && c->comm) ? r->CommDestroy(c->comm) : ncclSuccess;
  delete c;
  if (e != ncclSuccess) return rp::fail(RP_E_COMM, "ncclCommDestroy failed: %s", r->GetErrorString(e));
  return RP_OK;
}

extern "C" int32_t rp_comm_world(const RpComm* c) { return c ? c->world : 0; }
extern "C" int32_t rp_comm_rank(const RpComm* c) { return c ? c->rank : -1; }

extern "C" RpStatus rp_comm_allgather(RpComm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  RP_REQUIRE(c && send && recv, "rp_comm_allgather: null argument");
  RP_REQUIRE(bytes_per_rank % 4 == 0, "rp_comm_allgather: %zu bytes per rank (4-byte units)", bytes_per_rank);
  Rccl* r = rccl();
  RP_RCCL_BOUND(r);
  if (bytes_per_rank == 0) return RP_OK;
  rp::ProfScope ps((hipStream_t)stream, RP_K_COLLECTIVE);
  RP_RCCL(r, r->AllGather(send, recv, bytes_per_rank / 4, ncclInt32, c->comm, (hipStream_t)stream));
  return RP_OK;
}

extern "C" RpStatus rp_allgather_topk(RpComm* c, const void* send_block, void* recv_blocks, int32_t Bt, int32_t k,
                                      int32_t q0, int32_t B, float* out_scores, int32_t* out_ids, int32_t* out_count,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  RP_REQUIRE(c, "rp_allgather_topk: null communicator");
  RP_REQUIRE(Bt >= 0 && k >= 1 && q0 >= 0 && B >= 0 && q0 + B <= Bt, "rp_allgather_topk: queries [%d, %d) of %d, k %d",
             q0, q0 + B, Bt, k);
  const int64_t blk = (int64_t)Bt * (2 * k + 1);  // 4-byte units: [scores f32 [Bt,k] | ids i32 [Bt,k] | counts i32 [Bt]]
  RpStatus st = rp_comm_allgather(c, send_block, recv_blocks, (size_t)blk * 4, stream);
  if (st != RP_OK) return st;
  if (B == 0) return RP_OK;
  const int32_t* base = static_cast<const int32_t*>(recv_blocks);
  return rp_topk_merge_strided(reinterpret_cast<const float*>(base) + (int64_t)q0 * k, base + (int64_t)Bt * k + (int64_t)q0 * k,
                               base + 2 * (int64_t)Bt * k + q0, blk, c->world, B, k, out_scores, out_ids, out_count,
                               workspace, workspace_bytes, stream);
}
