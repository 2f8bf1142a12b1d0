// Gradient exchange through the C ABI (`..._allreduce(comm*, buf, nbytes, stream)`): RCCL communicators as opaque
// handles, one in-place all-reduce per gradient bucket on the caller's stream.  Stands for what the reference gets from
// Lightning's `ddp_find_unused_parameters_false` strategy (config/config.yaml:45, train.sh:6): torch DDP's bucketed
// all-reduce over NCCL.  mrmt3/ddp.py drives the same buckets either through torch.distributed (default) or through these
// entry points (MRMT3_DDP_NATIVE=1); a host that is not Python needs only these four calls and a way to hand the 128-byte id
// from rank 0 to the other ranks.
//
// RCCL is resolved at FIRST USE with dlopen / dlsym — the library has no link-time dependency on it, a process that never
// exchanges gradients never loads it, and inside a PyTorch process the RCCL that torch already mapped is the one used (two
// RCCL copies in one process would each claim the xGMI links).  Search order: $MRMT3_RCCL_LIB, an already loaded
// librccl.so.1 / librccl.so, then the loader's search path, then /opt/rocm/lib.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>      // types and prototypes only (decltype below); nothing is linked

#include "common.h"

namespace {
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

// (one host thread per process drives the GPU — SURVEY 8b; the function-local static still makes the first use thread-safe)
const Rccl* rccl() {
  static const Rccl r = [] {
    Rccl t;
    const char* env = getenv("MRMT3_RCCL_LIB");
    if (env && *env) t.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    const char* loaded[] = {"librccl.so.1", "librccl.so"};
    for (int i = 0; i < 2 && !t.handle; ++i) t.handle = dlopen(loaded[i], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (int i = 0; i < 4 && !t.handle; ++i) t.handle = dlopen(fresh[i], RTLD_NOW | RTLD_LOCAL);
    if (!t.handle) return t;
    t.get_unique_id = (decltype(t.get_unique_id))dlsym(t.handle, "ncclGetUniqueId");
    t.comm_init_rank = (decltype(t.comm_init_rank))dlsym(t.handle, "ncclCommInitRank");
    t.comm_destroy = (decltype(t.comm_destroy))dlsym(t.handle, "ncclCommDestroy");
    t.all_reduce = (decltype(t.all_reduce))dlsym(t.handle, "ncclAllReduce");
    t.error_string = (decltype(t.error_string))dlsym(t.handle, "ncclGetErrorString");
    return t;
  }();
  if (!r.handle || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
    mrmt3_set_error("comm: RCCL not found (set MRMT3_RCCL_LIB to librccl.so): %s", r.handle ? "missing symbol" : "dlopen failed");
    return nullptr;
  }
  return &r;
}

struct Comm {
  ncclComm_t nccl;
  int rank, world;
};
}  // namespace

#define MR_CHECK_RCCL(r, expr, what)                                            \
  do {                                                                          \
    const ncclResult_t e_ = (expr);                                             \
    if (e_ != ncclSuccess) {                                                    \
      mrmt3_set_error("%s: %s", what, (r)->error_string(e_));                   \
      return MRMT3_ERR_COMM;                                                 \
    }                                                                           \
  } while (0)

extern "C" int mrmt3_comm_unique_id(void* id_out) {
  MR_CHECK_ARG(id_out, "comm_unique_id: null pointer");
  const Rccl* r = rccl();
  if (!r) return MRMT3_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == MRMT3_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  MR_CHECK_RCCL(r, r->get_unique_id(&id), "comm_unique_id");
  memcpy(id_out, &id, sizeof(id));
  return MRMT3_OK;
}

extern "C" int mrmt3_comm_create(const void* id, int rank, int world, void** comm_out) {
  MR_CHECK_ARG(id && comm_out, "comm_create: null pointer");
  MR_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "comm_create: rank %d of %d", rank, world);
  const Rccl* r = rccl();
  if (!r) return MRMT3_ERR_COMM;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  Comm* c = new Comm{nullptr, rank, world};
  const ncclResult_t e = r->comm_init_rank(&c->nccl, world, uid, rank);      // blocks until every rank has called it
  if (e != ncclSuccess) {
    mrmt3_set_error("comm_create: ncclCommInitRank(rank %d of %d): %s", rank, world, r->error_string(e));
    delete c;
    return MRMT3_ERR_COMM;
  }
  *comm_out = c;
  return MRMT3_OK;
}

extern "C" int mrmt3_comm_destroy(void* comm) {
  if (!comm) return MRMT3_OK;
  Comm* c = (Comm*)comm;
  const Rccl* r = rccl();
  if (!r) return MRMT3_ERR_COMM;
  const ncclResult_t e = r->comm_destroy(c->nccl);
  delete c;
  if (e != ncclSuccess) {
    mrmt3_set_error("comm_destroy: %s", r->error_string(e));
    return MRMT3_ERR_COMM;
  }
  return MRMT3_OK;
}

// in place: buf[i] = sum over ranks (average != 0: mean over ranks) of buf[i]; asynchronous on `stream`
extern "C" int mrmt3_allreduce(void* comm, void* buf, size_t count, int dtype, int average, void* stream) {
  MR_CHECK_ARG(comm && (buf || count == 0), "allreduce: null pointer");
  MR_CHECK_ARG(dtype == MRMT3_F32 || dtype == MRMT3_BF16, "allreduce: unknown dtype %d", dtype);
  if (count == 0) return MRMT3_OK;
  const Rccl* r = rccl();
  if (!r) return MRMT3_ERR_COMM;
  Comm* c = (Comm*)comm;
  MR_CHECK_RCCL(r, r->all_reduce(buf, buf, count, dtype == MRMT3_F32 ? ncclFloat32 : ncclBfloat16, average ? ncclAvg : ncclSum,
                                 c->nccl, (hipStream_t)stream), "allreduce");
  return MRMT3_OK;
}
