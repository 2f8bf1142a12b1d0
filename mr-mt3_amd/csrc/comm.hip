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

// ---- counting hand-offs between two streams ---------------------------------------------------------------------------------
// In the product: the probe that tells whether two streams run side by side (Trainer._pick_collective_stream: a wait on one,
// its signal on the other; on a shared hardware queue the wait times out).  Built in round 5 for the data-parallel step as TWO
// hipGraphs on two streams (measured, lost, removed in round 6: profiles/tools/closed/trainer_captured_collectives_r5.py): the
// compute chain and the chain of the gradient buckets' all-reduces.  Each is captured as ONE linear chain (ROCm 7.2 maps forked graphs badly: a
// graph with a side branch replayed in 40.9 instead of 25 ms), and what one graph has to tell the other —
// "bucket i is complete", "every bucket is reduced" — goes through counting flags in device memory instead of events:
// an event-wait node waits for whatever record is the latest when it RUNS, which for two graphs replayed side by side may
// be the previous step's (already complete) one; a counter cannot be satisfied by an old signal.
//   signal: flag += 1 (release, agent scope), after everything before it on its stream
//   wait:   until flag >= seen + 1 (acquire, agent scope), then seen += 1.  `seen` belongs to the waiting side alone.
// A wait gives up after `timeout_ms` (the other graph was never launched, or failed): it raises *err and lets its stream go
// on, so that a broken step shows up as an error word the host checks — never as a GPU that spins for ever.
__global__ void flag_signal_kernel(int* flag) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_kernel(const int* flag, int* seen, int* err, unsigned long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  const int want = *seen + 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                  // 100 MHz
  // The poll is a RELAXED device-scope load: an acquire at this scope is an L2 invalidate of the XCD the wave sits on, and
  // one of those every microsecond through a whole backward pass cost the compute graph beside it 4 ms of a 24.7 ms step
  // (profiles/r05_collectives_ab.txt).  One acquire fence once the signal has been seen orders what follows.
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
    if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  *seen = want;
}

extern "C" int mrmt3_flag_signal(int32_t* flag, void* stream) {
  MR_CHECK_ARG(flag, "flag_signal: null pointer");
  hipLaunchKernelGGL(flag_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)flag);
  MR_CHECK_LAUNCH("flag_signal");
  return MRMT3_OK;
}

extern "C" int mrmt3_flag_wait(const int32_t* flag, int32_t* seen, int32_t* err, int timeout_ms, void* stream) {
  MR_CHECK_ARG(flag && seen && err && timeout_ms > 0, "flag_wait: null pointer or no timeout");
  hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)flag, (int*)seen, (int*)err,
                     (unsigned long long)timeout_ms * 100000ull);
  MR_CHECK_LAUNCH("flag_wait");
  return MRMT3_OK;
}

#ifdef MRMT3_DIAG
// Diagnostics build only (profiles/tools/overlap_emulation.py): a stand-in for an all-reduce on a box with ONE GPU.  `n_ctas`
// workgroups (RCCL's channels occupy a few dozen CUs) read and rewrite `buf` in place — values unchanged — until `seconds` have
// passed: a kernel that is resident on the collective stream's hardware queue, takes memory bandwidth and lasts as long as the
// real collective would at an assumed bus bandwidth.  What it cannot emulate: launch latency across ranks, stragglers, link errors.
__global__ void comm_emulate_kernel(float* buf, size_t n, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const size_t stride = (size_t)gridDim.x * blockDim.x, chunk = stride * 64;
  size_t base = 0;
  do {
    for (size_t i = base + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n && i < base + chunk; i += stride) {
      float v = buf[i];
      asm volatile("" : "+v"(v));
      buf[i] = v;
    }
    base += chunk;
    if (base >= n) base = 0;
  } while (__builtin_amdgcn_s_memrealtime() - t0 < ticks);
}
// (probe, profiles/tools/wait_value_probe.py) the signal with SYSTEM scope: for a word the command processor polls
__global__ void flag_signal_sys_kernel(int* flag) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int mrmt3_flag_signal_sys(int32_t* flag, void* stream) {
  MR_CHECK_ARG(flag, "flag_signal_sys: null pointer");
  hipLaunchKernelGGL(flag_signal_sys_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)flag);
  MR_CHECK_LAUNCH("flag_signal_sys");
  return MRMT3_OK;
}
extern "C" int mrmt3_comm_emulate(void* buf, size_t count, double seconds, int n_ctas, void* stream) {
  MR_CHECK_ARG(buf && count > 0 && seconds >= 0 && n_ctas > 0, "comm_emulate: bad arguments");
  hipLaunchKernelGGL(comm_emulate_kernel, dim3((unsigned)n_ctas), dim3(256), 0, (hipStream_t)stream, (float*)buf, count,
                     (unsigned long long)(seconds * 1e8));
  MR_CHECK_LAUNCH("comm_emulate");
  return MRMT3_OK;
}
#endif
