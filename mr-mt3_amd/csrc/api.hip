// Library-level entry points: version and thread-local error text.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void mrmt3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#ifdef MRMT3_DIAG
int mrmt3_diag_env(const char* name) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  if (v != 0) {
    static std::atomic<unsigned> warned{0};          // bit per switch name (two switches exist: hash by first letter after MRMT3_)
    const unsigned bit = 1u << ((unsigned char)name[6] & 31);
    if (!(warned.fetch_or(bit, std::memory_order_relaxed) & bit))
      fprintf(stderr, "mrmt3: %s=%d is a kernel DIAGNOSTIC switch (parts of a kernel are knocked out): the results of this process are NOT valid\n", name, v);
  }
  return v;
}
#endif

// ---- knobs (common.h): environment switches read once per process, overridable through the ABI ------------------------
namespace {
std::mutex g_knob_mu;
MrKnob* g_knob_sites = nullptr;                     // every MR_KNOB site that has been read
struct KnobOverride { char name[48]; int value; };
KnobOverride g_knob_over[32];
int g_knob_n_over = 0;

bool knob_parse(const char* e, int* out) {
  if (e == nullptr || *e == 0) return false;
  char* end = nullptr;
  const long v = strtol(e, &end, (e[0] == '0' && (e[1] == 'x' || e[1] == 'X')) ? 16 : 10);
  if (end == e) return false;
  *out = (int)v;
  return true;
}
// (lock held) the value the process should see for `name`: an override, else the environment, else `dflt`
int knob_resolve(const char* name, int dflt) {
  for (int i = 0; i < g_knob_n_over; ++i)
    if (strcmp(g_knob_over[i].name, name) == 0) return g_knob_over[i].value;
  int v;
  return knob_parse(getenv(name), &v) ? v : dflt;
}
}  // namespace

int mrmt3_knob_get(MrKnob* k, int dflt) {
#ifndef MRMT3_DIAG
  // the launch path after the first use: no lock, no environment access.  state is published with release order AFTER value
  // (below) and read with acquire order here, so a second launching thread never sees state == 1 with a stale value
  if (__atomic_load_n(&k->state, __ATOMIC_ACQUIRE) == 1) return __atomic_load_n(&k->value, __ATOMIC_RELAXED);
#endif
  std::lock_guard<std::mutex> lock(g_knob_mu);
  const int v = knob_resolve(k->name, dflt);
#ifndef MRMT3_DIAG
  if (!k->registered) { k->registered = 1; k->next = g_knob_sites; g_knob_sites = k; }
  __atomic_store_n(&k->value, v, __ATOMIC_RELAXED);
  __atomic_store_n(&k->state, 1, __ATOMIC_RELEASE);
#endif
  return v;
}

extern "C" int mrmt3_set_knob(const char* name, int value) {
  if (name == nullptr || strncmp(name, "MRMT3_", 6) != 0 || strlen(name) >= sizeof(g_knob_over[0].name)) {
    mrmt3_set_error("set_knob: a knob is named MRMT3_<SWITCH> (at most %zu characters)", sizeof(g_knob_over[0].name) - 1);
    return MRMT3_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lock(g_knob_mu);
  int i = 0;
  while (i < g_knob_n_over && strcmp(g_knob_over[i].name, name) != 0) ++i;
  if (i == g_knob_n_over) {
    if (g_knob_n_over == (int)(sizeof(g_knob_over) / sizeof(g_knob_over[0]))) {
      mrmt3_set_error("set_knob: more than %d overrides", g_knob_n_over);
      return MRMT3_ERR_INVALID_ARG;
    }
    strcpy(g_knob_over[g_knob_n_over++].name, name);
  }
  g_knob_over[i].value = value;
  for (MrKnob* k = g_knob_sites; k != nullptr; k = k->next)
    if (strcmp(k->name, name) == 0) __atomic_store_n(&k->state, 0, __ATOMIC_RELEASE);   // re-resolved (to the override) at the next launch that asks
  return MRMT3_OK;
}

extern "C" int mrmt3_reset_knobs(void) {
  std::lock_guard<std::mutex> lock(g_knob_mu);
  g_knob_n_over = 0;
  for (MrKnob* k = g_knob_sites; k != nullptr; k = k->next) __atomic_store_n(&k->state, 0, __ATOMIC_RELEASE);   // back to the environment's value (or the default)
  return MRMT3_OK;
}

extern "C" int mrmt3_version(void) { return 110; /* 0.1.10: round 6 (capture hygiene entry points, owned streams, abort trace, pair bf16 conversion); 108: round 5 (knobs read once per process + mrmt3_set_knob; kernel diagnostics only in the -DMRMT3_DIAG build); 107: round 4 */ }
extern "C" const char* mrmt3_last_error(void) { return g_err; }

// Page-locked host memory for tables the device reads through an async copy (the grouped weight-gradient plan): owned by
// the caller, released with mrmt3_host_free.  Not to be called while a stream of this thread is capturing.
extern "C" void* mrmt3_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    mrmt3_set_error("host_alloc: hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}
extern "C" void mrmt3_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// Diagnostics only: how often each kernel family was launched by this process (tests assert that a shape really
// dispatched to the kernel they mean to check).  Relaxed atomics; no effect on any result.
static std::atomic<unsigned long long> g_counts[MRMT3_CNT_N];
void mrmt3_count(int which) {
  if (which >= 0 && which < MRMT3_CNT_N) g_counts[which].fetch_add(1, std::memory_order_relaxed);
}
extern "C" int mrmt3_dispatch_counts(unsigned long long* out, int n, int reset) {
  if (out == nullptr || n < 0) {
    mrmt3_set_error("dispatch_counts: bad arguments");
    return -1;
  }
  for (int i = 0; i < n && i < MRMT3_CNT_N; ++i)
    out[i] = reset ? g_counts[i].exchange(0, std::memory_order_relaxed) : g_counts[i].load(std::memory_order_relaxed);
  for (int i = MRMT3_CNT_N; i < n; ++i) out[i] = 0;
  return MRMT3_CNT_N;
}

// ---- capture hygiene: what a host that captures the step into hipGraphs needs when a capture goes wrong -------------------
// (mrmt3/trainer.py: a failed capture must leave NO participating stream in capture mode and no stale error in the calling
// thread's HIP error slot before anything synchronises — include/mrmt3_hip.h)
extern "C" int mrmt3_stream_capture_status(void* stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipError_t e = hipStreamIsCapturing((hipStream_t)stream, &st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    // an INVALIDATED capture answers the query itself with an error on some runtimes: report it as invalidated
    if (e == hipErrorStreamCaptureInvalidated || e == hipErrorStreamCaptureImplicit) return 2;
    mrmt3_set_error("stream_capture_status: %s", hipGetErrorString(e));
    return MRMT3_ERR_HIP;
  }
  return st == hipStreamCaptureStatusNone ? 0 : st == hipStreamCaptureStatusActive ? 1 : 2;
}

extern "C" int mrmt3_stream_abandon_capture(void* stream) {
  const int before = mrmt3_stream_capture_status(stream);
  if (before > 0) {
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);       // an invalidated capture ends with an error and no graph
    if (e == hipSuccess && g != nullptr) (void)hipGraphDestroy(g);
  }
  (void)hipGetLastError();
  return before;
}

// A stream of the caller's own, outside every framework's stream pool: torch hands `torch.cuda.Stream()` out of a pool of 32
// per priority, round robin — a stream that a failed capture left invalidated for good comes BACK from that pool some
// dozens of streams later (seen in round 6: a fresh trainer's "new" capture stream was born invalidated).  The capture stream
// of the trainer is therefore created and destroyed here.
extern "C" int mrmt3_stream_create(void** stream_out, int priority) {
  MR_CHECK_ARG(stream_out != nullptr, "stream_create: null pointer");
  hipStream_t s = nullptr;
  MR_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
  *stream_out = (void*)s;
  return MRMT3_OK;
}
extern "C" int mrmt3_stream_destroy(void* stream) {
  if (stream == nullptr) return MRMT3_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    mrmt3_set_error("stream_destroy: %s", hipGetErrorString(e));
    return MRMT3_ERR_HIP;
  }
  return MRMT3_OK;
}

extern "C" int mrmt3_runtime_error_pop(char* text, int n) {
  const hipError_t e = hipGetLastError();                                     // returns AND clears the calling thread's error slot
  if (text != nullptr && n > 0) snprintf(text, (size_t)n, "%s", e == hipSuccess ? "" : hipGetErrorName(e));
  return (int)e;
}

// A process that is taken down by abort() inside a runtime library (HIP, RCCL) leaves only the Python frames that faulthandler
// prints.  This writes the NATIVE frames of the aborting thread first (backtrace_symbols_fd: async-signal-safe), then hands the
// signal back to whoever handled it before (faulthandler, the default action).  Opt-in: tests/conftest.py, profiles/tools.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
namespace {
int g_trace_fd = 2;
struct sigaction g_prev_abrt, g_prev_segv;
void abort_trace_handler(int sig, siginfo_t* info, void* uctx) {
  static const char head[] = "\nmrmt3: fatal signal, native frames of the faulting thread:\n";
  (void)!write(g_trace_fd, head, sizeof(head) - 1);
  void* frames[96];
  const int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, g_trace_fd);
  const struct sigaction* prev = sig == SIGABRT ? &g_prev_abrt : &g_prev_segv;
  if (prev->sa_flags & SA_SIGINFO) {
    if (prev->sa_sigaction != nullptr) { prev->sa_sigaction(sig, info, uctx); return; }
  } else if (prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN && prev->sa_handler != nullptr) {
    prev->sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
}  // namespace
extern "C" int mrmt3_abort_trace_install(const char* path) {
  if (path != nullptr && *path != 0) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd < 0) {
      mrmt3_set_error("abort_trace_install: cannot open %s", path);
      return MRMT3_ERR_INVALID_ARG;
    }
    g_trace_fd = fd;
  }
  void* warm[4];
  (void)backtrace(warm, 4);                        // loads libgcc now: not from inside the handler
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = abort_trace_handler;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_RESETHAND;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGABRT, &sa, &g_prev_abrt);
  sigaction(SIGSEGV, &sa, &g_prev_segv);
  return MRMT3_OK;
}
