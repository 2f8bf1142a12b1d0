// Library-level entry points: version and thread-local error text.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "common.h"

static thread_local char g_err[512] = "";

void mrmt3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

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

extern "C" int mrmt3_version(void) { return 107; /* 0.1.7: round 4 (gemm_rows: projection + row kernel in one launch; 106: activation helpers as explicit FMAs; 107: mrmt3_comm_*, mrmt3_allreduce) */ }
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
