// l2_persistence_probe.hip — does data read by one kernel stay in the XCD L2s for the NEXT kernel on the same stream?
// (decides whether a decode-step kernel can prefetch the next kernel's weights.)
// Kernel `reader` sums a 1 MiB slice-per-workgroup buffer with the same workgroup->slice mapping in every launch.
//   A: reader(W0) then reader(W0)            second launch: L2-warm if L2 survives the kernel boundary
//   B: reader(Wi) then reader(W0), Wi != W0  second launch: W0 last touched long ago -> from Infinity Cache / HBM
//   C: same as B but W0 was touched just before Wi (Infinity-Cache-warm, L2 holds Wi)
// build: hipcc --offload-arch=gfx950 -O3 -o l2_probe l2_persistence_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// out[wg] = shader-clock ticks from issuing this workgroup's loads to having the data (wave 0's view)
__global__ __launch_bounds__(256) void reader(const uint4* __restrict__ w, unsigned* __restrict__ out, int vec_per_wg) {
  const uint4* p = w + (size_t)blockIdx.x * vec_per_wg;
  unsigned acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = threadIdx.x; i < vec_per_wg; i += 256) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc) : "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)(t1 - t0) + (acc == 0x12345679u);
}

int main() {
  const int n_wg = 256, vec_per_wg = 256;                 // 256 WGs x 4 KiB = 1 MiB per buffer (a decode GEMV's weights)
  const size_t bytes = (size_t)n_wg * vec_per_wg * 16;
  const int n_buf = 600;                                  // 600 MiB rotating set > 256 MiB Infinity Cache
  std::vector<uint4*> bufs(n_buf);
  for (auto& b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 1, bytes)); }
  unsigned* out; CHECK(hipMalloc(&out, 4 * n_wg));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timed = [&](uint4* first, uint4* second, bool touch_second_before) {
    float tot = 0;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
      for (int i = 1; i < n_buf; ++i) reader<<<n_wg, 256>>>(bufs[i], out, vec_per_wg);   // evict everything
      if (touch_second_before) reader<<<n_wg, 256>>>(second, out, vec_per_wg);
      reader<<<n_wg, 256>>>(first, out, vec_per_wg);
      CHECK(hipEventRecord(e0));
      reader<<<n_wg, 256>>>(second, out, vec_per_wg);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      std::vector<unsigned> h(n_wg);
      CHECK(hipMemcpy(h.data(), out, 4 * n_wg, hipMemcpyDeviceToHost));
      double s = 0; for (unsigned v : h) s += v;
      tot += (float)(s / n_wg);
    }
    return tot / reps;
  };
  printf("A second launch re-reads the SAME 1 MiB as the launch before it : %7.0f ticks\n", timed(bufs[0], bufs[0], false));
  printf("B second launch reads 1 MiB evicted from every cache            : %7.0f ticks\n", timed(bufs[1], bufs[0], false));
  printf("C second launch reads 1 MiB touched two launches ago (IC-warm)  : %7.0f ticks\n", timed(bufs[1], bufs[0], true));
  return 0;
}
