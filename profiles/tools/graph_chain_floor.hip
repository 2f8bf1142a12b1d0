// Floor of a launch chain: N dependent near-empty kernels captured in one hipGraph, replayed R times.
// hipcc --offload-arch=gfx950 -O2 graph_chain_floor.hip -o graph_chain_floor && ./graph_chain_floor 66 1000
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void tick(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 66, reps = argc > 2 ? atoi(argv[2]) : 1000, wgs = argc > 3 ? atoi(argv[3]) : 128;
  int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tick, dim3(wgs), dim3(256), 0, s, d);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  for (int i = 0; i < 20; ++i) hipGraphLaunch(e, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) hipGraphLaunch(e, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("%d kernels x %d WGs per graph: %.1f us per replay, %.2f us per kernel\n", n, wgs, us / reps, us / reps / n);
  return 0;
}
