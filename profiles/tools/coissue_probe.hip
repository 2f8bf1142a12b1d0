// coissue_probe.hip — do MFMA and VALU from DIFFERENT waves of one SIMD overlap on gfx950?
// One 512-thread workgroup per CU: waves 0-3 run an MFMA loop, waves 4-7 a VALU loop (wave w and w+4 share a SIMD).
// Each role is also timed alone.  overlap => both ~ max(alone); no overlap => ~ sum.
// build: hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// roles: bit0 = MFMA waves active, bit1 = VALU waves active; KIND: 0 fma, 1 exp, 2 cvt+max mix
template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int roles, int it_mfma, int it_valu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    if (roles & 1) {
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 x, y;
#pragma unroll
      for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + lane); y[i] = (short)(0x3f80 + i); }
      for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else if (roles & 2) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 1.0f + 1e-6f * (lane + i);
    const float c1 = 0.999f, c2 = 1e-3f;
    for (int it = 0; it < it_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) a[i] = __builtin_fmaf(a[i], c1, c2);
        else if (KIND == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
        else a[i] = fmaxf(a[i] * c1, a[(i + 1) & 15]);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static float run(int roles, int it_mfma, int it_valu) {
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * 512 * 256));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  probe<KIND><<<256, 512>>>(out, roles, 10, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probe<KIND><<<256, 512>>>(out, roles, it_mfma, it_valu);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipFree(out));
  return ms * 1e3f;
}

int main() {
  const int im = 40000;                    // 320k MFMAs per wave
  const char* names[3] = {"v_fma_f32", "v_exp_f32", "v_mul+v_max"};
  for (int kind = 0; kind < 3; ++kind) {
    const int iv = kind == 1 ? 70000 : 250000;
    float m, v, b;
    if (kind == 0) { m = run<0>(1, im, iv); v = run<0>(2, im, iv); b = run<0>(3, im, iv); }
    else if (kind == 1) { m = run<1>(1, im, iv); v = run<1>(2, im, iv); b = run<1>(3, im, iv); }
    else { m = run<2>(1, im, iv); v = run<2>(2, im, iv); b = run<2>(3, im, iv); }
    printf("%-12s  mfma-waves alone %8.1f us   valu-waves alone %8.1f us   both %8.1f us   (sum %8.1f, max %8.1f)\n",
           names[kind], m, v, b, m + v, m > v ? m : v);
  }
  return 0;
}
