// issue_probe.hip — measures what the attention kernels' inner loop is made of on gfx950:
//   * shader clock under VALU / MFMA load (s_memtime ticks vs wall clock)
//   * issue cost of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_mul_lo_u32, v_cvt_pk_bf16_f32
//   * v_mfma_f32_16x16x32_bf16 back-to-back (independent accumulators)
//   * MFMA + VALU interleaved in ONE wave, and MFMA wave + VALU wave sharing a SIMD
// build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip ; run: ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE: 0 fma, 1 pk_fma, 2 exp, 3 mul_lo, 4 cvt_pk, 5 mfma, 6 mfma+fma interleaved (1:4), 7 waves alternate (even: mfma, odd: fma),
//       8 mfma+exp interleaved (1:1), 9 mfma+fma interleaved (1:8)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* ticks, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + 1e-6f * (lane + i);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + lane); y[i] = (short)(0x3f80 + i); }
  unsigned u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) u[i] = lane * 2654435761u + i;
  const float c1 = 0.999f, c2 = 1e-3f;
  const long long t0 = __builtin_readcyclecounter();
  const bool mf_wave = (MODE == 7) ? ((blockIdx.y + wave / 4) & 1) == 0 : true;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || (MODE == 7 && !mf_wave)) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], c1, c2);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f32x2 v = {a[i], a[i + 1]};
          v = __builtin_elementwise_fma(v, f32x2{c1, c1}, f32x2{c2, c2});
          a[i] = v.x; a[i + 1] = v.y;
        }
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    } else if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = (u[i] ^ (u[i] >> 15)) * 0x7feb352du;
    } else if (MODE == 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          bf2 b = __builtin_convertvector(f32x2{a[i], a[i + 1]}, bf2);
          unsigned w = __builtin_bit_cast(unsigned, b);
          a[i] = __uint_as_float(w | 0x3f000000u); a[i + 1] = __uint_as_float((w << 16) | 0x3f00u);
        }
    } else if (MODE == 5 || (MODE == 7 && mf_wave)) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i], 0, 0, 0);
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i & 7], 0, 0, 0);
        a[(4 * i) & 15] = __builtin_fmaf(a[(4 * i) & 15], c1, c2);
        a[(4 * i + 1) & 15] = __builtin_fmaf(a[(4 * i + 1) & 15], c1, c2);
        a[(4 * i + 2) & 15] = __builtin_fmaf(a[(4 * i + 2) & 15], c1, c2);
        a[(4 * i + 3) & 15] = __builtin_fmaf(a[(4 * i + 3) & 15], c1, c2);
      }
    } else if (MODE == 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i & 7], 0, 0, 0);
        a[i] = __builtin_amdgcn_exp2f(a[i]);
      }
    } else if (MODE == 9) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[(8 * i + j) & 15] = __builtin_fmaf(a[(8 * i + j) & 15], c1, c2);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) ticks[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, double n_valu_per_iter, double n_mfma_per_iter) {
  const int iters = 20000;
  float* out; long long* ticks;
  const int blocks = 256 * waves_per_simd;       // 256-thread workgroup = one wave per SIMD of a CU
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  CHECK(hipMalloc(&ticks, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  dim3 grid(256, waves_per_simd);
  probe<MODE><<<grid, 256>>>(out, ticks, 100);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probe<MODE><<<grid, 256>>>(out, ticks, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long t; CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
  const double sec = ms * 1e-3;
  printf("%-34s waves/SIMD=%d  %8.1f us  memtime %6.1f MHz", name, waves_per_simd, sec * 1e6, t / sec / 1e6);
  if (n_valu_per_iter > 0) printf("  %6.2f ns/valu-instr/SIMD", sec / (iters * n_valu_per_iter * waves_per_simd) * 1e9);
  if (n_mfma_per_iter > 0) printf("  %6.2f ns/mfma/SIMD (%.0f TF chip)", sec / (iters * n_mfma_per_iter * waves_per_simd) * 1e9,
                                  16384.0 * iters * n_mfma_per_iter * waves_per_simd * 1024 / sec / 1e12);
  printf("\n");
  CHECK(hipFree(out)); CHECK(hipFree(ticks));
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_fma_f32 x64", w, 64, 0);
    run<1>("v_pk_fma_f32 x32", w, 32, 0);
    run<2>("v_exp_f32 x64", w, 64, 0);
    run<3>("(v_lshr+v_xor+v_mul_lo_u32) x64", w, 64, 0);
    run<4>("v_cvt_pk_bf16_f32 x32 (+2 bitops)", w, 32, 0);
    run<5>("mfma 16x16x32 bf16 x16", w, 0, 16);
    run<6>("mfma x16 + fma x64 interleaved", w, 64, 16);
    run<9>("mfma x8 + fma x64 interleaved", w, 64, 8);
    run<8>("mfma x16 + exp x16 interleaved", w, 16, 16);
  }
  run<7>("2 waves/SIMD: one mfma x16, one fma x64", 2, 32, 8);
  return 0;
}
