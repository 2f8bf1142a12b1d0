// Shared device/host helpers for the MR-MT3 gfx950 kernels.  gfx950 (CDNA4) only: 64-wide waves,
// MFMA 16x16x32 bf16, 160 KiB LDS.  No CUDA shims, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mrmt3_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing (never abort / throw across the C ABI) --------------------------------------
void mrmt3_set_error(const char* fmt, ...);
#define MR_CHECK_ARG(cond, ...)                      \
  do {                                               \
    if (!(cond)) {                                   \
      mrmt3_set_error(__VA_ARGS__);                  \
      return MRMT3_ERR_INVALID_ARG;                  \
    }                                                \
  } while (0)
#define MR_CHECK_LAUNCH(name)                                                       \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      mrmt3_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
      return MRMT3_ERR_HIP;                                                         \
    }                                                                               \
  } while (0)
#define MR_CHECK_HIP(expr)                                                          \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      mrmt3_set_error("%s failed: %s", #expr, hipGetErrorString(e_));               \
      return MRMT3_ERR_HIP;                                                         \
    }                                                                               \
  } while (0)

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// ---- wave / block reductions (wave = 64) -----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based RNG for dropout: Philox4x32-10 --------------------------------------------------
// One call yields 4 x 32 random bits for (seed, offset, idx).  Every dropout site passes a distinct
// `offset` stream id so forward and backward regenerate identical masks without storing them.
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long ctr_lo,
                                              unsigned ctr_hi, unsigned out[4]) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr_lo, c1 = (unsigned)(ctr_lo >> 32), c2 = ctr_hi, c3 = 0x9E3779B9u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    unsigned n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// keep-scale for 4 consecutive elements whose first flat index is idx4*4.
// keep iff u32 >= thresh where thresh = p * 2^32;  scale = 1/(1-p)
struct DropCfg {
  unsigned long long seed;
  unsigned stream;   // site id
  unsigned thresh;   // p * 2^32 (0 => dropout off)
  float scale;       // 1/(1-p)
};
__device__ __forceinline__ void drop_mask4(const DropCfg& d, unsigned long long idx4, float m[4]) {
  unsigned r[4];
  philox4x32_10(d.seed, idx4, d.stream, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = (r[i] >= d.thresh) ? d.scale : 0.f;
}
__host__ inline DropCfg make_drop(float p, unsigned long long seed, unsigned stream) {
  DropCfg d;
  d.seed = seed;
  d.stream = stream;
  if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; }
  else { double t = (double)p * 4294967296.0; d.thresh = (unsigned)(t > 4294967295.0 ? 4294967295.0 : t); d.scale = 1.f / (1.f - p); }
  return d;
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
