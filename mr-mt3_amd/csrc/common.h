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
void mrmt3_count(int which);   // diagnostics: launches per kernel family (MRMT3_CNT_*), read by mrmt3_dispatch_counts
// ---- dispatch / tuning switches ("knobs") -------------------------------------------------------------------------------
// A knob is an environment variable (MRMT3_*) that picks between kernels or tile shapes for A/B runs and parity tests.  The
// product build reads each one ONCE per process — at the first launch that asks — and never touches the environment on a
// launch path again; mrmt3_set_knob() / mrmt3_reset_knobs() (include/mrmt3_hip.h) override them in-process (tests, tuning).
// MR_KNOB(name, default) is the cached value at its call site.  The -DMRMT3_DIAG build (libmrmt3_hip_diag.so, loaded by
// profiles/tools through MRMT3_TOOL_LIB) re-reads the environment on every call so that one process can A/B a switch.
struct MrKnob {
  const char* name;
  int state;      // 0 not read yet, 1 value cached
  int value;
  int registered;
  MrKnob* next;   // registry (api.hip): every site that has been read, so that an override reaches it
};
int mrmt3_knob_get(MrKnob* k, int dflt);
#define MR_KNOB(NAME, DFLT) ([]() -> int { static MrKnob k_ = {NAME, 0, 0, 0, nullptr}; return mrmt3_knob_get(&k_, (DFLT)); }())

// Kernel DIAGNOSTICS (knock-outs of loads / MFMAs / stores, start skews, per-workgroup time stamps: the experiments DESIGN §0d
// lists as measured and closed) exist only in the -DMRMT3_DIAG build; in the product MR_DIAG(x) is the constant 0 and the
// branches fold away.  mrmt3_diag_env: value of such a switch, announced on stderr once (results of that process are wrong).
#ifdef MRMT3_DIAG
#define MR_DIAG(expr) (expr)
int mrmt3_diag_env(const char* name);
#else
#define MR_DIAG(expr) 0
#endif
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
// a PAIR in one v_cvt_pk_bf16_f32 (the vector convert); converting the halves apart and joining them compiled to two
// conversions + a shift + an or (round 6, seen in the epilogue of profiles/tools/gemm_w4_probe.hip).  Same bits.
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
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

// ---- counter-based mask generator for the element-wise dropout sites ------------------------------------------
// mask(seed, site, element index): forward and backward regenerate the same mask, nothing is stored.  Two 32-bit
// mixes per 4 elements, 16 bits per element: keep iff u16 >= thresh16, thresh16 = round(p * 65536) (p = 0.1 ->
// 6554/65536 = 0.100006) and the keep scale is 65536 / (65536 - thresh16), exactly unbiased.  The mix is the
// attention kernels' one (xor-shift, two full-rate 24-bit multiplies) with a final fold so the low half is as good
// as the high one; checked on row-major [4096 x 512] masks for several (seed, site): keep rate 0.9000, byte
// histograms chi^2 ~ 255, |correlation| <= 0.002 at lags 1-8 along rows and columns, row / column keep counts
// binomial, sites and seeds uncorrelated.  (Round 1 first used Philox4x32-10: 40 quarter-rate multiplies per 4
// elements, which made the mask, not HBM, a third of the GEGLU / norm kernels' time: geglu_fwd 79 -> 105 us.)
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
  x ^= x >> 16; x = __umul24(x, 0x7feb35u); x ^= x >> 15; x = __umul24(x, 0x6ca68bu); x ^= x >> 16;
  return x;
}
struct DropCfg {
  unsigned key;      // seed and site id folded together
  unsigned thresh;   // 16-bit threshold (0 => dropout off)
  float scale;       // 65536 / (65536 - thresh)
  const int* step;   // nullable DEVICE step counter: salts the key in-kernel, so a replayed hipGraph (whose by-value
                     // arguments are frozen at capture) still draws new masks every optimizer step
};
// salt of the optimizer step read from device memory (0 when there is no counter: masks as before)
__device__ __forceinline__ unsigned step_salt(const int* step) {
  return step ? drop_mix((unsigned)(*step) * 0x9E3779B1u + 0x7F4A7C15u) : 0u;
}
#define DROP_STEP(d) (d).key += step_salt((d).step)
// keep-scale for 4 consecutive elements whose first flat index is idx4*4
__device__ __forceinline__ void drop_mask4(const DropCfg& d, unsigned long long idx4, float m[4]) {
  const unsigned c = ((unsigned)idx4 << 1) ^ ((unsigned)(idx4 >> 31) * 0xC2B2AE35u);
  const unsigned h0 = drop_mix(d.key + c * 0x9E3779B1u), h1 = drop_mix(d.key + (c + 1u) * 0x9E3779B1u);
  m[0] = (h0 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
  m[1] = (h0 >> 16) >= d.thresh ? d.scale : 0.f;
  m[2] = (h1 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
  m[3] = (h1 >> 16) >= d.thresh ? d.scale : 0.f;
}
__host__ inline DropCfg make_drop(float p, unsigned long long seed, unsigned stream, const int* step = nullptr) {
  DropCfg d;
  d.step = p > 0.f ? step : nullptr;
  d.key = ((unsigned)seed ^ ((unsigned)(seed >> 32) * 0x9E3779B1u)) + stream * 0x85EBCA6Bu;
  if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; }
  else {
    unsigned t = (unsigned)((double)p * 65536.0 + 0.5);
    if (t < 1) t = 1;
    if (t > 65535) t = 65535;
    d.thresh = t;
    d.scale = 65536.0f / (65536.0f - (float)t);
  }
  return d;
}

// tanh(u) = 1 - 2 / (exp(2u) + 1): one v_exp_f32 and one v_rcp_f32 instead of libm's branchy tanhf (~3x the
// instructions; with the activations served from the Infinity Cache the GEGLU kernels were as much VALU as memory).
// Saturates cleanly (exp -> inf gives 1, exp -> 0 gives -1); absolute error <= 2e-7.
// Every a * b + c of the activation helpers is an EXPLICIT fmaf and contraction is off around them (as it is at the top of
// rowops.hip / gemm_rows.hip): the fused kernels promise the SAME BITS as the element-wise kernels they replace, and which
// a * b + c becomes an FMA is otherwise the compiler's choice per call site — it differed between two files as soon as the
// build flags changed.  Written out, the helpers are also shorter than what either setting gave: gelu_new 9 vector
// instructions + v_exp + v_rcp (contraction off, no fmaf: 13 + 2), gelu_new with its derivative 17 + 2 (was 29 + 2); the
// GEGLU epilogues are vector-issue time that nothing hides (profiles/r04_geglu_store_probe.txt).
__device__ __forceinline__ float fast_tanh(float u) {
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(u * 2.885390081777927f);   // exp(2u)
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
__device__ __forceinline__ float gelu_new_f(float x) {
#pragma clang fp contract(off)
  const float c = 0.7978845608028654f;  // sqrt(2/pi)
  const float t = fast_tanh(c * fmaf(0.044715f, x * x * x, x));
  const float hx = 0.5f * x;
  return fmaf(hx, t, hx);                                            // 0.5 x (1 + t)
}

// gelu_new and its derivative (HF NewGELUActivation; the GEGLU backward, element-wise kernel and GEMM epilogue alike)
__device__ __forceinline__ void gelu_new_fd(float x, float* f, float* d) {
#pragma clang fp contract(off)
  const float c = 0.7978845608028654f;
  const float x2 = x * x;
  const float t = fast_tanh(c * fmaf(0.044715f, x2 * x, x));
  const float hx = 0.5f * x;
  *f = fmaf(hx, t, hx);
  // d = 0.5 (1 + t) + 0.5 x (1 - t^2) c (1 + 3 * 0.044715 x^2)
  const float p = fmaf(0.5f, t, 0.5f), q = fmaf(-t, t, 1.0f), w = fmaf(3.0f * 0.044715f, x2, 1.0f);
  *d = fmaf(hx * q * c, w, p);
}

// norm-weight gradient: per-workgroup partial rows are summed by DW_CHUNKS row chunks (rowops.hip: dw_reduce_body);
// the workspace of a norm site is [partial rows | DW_CHUNKS chunk sums][cols] f32 + one arrival counter per 64 columns
#define DW_CHUNKS 16

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
