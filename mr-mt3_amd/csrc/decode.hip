// K12 — greedy autoregressive decode with a self-attention KV cache, one hipGraph replay per token.
//
// Stands for the decoder loop of models/t5.py:267-295 (batched, MT3Net) and
// models/t5_segmem_v2_with_prev.py:273-291 (one segment at a time).  The reference re-runs the
// whole decoder over the growing prefix for every token (no cache, O(L^2) GEMM work) and, in the
// segment-memory model, synchronises with the host once per token (`.item()`, :284).  Here a step
// is 66 small kernels captured once into a hipGraph:
//   per layer  norm+QKV gemv (K/V appended to the cache) -> self-attention over the cache ->
//              O gemv + residual -> norm+Q gemv -> cross-attention over the projected encoder
//              states -> O gemv + residual -> norm + wi gemv + gated-GELU -> wo gemv + residual
//   then       final norm + lm_head gemv -> argmax / EOS bookkeeping / next-token embedding.
// The step index and the finished flags live in device memory, so replays need no host
// interaction; the host polls an 12-byte status block only when it wants to stop early.
// A step is a chain of dependent launches (1.77 us each at best on this runtime), so every kernel is built to be
// short rather than frugal: up to 8 sequences run one weight row x one sequence per wave (batch on gridDim.y;
// the re-read of a weight row by the other sequences is an L2 hit), larger batches (<= 256) multiply 16 rows by
// 16 sequences on the matrix cores with K split over a workgroup; the 45.6 MB (bf16) of per-step weights stay
// resident in the 256 MB Infinity Cache.  mrmt3_decoder_set_prefix feeds memory rows before the start token
// (the V1 segment-memory decode).
#include "common.h"

#define DMODEL 512
#define DEC_MAXB 256
#ifndef DEC_MFMA_ABOVE
#define DEC_MFMA_ABOVE 8   // batches larger than this use the 16-sequence MFMA projections (bf16 weights)
#endif
#define ST_CNT (ST_FLAGS + DEC_MAXB)   // workgroups of the current argmax launch that have finished

template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float v[8]) {
  u32x4 t = *(const u32x4*)p;
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
  v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xFFFF0000u);
  v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xFFFF0000u);
}
// activations entering a projection are rounded to the weights' dtype, as in the engine's bf16 GEMMs
// (identity for the fp32 parity path)
template <typename TW> __device__ __forceinline__ float act_round(float v);
template <> __device__ __forceinline__ float act_round<float>(float v) { return v; }
template <> __device__ __forceinline__ float act_round<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

__device__ __forceinline__ float gelu_new_d(float x) {
  const float c = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * (x * x * x))));
}

// state block: [0] step t, [1] all finished, [2] step at which the last row finished (-1), [4+b] finished[b]
#define ST_T 0
#define ST_ALL 1
#define ST_FIN 2
#define ST_NPRE 3   // number of prefix (memory) positions fed before the start token
#define ST_FLAGS 4

// ---- norm + gemv -------------------------------------------------------------------------------------
// One wave = one weight row (two for the gated FFN) x ONE sequence; the batch is gridDim.y.  Measured on
// MI355X: a decode step is a chain of ~66 dependent launches whose length is set by the slowest wave of
// each, so the lightest possible wave wins — carrying 2 rows or up to 8 sequences per wave (to stream a
// weight row once) cost 255 / 607 us per step at batch 1 / 8 against 235 / 277 this way; the re-read of
// a weight row by the other sequences' waves is an L2 hit.
// Every wave normalises x[b] for itself from registers (8 elements per lane, one wave_sum) — no LDS, no
// workgroup barrier — and requests its weight row before anything else.
// MODE 0: out[b][n] (f32, ld = N)      — cross-attention q, lm_head logits
// MODE 1: fused q|k|v: n < inner -> q scratch; else K / V cache row t of this layer
// MODE 2: gated GELU: rows n and n+N of W ([2N][512]) -> out[b][n] = gelu_new(h0) * h1
#ifndef DEC_WPG
#define DEC_WPG 4   // waves (rows) per workgroup
#endif
template <typename TW, int MODE>
__global__ __launch_bounds__(64 * DEC_WPG) void dec_norm_gemv(const float* __restrict__ x, const float* __restrict__ lnw,
                                                             const TW* __restrict__ W, int N, float eps,
                                                             float* __restrict__ out, TW* __restrict__ kc,
                                                             TW* __restrict__ vc, int inner, size_t cache_bstride,
                                                             const int* __restrict__ state) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * DEC_WPG + wave, b = blockIdx.y;
  if (n >= N) return;
  float w0[8], w1[8], lw[8], xn[8];
  load8<TW>(W + (size_t)n * DMODEL + lane * 8, w0);
  if (MODE == 2) load8<TW>(W + (size_t)(n + N) * DMODEL + lane * 8, w1);
  load8<float>(lnw + lane * 8, lw);
  load8<float>(x + (size_t)b * DMODEL + lane * 8, xn);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss = fmaf(xn[e], xn[e], ss);
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)DMODEL + eps);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = act_round<TW>(lw[e] * (xn[e] * rstd));
    s0 = fmaf(w0[e], v, s0);
    if (MODE == 2) s1 = fmaf(w1[e], v, s1);
  }
  s0 = wave_sum(s0);
  if (MODE == 2) s1 = wave_sum(s1);
  if (lane == 0) {
    if (MODE == 0) out[(size_t)b * N + n] = s0;
    else if (MODE == 2) out[(size_t)b * N + n] = gelu_new_d(s0) * s1;
    else {
      const int t = state[ST_T];
      if (n < inner) out[(size_t)b * inner + n] = s0;
      else if (n < 2 * inner) stf<TW>(kc + b * cache_bstride + (size_t)t * inner + (n - inner), s0);
      else stf<TW>(vc + b * cache_bstride + (size_t)t * inner + (n - 2 * inner), s0);
    }
  }
}

// x[b][n] += sum_k a[b][k] * W[n][k]   (O projections and FFN wo, residual add fused); K <= 1024.
// Weight row, activation vector and the residual value to update are all requested up front.
template <typename TW, int KCH>   // KCH = number of 512-element chunks of K (1 or 2)
__global__ __launch_bounds__(64 * DEC_WPG) void dec_gemv_res(const float* __restrict__ a, const TW* __restrict__ W,
                                                            float* __restrict__ x, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * DEC_WPG + wave, b = blockIdx.y;
  if (n >= N) return;
  float w[KCH][8], av[KCH][8];
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int k0 = c * 512 + lane * 8;
    if (k0 < K) {
      load8<TW>(W + (size_t)n * K + k0, w[c]);
      load8<float>(a + (size_t)b * K + k0, av[c]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { w[c][e] = 0.f; av[c][e] = 0.f; }
    }
  }
  float* xdst = x + (size_t)b * N + n;
  const float xold = (lane == 0) ? *xdst : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(w[c][e], act_round<TW>(av[c][e]), acc);
  acc = wave_sum(acc);
  if (lane == 0) *xdst = xold + acc;
}

// ---- the same two projections for groups of 16 sequences on the matrix cores (bf16 weights, batch > 8) ----
// With one sequence per wave every sequence re-reads every weight row from L2 (45.6 MB x B per step: at 64
// sequences that, not the launch chain, set the step time).  Here a wave multiplies 16 weight rows by 16
// sequences: D[16 rows][16 seq] = W[16][K] . A^T[K][16] as K/32 v_mfma_f32_16x16x32_bf16, operands loaded
// straight from global memory in fragment layout (lane = (row or sequence) & 15, k-group = lane >> 4 holds
// 8 consecutive k).  Same operand precision and the same products as the single-sequence kernels above; only
// the order of the f32 additions differs.
__device__ __forceinline__ bf16x8 pack8(const float v[8]) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (short)f2bf(v[e]);
  return r;
}

// A workgroup = 16 weight rows x 16 sequences; its 4 waves split K (a projection has only 384-2048 rows, so
// 16 rows per WAVE would leave most CUs idle and make each busy one pull hundreds of KB), partial tiles are
// summed through LDS in wave order.
template <int MODE>
__global__ __launch_bounds__(256) void dec_norm_gemm16(const float* __restrict__ x, const float* __restrict__ lnw,
                                                       const bf16_t* __restrict__ W, int N, float eps,
                                                       float* __restrict__ out, bf16_t* __restrict__ kc,
                                                       bf16_t* __restrict__ vc, int inner, size_t cache_bstride,
                                                       const int* __restrict__ state, int B) {
  constexpr int KB = DMODEL / 32 / 4;          // k-blocks of 32 per wave
  __shared__ float ssq[4][16];
  __shared__ float part[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int seq = blockIdx.y * 16 + r;
  const int k0 = wave * (DMODEL / 4) + g * 8;   // this lane's first k
  const bf16_t* wrow = W + (size_t)min(n0 + r, N - 1) * DMODEL + k0;
  bf16x8 a0[KB], a1[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    a0[kb] = *(const bf16x8*)(wrow + kb * 32);
    if (MODE == 2) a1[kb] = *(const bf16x8*)(wrow + (size_t)N * DMODEL + kb * 32);
  }
  const float* xs = x + (size_t)min(seq, B - 1) * DMODEL + k0;
  float xv[KB][8], lw[KB][8];
  float ss = 0.f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    load8<float>(xs + kb * 32, xv[kb]);
    load8<float>(lnw + k0 + kb * 32, lw[kb]);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(xv[kb][e], xv[kb][e], ss);
  }
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  if (g == 0) ssq[wave][r] = ss;
  __syncthreads();
  const float rstd = rsqrtf(((ssq[0][r] + ssq[1][r]) + (ssq[2][r] + ssq[3][r])) / (float)DMODEL + eps);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = lw[kb][e] * (xv[kb][e] * rstd);
    const bf16x8 bfr = pack8(v);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[kb], bfr, acc0, 0, 0, 0);
    if (MODE == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[kb], bfr, acc1, 0, 0, 0);
  }
  // D: column = lane & 15 = sequence, rows 4 g + (0..3); element id = (4 g + rr) * 16 + r
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    part[0][wave][(g * 4 + rr) * 16 + r] = acc0[rr];
    if (MODE == 2) part[1][wave][(g * 4 + rr) * 16 + r] = acc1[rr];
  }
  __syncthreads();
  const int id = threadIdx.x;                   // one output element per thread: row id / 16, sequence id % 16
  const int n = n0 + (id >> 4), sq = blockIdx.y * 16 + (id & 15);
  if (n >= N || sq >= B) return;
  const float s0 = (part[0][0][id] + part[0][1][id]) + (part[0][2][id] + part[0][3][id]);
  if (MODE == 0) out[(size_t)sq * N + n] = s0;
  else if (MODE == 2)
    out[(size_t)sq * N + n] = gelu_new_d(s0) * ((part[1][0][id] + part[1][1][id]) + (part[1][2][id] + part[1][3][id]));
  else {
    const int t = state[ST_T];
    if (n < inner) out[(size_t)sq * inner + n] = s0;
    else if (n < 2 * inner) kc[sq * cache_bstride + (size_t)t * inner + (n - inner)] = f2bf(s0);
    else vc[sq * cache_bstride + (size_t)t * inner + (n - 2 * inner)] = f2bf(s0);
  }
}

template <int KB>   // K / 32 / 4 : k-blocks per wave
__global__ __launch_bounds__(256) void dec_gemm16_res(const float* __restrict__ a, const bf16_t* __restrict__ W,
                                                      float* __restrict__ x, int N, int B) {
  constexpr int K = KB * 32 * 4;
  __shared__ float part[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int seq = blockIdx.y * 16 + r;
  const int k0 = wave * (K / 4) + g * 8;
  const bf16_t* wrow = W + (size_t)min(n0 + r, N - 1) * K + k0;
  const float* as = a + (size_t)min(seq, B - 1) * K + k0;
  const int id = threadIdx.x;
  const int n = n0 + (id >> 4), sq = blockIdx.y * 16 + (id & 15);
  const bool live = n < N && sq < B;
  const float xold = live ? x[(size_t)sq * N + n] : 0.f;
  bf16x8 wf[KB];
  float av[KB][8];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    wf[kb] = *(const bf16x8*)(wrow + kb * 32);
    load8<float>(as + kb * 32, av[kb]);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kb], pack8(av[kb]), acc, 0, 0, 0);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) part[wave][(g * 4 + rr) * 16 + r] = acc[rr];
  __syncthreads();
  if (live) x[(size_t)sq * N + n] = xold + ((part[0][id] + part[1][id]) + (part[2][id] + part[3][id]));
}

// one (head, batch) per workgroup: softmax(q.K^T) V over `len` cached rows (len = t+1 or fixed).
// The kernel is a chain of L2 round trips, so every phase issues all the loads it can before it waits:
// scores: one key per thread (8 x 16-byte loads of the key row), the first batch requested before q is
// staged and each later batch requested before the previous one is consumed.  PV: thread = (32 key
// lanes) x (8 dim-groups of 8), eight predicated 16-byte V loads in flight per thread, 32 independent
// partial sums reduced through LDS.
// (Computing the cross-attention query inside this kernel, 16 rows per wave while the key rows are in
// flight, saved a launch per layer and cost the same time: reverted.)
template <typename TC>
__global__ __launch_bounds__(256) void dec_attn(const float* __restrict__ q, const TC* __restrict__ kb,
                                                const TC* __restrict__ vb, int ld, size_t bstride, int fixed_len,
                                                const int* __restrict__ state, float* __restrict__ o, int inner) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // scores[len] | q[64] | red[8] | part[32][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int len = fixed_len > 0 ? fixed_len : state[ST_T] + 1;
  float* sc = sm;
  float* qs = sm + ((len + 3) & ~3);
  float* red = qs + 64;
  float* part = red + 8;
  const TC* kp = kb + b * bstride + h * 64;
  const TC* vp = vb + b * bstride + h * 64;
  const int nb = (len + 255) >> 8;            // key batches of 256 (block-uniform)
  float kva[8][8], kvb[8][8];
  auto fetch = [&](float (&dst)[8][8], int key) {
    const TC* kr = kp + (size_t)min(key, len - 1) * ld;
#pragma unroll
    for (int d0 = 0; d0 < 8; ++d0) load8<TC>(kr + d0 * 8, dst[d0]);
  };
  fetch(kva, tid);
  if (tid < 64) qs[tid] = q[(size_t)b * inner + h * 64 + tid];
  __syncthreads();
  float mx = -INFINITY;
  auto score = [&](const float (&src)[8][8], int key) {
    float s = 0.f;
#pragma unroll
    for (int d0 = 0; d0 < 8; ++d0)
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(qs[d0 * 8 + e], src[d0][e], s);
    if (key < len) {
      sc[key] = s;
      mx = fmaxf(mx, s);
    }
  };
  for (int it = 0; it < nb; it += 2) {
    const int key = it * 256 + tid;
    if (it + 1 < nb) fetch(kvb, key + 256);
    score(kva, key);
    if (it + 1 < nb) {
      if (it + 2 < nb) fetch(kva, key + 512);
      score(kvb, key + 256);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // V rows of the first PV batch are requested before the exponentials (requesting them at kernel entry,
  // next to the first key batch, measured the same)
  const int dg = tid & 7, kl = tid >> 3;   // 8 dims per thread, 32 key lanes
  float vv[8][8];
#pragma unroll
  for (int u = 0; u < 8; ++u) load8<TC>(vp + (size_t)min(kl + 32 * u, len - 1) * ld + dg * 8, vv[u]);
  float se = 0.f;
  for (int key = tid; key < len; key += 256) {
    const float p = expf(sc[key] - mx);
    sc[key] = p;
    se += p;
  }
  se = wave_sum(se);
  if (lane == 0) red[4 + wave] = se;
  __syncthreads();
  se = (red[4] + red[5]) + (red[6] + red[7]);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int k0 = kl; k0 < len; k0 += 256) {
    float pr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pr[u] = (k0 + 32 * u < len) ? sc[k0 + 32 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pr[u], vv[u][e], acc[e]);
    if (k0 + 256 < len) {
#pragma unroll
      for (int u = 0; u < 8; ++u) load8<TC>(vp + (size_t)min(k0 + 256 + 32 * u, len - 1) * ld + dg * 8, vv[u]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[kl * 64 + dg * 8 + e] = acc[e];
  __syncthreads();
  if (tid < 64) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) r += part[k * 64 + tid];
    o[(size_t)b * inner + h * 64 + tid] = r / se;
  }
}

// argmax + EOS bookkeeping (models/t5.py:286-295) + embedding of the next token.  One wave per sequence,
// 8 sequences per workgroup; the workgroup that finishes last (agent-scope counter) folds the finished
// flags into the "all done" state and advances the step counter — every other workgroup has read it by then.
__device__ __forceinline__ void dec_step_close(int* state, int B, int p, int t, bool token_step) {
  // no __threadfence() (on gfx950: write-back + invalidate of the XCD's L2, which would drop the weights the next
  // step is about to reread): the finished flags are agent-scope atomic stores, complete before the counter is bumped
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int old = __hip_atomic_fetch_add(&state[ST_CNT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old != (int)gridDim.x - 1) return;
  __hip_atomic_store(&state[ST_CNT], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (token_step) {
    int all_done = 1;
    for (int b = 0; b < B; ++b) all_done &= __hip_atomic_load(&state[ST_FLAGS + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (all_done && !state[ST_ALL]) { state[ST_ALL] = 1; state[ST_FIN] = t; }
  }
  state[ST_T] = p + 1;
}

__global__ __launch_bounds__(512) void dec_argmax(const float* __restrict__ logits, int V, int B, int64_t* __restrict__ tokens,
                                                  int tok_ld, const float* __restrict__ embed, const float* __restrict__ pos,
                                                  float* __restrict__ x, int* __restrict__ state, int eos, int pad,
                                                  const float* __restrict__ prefix) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = state[ST_T];            // position just processed
  const int npre = state[ST_NPRE];
  if (p + 1 <= npre) {
    // still inside the prefix: the next input is the next memory row, or the start token right after
    // the last one (models/t5_segmem.py:203-213); this step's logits are discarded
    for (int b = blockIdx.x * 8; b < min(B, (int)blockIdx.x * 8 + 8); ++b) {
      const float* er = (p + 1 < npre) ? prefix + ((size_t)b * npre + p + 1) * DMODEL
                                       : embed + (size_t)tokens[(size_t)b * tok_ld] * DMODEL;
      const float* pr = pos + (size_t)(p + 1) * DMODEL;
      x[b * DMODEL + tid] = er[tid] + pr[tid];          // 512 threads = DMODEL
    }
    __syncthreads();
    if (tid == 0) dec_step_close(state, B, p, 0, false);
    return;
  }
  const int t = p - npre;               // token step
  // first-maximum argmax, EOS bookkeeping, next embedding.  Everything that does not depend on the winner
  // (finished flag, positional row) is requested up front.
  const int b = blockIdx.x * 8 + wave;
  if (b < B) {
    const int was_done = state[ST_FLAGS + b];
    const float* pr = pos + (size_t)(p + 1) * DMODEL;
    float pv[DMODEL / 64];
#pragma unroll
    for (int c = 0; c < DMODEL / 64; ++c) pv[c] = pr[c * 64 + lane];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c0 = lane; c0 < V; c0 += 512) {
      float lv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) lv[u] = (c0 + 64 * u < V) ? logits[(size_t)b * V + c0 + 64 * u] : -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (lv[u] > best) { best = lv[u]; idx = c0 + 64 * u; }   // ascending c per lane: first maximum wins
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(idx, off, 64);
      if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int nxt = was_done ? pad : idx;
    const float* er = embed + (size_t)nxt * DMODEL;
#pragma unroll
    for (int c = 0; c < DMODEL / 64; ++c) x[b * DMODEL + c * 64 + lane] = er[c * 64 + lane] + pv[c];
    if (lane == 0) {
      if (!was_done && nxt == eos) __hip_atomic_store(&state[ST_FLAGS + b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tokens[(size_t)b * tok_ld + t + 1] = nxt;
    }
  }
  __syncthreads();
  if (tid == 0) dec_step_close(state, B, p, t, true);
}

// position 0 becomes the first memory row instead of the start token
__global__ void dec_prefix_kernel(int B, int n_prefix, const float* prefix, const float* pos, float* x, int* state) {
  const int tid = threadIdx.x;
  if (tid == 0) state[ST_NPRE] = n_prefix;
  for (int b = 0; b < B; ++b) {
    const float* er = prefix + (size_t)b * n_prefix * DMODEL;
    x[b * DMODEL + tid] = er[tid] + pos[tid];
    x[b * DMODEL + 256 + tid] = er[256 + tid] + pos[256 + tid];
  }
}

__global__ void dec_begin_kernel(int B, int64_t* tokens, int tok_ld, const float* embed, const float* pos, float* x,
                                 int* state, int start_id) {
  const int tid = threadIdx.x;
  for (int i = tid; i <= ST_CNT; i += (int)blockDim.x) state[i] = (i == ST_FIN) ? -1 : 0;
  for (int b = 0; b < B; ++b) {
    if (tid == 0) tokens[(size_t)b * tok_ld] = start_id;
    x[b * DMODEL + tid] = embed[(size_t)start_id * DMODEL + tid] + pos[tid];
    x[b * DMODEL + 256 + tid] = embed[(size_t)start_id * DMODEL + 256 + tid] + pos[256 + tid];
  }
}

// ------------------------------------------------------------------------------------------------
struct mrmt3_decoder {
  int L, d, H, dff, V, maxB, maxLen, maxEnc, wdt, inner;
  float eps;
  void *kc, *vc;  // [L][maxB][maxLen][inner]
  float *x, *q, *o, *g, *logits;
  int* state;
  // per-batch
  mrmt3_decoder_weights w;
  const void *ln_self[64], *w_qkv[64], *w_o_self[64], *ln_cross[64], *w_q_cross[64], *w_o_cross[64], *ln_ff[64],
      *w_wi[64], *w_wo[64];
  const void* cross_kv;
  const float* prefix;   // [B][n_prefix][d] f32 memory rows fed before the start token (or null)
  int B, encLen, eos, pad;
  int64_t* tokens;
  hipGraph_t graph;
  hipGraphExec_t exec;
  int captured;     // 1 = exec valid for the current (B, encLen, pointers)
  int graph_failed;  // 1 = capture failed once; run with plain launches
};

extern "C" int mrmt3_decoder_create(mrmt3_decoder** out, int n_layers, int d_model, int n_heads, int d_ff, int vocab,
                                    int max_batch, int max_len, int max_enc_len, int w_dtype, float eps) {
  MR_CHECK_ARG(out, "decoder_create: null out");
  *out = nullptr;
  MR_CHECK_ARG(d_model == DMODEL, "decoder_create: kernels are specialised for d_model = 512");
  MR_CHECK_ARG(n_layers > 0 && n_layers <= 64 && max_batch > 0 && max_batch <= DEC_MAXB, "decoder_create: need 1..64 layers, batch <= 256");
  MR_CHECK_ARG(d_ff % 8 == 0 && d_ff <= 1024 && n_heads * 64 <= 512 && vocab > 0 && max_len > 0 && max_enc_len > 0, "decoder_create: need d_ff <= 1024, heads*64 <= 512");
  MR_CHECK_ARG(w_dtype == MRMT3_F32 || w_dtype == MRMT3_BF16, "decoder_create: bad dtype");
  mrmt3_decoder* D = new mrmt3_decoder();
  memset(D, 0, sizeof(*D));
  D->L = n_layers; D->d = d_model; D->H = n_heads; D->dff = d_ff; D->V = vocab; D->maxB = max_batch;
  D->maxLen = max_len; D->maxEnc = max_enc_len; D->wdt = w_dtype; D->inner = n_heads * 64; D->eps = eps;
  const size_t esz = w_dtype == MRMT3_BF16 ? 2 : 4;
  const size_t cache = (size_t)n_layers * max_batch * max_len * D->inner * esz;
  hipError_t e = hipMalloc(&D->kc, cache);
  if (e == hipSuccess) e = hipMalloc(&D->vc, cache);
  if (e == hipSuccess) e = hipMalloc((void**)&D->x, sizeof(float) * max_batch * DMODEL);
  if (e == hipSuccess) e = hipMalloc((void**)&D->q, sizeof(float) * max_batch * D->inner);
  if (e == hipSuccess) e = hipMalloc((void**)&D->o, sizeof(float) * max_batch * D->inner);
  if (e == hipSuccess) e = hipMalloc((void**)&D->g, sizeof(float) * max_batch * d_ff);
  if (e == hipSuccess) e = hipMalloc((void**)&D->logits, sizeof(float) * max_batch * vocab);
  if (e == hipSuccess) e = hipMalloc((void**)&D->state, sizeof(int) * (ST_CNT + 1));
  if (e != hipSuccess) {
    mrmt3_set_error("decoder_create: hipMalloc failed: %s", hipGetErrorString(e));
    mrmt3_decoder_destroy(D);
    return MRMT3_ERR_HIP;
  }
  *out = D;
  return MRMT3_OK;
}

extern "C" void mrmt3_decoder_destroy(mrmt3_decoder* D) {
  if (!D) return;
  if (D->exec) (void)hipGraphExecDestroy(D->exec);
  if (D->graph) (void)hipGraphDestroy(D->graph);
  void* bufs[] = {D->kc, D->vc, D->x, D->q, D->o, D->g, D->logits, D->state};
  for (void* b : bufs) if (b) (void)hipFree(b);
  delete D;
}

extern "C" int mrmt3_decoder_begin(mrmt3_decoder* D, const mrmt3_decoder_weights* w, const void* cross_kv, int batch,
                                   int enc_len, int64_t* tokens_out, int start_id, int eos_id, int pad_id,
                                   void* stream) {
  MR_CHECK_ARG(D && w && cross_kv && tokens_out, "decoder_begin: null pointer");
  MR_CHECK_ARG(batch > 0 && batch <= D->maxB && enc_len > 0 && enc_len <= D->maxEnc, "decoder_begin: batch/enc_len out of range");
  bool same = D->captured && D->B == batch && D->encLen == enc_len && D->cross_kv == cross_kv &&
              D->tokens == tokens_out && D->eos == eos_id && D->pad == pad_id && D->w.embed == w->embed &&
              D->w.lm_head == w->lm_head && D->w.pos == w->pos && D->w.final_ln == w->final_ln;
  for (int l = 0; l < D->L && same; ++l)
    same = D->w_qkv[l] == w->w_qkv[l] && D->w_wi[l] == w->w_wi[l] && D->w_wo[l] == w->w_wo[l] &&
           D->w_o_self[l] == w->w_o_self[l] && D->w_q_cross[l] == w->w_q_cross[l] && D->w_o_cross[l] == w->w_o_cross[l];
  D->w = *w;
  for (int l = 0; l < D->L; ++l) {
    D->ln_self[l] = w->ln_self[l]; D->w_qkv[l] = w->w_qkv[l]; D->w_o_self[l] = w->w_o_self[l];
    D->ln_cross[l] = w->ln_cross[l]; D->w_q_cross[l] = w->w_q_cross[l]; D->w_o_cross[l] = w->w_o_cross[l];
    D->ln_ff[l] = w->ln_ff[l]; D->w_wi[l] = w->w_wi[l]; D->w_wo[l] = w->w_wo[l];
  }
  D->cross_kv = cross_kv; D->B = batch; D->encLen = enc_len; D->tokens = tokens_out; D->eos = eos_id; D->pad = pad_id;
  if (!same) D->captured = 0;
  hipLaunchKernelGGL(dec_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, batch, tokens_out, D->maxLen + 1,
                     (const float*)w->embed, w->pos, D->x, D->state, start_id);
  MR_CHECK_LAUNCH("decoder_begin");
  return MRMT3_OK;
}

extern "C" int mrmt3_decoder_set_prefix(mrmt3_decoder* D, const float* prefix, int n_prefix, void* stream) {
  MR_CHECK_ARG(D && D->tokens, "decoder_set_prefix: call decoder_begin first");
  MR_CHECK_ARG(prefix && n_prefix > 0 && n_prefix < D->maxLen, "decoder_set_prefix: need 0 < n_prefix < max_len rows");
  if (D->prefix != prefix) { D->prefix = prefix; D->captured = 0; }   // pointer is baked into the graph
  hipLaunchKernelGGL(dec_prefix_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, D->B, n_prefix, prefix, D->w.pos,
                     D->x, D->state);
  MR_CHECK_LAUNCH("decoder_set_prefix");
  return MRMT3_OK;
}

template <typename TW>
static int launch_step(mrmt3_decoder* D, hipStream_t s);

// batch > 8, bf16 weights: projections on the matrix cores, 16 sequences per wave
static int launch_step_mfma(mrmt3_decoder* D, hipStream_t s) {
  typedef bf16_t TW;
  const int B = D->B, inner = D->inner, dff = D->dff, V = D->V;
  const size_t cache_b = (size_t)D->maxLen * inner;
  const size_t cache_l = (size_t)D->maxB * cache_b;
  const size_t attn_extra = (size_t)(64 + 8 + 32 * 64) * sizeof(float);
  const size_t attn_shm_self = (size_t)((D->maxLen + 3) & ~3) * sizeof(float) + attn_extra;
  const size_t attn_shm_cross = (size_t)((D->encLen + 3) & ~3) * sizeof(float) + attn_extra;
  const TW* ckv = (const TW*)D->cross_kv;
  const dim3 blk(256);
  auto rows = [&](int n) { return dim3((unsigned)ceil_div(n, 16), (unsigned)ceil_div(B, 16)); };
  for (int l = 0; l < D->L; ++l) {
    TW* kc = (TW*)D->kc + l * cache_l;
    TW* vc = (TW*)D->vc + l * cache_l;
    hipLaunchKernelGGL((dec_norm_gemm16<1>), rows(3 * inner), blk, 0, s, D->x, (const float*)D->ln_self[l],
                       (const TW*)D->w_qkv[l], 3 * inner, D->eps, D->q, kc, vc, inner, cache_b, D->state, B);
    hipLaunchKernelGGL((dec_attn<TW>), dim3(D->H, B), dim3(256), attn_shm_self, s, D->q, (const TW*)kc, (const TW*)vc,
                       inner, cache_b, 0, D->state, D->o, inner);
    hipLaunchKernelGGL((dec_gemm16_res<3>), rows(DMODEL), blk, 0, s, D->o, (const TW*)D->w_o_self[l], D->x, DMODEL, B);
    hipLaunchKernelGGL((dec_norm_gemm16<0>), rows(inner), blk, 0, s, D->x, (const float*)D->ln_cross[l],
                       (const TW*)D->w_q_cross[l], inner, D->eps, D->q, (TW*)nullptr, (TW*)nullptr, inner, (size_t)0,
                       D->state, B);
    const TW* ck = ckv + (size_t)l * B * D->encLen * 2 * inner;
    hipLaunchKernelGGL((dec_attn<TW>), dim3(D->H, B), dim3(256), attn_shm_cross, s, D->q, ck, ck + inner, 2 * inner,
                       (size_t)D->encLen * 2 * inner, D->encLen, D->state, D->o, inner);
    hipLaunchKernelGGL((dec_gemm16_res<3>), rows(DMODEL), blk, 0, s, D->o, (const TW*)D->w_o_cross[l], D->x, DMODEL, B);
    hipLaunchKernelGGL((dec_norm_gemm16<2>), rows(dff), blk, 0, s, D->x, (const float*)D->ln_ff[l],
                       (const TW*)D->w_wi[l], dff, D->eps, D->g, (TW*)nullptr, (TW*)nullptr, inner, (size_t)0, D->state, B);
    hipLaunchKernelGGL((dec_gemm16_res<8>), rows(DMODEL), blk, 0, s, D->g, (const TW*)D->w_wo[l], D->x, DMODEL, B);
  }
  hipLaunchKernelGGL((dec_norm_gemm16<0>), rows(V), blk, 0, s, D->x, D->w.final_ln, (const TW*)D->w.lm_head, V, D->eps,
                     D->logits, (TW*)nullptr, (TW*)nullptr, inner, (size_t)0, D->state, B);
  hipLaunchKernelGGL(dec_argmax, dim3((unsigned)ceil_div(B, 8)), dim3(512), 0, s, D->logits, V, B, D->tokens,
                     D->maxLen + 1, (const float*)D->w.embed, D->w.pos, D->x, D->state, D->eos, D->pad, D->prefix);
  MR_CHECK_LAUNCH("decoder step (mfma)");
  return MRMT3_OK;
}

template <typename TW>
static int launch_step(mrmt3_decoder* D, hipStream_t s) {
  const int B = D->B, inner = D->inner, dff = D->dff, V = D->V;
  const size_t cache_b = (size_t)D->maxLen * inner;             // elements per batch row of a layer's cache
  const size_t cache_l = (size_t)D->maxB * cache_b;             // elements per layer
  const size_t attn_extra = (size_t)(64 + 8 + 32 * 64) * sizeof(float);
  const size_t attn_shm_self = (size_t)((D->maxLen + 3) & ~3) * sizeof(float) + attn_extra;
  const size_t attn_shm_cross = (size_t)((D->encLen + 3) & ~3) * sizeof(float) + attn_extra;
  const TW* ckv = (const TW*)D->cross_kv;
  const dim3 blk(64 * DEC_WPG);
  auto rows = [&](int n) { return dim3((unsigned)ceil_div(n, DEC_WPG), (unsigned)B); };
  for (int l = 0; l < D->L; ++l) {
    TW* kc = (TW*)D->kc + l * cache_l;
    TW* vc = (TW*)D->vc + l * cache_l;
    hipLaunchKernelGGL((dec_norm_gemv<TW, 1>), rows(3 * inner), blk, 0, s, D->x, (const float*)D->ln_self[l],
                       (const TW*)D->w_qkv[l], 3 * inner, D->eps, D->q, kc, vc, inner, cache_b, D->state);
    hipLaunchKernelGGL((dec_attn<TW>), dim3(D->H, B), dim3(256), attn_shm_self, s, D->q, (const TW*)kc, (const TW*)vc,
                       inner, cache_b, 0, D->state, D->o, inner);
    hipLaunchKernelGGL((dec_gemv_res<TW, 1>), rows(DMODEL), blk, 0, s, D->o, (const TW*)D->w_o_self[l], D->x, DMODEL,
                       inner);
    hipLaunchKernelGGL((dec_norm_gemv<TW, 0>), rows(inner), blk, 0, s, D->x, (const float*)D->ln_cross[l],
                       (const TW*)D->w_q_cross[l], inner, D->eps, D->q, (TW*)nullptr, (TW*)nullptr, inner, (size_t)0,
                       D->state);
    const TW* ck = ckv + (size_t)l * B * D->encLen * 2 * inner;
    hipLaunchKernelGGL((dec_attn<TW>), dim3(D->H, B), dim3(256), attn_shm_cross, s, D->q, ck, ck + inner, 2 * inner,
                       (size_t)D->encLen * 2 * inner, D->encLen, D->state, D->o, inner);
    hipLaunchKernelGGL((dec_gemv_res<TW, 1>), rows(DMODEL), blk, 0, s, D->o, (const TW*)D->w_o_cross[l], D->x, DMODEL,
                       inner);
    hipLaunchKernelGGL((dec_norm_gemv<TW, 2>), rows(dff), blk, 0, s, D->x, (const float*)D->ln_ff[l],
                       (const TW*)D->w_wi[l], dff, D->eps, D->g, (TW*)nullptr, (TW*)nullptr, inner, (size_t)0, D->state);
    hipLaunchKernelGGL((dec_gemv_res<TW, 2>), rows(DMODEL), blk, 0, s, D->g, (const TW*)D->w_wo[l], D->x, DMODEL, dff);
  }
  hipLaunchKernelGGL((dec_norm_gemv<TW, 0>), rows(V), blk, 0, s, D->x, D->w.final_ln, (const TW*)D->w.lm_head, V, D->eps,
                     D->logits, (TW*)nullptr, (TW*)nullptr, inner, (size_t)0, D->state);
  hipLaunchKernelGGL(dec_argmax, dim3((unsigned)ceil_div(B, 8)), dim3(512), 0, s, D->logits, V, B, D->tokens, D->maxLen + 1,
                     (const float*)D->w.embed, D->w.pos, D->x, D->state, D->eos, D->pad, D->prefix);
  MR_CHECK_LAUNCH("decoder step");
  return MRMT3_OK;
}

static int step(mrmt3_decoder* D, hipStream_t s) {
  if (D->wdt == MRMT3_BF16 && D->B > DEC_MFMA_ABOVE && D->inner == 384 && D->dff == 1024 && D->d == DMODEL)
    return launch_step_mfma(D, s);
  return D->wdt == MRMT3_BF16 ? launch_step<bf16_t>(D, s) : launch_step<float>(D, s);
}

extern "C" int mrmt3_decoder_run(mrmt3_decoder* D, int n_steps, void* stream) {
  MR_CHECK_ARG(D && D->tokens && n_steps >= 0, "decoder_run: call decoder_begin first");
  hipStream_t s = (hipStream_t)stream;
  if (!D->captured && !D->graph_failed) {
    if (D->exec) { (void)hipGraphExecDestroy(D->exec); D->exec = nullptr; }
    if (D->graph) { (void)hipGraphDestroy(D->graph); D->graph = nullptr; }
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      int rc = step(D, s);
      hipError_t e2 = hipStreamEndCapture(s, &D->graph);
      if (rc == MRMT3_OK && e2 == hipSuccess && D->graph) e = hipGraphInstantiate(&D->exec, D->graph, nullptr, nullptr, 0);
      else e = hipErrorUnknown;
    }
    if (e == hipSuccess && D->exec) D->captured = 1;
    else { D->graph_failed = 1; (void)hipGetLastError(); }
  }
  for (int i = 0; i < n_steps; ++i) {
    if (D->captured) {
      MR_CHECK_HIP(hipGraphLaunch(D->exec, s));
    } else {
      int rc = step(D, s);
      if (rc != MRMT3_OK) return rc;
    }
  }
  return MRMT3_OK;
}

extern "C" int mrmt3_decoder_graph_captured(const mrmt3_decoder* D) { return D ? D->captured : 0; }

extern "C" int mrmt3_decoder_poll(mrmt3_decoder* D, int32_t* state_out_pinned, void* stream) {
  MR_CHECK_ARG(D && state_out_pinned, "decoder_poll: null pointer");
  MR_CHECK_HIP(hipMemcpyAsync(state_out_pinned, D->state, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return MRMT3_OK;
}
