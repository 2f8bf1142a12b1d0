// K3 / K7 / K8 / K9 / K11 — the HBM-bound row and element-wise kernels of the T5 block:
// fused residual-add + dropout + T5LayerNorm (RMS), gated-GELU, embedding gather / scatter-add,
// sinusoid add, cross-entropy, AdamW, cast / transpose helpers.
//
// All of them stream their rows once with 16-byte accesses (float4 / 8 x bf16 per lane), keep the
// row statistics in fp32 and reduce with 64-lane wave shuffles.  Reference lines are cited at each
// entry point.
#include "common.h"

// no implicit FMA formation in this file: gemm_rows.hip restates these kernels' arithmetic and must land on the same bits
#pragma clang fp contract(off)

template <typename T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
  f32x4 t = *(const f32x4*)p;
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
  u32x2 t = *(const u32x2*)p;
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
  *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
  *(u32x2*)p = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}

// streaming (non-temporal) forms: for operands a kernel reads once and nobody reads again soon (the saved activations of
// the backward), and for outputs whose reader is several kernels away (the residual stream) — they then do not push
// the tensors the NEXT kernel wants out of the 256 MB Infinity Cache
template <typename T> __device__ __forceinline__ void load4_nt(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4_nt<float>(const float* p, float v[4]) {
  const f32x4 t = __builtin_nontemporal_load((const f32x4*)p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4_nt<bf16_t>(const bf16_t* p, float v[4]) {
  const u32x2 t = __builtin_nontemporal_load((const u32x2*)p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
}
template <typename T> __device__ __forceinline__ void store4_nt(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4_nt<float>(float* p, const float v[4]) {
  __builtin_nontemporal_store((f32x4{v[0], v[1], v[2], v[3]}), (f32x4*)p);
}
template <> __device__ __forceinline__ void store4_nt<bf16_t>(bf16_t* p, const float v[4]) {
  __builtin_nontemporal_store((u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}), (u32x2*)p);
}

// ------------------------------------------------------------------------------------------------
// fused add + dropout + RMS norm, forward.   one wave per row, 4 rows per workgroup
// HF T5LayerNorm: w * (x * rsqrt(mean(x^2) + eps)), fp32 statistics.
// ------------------------------------------------------------------------------------------------
#define NORM_MAXV 8  // float4 groups per lane: cols <= 64*4*8 = 2048

template <typename TY, typename TN, int NV>
__global__ __launch_bounds__(256) void add_rmsnorm_fwd_kernel(const float* __restrict__ x0, const TY* __restrict__ y,
                                                              const float* __restrict__ w, float eps,
                                                              float* __restrict__ x1, TN* __restrict__ xn,
                                                              float* __restrict__ rstd_out, int rows, int cols,
                                                              DropCfg dy, DropCfg dout, int out_drop, int cache_mode) {
  DROP_STEP(dy); DROP_STEP(dout);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int nv = NV;  // float4 groups per lane (cols = NV*256)
  float v[NV][4];
  float ss = 0.f;
  const size_t base = (size_t)row * cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nv) {
      const int col = i * 256 + lane * 4;
      if (cache_mode & 2) load4_nt<float>(x0 + base + col, v[i]);
      else load4<float>(x0 + base + col, v[i]);
      if (y != nullptr) {
        float yv[4];
        load4<TY>(y + base + col, yv);
        if (dy.thresh) {
          float m[4];
          drop_mask4(dy, (base + col) >> 2, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) yv[e] *= m[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] += yv[e];
      }
      if (x1 != nullptr) {
        if (cache_mode & 1) store4_nt<float>(x1 + base + col, v[i]);
        else store4<float>(x1 + base + col, v[i]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ss = fmaf(v[i][e], v[i][e], ss);
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)cols + eps);
  if (lane == 0 && rstd_out != nullptr) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nv) {
      const int col = i * 256 + lane * 4;
      float wv[4], o[4];
      load4<float>(w + col, wv);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = wv[e] * (v[i][e] * rstd);
      if (out_drop && dout.thresh) {
        float m[4];
        drop_mask4(dout, (base + col) >> 2, m);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= m[e];
      }
      store4<TN>(xn + base + col, o);
    }
  }
}

extern "C" int mrmt3_add_rmsnorm_fwd(const float* x0, const void* y, int y_dtype, const float* w, float eps,
                                     float* x1, void* xn, int xn_dtype, float* rstd, int rows, int cols,
                                     float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_y, uint32_t stream_out,
                                     int out_drop, void* stream) {
  MR_CHECK_ARG(x0 && w && xn, "add_rmsnorm_fwd: null pointer");
  MR_CHECK_ARG(rows > 0 && (cols == 256 || cols == 512 || cols == 1024 || cols == 2048),
               "add_rmsnorm_fwd: cols must be 256, 512, 1024 or 2048");
  DropCfg dy = make_drop(p_drop, seed, stream_y, step_dev), dn = make_drop(p_drop, seed, stream_out, step_dev);
  dim3 grid((unsigned)ceil_div(rows, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  // bit 0: streaming store of the residual stream x1 (its reader is the next norm, three kernels on), bit 1: streaming
  // load of x0.  64 segments, same box, three alternations: 0 26.00 ms, 1 25.81, 2 25.97, 3 25.79 (MRMT3_NORM_NT)
  const int cache_mode = MR_KNOB("MRMT3_NORM_NT", 3);
#define LAUNCH2(TY, TN, NV)                                                                                 \
  hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<TY, TN, NV>), grid, block, 0, s, x0, (const TY*)y, w, eps, x1, \
                     (TN*)xn, rstd, rows, cols, dy, dn, out_drop, cache_mode)
#define LAUNCH(TY, TN)                                                          \
  do {                                                                          \
    if (cols == 512) LAUNCH2(TY, TN, 2);                                        \
    else if (cols == 256) LAUNCH2(TY, TN, 1);                                   \
    else if (cols == 1024) LAUNCH2(TY, TN, 4);                                  \
    else LAUNCH2(TY, TN, 8);                                                    \
  } while (0)
  if (y_dtype == MRMT3_BF16 && xn_dtype == MRMT3_BF16) LAUNCH(bf16_t, bf16_t);
  else if (y_dtype == MRMT3_F32 && xn_dtype == MRMT3_BF16) LAUNCH(float, bf16_t);
  else if (y_dtype == MRMT3_BF16 && xn_dtype == MRMT3_F32) LAUNCH(bf16_t, float);
  else LAUNCH(float, float);
#undef LAUNCH
#undef LAUNCH2
  MR_CHECK_LAUNCH("add_rmsnorm_fwd");
  return MRMT3_OK;
}

// backward: each workgroup owns NB_ROWS consecutive rows (wave w takes rows w, w+4, ...), keeps the
// per-column dw partial sums in registers and issues one f32 atomic per column at the end.
// rows per workgroup: 32 for the tall decoder shapes (2048 workgroups at 65536 rows), fewer for shorter inputs so that
// a launch still has >= ~2048 workgroups (12 segments per GPU: 12288 rows ran 384 workgroups at 3.1 TB/s of 6-7)
static inline int nb_rows(int rows) {
  int r = 32;
  while (r > 4 && rows / r < 2048) r >>= 1;
  return r;
}
template <int NV, typename TG, typename TRI, typename TRO>
__global__ __launch_bounds__(256) void add_rmsnorm_bwd_kernel(const TG* __restrict__ dxn, const TRI* __restrict__ dres,
                                                              const float* __restrict__ x1, const float* __restrict__ rstd_in,
                                                              const float* __restrict__ w, TRO* __restrict__ dx1,
                                                              bf16_t* __restrict__ dy, float* __restrict__ dw_part, int rows,
                                                              int cols, DropCfg ddy, DropCfg dout, int out_drop,
                                                              int* __restrict__ dw_counters, int nb_rows_, int cache_mode) {
  DROP_STEP(ddy); DROP_STEP(dout);
  __shared__ float red[4 * 256 * NV];  // [wave][col]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (dw_counters != nullptr && blockIdx.x == 0 && (int)threadIdx.x < (cols + 63) / 64) dw_counters[threadIdx.x] = 0;
  constexpr int nv = NV;
  float dwp[NV][4];
  float wv[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dwp[i][e] = 0.f;
    if (i < nv) load4<float>(w + i * 256 + lane * 4, wv[i]);
  }
  const int row_end = min(rows, (int)(blockIdx.x + 1) * nb_rows_);
  for (int row = blockIdx.x * nb_rows_ + wave; row < row_end; row += 4) {
    const size_t base = (size_t)row * cols;
    const float rstd = rstd_in[row];
    float g[NV][4], xh[NV][4], rr[NV][4];
    float dot = 0.f;
    // the residual gradient is requested together with the other operands: one memory round trip per row, not two
    if (dres != nullptr) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < nv) {
          if (cache_mode & 4) load4_nt<TRI>(dres + base + i * 256 + lane * 4, rr[i]);
          else load4<TRI>(dres + base + i * 256 + lane * 4, rr[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nv) {
        const int col = i * 256 + lane * 4;
        load4<TG>(dxn + base + col, g[i]);
        if (out_drop && dout.thresh) {
          float m[4];
          drop_mask4(dout, (base + col) >> 2, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) g[i][e] *= m[e];
        }
        if (cache_mode & 1) load4_nt<float>(x1 + base + col, xh[i]);
        else load4<float>(x1 + base + col, xh[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] *= rstd;
          dwp[i][e] = fmaf(g[i][e], xh[i][e], dwp[i][e]);
          g[i][e] *= wv[i][e];
          dot = fmaf(g[i][e], xh[i][e], dot);
        }
      }
    }
    dot = wave_sum(dot) / (float)cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nv) {
        const int col = i * 256 + lane * 4;
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = rstd * fmaf(-xh[i][e], dot, g[i][e]);
        if (dres != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] += rr[i][e];
        }
        if (cache_mode & 2) store4_nt<TRO>(dx1 + base + col, d);
        else store4<TRO>(dx1 + base + col, d);
        if (dy != nullptr) {
          if (ddy.thresh) {
            float m[4];
            drop_mask4(ddy, (base + col) >> 2, m);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] *= m[e];
          }
          store4<bf16_t>(dy + base + col, d);
        }
      }
    }
  }
  // reduce dw partials across the 4 waves, then one atomic per column
  float* redf = &red[0];
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) redf[wave * cols + i * 256 + lane * 4 + e] = dwp[i][e];
    }
  __syncthreads();
  // per-workgroup partial -> workspace row (no atomics: 2048 workgroups hammering the same 512
  // addresses ran at the contended-atomic rate); dw_reduce_kernel sums the rows in a fixed order
  if (dw_part != nullptr)
    for (int c = threadIdx.x; c < cols; c += 256)
      dw_part[(size_t)blockIdx.x * cols + c] = redf[c] + redf[cols + c] + redf[2 * cols + c] + redf[3 * cols + c];
}

// dw[c] += sum over partial rows.  grid = (cols/64, DW_CHUNKS row chunks); a thread = (column, 1 of 4 row
// groups) streams its rows with independent loads, the workgroup combines through LDS and leaves its 64 sums in
// scratch[chunk]; the chunk workgroup that finishes last (agent-scope counter, zeroed by the norm-backward
// kernel before) adds the DW_CHUNKS partials in chunk order: no float atomics, bitwise reproducible.
__device__ __forceinline__ void dw_reduce_body(const float* __restrict__ part, float* __restrict__ dw, int n_part, int cols,
                                               float* __restrict__ scratch, int* __restrict__ counters) {
  // workgroup = 64 columns x one of DW_CHUNKS row chunks.  A thread = 4 columns (one 16-byte load per row) x 1 of 16
  // row groups, its rows requested eight at a time: the kernel is a handful of wide, independent loads per thread
  // instead of a long chain of 4-byte ones (550 -> 60 us for the 42 norm sites of a training step).
  __shared__ float red[16][64];
  __shared__ int last;
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c4 = blockIdx.x * 64 + q * 4;
  const int per = (n_part + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(n_part, r0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 < cols) {
    int i = r0 + g;
    for (; i + 7 * 16 < r1; i += 8 * 16) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(part + (size_t)(i + u * 16) * cols + c4);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < r1; i += 16) acc += *(const f32x4*)(part + (size_t)i * cols + c4);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[g][q * 4 + e] = acc[e];
  __syncthreads();
  const int c = blockIdx.x * 64 + (int)threadIdx.x;
  if (threadIdx.x < 64 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    __hip_atomic_store(&scratch[(size_t)blockIdx.y * cols + c], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // hand-off to the last-arriving chunk without __threadfence(): on gfx950 that is a write-back + invalidate of the
  // XCD's whole L2 per workgroup (5376 of them in the batched launch: 500-900 us for 168 MB).  Agent-scope relaxed
  // atomics (sc1: written through / read around the L2 per instruction), stores complete before the barrier.
  // This is NOT a release/acquire pair in the HIP memory model: it leans on gfx950 behaviour — an
  // acknowledged sc1 store (s_waitcnt vmcnt(0)) is visible to every later sc1 load of the device, and the inline-asm
  // wait is a compiler barrier — hence the architecture check below; profiles/tools/norm_dw_stress.py is the stress
  // test (900 iterations x 20 sites, alone and beside a second process: every result bit-identical).
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "dw_reduce_body's fence-free hand-off is validated on gfx950 only"
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
  __syncthreads();
  if (!last) return;
  if (threadIdx.x < 64 && c < cols) {
    float a = dw[c];
    for (int k = 0; k < (int)gridDim.y; ++k)
      a += __hip_atomic_load(&scratch[(size_t)k * cols + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dw[c] = a;
  }
  if (threadIdx.x == 0) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                        int n_part, int cols, float* __restrict__ scratch,
                                                        int* __restrict__ counters) {
  dw_reduce_body(part, dw, n_part, cols, scratch, counters);
}
// the same reduction for several norm sites in one launch (blockIdx.z = site): the engine leaves each site's partial
// rows in a workspace of its own during backward and sums them all when a gradient bucket is about to be sent
__global__ __launch_bounds__(256) void dw_reduce_sites_kernel(const unsigned long long* __restrict__ ws, const unsigned long long* __restrict__ dws,
                                                              const int* __restrict__ n_parts, int cols) {
  const int site = blockIdx.z, n_part = n_parts[site];
  float* part = (float*)ws[site];
  float* scratch = part + (size_t)n_part * cols;
  dw_reduce_body(part, (float*)dws[site], n_part, cols, scratch, (int*)(scratch + (size_t)DW_CHUNKS * cols));
}

extern "C" int mrmt3_add_rmsnorm_bwd_partial_rows(int rows) { return ceil_div(rows, nb_rows(rows)); }

extern "C" int mrmt3_norm_dw_reduce(const void* workspaces, const void* dws, const int* partial_rows, int n_sites, int cols,
                                    void* stream) {
  MR_CHECK_ARG(workspaces && dws && partial_rows, "norm_dw_reduce: null pointer");
  MR_CHECK_ARG(n_sites > 0 && n_sites <= 65535 && cols > 0, "norm_dw_reduce: bad sizes");
  hipLaunchKernelGGL(dw_reduce_sites_kernel, dim3((unsigned)ceil_div(cols, 64), DW_CHUNKS, (unsigned)n_sites), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned long long*)workspaces, (const unsigned long long*)dws, partial_rows, cols);
  MR_CHECK_LAUNCH("norm_dw_reduce");
  return MRMT3_OK;
}

extern "C" size_t mrmt3_add_rmsnorm_bwd_workspace_bytes(int rows, int cols) {
  // per-workgroup partial rows | DW_CHUNKS chunk sums | one arrival counter per 64 columns
  return ((size_t)ceil_div(rows, nb_rows(rows)) + DW_CHUNKS) * cols * sizeof(float) + (size_t)ceil_div(cols, 64) * sizeof(int);
}

extern "C" int mrmt3_add_rmsnorm_bwd(const void* dxn, int dxn_dtype, const void* dres, int dres_dtype, const float* x1,
                                     const float* rstd, const float* w, void* dx1, int dx1_dtype, void* dy_bf16,
                                     float* dw, int rows, int cols,
                                     float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_y, uint32_t stream_out,
                                     int out_drop, void* workspace, size_t workspace_bytes, void* stream) {
  MR_CHECK_ARG(dw == nullptr || workspace, "add_rmsnorm_bwd: dw needs a workspace");
  MR_CHECK_ARG(workspace == nullptr || workspace_bytes >= mrmt3_add_rmsnorm_bwd_workspace_bytes(rows, cols),
               "add_rmsnorm_bwd: workspace too small");
  // workspace without dw: the partial rows are left for mrmt3_norm_dw_reduce (deferred, batched over sites)
  float* dw_part = (float*)workspace;
  const int nbr = nb_rows(rows);
  float* dw_scratch = workspace ? dw_part + (size_t)ceil_div(rows, nbr) * cols : nullptr;
  int* dw_counters = workspace ? (int*)(dw_scratch + (size_t)DW_CHUNKS * cols) : nullptr;
  MR_CHECK_ARG(dxn && x1 && rstd && w && dx1, "add_rmsnorm_bwd: null pointer");
  MR_CHECK_ARG(rows > 0 && (cols == 256 || cols == 512 || cols == 1024 || cols == 2048),
               "add_rmsnorm_bwd: cols must be 256, 512, 1024 or 2048");
  DropCfg dy = make_drop(p_drop, seed, stream_y, step_dev), dn = make_drop(p_drop, seed, stream_out, step_dev);
  // bit 0: streaming load of x1 (the saved residual stream: read once), bit 1: streaming store of dx1 (the residual
  // gradient: its reader is the next norm backward), bit 2: streaming load of dres (MRMT3_NORMB_NT).  Measured: none of
  // them moves the 64-segment step (25.61-25.80 ms against 25.61-25.68, two alternations) — off.
  const int cache_mode = MR_KNOB("MRMT3_NORMB_NT", 0);
#define LAUNCH3(NV, TG, TRI, TRO)                                                                                 \
  hipLaunchKernelGGL((add_rmsnorm_bwd_kernel<NV, TG, TRI, TRO>), dim3((unsigned)ceil_div(rows, nbr)), dim3(256), 0, \
                     (hipStream_t)stream, (const TG*)dxn, (const TRI*)dres, x1, rstd, w, (TRO*)dx1, (bf16_t*)dy_bf16,  \
                     dw_part, rows, cols, dy, dn, out_drop, dw_counters, nbr, cache_mode)
#define LAUNCH2(NV, TG) LAUNCH3(NV, TG, float, float)
#define LAUNCH(NV)                                       \
  do {                                                   \
    if (dxn_dtype == MRMT3_BF16) LAUNCH2(NV, bf16_t);    \
    else LAUNCH2(NV, float);                             \
  } while (0)
  const bool ri16 = dres != nullptr && dres_dtype == MRMT3_BF16, ro16 = dx1_dtype == MRMT3_BF16;
  if (ri16 || ro16) {
    // bf16 residual-gradient stream (the engine's bf16 path): model width 512 only
    MR_CHECK_ARG(cols == 512, "add_rmsnorm_bwd: a bf16 residual gradient needs cols == 512");
    if (dxn_dtype == MRMT3_BF16) {
      if (ri16 && ro16) LAUNCH3(2, bf16_t, bf16_t, bf16_t);
      else if (ri16) LAUNCH3(2, bf16_t, bf16_t, float);
      else LAUNCH3(2, bf16_t, float, bf16_t);
    } else {
      if (ri16 && ro16) LAUNCH3(2, float, bf16_t, bf16_t);
      else if (ri16) LAUNCH3(2, float, bf16_t, float);
      else LAUNCH3(2, float, float, bf16_t);
    }
  } else if (cols == 512) LAUNCH(2);
  else if (cols == 256) LAUNCH(1);
  else if (cols == 1024) LAUNCH(4);
  else LAUNCH(8);
#undef LAUNCH
#undef LAUNCH2
#undef LAUNCH3
  MR_CHECK_LAUNCH("add_rmsnorm_bwd");
  if (dw) {
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)ceil_div(cols, 64), DW_CHUNKS), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dw_part, dw, ceil_div(rows, nbr), cols, dw_scratch, dw_counters);
    MR_CHECK_LAUNCH("add_rmsnorm_bwd dw reduce");
  }
  return MRMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// gated GELU (HF T5DenseGatedGeluDense + NewGELUActivation)
// ------------------------------------------------------------------------------------------------
// (fast_tanh / gelu_new_f live in common.h: the GEGLU epilogue of gemm8.hip uses the same arithmetic)
// (gelu_new_fd lives in common.h too: the GEGLU-backward epilogue of gemm_rows.hip uses the same arithmetic)

// Both kernels move 8 elements (two dropout quads) per thread and iteration: 16-byte accesses for bf16.
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ g, int rows, int dff, DropCfg d) {
  DROP_STEP(d);
  const size_t n8 = (size_t)rows * dff / 8;
  const int dff8 = dff / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / dff8;
    const int col = (int)(i % dff8) * 8;
    float a[8], b[8], o[8];
    load4<T>(h + row * 2 * dff + col, a);
    load4<T>(h + row * 2 * dff + col + 4, a + 4);
    load4<T>(h + row * 2 * dff + dff + col, b);
    load4<T>(h + row * 2 * dff + dff + col + 4, b + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = gelu_new_f(a[e]) * b[e];
    if (d.thresh) {
      float m[8];
      drop_mask4(d, 2 * i, m);
      drop_mask4(d, 2 * i + 1, m + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= m[e];
    }
    store4<T>(g + row * dff + col, o);
    store4<T>(g + row * dff + col + 4, o + 4);
  }
}

template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dg, T* __restrict__ dh,
                                 int rows, int dff, DropCfg d, int cache_mode) {
  DROP_STEP(d);
  const size_t n8 = (size_t)rows * dff / 8;
  const int dff8 = dff / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / dff8;
    const int col = (int)(i % dff8) * 8;
    float a[8], b[8], go[8], da[8], db[8];
    if (cache_mode & 1) {                       // h: the saved projection, read once
      load4_nt<T>(h + row * 2 * dff + col, a);
      load4_nt<T>(h + row * 2 * dff + col + 4, a + 4);
      load4_nt<T>(h + row * 2 * dff + dff + col, b);
      load4_nt<T>(h + row * 2 * dff + dff + col + 4, b + 4);
    } else {
      load4<T>(h + row * 2 * dff + col, a);
      load4<T>(h + row * 2 * dff + col + 4, a + 4);
      load4<T>(h + row * 2 * dff + dff + col, b);
      load4<T>(h + row * 2 * dff + dff + col + 4, b + 4);
    }
    load4<T>(dg + row * dff + col, go);
    load4<T>(dg + row * dff + col + 4, go + 4);
    if (d.thresh) {
      float m[8];
      drop_mask4(d, 2 * i, m);
      drop_mask4(d, 2 * i + 1, m + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) go[e] *= m[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f, fd;
      gelu_new_fd(a[e], &f, &fd);
      da[e] = go[e] * b[e] * fd;
      db[e] = go[e] * f;
    }
    store4<T>(dh + row * 2 * dff + col, da);
    store4<T>(dh + row * 2 * dff + col + 4, da + 4);
    store4<T>(dh + row * 2 * dff + dff + col, db);
    store4<T>(dh + row * 2 * dff + dff + col + 4, db + 4);
  }
}

static inline int ew_blocks(size_t n_items) {
  size_t b = (n_items + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int mrmt3_geglu_fwd(const void* h, void* g, int rows, int dff, int dtype, float p_drop, uint64_t seed, const int32_t* step_dev,
                               uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(h && g && rows > 0 && dff % 8 == 0, "geglu_fwd: bad args");
  DropCfg d = make_drop(p_drop, seed, stream_id, step_dev);
  const int blocks = ew_blocks((size_t)rows * dff / 8);
  if (dtype == MRMT3_BF16)
    hipLaunchKernelGGL(geglu_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                       (bf16_t*)g, rows, dff, d);
  else
    hipLaunchKernelGGL(geglu_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)h,
                       (float*)g, rows, dff, d);
  MR_CHECK_LAUNCH("geglu_fwd");
  return MRMT3_OK;
}

extern "C" int mrmt3_geglu_bwd(const void* h, const void* dg, void* dh, int rows, int dff, int dtype, float p_drop,
                               uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(h && dg && dh && rows > 0 && dff % 8 == 0, "geglu_bwd: bad args");
  DropCfg d = make_drop(p_drop, seed, stream_id, step_dev);
  const int cm = MR_KNOB("MRMT3_GEGLUB_NT", 0);      // bit 0: streaming load of h (measured: no change; off)
  if (dtype == MRMT3_F32)
    hipLaunchKernelGGL(geglu_bwd_kernel<float>, dim3(ew_blocks((size_t)rows * dff / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)h, (const float*)dg, (float*)dh, rows, dff, d, cm);
  else
    hipLaunchKernelGGL(geglu_bwd_kernel<bf16_t>, dim3(ew_blocks((size_t)rows * dff / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)h, (const bf16_t*)dg, (bf16_t*)dh, rows, dff, d, cm);
  MR_CHECK_LAUNCH("geglu_bwd");
  return MRMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// embedding gather (+ _shift_right) + sinusoid add + dropout; scatter-add backward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t token_at(const int64_t* ids, int row, int seq_len, int shift, int start_id,
                                            int pad_id, int vocab) {
  int64_t id;
  if (shift) {
    const int t = row % seq_len;
    id = (t == 0) ? start_id : ids[row - 1];
    if (id == -100) id = pad_id;
  } else {
    id = ids[row];
  }
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  return id;
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                        const float* __restrict__ pos, float* __restrict__ x, int rows,
                                                        int seq_len, int d, int vocab, int shift, int start_id,
                                                        int pad_id, int pos_offset, DropCfg dc) {
  DROP_STEP(dc);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t id = token_at(ids, row, seq_len, shift, start_id, pad_id, vocab);
  const float* trow = table + (size_t)id * d;
  const float* prow = pos ? pos + (size_t)((row % seq_len) + pos_offset) * d : nullptr;
  const size_t base = (size_t)row * d;
  for (int col = lane * 4; col < d; col += 256) {
    float a[4], p[4];
    load4<float>(trow + col, a);
    if (prow) {
      load4<float>(prow + col, p);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += p[e];
    }
    if (dc.thresh) {
      float m[4];
      drop_mask4(dc, (base + col) >> 2, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= m[e];
    }
    store4<float>(x + base + col, a);
  }
}

// ---- embedding gradient: dtable[id] += dropmask(dx[row]) without atomics -------------------------------------
// Token ids of a transcription target are heavily skewed (a handful of velocity / program ids, and every padded
// position maps to id 0), so a scatter-add with f32 atomics serialises on a few table rows and is order-dependent.
// Instead the rows are ranked by id with a counting sort (per-workgroup histograms -> column scan -> stable
// scatter), chunks of EB_CHUNK sorted rows are summed run by run, and the runs that cross a chunk boundary are
// combined per id in chunk order: every table row has one writer and the result is bitwise reproducible.
#define EB_ROWS 256     // rows ranked per workgroup
#define EB_CHUNK 32     // sorted rows summed per workgroup
struct EbPlan { int n_wg, n_chunk; size_t off_hist, off_base, off_order, off_tok, off_part, total; };
static EbPlan eb_plan(int rows, int vocab, int d) {
  EbPlan p;
  p.n_wg = ceil_div(rows, EB_ROWS);
  p.n_chunk = ceil_div(rows, EB_CHUNK);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  p.off_hist = take((size_t)p.n_wg * vocab * 4);          // [n_wg][vocab] counts, then exclusive offsets inside an id
  p.off_base = take((size_t)(vocab + 1) * 4);             // first sorted position of every id
  p.off_order = take((size_t)rows * 4);                   // sorted position -> row
  p.off_tok = take((size_t)rows * 4);                     // row -> id
  p.off_part = take((size_t)p.n_chunk * 2 * d * 4);       // [chunk][first|last run][d] partial sums
  p.total = o;
  return p;
}

__global__ __launch_bounds__(EB_ROWS) void eb_hist_kernel(const int64_t* __restrict__ ids, int* __restrict__ tok,
                                                           int* __restrict__ hist, int rows, int seq_len, int vocab,
                                                           int shift, int start_id, int pad_id) {
  extern __shared__ int h[];
  for (int v = threadIdx.x; v < vocab; v += EB_ROWS) h[v] = 0;
  __syncthreads();
  const int row = blockIdx.x * EB_ROWS + threadIdx.x;
  if (row < rows) {
    const int id = (int)token_at(ids, row, seq_len, shift, start_id, pad_id, vocab);
    tok[row] = id;
    atomicAdd(&h[id], 1);                                   // LDS counter: only the count matters here
  }
  __syncthreads();
  for (int v = threadIdx.x; v < vocab; v += EB_ROWS) hist[(size_t)blockIdx.x * vocab + v] = h[v];
}

// per id: exclusive scan of the workgroup counts (in place) and the id's total.  A workgroup owns 32 ids; the
// n_wg counts of an id are cut into 8 segments scanned by 8 threads (sum, scan of the 8 sums in LDS, rescan).
__global__ __launch_bounds__(256) void eb_scan_wg_kernel(int* __restrict__ hist, int* __restrict__ base, int n_wg, int vocab) {
  __shared__ int seg_sum[8][32];
  const int vi = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int v = blockIdx.x * 32 + vi;
  const int per = ceil_div(n_wg, 8), g0 = sg * per, g1 = min(n_wg, g0 + per);
  int s = 0;
  if (v < vocab)
    for (int g = g0; g < g1; ++g) s += hist[(size_t)g * vocab + v];
  seg_sum[sg][vi] = s;
  __syncthreads();
  int run = 0;
  for (int k = 0; k < sg; ++k) run += seg_sum[k][vi];
  if (v >= vocab) return;
  for (int g = g0; g < g1; ++g) {
    const int c = hist[(size_t)g * vocab + v];
    hist[(size_t)g * vocab + v] = run;
    run += c;
  }
  if (sg == 7) base[v] = run;                               // totals for now
}
// exclusive scan of the totals over the ids (one workgroup; vocab is a few thousand at most)
__global__ __launch_bounds__(1024) void eb_scan_ids_kernel(int* __restrict__ base, int vocab) {
  __shared__ int part[1024];
  const int per = ceil_div(vocab, 1024);
  const int lo = threadIdx.x * per, hi = min(vocab, lo + per);
  int s = 0;
  for (int v = lo; v < hi; ++v) s += base[v];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) { const int c = part[i]; part[i] = run; run += c; }
    base[vocab] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int v = lo; v < hi; ++v) { const int c = base[v]; base[v] = run; run += c; }
}

__global__ __launch_bounds__(EB_ROWS) void eb_scatter_kernel(const int* __restrict__ tok, const int* __restrict__ hist,
                                                              const int* __restrict__ base, int* __restrict__ order,
                                                              int rows, int vocab) {
  __shared__ int t[EB_ROWS];
  const int row = blockIdx.x * EB_ROWS + threadIdx.x;
  const int mine = row < rows ? tok[row] : -1;
  t[threadIdx.x] = mine;
  __syncthreads();
  if (row >= rows) return;
  int rank = 0;                                             // earlier rows of this workgroup with the same id: stable
  for (int j = 0; j < (int)threadIdx.x; ++j) rank += (t[j] == mine);
  order[base[mine] + hist[(size_t)blockIdx.x * vocab + mine] + rank] = row;
}

// one workgroup sums EB_CHUNK consecutive sorted rows run by run.  A run that starts and ends inside the chunk has
// all of its id's rows here: it is added to the table directly.  The chunk's first and last run may continue in
// the neighbouring chunks: their sums go to part[chunk][0 / 1] and eb_combine_kernel adds them per id.
__global__ __launch_bounds__(256) void eb_sum_kernel(const int* __restrict__ order, const int* __restrict__ tok,
                                                      const int* __restrict__ base, const float* __restrict__ dx,
                                                      float* __restrict__ dtable, float* __restrict__ part, int rows,
                                                      int d, DropCfg dc) {
  DROP_STEP(dc);
  __shared__ int srow[EB_CHUNK], sid[EB_CHUNK];
  const int p0 = blockIdx.x * EB_CHUNK, n = min(EB_CHUNK, rows - p0);
  if (threadIdx.x < n) {
    const int r = order[p0 + threadIdx.x];
    srow[threadIdx.x] = r;
    sid[threadIdx.x] = tok[r];
  }
  __syncthreads();
  for (int col = threadIdx.x * 4; col < d; col += 1024) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int run_start = 0;
    for (int i = 0; i < n; ++i) {
      float a[4];
      const size_t at = (size_t)srow[i] * d + col;
      load4<float>(dx + at, a);
      if (dc.thresh) {
        float m[4];
        drop_mask4(dc, at >> 2, m);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] *= m[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += a[e];
      if (i + 1 == n || sid[i + 1] != sid[i]) {             // the run [run_start, i] of id sid[i] ends here
        const int id = sid[i];
        const bool whole = base[id] >= p0 + run_start && base[id + 1] <= p0 + i + 1;   // all rows of the id are in it
        float* dst = whole ? dtable + (size_t)id * d + col
                           : part + ((size_t)blockIdx.x * 2 + (run_start == 0 ? 0 : 1)) * d + col;
        if (whole) {
          float o[4];
          load4<float>(dst, o);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += o[e];
        }
        store4<float>(dst, acc);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = 0.f;
        run_start = i + 1;
      }
    }
  }
}

// ids whose rows span several chunks: add the chunks' partial sums in chunk order
__global__ __launch_bounds__(128) void eb_combine_kernel(const int* __restrict__ base, const float* __restrict__ part,
                                                          float* __restrict__ dtable, int d) {
  const int id = blockIdx.x;
  const int lo = base[id], hi = base[id + 1];
  if (hi <= lo) return;
  const int c0 = lo / EB_CHUNK, c1 = (hi - 1) / EB_CHUNK;
  if (c0 == c1 && lo >= c0 * EB_CHUNK && hi <= (c0 + 1) * EB_CHUNK) return;      // a whole run: already in the table
  for (int col = threadIdx.x * 4; col < d; col += 512) {
    float acc[4];
    load4<float>(dtable + (size_t)id * d + col, acc);
    for (int c = c0; c <= c1; ++c) {
      // in chunk c the id's run is the first one unless it starts inside the chunk
      const int slot = (lo > c * EB_CHUNK) ? 1 : 0;
      float a[4];
      load4<float>(part + ((size_t)c * 2 + slot) * d + col, a);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += a[e];
    }
    store4<float>(dtable + (size_t)id * d + col, acc);
  }
}

extern "C" int mrmt3_embed_fwd(const int64_t* ids, const float* table, const float* pos, float* x, int rows,
                               int seq_len, int d, int vocab, int shift, int start_id, int pad_id, int pos_offset,
                               float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(ids && table && x && rows > 0 && seq_len > 0 && d % 4 == 0, "embed_fwd: bad args");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, ids,
                     table, pos, x, rows, seq_len, d, vocab, shift, start_id, pad_id, pos_offset,
                     make_drop(p_drop, seed, stream_id, step_dev));
  MR_CHECK_LAUNCH("embed_fwd");
  return MRMT3_OK;
}

extern "C" size_t mrmt3_embed_bwd_workspace_bytes(int rows, int vocab, int d) { return eb_plan(rows, vocab, d).total; }

extern "C" int mrmt3_embed_bwd(const int64_t* ids, const float* dx, float* dtable, int rows, int seq_len, int d,
                               int vocab, int shift, int start_id, int pad_id, float p_drop, uint64_t seed, const int32_t* step_dev,
                               uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream) {
  MR_CHECK_ARG(ids && dx && dtable && workspace && rows > 0 && seq_len > 0 && d % 4 == 0 && vocab > 0,
               "embed_bwd: bad args");
  MR_CHECK_ARG(vocab * (int)sizeof(int) <= 64 * 1024, "embed_bwd: vocabulary too large for the LDS histogram");
  const EbPlan P = eb_plan(rows, vocab, d);
  MR_CHECK_ARG(workspace_bytes >= P.total, "embed_bwd: workspace too small");
  unsigned char* ws = (unsigned char*)workspace;
  int* hist = (int*)(ws + P.off_hist);
  int* base = (int*)(ws + P.off_base);
  int* order = (int*)(ws + P.off_order);
  int* tok = (int*)(ws + P.off_tok);
  float* part = (float*)(ws + P.off_part);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(eb_hist_kernel, dim3(P.n_wg), dim3(EB_ROWS), vocab * sizeof(int), s, ids, tok, hist, rows, seq_len,
                     vocab, shift, start_id, pad_id);
  hipLaunchKernelGGL(eb_scan_wg_kernel, dim3(ceil_div(vocab, 32)), dim3(256), 0, s, hist, base, P.n_wg, vocab);
  hipLaunchKernelGGL(eb_scan_ids_kernel, dim3(1), dim3(1024), 0, s, base, vocab);
  hipLaunchKernelGGL(eb_scatter_kernel, dim3(P.n_wg), dim3(EB_ROWS), 0, s, tok, hist, base, order, rows, vocab);
  hipLaunchKernelGGL(eb_sum_kernel, dim3(P.n_chunk), dim3(256), 0, s, order, tok, base, dx, dtable, part, rows, d,
                     make_drop(p_drop, seed, stream_id, step_dev));
  hipLaunchKernelGGL(eb_combine_kernel, dim3(vocab), dim3(128), 0, s, base, part, dtable, d);
  MR_CHECK_LAUNCH("embed_bwd");
  return MRMT3_OK;
}

template <typename T>
__global__ void addpos_fwd_kernel(const T* __restrict__ src, const float* __restrict__ pos, float* __restrict__ x,
                                  size_t n4, int seq_len, int d, int pos_offset, DropCfg dc) {
  DROP_STEP(dc);
  const int d4 = d / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / d4;
    const int col = (int)(i % d4) * 4;
    float a[4], p[4];
    load4<T>(src + i * 4, a);
    load4<float>(pos + (size_t)((row % seq_len) + pos_offset) * d + col, p);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += p[e];
    if (dc.thresh) {
      float m[4];
      drop_mask4(dc, i, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= m[e];
    }
    store4<float>(x + i * 4, a);
  }
}

extern "C" int mrmt3_addpos_fwd(const void* src, int src_dtype, const float* pos, float* x, int rows, int seq_len,
                                int d, int pos_offset, float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id,
                                void* stream) {
  MR_CHECK_ARG(src && pos && x && rows > 0 && seq_len > 0 && d % 4 == 0, "addpos_fwd: bad args");
  const size_t n4 = (size_t)rows * d / 4;
  DropCfg dc = make_drop(p_drop, seed, stream_id, step_dev);
  if (src_dtype == MRMT3_BF16)
    hipLaunchKernelGGL(addpos_fwd_kernel<bf16_t>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, pos, x, n4, seq_len, d, pos_offset, dc);
  else
    hipLaunchKernelGGL(addpos_fwd_kernel<float>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, pos, x, n4, seq_len, d, pos_offset, dc);
  MR_CHECK_LAUNCH("addpos_fwd");
  return MRMT3_OK;
}

template <typename TO>
__global__ void dropmask_cast_kernel(const float* __restrict__ dx, TO* __restrict__ out, size_t n4, DropCfg dc) {
  DROP_STEP(dc);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float a[4];
    load4<float>(dx + i * 4, a);
    if (dc.thresh) {
      float m[4];
      drop_mask4(dc, i, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= m[e];
    }
    store4<TO>(out + i * 4, a);
  }
}

extern "C" int mrmt3_dropmask_cast(const float* dx, void* out, int out_dtype, size_t n, float p_drop, uint64_t seed,
                                   const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(dx && out && n % 4 == 0, "dropmask_cast: bad args");
  if (out_dtype == MRMT3_F32)
    hipLaunchKernelGGL(dropmask_cast_kernel<float>, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dx,
                       (float*)out, n / 4, make_drop(p_drop, seed, stream_id, step_dev));
  else
    hipLaunchKernelGGL(dropmask_cast_kernel<bf16_t>, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dx,
                       (bf16_t*)out, n / 4, make_drop(p_drop, seed, stream_id, step_dev));
  MR_CHECK_LAUNCH("dropmask_cast");
  return MRMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// cross-entropy (tasks/mt3_net.py:32-35; weighted variant tasks/mt3_net.py:96-108)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ce_weights(int64_t t, int weighted, int lo, int hi, float* w, float* n) {
  if (t == -100) { *w = 0.f; *n = 0.f; return; }
  if (weighted && t >= lo && t <= hi) { *w = 3.f; *n = 2.f; return; }
  *w = 1.f; *n = 1.f;
}

__global__ void ce_count_kernel(const int64_t* __restrict__ targets, int rows, int weighted, int lo, int hi,
                                float* __restrict__ denom) {
  float s = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
    float w, n;
    ce_weights(targets[i], weighted, lo, hi, &w, &n);
    s += n;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(denom, s);
}

extern "C" int mrmt3_ce_count(const int64_t* targets, int rows, int weighted, int inst_lo, int inst_hi,
                              float* denom_dev, void* stream) {
  MR_CHECK_ARG(targets && denom_dev && rows > 0, "ce_count: bad args");
  int blocks = ceil_div(rows, 256);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(ce_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, targets, rows, weighted,
                     inst_lo, inst_hi, denom_dev);
  MR_CHECK_LAUNCH("ce_count");
  return MRMT3_OK;
}

template <typename TD>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                 const float* __restrict__ denom, double* __restrict__ loss,
                                                 TD* __restrict__ dlogits, int rows, int V, int weighted, int lo,
                                                 int hi, float grad_scale) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lp = logits + (size_t)row * V;
  const int64_t t = targets[row];
  float w, n;
  ce_weights(t, weighted, lo, hi, &w, &n);
  TD* dp = dlogits ? dlogits + (size_t)row * V : nullptr;
  if (w == 0.f) {
    if (dp) for (int c = tid * 4; c < V; c += 1024) { float z[4] = {0.f, 0.f, 0.f, 0.f}; store4<TD>(dp + c, z); }
    return;
  }
  // one pass over the row: running (max, sum of exp) per thread, merged across the workgroup
  float mx = -INFINITY, se = 0.f;
  for (int c = tid * 4; c < V; c += 1024) {
    float v[4];
    load4<float>(lp + c, v);
    const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    const float mn = fmaxf(mx, m4);
    se = se * expf(mx - mn) + ((expf(v[0] - mn) + expf(v[1] - mn)) + (expf(v[2] - mn) + expf(v[3] - mn)));
    mx = mn;
  }
  const float wmx = wave_max(mx);
  se = wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
  if (lane == 0) { red[wave] = wmx; red[4 + wave] = se; }
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  se = 0.f;
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) se += red[4 + w4] * expf(red[w4] - mx);
  const float lse = mx + logf(se);
  const float inv_den = 1.f / denom[0];
  if (tid == 0) atomicAdd(loss, (double)(w * (lse - lp[t]) * inv_den));
  if (dp) {
    const float gs = w * inv_den * grad_scale;
    for (int c = tid * 4; c < V; c += 1024) {
      float v[4];
      load4<float>(lp + c, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p = expf(v[e] - lse);
        if (c + e == t) p -= 1.f;
        v[e] = p * gs;
      }
      store4<TD>(dp + c, v);
    }
  }
}

// fast path for V = NV*256 <= 2048 (MT3: 1536): one WAVE per row, the row lives in registers (one
// HBM read), statistics by wave shuffles.  A wave walks CE_RPW consecutive rows and the workgroup adds its
// loss contribution with ONE atomic: one atomic per row (65 536 of them on a single address) cost more
// than the whole rest of the kernel.
// Rows per wave: as few as keep the launch at ~2048 workgroups (a wave walks its rows one after the other, each a
// load -> max -> exp -> sum -> store chain of ~3.5 us: at 16 rows per wave a 16 384-row chunk ran 4 waves per CU and
// 2.6 TB/s; at 2 rows per wave the CUs are full), at most 16.
static inline int ce_rows_per_wave(int rows) {
  int r = rows / (4 * 2048);
  return r < 1 ? 1 : (r > 16 ? 16 : r);
}
template <typename TD, int NV>
__global__ __launch_bounds__(256) void ce_wave_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                      const float* __restrict__ denom, double* __restrict__ loss,
                                                      TD* __restrict__ dlogits, int rows, int weighted, int lo, int hi,
                                                      float grad_scale, int rpw) {
  constexpr int V = NV * 256;
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * rpw;
  const float inv_den = 1.f / denom[0];
  float acc = 0.f;
  for (int row = row0; row < min(row0 + rpw, rows); ++row) {
    const float* lp = logits + (size_t)row * V;
    const int64_t t = targets[row];
    float w, n;
    ce_weights(t, weighted, lo, hi, &w, &n);
    TD* dp = dlogits ? dlogits + (size_t)row * V : nullptr;
    if (w == 0.f) {
      if (dp) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NV; ++i) store4<TD>(dp + (i * 64 + lane) * 4, z);
      }
      continue;
    }
    float v[NV][4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      load4<float>(lp + (i * 64 + lane) * 4, v[i]);
      mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
    }
    const float zt = lp[t];
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = expf(v[i][e] - mx);
        se += v[i][e];
      }
    se = wave_sum(se);
    const float lse = mx + logf(se);
    acc += w * (lse - zt) * inv_den;
    if (dp) {
      const float gs = w * inv_den * grad_scale / se;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = v[i][e] * gs - ((c + e == t) ? w * inv_den * grad_scale : 0.f);
        store4<TD>(dp + c, g);
      }
    }
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    // the loss scalar is a DOUBLE: the order in which the workgroups' sums arrive perturbs it at 1e-16, far below the
    // f32 value that is logged — two runs log the same float (a float accumulator wandered by a few ulp per run)
    const double tot = ((double)part[0] + (double)part[1]) + ((double)part[2] + (double)part[3]);
    if (tot != 0.0) atomicAdd(loss, tot);
  }
}

extern "C" int mrmt3_ce_fwd_bwd(const float* logits, const int64_t* targets, const float* denom_dev, double* loss_dev,
                                void* dlogits, int dl_dtype, int rows, int V, int weighted, int inst_lo,
                                int inst_hi, float grad_scale, void* stream) {
  MR_CHECK_ARG(logits && targets && denom_dev && loss_dev && rows > 0 && V % 4 == 0, "ce_fwd_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (V == 1536 || V == 1024 || V == 2048 || V == 512) {
    const int rpw = ce_rows_per_wave(rows);
    dim3 grid((unsigned)ceil_div(rows, 4 * rpw)), block(256);
#define CEW(TD, NV)                                                                                              \
  hipLaunchKernelGGL((ce_wave_kernel<TD, NV>), grid, block, 0, s, logits, targets, denom_dev, loss_dev, (TD*)dlogits, \
                     rows, weighted, inst_lo, inst_hi, grad_scale, rpw)
#define CEV(TD)                                  \
  do {                                           \
    if (V == 1536) CEW(TD, 6);                   \
    else if (V == 1024) CEW(TD, 4);              \
    else if (V == 2048) CEW(TD, 8);              \
    else CEW(TD, 2);                             \
  } while (0)
    if (dl_dtype == MRMT3_BF16) CEV(bf16_t);
    else CEV(float);
#undef CEV
#undef CEW
    MR_CHECK_LAUNCH("ce_fwd_bwd");
    return MRMT3_OK;
  }
  if (dl_dtype == MRMT3_BF16)
    hipLaunchKernelGGL(ce_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, logits, targets, denom_dev, loss_dev,
                       (bf16_t*)dlogits, rows, V, weighted, inst_lo, inst_hi, grad_scale);
  else
    hipLaunchKernelGGL(ce_kernel<float>, dim3(rows), dim3(256), 0, s, logits, targets, denom_dev, loss_dev,
                       (float*)dlogits, rows, V, weighted, inst_lo, inst_hi, grad_scale);
  MR_CHECK_LAUNCH("ce_fwd_bwd");
  return MRMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// AdamW over the flat parameter buffer (torch.optim.AdamW single-tensor semantics)
// ------------------------------------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n4, const float* __restrict__ lr_dev,
                             const int32_t* __restrict__ step_dev, float b1, float b2, float eps, float wd,
                             float gscale, bf16_t* __restrict__ shadow) {
  const float lr = lr_dev[0];
  const int step = step_dev[0] + 1;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float decay = 1.f - lr * wd;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float pv[4], gv[4], mv[4], vv[4];
    load4<float>(p + i * 4, pv);
    load4<float>(g + i * 4, gv);
    load4<float>(m + i * 4, mv);
    load4<float>(v + i * 4, vv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gv[e] * gscale;
      pv[e] *= decay;
      mv[e] = mv[e] + (gr - mv[e]) * (1.f - b1);           // exp_avg.lerp_(grad, 1-beta1)
      vv[e] = vv[e] * b2 + (1.f - b2) * gr * gr;           // mul_(beta2).addcmul_(g, g, 1-beta2)
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] -= step_size * (mv[e] / denom);
    }
    store4<float>(p + i * 4, pv);
    store4<float>(m + i * 4, mv);
    store4<float>(v + i * 4, vv);
    if (shadow) store4<bf16_t>(shadow + i * 4, pv);
  }
}
__global__ void step_inc_kernel(int32_t* step) { step[0] += 1; }

extern "C" int mrmt3_adamw_step(float* p, const float* g, float* m, float* v, size_t n, const float* lr_dev,
                                int32_t* step_dev, float beta1, float beta2, float eps, float weight_decay,
                                float grad_scale, void* shadow_bf16, void* stream) {
  MR_CHECK_ARG(p && g && m && v && lr_dev && step_dev && n % 4 == 0, "adamw_step: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, p, g, m, v, n / 4, lr_dev, step_dev,
                     beta1, beta2, eps, weight_decay, grad_scale, (bf16_t*)shadow_bf16);
  MR_CHECK_LAUNCH("adamw_step");
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step_dev);
  MR_CHECK_LAUNCH("adamw_step inc");
  return MRMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// cast / transpose helpers (bf16 shadow weights and their pre-transposed dgrad copies)
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float a[4];
    load4<TI>(in + i * 4, a);
    store4<TO>(out + i * 4, a);
  }
}

extern "C" int mrmt3_cast(const void* in, int in_dtype, void* out, int out_dtype, size_t n, void* stream) {
  MR_CHECK_ARG(in && out && n % 4 == 0, "cast: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int b = ew_blocks(n / 4);
  if (in_dtype == MRMT3_F32 && out_dtype == MRMT3_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(b), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n / 4);
  else if (in_dtype == MRMT3_BF16 && out_dtype == MRMT3_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(b), dim3(256), 0, s, (const bf16_t*)in, (float*)out, n / 4);
  else if (in_dtype == MRMT3_F32 && out_dtype == MRMT3_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(b), dim3(256), 0, s, (const float*)in, (float*)out, n / 4);
  else
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(b), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, n / 4);
  MR_CHECK_LAUNCH("cast");
  return MRMT3_OK;
}

template <typename TI, typename TO>
__global__ void transpose_kernel(const TI* __restrict__ in, TO* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      if constexpr (sizeof(TI) == 2) v = bf2f(in[(size_t)r * cols + c]);
      else v = in[(size_t)r * cols + c];
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;  // out[c][r]
    if (r < rows && c < cols) {
      const float v = tile[tx][j];
      if constexpr (sizeof(TO) == 2) out[(size_t)c * rows + r] = f2bf(v);
      else out[(size_t)c * rows + r] = v;
    }
  }
}

// batched bf16 transpose of many small matrices described by a device table of
// {src_off, dst_off, rows, cols} (element offsets) — one launch refreshes every pre-transposed weight.
// 64x64 tiles; both sides move element PAIRS (4 bytes per lane, 128 contiguous bytes per 32 lanes) when the
// matrix dimensions and offsets are even, single elements otherwise.
#define TRB 64
struct TrDesc { long long src, dst; int rows, cols; };
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                const TrDesc* __restrict__ tab,
                                                                const int* __restrict__ tile_start, int n_mats) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[TRB][TRB + 2];
  // find the matrix this block belongs to (tile_start is a prefix sum of 64x64 tile counts)
  int m = 0;
  while (m + 1 < n_mats && (int)blockIdx.x >= tile_start[m + 1]) ++m;
  const TrDesc d = tab[m];
  const int local = blockIdx.x - tile_start[m];
  const int tiles_x = (d.cols + TRB - 1) / TRB;
  const int bx = (local % tiles_x) * TRB, by = (local / tiles_x) * TRB;
  // fast path (every weight of the model): a tile inside a matrix whose dimensions and offsets are multiples of 8 moves
  // 16 bytes per lane on both sides — a lane loads 8 consecutive columns of a row, stores them as four dwords (row stride
  // 66 elements = 33 dwords: the transposed 2-byte reads of a wave then touch 32 different banks), reads 8 consecutive
  // ROWS of one column back and stores them as 16 bytes of the transposed row.
  if ((((d.rows | d.cols) & 7) == 0) && (((d.src | d.dst) & 7) == 0) && bx + TRB <= d.cols && by + TRB <= d.rows) {
    const int r = threadIdx.x >> 3, c8 = threadIdx.x & 7;
    unsigned* t32 = (unsigned*)&tile[0][0];                     // row stride 33 dwords
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int rr = r + 32 * hlf;
      const u32x4 v = *(const u32x4*)(src + d.src + (size_t)(by + rr) * d.cols + bx + c8 * 8);
      unsigned* q = t32 + rr * 33 + c8 * 4;
      q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int oc = r + 32 * hlf;                              // output row = source column
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        w[k] = (unsigned)tile[c8 * 8 + 2 * k][oc] | ((unsigned)tile[c8 * 8 + 2 * k + 1][oc] << 16);
      *(u32x4*)(dst + d.dst + (size_t)(bx + oc) * d.rows + by + c8 * 8) = u32x4{w[0], w[1], w[2], w[3]};
    }
    return;
  }
  const int tp = threadIdx.x & 31, ty = threadIdx.x >> 5;      // pair index, 8 row groups
  const bool even = ((d.rows | d.cols) & 1) == 0 && ((d.src | d.dst) & 1) == 0;
  for (int j = ty; j < TRB; j += 8) {
    const int r = by + j, c = bx + 2 * tp;
    bf16_t v0 = 0, v1 = 0;
    if (r < d.rows) {
      const bf16_t* p = src + d.src + (size_t)r * d.cols + c;
      if (even && c + 1 < d.cols) {
        const unsigned u = *(const unsigned*)p;
        v0 = (bf16_t)(u & 0xFFFFu);
        v1 = (bf16_t)(u >> 16);
      } else {
        if (c < d.cols) v0 = p[0];
        if (c + 1 < d.cols) v1 = p[1];
      }
    }
    tile[j][2 * tp] = v0;
    tile[j][2 * tp + 1] = v1;
  }
  __syncthreads();
  for (int j = ty; j < TRB; j += 8) {
    const int c = bx + j, r = by + 2 * tp;                    // output row c holds source rows r, r+1 side by side
    if (c >= d.cols) continue;
    bf16_t* q = dst + d.dst + (size_t)c * d.rows + r;
    const bf16_t v0 = tile[2 * tp][j], v1 = tile[2 * tp + 1][j];
    if (even && r + 1 < d.rows) *(unsigned*)q = (unsigned)v0 | ((unsigned)v1 << 16);
    else {
      if (r < d.rows) q[0] = v0;
      if (r + 1 < d.rows) q[1] = v1;
    }
  }
}

extern "C" int mrmt3_transpose_batched(const void* src_bf16, void* dst_bf16, const void* desc_table,
                                       const int* tile_start, int n_mats, int total_tiles, void* stream) {
  MR_CHECK_ARG(src_bf16 && dst_bf16 && desc_table && tile_start && n_mats > 0 && total_tiles > 0, "transpose_batched: bad args");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src_bf16, (bf16_t*)dst_bf16, (const TrDesc*)desc_table, tile_start, n_mats);
  MR_CHECK_LAUNCH("transpose_batched");
  return MRMT3_OK;
}

extern "C" int mrmt3_transpose(const void* in, int in_dtype, void* out, int out_dtype, int rows, int cols,
                               void* stream) {
  MR_CHECK_ARG(in && out && rows > 0 && cols > 0, "transpose: bad args");
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == MRMT3_F32 && out_dtype == MRMT3_BF16)
    hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, block, 0, s, (const float*)in, (bf16_t*)out, rows, cols);
  else if (in_dtype == MRMT3_F32 && out_dtype == MRMT3_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, block, 0, s, (const float*)in, (float*)out, rows, cols);
  else if (in_dtype == MRMT3_BF16 && out_dtype == MRMT3_BF16)
    hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, rows, cols);
  else
    hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)in, (float*)out, rows, cols);
  MR_CHECK_LAUNCH("transpose");
  return MRMT3_OK;
}
