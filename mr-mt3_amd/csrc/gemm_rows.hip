// NT GEMM with a ROW epilogue: the projection and the row kernel behind it in one launch.
//
// Every o / co / wo projection of a T5 block is followed by "residual add + dropout + T5LayerNorm" (HF
// T5LayerSelfAttention / T5LayerCrossAttention / T5LayerFF as called at models/t5.py:636-648), every data-gradient
// product with 512 output columns by the backward of that norm, and the wo data gradient by the gated-GELU backward.
// As separate kernels the product's output makes an HBM round trip (y 3.9 GB, dxn 5.4 GB, dg 4.3 GB per 64-segment step,
// profiles/r03_pmc_step_traffic.txt) and the row kernels — already at 5.8 TB/s — are 24 % of the step.  Here the
// workgroup that computes 64 rows x 512 columns of the product keeps them on chip:
//
//   K loop   64 x 512 tile, 8 waves, wave w owns columns [64 w, 64 w + 64) and ALL 64 rows: accumulators 4 x 4 MFMA
//            tiles (64 VGPRs), two workgroups per CU.  The weight rows of a wave's columns are read by that wave only, so
//            they are staged PRIVATELY: per 64-deep K chunk a wave requests its 2 x 32 weight rows x 128 B (whole lines)
//            by LDS-DMA into two 4-KiB slots of its own — one per column half, consumed in two phases of 16 MFMAs — and
//            waits for them with a counted vmcnt: no barrier for 8/9 of the staged bytes.  The 64 activation rows are
//            shared: 8 KiB per K chunk, one 1-KiB piece per wave, double buffered, one s_barrier per chunk.  Every load
//            has a whole chunk (two phases) to land.
//            MFMAs are issued with the weights as the first operand, so a lane holds 4 consecutive columns of one row.
//   hand-off the tile is rounded to bf16 — exactly what the stand-alone product writes — into an LDS image
//            [64 rows][512 columns] (row stride 1040 B: conflict-free 8-byte writes), the staging memory being free.
//   rows     each wave then runs the ROW kernel's body on 8 rows of the tile, reading y from LDS instead of HBM: the
//            same per-lane column assignment, the same arithmetic in the same order as rowops.hip, so the results equal
//            the two-kernel form bit for bit (forward; the norm-weight partial sums are grouped by 64 rows instead of 32).
//            All of a wave's global loads (8 rows x 2 KiB) are requested before the first is used.
//
// The launch is bound by HBM (o projection: 25.8 GFLOP against 385 MB): with two workgroups per CU one streams its rows
// while the other runs its K loop.
//   EPI 0  mrmt3_gemm_nt_addnorm    x1 = x0 + dropout(A W^T);  xn = w * x1 * rsqrt(mean(x1^2) + eps)   (K3 behind K2/K6)
//   EPI 1  mrmt3_gemm_nt_normbwd    backward of that norm on dxn = A WT^T (+ residual gradient), dx1 / masked dy / dw rows
//   EPI 2  mrmt3_gemm_nt_geglubwd   dh = gated-GELU backward of dg = dy Wo  (K7 backward behind the wo data gradient)
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// no implicit FMA formation: the row phases restate rowops.hip's arithmetic and must land on the same bits
#pragma clang fp contract(off)

#define GR_OOB 0x7FFF0000
#define GR_YLD 1040                       // bytes per row of the LDS image of the output tile
#define GR_B_BYTES 65536                  // 8 waves x 2 slots x 4 KiB of private weight rows

enum { GR_ADDNORM = 0, GR_NORMBWD = 1, GR_GEGLUBWD = 2 };

struct GRParams {
  const bf16_t* A;
  const bf16_t* B;
  int lda, ldb, M, K, n_ctiles;
  // row operands (which ones depends on the epilogue)
  const float* xin;        // ADDNORM: residual stream in [M][512] f32;  NORMBWD: x1, the saved residual stream
  const float* wn;         // norm weight [512]
  float eps;
  float* x1;               // ADDNORM: residual stream out (may be xin)
  bf16_t* xn;              // ADDNORM: normalised output
  float* rstd;             // ADDNORM: out [M];  NORMBWD: in
  const void* dres;        // NORMBWD: residual gradient in (f32 or bf16)
  void* dx1;               // NORMBWD: residual gradient out (f32 or bf16; may be dres)
  bf16_t* dy;              // NORMBWD: dropout-masked bf16 gradient of the sublayer output below (nullable)
  float* dw_part;          // NORMBWD: [gridDim.x][512] partial sums of the norm-weight gradient (nullable)
  int* dw_counters;        // NORMBWD: arrival counters of the reduction (zeroed here, like add_rmsnorm_bwd_kernel)
  const bf16_t* h;         // GEGLUBWD: saved [M][2 dff]
  bf16_t* dh;              // GEGLUBWD: out [M][2 dff]
  int dff;
  DropCfg d0, d1;          // ADDNORM: d0 mask of y, d1 mask of the output (out_drop);  NORMBWD: d0 mask of dy;  GEGLUBWD: d0
  int out_drop, cache_mode;
#ifdef MRMT3_DIAG
  // The round-4 experiments (measured and closed) live in the diagnostics build only, libmrmt3_hip_diag.so:
  int skew_fine;           // start delay (10-ns ticks of s_memrealtime): see the kernel
  unsigned long long* trace;   // mrmt3_gemm_rows_trace: 8 timestamps per workgroup
  int dbg;                 // MRMT3_ROWS_DBG: 1 no K loop, 2 no row epilogue, 4 print the occupancy, 16 / 32 weight / activation loads
                           // switched off (zero fill), 64 no MFMAs, 128 no fragment reads, 256 K chunks walked from a per-workgroup start
#endif
};
#ifdef MRMT3_DIAG
#define GR_DBG(bit) (P.dbg & (bit))
#else
#define GR_DBG(bit) 0
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gr_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void gr_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
#define GR_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// The lane's two 8-byte pieces of an image row (columns 4 lane .. + 3 and 256 + 4 lane .. + 3, 512 bytes apart) in ONE
// ds_read2st64_b64.
__device__ __forceinline__ u32x4 gr_img_pair(const unsigned char* p) {
  u32x4 r;
  asm volatile("ds_read2st64_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(uintptr_t)p) : "memory");
  return r;
}
__device__ __forceinline__ void gr_unpack4(u32x2 t, float v[4]) {
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
}
template <bool BF> __device__ __forceinline__ void gr_load4(const void* p, size_t idx, float v[4]) {
  if constexpr (BF) {
    gr_unpack4(*(const u32x2*)((const bf16_t*)p + idx), v);
  } else {
    const f32x4 t = *(const f32x4*)((const float*)p + idx);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
}
template <bool BF> __device__ __forceinline__ void gr_store4(void* p, size_t idx, const float v[4]) {
  if constexpr (BF) *(u32x2*)((bf16_t*)p + idx) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  else *(f32x4*)((float*)p + idx) = f32x4{v[0], v[1], v[2], v[3]};
}

// RI / RO: the residual gradient in / out is bf16 (NORMBWD only).  BM: rows of a tile — 64 (8 waves x 64 accumulator
// registers, two workgroups per CU) or 128 (8 waves x 128, one workgroup per CU, half the weight bytes staged per row).
template <int BM> struct GRCfg {
  static constexpr int RT = BM / 16;                 // 16-row MFMA tiles per wave
  static constexpr int NAP = BM / 64;                // activation pieces (8 rows x 128 B) per wave and K chunk
  static constexpr int A_SLOT = BM * 128;            // bytes of one K chunk of activations
  static constexpr int NSLOT = BM == 64 ? 2 : 4;     // activation slots; a chunk is requested NSLOT - 1 chunks ahead
  static constexpr int LA = NSLOT - 1;
  static constexpr int STAGE = GR_B_BYTES + NSLOT * A_SLOT;
  static constexpr int IMAGE = BM * GR_YLD;
  static constexpr int LDS = STAGE > IMAGE ? STAGE : IMAGE;
  static constexpr int RPW = BM / 8;                 // rows per wave in the row phase
};

template <int EPI, bool RI, bool RO, int BM>
__global__ __launch_bounds__(512, BM == 64 ? 4 : 2) void gemm_rows_kernel(GRParams P) {
  using Cfg = GRCfg<BM>;
  constexpr int RT = Cfg::RT, NAP = Cfg::NAP, A_SLOT = Cfg::A_SLOT, NSLOT = Cfg::NSLOT, LA = Cfg::LA, RPW = Cfg::RPW;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[Cfg::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- tile.  Two column tiles (the wo data gradient, dff = 1024) of the same rows read the same activations: they run
  // as blocks b and b + 8, which share an XCD (round-robin dispatch; a speed assumption only).
  int mt, nt;
  if (P.n_ctiles == 2) {
    const int b = (int)blockIdx.x;
    mt = (b >> 4) * 8 + (b & 7);
    nt = (b >> 3) & 1;
  } else {
    mt = (int)blockIdx.x / P.n_ctiles;
    nt = (int)blockIdx.x - mt * P.n_ctiles;
  }
  const int m0 = mt * BM, n0 = nt * 512;
#ifdef MRMT3_DIAG
#define GR_STAMP(i) do { if (P.trace && tid == 0) P.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GR_STAMP(0);
  // (experiment, MRMT3_ROWS_SKEW_FINE: two start phases so that not every CU is in its K loop — HBM idle — and then in
  // its row phase at the same time)
  if (P.skew_fine > 0) {
    const unsigned long long wait = (unsigned long long)(((int)blockIdx.x >> 3) & 1) * P.skew_fine;
    if (wait) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }
#else
#define GR_STAMP(i) do { } while (0)
#endif
  if (EPI == GR_NORMBWD && P.dw_counters != nullptr && blockIdx.x == 0 && tid < 8) P.dw_counters[tid] = 0;
  if (m0 >= P.M) return;

  // (the step salt is a global load: requested here, long before the row phase needs it)
  DropCfg dc0 = P.d0, dc1 = P.d1;
  DROP_STEP(dc0);
  if (EPI == GR_ADDNORM) DROP_STEP(dc1);
  const __amdgpu_buffer_rsrc_t ra = gr_rsrc(P.A, ((size_t)(P.M - 1) * P.lda + P.K) * 2);
  const __amdgpu_buffer_rsrc_t rb = gr_rsrc(P.B, ((size_t)(P.n_ctiles * 512 - 1) * P.ldb + P.K) * 2);
  // LDS-DMA sources: WHOLE 128-byte lines only.  (The first version staged the weights in 64-byte K steps, half a line
  // per row and request: a CU's memory path moves lines, used in full or not.)  A piece q of wave w = tile rows
  // 64 q + 8 w .. + 7 x 128 B; B piece q of wave w and column half ch = its columns 32 ch + 8 q .. + 7 x 128 B.
  // Lane p -> (row p/8, 16-byte chunk (p%8) ^ (p/8)).
  const int sw16 = ((lane & 7) ^ (lane >> 3)) << 4;
  const unsigned voffA = (unsigned)((m0 + 8 * w + (lane >> 3)) * P.lda * 2 + sw16);
  const unsigned voffB = (unsigned)((n0 + 64 * w + (lane >> 3)) * P.ldb * 2 + sw16);
  const int qstride = 8 * P.ldb * 2, astride = 64 * P.lda * 2;
  unsigned char* const ldsA = lds + GR_B_BYTES + w * 1024;
  unsigned char* const ldsB = lds + w * 8192;
  const int np = GR_DBG(1) ? 0 : P.K >> 6;
#ifdef MRMT3_DIAG
  // (experiment, MRMT3_ROWS_DBG bit 256: every workgroup walks the K chunks from a different starting chunk, so that the
  // CUs of an XCD do not all ask the L2 for the same weight lines at the same time)
  const int rot = (P.dbg & 256) && np > 0 ? (int)((blockIdx.x >> 3) % (unsigned)np) : 0;   // (blocks b, b + 8, ... share an XCD)
  auto koff = [&](int chunk) __attribute__((always_inline)) -> int {
    int c = chunk + rot;
    if (c >= np) c -= np;
    return c * 128;
  };
#else
  auto koff = [&](int chunk) __attribute__((always_inline)) -> int { return chunk * 128; };
#endif

  auto ldA = [&](int slot, int chunk) __attribute__((always_inline)) {
    const bool on = chunk < np && !GR_DBG(32);
#pragma unroll
    for (int q = 0; q < NAP; ++q)
      gr_dma16(ra, ldsA + slot * A_SLOT + q * 8192, voffA, on ? koff(chunk) + q * astride : GR_OOB);
  };
  // B of K chunk `chunk` (64 deep), column half ch -> this wave's slot ch
  auto ldB = [&](int ch, int chunk) __attribute__((always_inline)) {
    const bool on = chunk < np && !GR_DBG(16);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      gr_dma16(rb, ldsB + ch * 4096 + q * 1024, voffB, on ? koff(chunk) + (ch * 4 + q) * qstride : GR_OOB);
  };

  // fragment read offsets (128-byte rows, chunk c of row r at c ^ (r & 7); the second 32-deep K step is the offset ^ 64)
  const int f_off = fr * 128 + ((fg ^ (fr & 7)) << 4);

  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one phase = the wave's column half ch over one 64-deep K chunk: 2 x (RT A + 2 B fragments, 2 RT MFMAs)
  auto phase = [&](int aslot, int ch, int p) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[RT], bf_[2];
      const int fo = ks ? (f_off ^ 64) : f_off;     // (an integer offset: XOR on the pointer itself turns the reads into flat loads)
      const unsigned char* pa = lds + GR_B_BYTES + aslot * A_SLOT + fo;
      const unsigned char* pb = lds + w * 8192 + ch * 4096 + fo;
      if (!GR_DBG(128)) {
#pragma unroll
        for (int i = 0; i < RT; ++i) af[i] = *(const bf16x8*)(pa + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf_[j] = *(const bf16x8*)(pb + j * 2048);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 1) {
        ldB(ch, p + 1);                            // this wave's own slot: its reads above have completed
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!GR_DBG(64)) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][ch * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[j], af[i], acc[i][ch * 2 + j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);            // (the counted waits / the barrier that follow stay BEHIND these MFMAs)
    }
  };
  auto chunk = [&](int aslot, int p) __attribute__((always_inline)) {
    // outstanding, oldest first: A(p + LA - 1) x NAP, B(p, 0) x 4, B(p, 1) x 4
    GR_VMCNT(4);
    __builtin_amdgcn_s_barrier();               // every wave's pieces of A(p) have landed; all are done reading A(p-1)
    __builtin_amdgcn_sched_barrier(0);
    ldA((aslot + LA) % NSLOT, p + LA);          // (into the slot of chunk p - 1)
    phase(aslot, 0, p);
    // outstanding: B(p, 1) x 4, A(p + LA) x NAP, B(p+1, 0) x 4
    if (NAP == 1) GR_VMCNT(5); else GR_VMCNT(6);
    __builtin_amdgcn_sched_barrier(0);
    phase(aslot, 1, p);
  };

  GR_STAMP(1);
#pragma unroll
  for (int c = 0; c < LA; ++c) ldA(c, c);
  ldB(0, 0);
  ldB(1, 0);
  if constexpr (NSLOT == 2) {
    for (int p = 0; p < np; p += 2) {             // K is a multiple of 128: an even number of 64-deep chunks
      chunk(0, p);
      chunk(1, p + 1);
    }
  } else {
    int p = 0;
    for (; p + 4 <= np; p += 4) { chunk(0, p); chunk(1, p + 1); chunk(2, p + 2); chunk(3, p + 3); }
    if (p < np) { chunk(0, p); chunk(1, p + 1); }   // (np is even)
  }
  GR_VMCNT(0);                                   // (the switched-off tail loads still write zeros into their slots)
  __builtin_amdgcn_s_barrier();
  GR_STAMP(2);

  // ---- hand-off: the tile, rounded to bf16, as an LDS image [BM][GR_YLD]
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(u32x2*)(lds + (16 * i + fr) * GR_YLD + (64 * w + 16 * j + 4 * fg) * 2) =
          u32x2{pack_bf2(acc[i][j][0], acc[i][j][1]), pack_bf2(acc[i][j][2], acc[i][j][3])};
  __syncthreads();

  GR_STAMP(3);
  if (GR_DBG(2)) return;
  // ---- rows: wave w takes tile rows w, w + 8, ...
  if constexpr (EPI == GR_ADDNORM) {
    // The stand-alone row kernel hides its per-row dependency chains (LDS read -> mask -> sum of squares -> six shuffle
    // steps -> rsqrt -> stores) behind 32 waves per CU; here 8 or 16 waves run the rows, so the chains of ALL of a wave's
    // rows are walked side by side: one pass per stage over the rows (same arithmetic per row, same bits), every global
    // load of the wave's rows requested before the first is used.
    const DropCfg dy = dc0, dout = dc1;
    float wv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) gr_load4<false>(P.wn, i * 256 + lane * 4, wv[i]);
    constexpr int G = RPW;                       // rows per group: all of the wave's rows (8 or 16)
    // the wave's pieces of the tile image first (LDS reads while none of this wave's global loads are in flight)
    u32x4 yimg[G];
#pragma unroll
    for (int t = 0; t < G; ++t) yimg[t] = gr_img_pair(lds + (w + 8 * t) * GR_YLD + lane * 8);
    f32x4 xv[G][2];
#pragma unroll
    for (int t = 0; t < G; ++t) {
      const int row = m0 + w + 8 * t;
      if (row < P.M) {
        const float* px = P.xin + (size_t)row * 512 + lane * 4;
        if (P.cache_mode & 2) { xv[t][0] = __builtin_nontemporal_load((const f32x4*)px); xv[t][1] = __builtin_nontemporal_load((const f32x4*)(px + 256)); }
        else { xv[t][0] = *(const f32x4*)px; xv[t][1] = *(const f32x4*)(px + 256); }
      }
    }
    float ss[G];
#pragma unroll
    for (int t = 0; t < G; ++t) {
      const int lr = w + 8 * t, row = m0 + lr;
      ss[t] = 0.f;
      if (row >= P.M) continue;
      const size_t base = (size_t)row * 512;
      const u32x4 y4 = yimg[t];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = i * 256 + lane * 4;
        float v[4] = {xv[t][i].x, xv[t][i].y, xv[t][i].z, xv[t][i].w};
        float yv[4];
        gr_unpack4(i == 0 ? u32x2{y4.x, y4.y} : u32x2{y4.z, y4.w}, yv);
        if (dy.thresh) {
          float m[4];
          drop_mask4(dy, (base + col) >> 2, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) yv[e] *= m[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += yv[e];
        xv[t][i] = f32x4{v[0], v[1], v[2], v[3]};
        if (P.x1 != nullptr) {
          if (P.cache_mode & 1) __builtin_nontemporal_store(xv[t][i], (f32x4*)(P.x1 + base + col));
          else *(f32x4*)(P.x1 + base + col) = xv[t][i];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ss[t] = fmaf(v[e], v[e], ss[t]);
      }
    }
    // wave_sum of every row, the rows' butterflies interleaved (per row: the same six steps in the same order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int t = 0; t < G; ++t) ss[t] += __shfl_xor(ss[t], o, 64);
#pragma unroll
    for (int t = 0; t < G; ++t) {
      const int row = m0 + w + 8 * t;
      if (row >= P.M) continue;
      const size_t base = (size_t)row * 512;
      const float rstd = rsqrtf(ss[t] / 512.0f + P.eps);
      if (lane == 0 && P.rstd != nullptr) P.rstd[row] = rstd;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = i * 256 + lane * 4;
        const float v[4] = {xv[t][i].x, xv[t][i].y, xv[t][i].z, xv[t][i].w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = wv[i][e] * (v[e] * rstd);
        if (P.out_drop && dout.thresh) {
          float m[4];
          drop_mask4(dout, (base + col) >> 2, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= m[e];
        }
        gr_store4<true>(P.xn, base + col, o);
      }
    }
  } else if constexpr (EPI == GR_NORMBWD) {
    const DropCfg ddy = dc0;
    float wv[2][4], dwp[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      gr_load4<false>(P.wn, i * 256 + lane * 4, wv[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) dwp[i][e] = 0.f;
    }
    // groups of G rows, their dependency chains side by side (see the forward).  G = 4 with 128 VGPRs (64-row tiles): x1
    // (f32) + the residual gradient of more rows would not fit; 8 with 256
    constexpr int G = BM == 128 ? (RI ? 8 : 4) : 2;
#pragma unroll
    for (int grp = 0; grp < RPW / G; ++grp) {
      typedef typename std::conditional<RI, u32x2, f32x4>::type RawR;
      f32x4 xv[G][2];
      RawR rr[G][2];
      float rs[G];
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int row = m0 + w + 8 * (grp * G + t);
        if (row < P.M) {
          const size_t base = (size_t)row * 512;
          xv[t][0] = *(const f32x4*)(P.xin + base + lane * 4);
          xv[t][1] = *(const f32x4*)(P.xin + base + 256 + lane * 4);
          rr[t][0] = *(const RawR*)((const unsigned char*)P.dres + (base + lane * 4) * (RI ? 2 : 4));
          rr[t][1] = *(const RawR*)((const unsigned char*)P.dres + (base + 256 + lane * 4) * (RI ? 2 : 4));
          rs[t] = P.rstd[row];
        }
      }
      float g[G][2][4], dot[G];
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int lr = w + 8 * (grp * G + t), row = m0 + lr;
        dot[t] = 0.f;
        if (row >= P.M) continue;
        const float rstd = rs[t];
        const u32x4 g4 = gr_img_pair(lds + lr * GR_YLD + lane * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          gr_unpack4(i == 0 ? u32x2{g4.x, g4.y} : u32x2{g4.z, g4.w}, g[t][i]);
          float xh[4] = {xv[t][i].x, xv[t][i].y, xv[t][i].z, xv[t][i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[e] *= rstd;
            dwp[i][e] = fmaf(g[t][i][e], xh[e], dwp[i][e]);
            g[t][i][e] *= wv[i][e];
            dot[t] = fmaf(g[t][i][e], xh[e], dot[t]);
          }
          xv[t][i] = f32x4{xh[0], xh[1], xh[2], xh[3]};
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int t = 0; t < G; ++t) dot[t] += __shfl_xor(dot[t], o, 64);
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int row = m0 + w + 8 * (grp * G + t);
        if (row >= P.M) continue;
        const size_t base = (size_t)row * 512;
        const float rstd = rs[t], dt = dot[t] / 512.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int col = i * 256 + lane * 4;
          const float xh[4] = {xv[t][i].x, xv[t][i].y, xv[t][i].z, xv[t][i].w};
          float d[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = rstd * fmaf(-xh[e], dt, g[t][i][e]);
          float rv[4];
          if constexpr (RI) gr_unpack4(rr[t][i], rv);
          else { rv[0] = rr[t][i].x; rv[1] = rr[t][i].y; rv[2] = rr[t][i].z; rv[3] = rr[t][i].w; }
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] += rv[e];
          gr_store4<RO>(P.dx1, base + col, d);
          if (P.dy != nullptr) {
            if (ddy.thresh) {
              float m[4];
              drop_mask4(ddy, (base + col) >> 2, m);
#pragma unroll
              for (int e = 0; e < 4; ++e) d[e] *= m[e];
            }
            gr_store4<true>(P.dy, base + col, d);
          }
        }
      }
    }
    // norm-weight gradient: the 8 waves' partial sums in wave order -> this workgroup's partial row
    if (P.dw_part != nullptr) {
      __syncthreads();                                     // every wave is done with the tile image
      float* red = (float*)lds;                            // [8][512]
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *(f32x4*)(red + w * 512 + i * 256 + lane * 4) = f32x4{dwp[i][0], dwp[i][1], dwp[i][2], dwp[i][3]};
      __syncthreads();
      float s = red[tid];
#pragma unroll
      for (int k = 1; k < 8; ++k) s += red[k * 512 + tid];
      P.dw_part[(size_t)mt * 512 + tid] = s;
    }
  } else {
    const DropCfg d = dc0;
    const int dff = P.dff;
    constexpr int G = BM == 128 ? 8 : 4;
#pragma unroll
    for (int grp = 0; grp < RPW / G; ++grp) {
      u32x4 hv[G][2];
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int row = m0 + w + 8 * (grp * G + t);
        if (row < P.M) {
          const bf16_t* ph = P.h + (size_t)row * 2 * dff + n0 + lane * 8;
          if (P.cache_mode & 1) { hv[t][0] = __builtin_nontemporal_load((const u32x4*)ph); hv[t][1] = __builtin_nontemporal_load((const u32x4*)(ph + dff)); }
          else { hv[t][0] = *(const u32x4*)ph; hv[t][1] = *(const u32x4*)(ph + dff); }
        }
      }
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int lr = w + 8 * (grp * G + t), row = m0 + lr;
        if (row >= P.M) continue;
        const u32x4 gv = *(const u32x4*)(lds + lr * GR_YLD + lane * 16);
        float a[8], b[8], go[8], da[8], db[8];
        gr_unpack4(u32x2{hv[t][0].x, hv[t][0].y}, a); gr_unpack4(u32x2{hv[t][0].z, hv[t][0].w}, a + 4);
        gr_unpack4(u32x2{hv[t][1].x, hv[t][1].y}, b); gr_unpack4(u32x2{hv[t][1].z, hv[t][1].w}, b + 4);
        gr_unpack4(u32x2{gv.x, gv.y}, go); gr_unpack4(u32x2{gv.z, gv.w}, go + 4);
        if (d.thresh) {
          float m[8];
          const unsigned long long i8 = ((unsigned long long)row * dff + n0 + lane * 8) >> 3;
          drop_mask4(d, 2 * i8, m);
          drop_mask4(d, 2 * i8 + 1, m + 4);
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
        bf16_t* pd = P.dh + (size_t)row * 2 * dff + n0 + lane * 8;
        *(u32x4*)pd = u32x4{pack_bf2(da[0], da[1]), pack_bf2(da[2], da[3]), pack_bf2(da[4], da[5]), pack_bf2(da[6], da[7])};
        *(u32x4*)(pd + dff) = u32x4{pack_bf2(db[0], db[1]), pack_bf2(db[2], db[3]), pack_bf2(db[4], db[5]), pack_bf2(db[6], db[7])};
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GR_STAMP(4);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
#ifdef MRMT3_DIAG
static unsigned long long* g_rows_trace = nullptr;
// Diagnostics build only (not in include/mrmt3_hip.h): per-workgroup timestamps (s_memrealtime, 10-ns ticks) of the fused
// kernels' phases are written to `buf` ([grid][8] uint64: start, K loop start, K loop end, tile image written, end) until
// it is set back to NULL.
extern "C" int mrmt3_gemm_rows_trace(void* buf) { g_rows_trace = (unsigned long long*)buf; return MRMT3_OK; }
#endif

// 1 when the fused kernel takes (M rows, N = n_ctiles * 512 columns, K): 16-byte aligned rows, K a multiple of 128 (an
// even number of 64-deep pairs), every buffer offset below 2^31.
extern "C" int mrmt3_gemm_rows_ok(int M, int N, int K, int lda, int ldw) {
  if (M <= 0 || N <= 0 || K < 128 || K % 128 != 0 || N % 512 != 0 || (N != 512 && N != 1024)) return 0;
  if (lda % 8 != 0 || ldw % 8 != 0 || lda < K || ldw < K) return 0;
  const size_t mpad = (size_t)ceil_div(M, 128) * 128;
  if ((mpad * lda + K) * 2 >= 0x7FFF0000ull || ((size_t)N * ldw + K) * 2 >= 0x7FFF0000ull) return 0;
  return 1;
}

static int gr_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
// Tile height: 128 rows (one workgroup per CU; half the weight bytes staged per row: a CU takes in ~60 GB/s of LDS-DMA,
// which is what the 64-row K loop ran at) once there are enough of them to fill the chip, else 64 rows (two per CU).
// A function of the row count alone: the norm-weight partial rows (one per tile) follow it.  MRMT3_ROWS_BM forces one.
static int gr_bm(int rows) {
  const int f = MR_KNOB("MRMT3_ROWS_BM", 0);
  if (f == 64 || f == 128) return f;
  return ceil_div(rows, 128) >= gr_cus() ? 128 : 64;
}

extern "C" int mrmt3_gemm_nt_normbwd_partial_rows(int rows) { return ceil_div(rows, gr_bm(rows)); }

static unsigned gr_grid(int M, int n_ctiles, int bm) {
  const int mt = ceil_div(M, bm);
  return n_ctiles == 2 ? (unsigned)(ceil_div(mt, 8) * 16) : (unsigned)(mt * n_ctiles);
}
static void gr_base(GRParams& P, const void* A, int lda, const void* W, int ldw, int M, int K, int n_ctiles) {
  memset(&P, 0, sizeof(P));
  P.A = (const bf16_t*)A; P.B = (const bf16_t*)W; P.lda = lda; P.ldb = ldw; P.M = M; P.K = K; P.n_ctiles = n_ctiles;
#ifdef MRMT3_DIAG
  P.dbg = mrmt3_diag_env("MRMT3_ROWS_DBG");
  P.trace = g_rows_trace;
  P.skew_fine = MR_KNOB("MRMT3_ROWS_SKEW_FINE", 0) / 10;
  if (P.dbg & 4) {
    int n64 = -1, n128 = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n64, gemm_rows_kernel<GR_ADDNORM, false, false, 64>, 512, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n128, gemm_rows_kernel<GR_ADDNORM, false, false, 128>, 512, 0);
    fprintf(stderr, "gemm_rows: workgroups per CU: %d (64-row tiles, LDS %d B), %d (128-row tiles, LDS %d B)\n", n64,
            GRCfg<64>::LDS, n128, GRCfg<128>::LDS);
  }
#endif
}
#define GR_LAUNCH(EPI, RI, RO, rows, nct)                                                                                      \
  do {                                                                                                                         \
    if (gr_bm(rows) == 128)                                                                                                    \
      hipLaunchKernelGGL((gemm_rows_kernel<EPI, RI, RO, 128>), dim3(gr_grid(rows, nct, 128)), dim3(512), 0, (hipStream_t)stream, P); \
    else                                                                                                                       \
      hipLaunchKernelGGL((gemm_rows_kernel<EPI, RI, RO, 64>), dim3(gr_grid(rows, nct, 64)), dim3(512), 0, (hipStream_t)stream, P);   \
  } while (0)

extern "C" int mrmt3_gemm_nt_addnorm(const void* A, int lda, const void* W, int ldw, int rows, int K, const float* x0,
                                     const float* w_norm, float eps, float* x1, void* xn_bf16, float* rstd, float p_drop,
                                     uint64_t seed, const int32_t* step_dev, uint32_t stream_y, uint32_t stream_out,
                                     int out_drop, void* stream) {
  MR_CHECK_ARG(A && W && x0 && w_norm && xn_bf16, "gemm_nt_addnorm: null pointer");
  MR_CHECK_ARG(mrmt3_gemm_rows_ok(rows, 512, K, lda, ldw), "gemm_nt_addnorm: unsupported shape rows=%d K=%d lda=%d ldw=%d", rows, K, lda, ldw);
  MR_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)x0 % 16) == 0 && ((uintptr_t)xn_bf16 % 16) == 0 &&
               ((uintptr_t)x1 % 16) == 0 && ((uintptr_t)w_norm % 16) == 0, "gemm_nt_addnorm: operands must be 16-byte aligned");
  GRParams P;
  gr_base(P, A, lda, W, ldw, rows, K, 1);
  P.xin = x0; P.wn = w_norm; P.eps = eps; P.x1 = x1; P.xn = (bf16_t*)xn_bf16; P.rstd = rstd;
  P.d0 = make_drop(p_drop, seed, stream_y, step_dev);
  P.d1 = make_drop(p_drop, seed, stream_out, step_dev);
  P.out_drop = out_drop;
  P.cache_mode = MR_KNOB("MRMT3_NORM_NT", 3);             // the stand-alone kernel's switch: streaming x0 load / x1 store
  GR_LAUNCH(GR_ADDNORM, false, false, rows, 1);
  MR_CHECK_LAUNCH("gemm_nt_addnorm");
  mrmt3_count(MRMT3_CNT_GEMM_NT_ADDNORM);
  return MRMT3_OK;
}

extern "C" int mrmt3_gemm_nt_normbwd(const void* A, int lda, const void* WT, int ldw, int rows, int K, const void* dres,
                                     int dres_dtype, const float* x1, const float* rstd, const float* w_norm, void* dx1,
                                     int dx1_dtype, void* dy_bf16, float p_drop, uint64_t seed, const int32_t* step_dev,
                                     uint32_t stream_y, void* workspace, size_t workspace_bytes, void* stream) {
  MR_CHECK_ARG(A && WT && dres && x1 && rstd && w_norm && dx1, "gemm_nt_normbwd: null pointer");
  MR_CHECK_ARG(mrmt3_gemm_rows_ok(rows, 512, K, lda, ldw), "gemm_nt_normbwd: unsupported shape rows=%d K=%d lda=%d ldw=%d", rows, K, lda, ldw);
  MR_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)WT % 16) == 0 && ((uintptr_t)dres % 16) == 0 && ((uintptr_t)x1 % 16) == 0 &&
               ((uintptr_t)dx1 % 16) == 0 && ((uintptr_t)dy_bf16 % 16) == 0 && ((uintptr_t)w_norm % 16) == 0,
               "gemm_nt_normbwd: operands must be 16-byte aligned");
  MR_CHECK_ARG((dres_dtype == MRMT3_F32 || dres_dtype == MRMT3_BF16) && (dx1_dtype == MRMT3_F32 || dx1_dtype == MRMT3_BF16),
               "gemm_nt_normbwd: unknown dtype code (dres %d, dx1 %d)", dres_dtype, dx1_dtype);
  const int n_part = mrmt3_gemm_nt_normbwd_partial_rows(rows);
  MR_CHECK_ARG(workspace == nullptr || workspace_bytes >= ((size_t)n_part + DW_CHUNKS) * 512 * sizeof(float) + 8 * sizeof(int),
               "gemm_nt_normbwd: workspace too small");
  GRParams P;
  gr_base(P, A, lda, WT, ldw, rows, K, 1);
  P.xin = x1; P.rstd = (float*)rstd; P.wn = w_norm; P.dres = dres; P.dx1 = dx1; P.dy = (bf16_t*)dy_bf16;
  P.dw_part = (float*)workspace;
  P.dw_counters = workspace ? (int*)((float*)workspace + ((size_t)n_part + DW_CHUNKS) * 512) : nullptr;
  P.d0 = make_drop(p_drop, seed, stream_y, step_dev);
  const bool ri = dres_dtype == MRMT3_BF16, ro = dx1_dtype == MRMT3_BF16;
  if (ri && ro) GR_LAUNCH(GR_NORMBWD, true, true, rows, 1);
  else if (ri) GR_LAUNCH(GR_NORMBWD, true, false, rows, 1);
  else if (ro) GR_LAUNCH(GR_NORMBWD, false, true, rows, 1);
  else GR_LAUNCH(GR_NORMBWD, false, false, rows, 1);
  MR_CHECK_LAUNCH("gemm_nt_normbwd");
  mrmt3_count(MRMT3_CNT_GEMM_NT_NORMBWD);
  return MRMT3_OK;
}

extern "C" int mrmt3_gemm_nt_geglubwd(const void* dy, int ldy, const void* WT, int ldw, const void* h, void* dh, int rows,
                                      int dff, int K, float p_drop, uint64_t seed, const int32_t* step_dev,
                                      uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(dy && WT && h && dh, "gemm_nt_geglubwd: null pointer");
  MR_CHECK_ARG(mrmt3_gemm_rows_ok(rows, dff, K, ldy, ldw), "gemm_nt_geglubwd: unsupported shape rows=%d dff=%d K=%d", rows, dff, K);
  MR_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)WT % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)dh % 16) == 0,
               "gemm_nt_geglubwd: operands must be 16-byte aligned");
  GRParams P;
  gr_base(P, dy, ldy, WT, ldw, rows, K, dff / 512);
  P.h = (const bf16_t*)h; P.dh = (bf16_t*)dh; P.dff = dff;
  P.d0 = make_drop(p_drop, seed, stream_id, step_dev);
  P.cache_mode = MR_KNOB("MRMT3_GEGLUB_NT", 0);
  GR_LAUNCH(GR_GEGLUBWD, false, false, rows, dff / 512);
  MR_CHECK_LAUNCH("gemm_nt_geglubwd");
  mrmt3_count(MRMT3_CNT_GEMM_NT_GEGLUBWD);
  return MRMT3_OK;
}
