// K2/K4/K6/K7/K9/K10 — bias-free Linear layers on gfx950 MFMA.
//
// Stands for every nn.Linear(bias=False) in the reference's T5 stack (HF T5Attention q/k/v/o,
// T5DenseGatedGeluDense wi_0/wi_1/wo, models/t5.py:51 proj, :72 lm_head) and their autograd
// backward.  Three products cover forward, dgrad and wgrad:
//   NT  C[M,N]   = A[M,K]  . B[N,K]^T     y = x W^T ; dx = dy (W^T)^T with W^T kept pre-transposed
//   TN  C[N1,N2] = A[M,N1]^T . B[M,N2]    dW = dy^T x, reduction over the M (token) rows
//
// NT kernel (gemm_nt_kernel, templated on the wave grid): a wave owns a 64x64 output sub-tile = 4x4
// v_mfma_f32_16x16x32_bf16 tiles; two instantiations are launched — 256x256 (16 waves, 2 x 64 KiB stages of
// 128-B K-steps, one workgroup per CU) when N is a multiple of 256, 256x128 (8 waves, 3 x 24 KiB stages of
// 64-B K-steps, two workgroups per CU) otherwise.  Operands are staged L2 -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write: the register-staged version was LDS-write bound)
// behind raw s_barrier + counted vmcnt.  The LDS image of a wave-instruction is linear (lane x 16 B), so the
// bank swizzle is applied to each lane's SOURCE address and again on the ds_read_b128 fragment reads, which are
// conflict-free.  Workgroup ids are remapped so the 8 XCDs each walk a contiguous run of (m-tile, n-tile) pairs
// and an A row-panel stays in one XCD's L2; the epilogue transposes each wave's tile through the idle staging
// LDS and stores whole rows with 16 bytes per lane.  f32 inputs use v_mfma_f32_16x16x4_f32 (exact f32 FMA
// chains) on the same LDS image.
//
// TN kernel: 128x128 tile of dW per workgroup, 64 token rows per step; operand tiles are stored
// row-major [64][128] (256-B rows, 32-B units XOR-swizzled by row & 7, again via the glds source
// address) and consumed through ds_read_b64_tr_b16 (hardware transpose) so both MFMA operands get their 8
// consecutive reduction indices per lane; the token range is split over workgroups — a split count that fills
// whole resident waves of 512 workgroups — into f32 slabs that a second kernel sums in a fixed order
// (bitwise reproducible, no atomics).
#include <stdlib.h>

#include "common.h"

#define TILE 128
#define ROWB 128  // bytes per LDS row

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective remap: blocks with equal bid%8 share an XCD (observed round-robin dispatch; only a
  // speed assumption).  Give each XCD a contiguous chunk of the logical tile order.
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + slot;
}

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

__device__ __forceinline__ f32x4 mfma_step(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_step(f32x4 a, f32x4 b, f32x4 c) {
  // 16 k-values per fragment pair: lane group g holds k = 4g..4g+3; MFMA s pairs element s of
  // every lane (k-slot g <-> k = 4g+s) — A and B use the same permutation.
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// NT main kernel, templated on the wave grid WM x WN (each wave owns a 64x64 output sub-tile = 4x4 MFMA
// tiles): workgroup tile = 64*WM x 64*WN.  K step = 64 bytes of each row (32 bf16 / 16 f32 = one MFMA
// k-step), 3 LDS stages, two K-steps of global_load_lds in flight across a raw s_barrier (counted vmcnt,
// never drained in the loop).
// Measured on MI355X: this kernel family runs at a constant ~9 TB/s of L2->LDS staging traffic whatever
// the tile (128^2: 64 FLOP/B -> 570 TF, 256x128: 85 FLOP/B -> 720-820 TF), so the lever is FLOP per staged
// byte = BM*BN/(BM+BN):
//   <4,4>  256x256 (128 FLOP/B), 16 waves, 2 x 64 KiB stages (128-B K steps), one workgroup per CU — N % 256 == 0
//   <4,2>  256x128 ( 85 FLOP/B),  8 waves, 3 x 24 KiB stages, two workgroups per CU — the other shapes
// LDS rows are 64 B; chunk c (16 B) of row r is stored at c ^ (2 * ((r >> 3) & 1)), applied on the
// glds source address and on the ds_read_b128 fragment reads (conflict-free for the b128 lane groups).
// pipeline of the 256x256 tile: 128-B K steps x 2 stages measured 2-9 % faster than 64 B x 3 stages
#ifndef NT_RB_BIG
#define NT_RB_BIG 128
#define NT_ST_BIG 2
#endif

template <typename TIN, typename TOUT, bool ACCUM, int WM, int WN, int RB, int NST>
__global__ __launch_bounds__(64 * WM * WN, 4) void gemm_nt_kernel(const TIN* __restrict__ A, int lda,
                                                                   const TIN* __restrict__ B, int ldb,
                                                                   TOUT* __restrict__ C, int ldc, int M, int N,
                                                                   int K, int tiles_n) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NWAVES = WM * WN;
  constexpr int STAGE_BYTES = (BM + BN) * RB;
  constexpr int RPI = 1024 / RB;              // rows per 1-KiB wave-instruction (16 or 8)
  constexpr int CPR = RB / 16;                // 16-B chunks per row (4 or 8)
  constexpr int A_INSTR = BM / RPI, TOT_INSTR = (BM + BN) / RPI, PER_WAVE = TOT_INSTR / NWAVES;
  static_assert(TOT_INSTR % NWAVES == 0, "staging instructions must divide evenly over the waves");
  static_assert((RB == 64 && (NST == 3 || NST == 4)) || (RB == 128 && NST == 2), "supported pipelines: 64 B x 3/4 stages, 128 B x 2 stages");
  static_assert(RB != 64 || PER_WAVE == 2 || PER_WAVE == 3, "vmcnt immediates below assume 2 or 3 staging instructions per wave");
  constexpr int EPC = 16 / sizeof(TIN);       // elements per 16-B chunk
  constexpr int BK = RB / sizeof(TIN);        // elements per K step
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_BYTES];  // one object
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int wr = uw / WN, wc = uw % WN;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

  // staging: one wave-instruction = 1 KiB = 16 rows x 64 B; instruction q < A_INSTR fills A rows 16q..,
  // the rest B rows.  Wave w issues q = w, w + NWAVES, ...  Lane p fills (row = 16q + p/4, c' = p%4)
  // with global chunk c' ^ (2*((row>>3)&1)).
  // (128-B rows: 8 rows per instruction, chunk c stored at c ^ (row & 7).)
  const int srow = lane / CPR;
  const int schunk = (RB == 64) ? ((lane & 3) ^ (((lane >> 5) & 1) << 1)) : ((lane & 7) ^ (lane >> 3));
  const TIN* gsrc[PER_WAVE];
  int ldst[PER_WAVE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int q = uw + NWAVES * i;
    if (q < A_INSTR) gsrc[i] = A + (size_t)min(m0 + q * RPI + srow, M - 1) * lda + schunk * EPC;
    else gsrc[i] = B + (size_t)min(n0 + (q - A_INSTR) * RPI + srow, N - 1) * ldb + schunk * EPC;
    ldst[i] = q * 1024;
  }
  auto stage = [&](int buf, int k0) {
    unsigned char* base = lds + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + ldst[i]), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int nk = K / BK;
  if constexpr (RB == 64) {
    // fragment byte offsets inside a stage (row & 8 is the same for rows i*16 + fr, i = 0..3)
    const int fswz = (fg ^ (((fr >> 3) & 1) << 1)) << 4;
    const int offa = (wr * 64 + fr) * RB + fswz, offb = BM * RB + (wc * 64 + fr) * RB + fswz;
    // NST stages, NST-1 K-steps of global_load_lds in flight
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
      if (st < nk) stage(st, st * BK);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed once at most the loads of the (up to NST-2) later tiles are still outstanding
      const int later = min(NST - 2, nk - 1 - kt);
      if (later >= 2) {
        if (PER_WAVE == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else if (later == 1) {
        if (PER_WAVE == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // everyone's part of tile kt landed; everyone is done reading stage (kt-1)%NST
      if (kt + NST - 1 < nk) stage(cur == 0 ? NST - 1 : cur - 1, (kt + NST - 1) * BK);   // (kt+NST-1)%NST == (cur-1)%NST
      const unsigned char* ls = lds + cur * STAGE_BYTES;
      typename Frag<TIN>::type af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *(const typename Frag<TIN>::type*)(ls + offa + i * 16 * RB);
        bfr[i] = *(const typename Frag<TIN>::type*)(ls + offb + i * 16 * RB);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_step(af[i], bfr[j], acc[i][j]);
      cur = cur == NST - 1 ? 0 : cur + 1;
    }
  } else {
    // 128-B rows, two stages: 32 MFMAs per wave per barrier; the next K step is issued right after the
    // barrier and lands during this step's MFMAs
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
      const unsigned char* la = lds + cur * STAGE_BYTES;
      const unsigned char* lb = la + BM * RB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        typename Frag<TIN>::type af[4], bfr[4];
        const int c = ks * 4 + fg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rowa = wr * 64 + i * 16 + fr, rowb = wc * 64 + i * 16 + fr;
          af[i] = *(const typename Frag<TIN>::type*)(la + rowa * RB + ((c ^ (rowa & 7)) << 4));
          bfr[i] = *(const typename Frag<TIN>::type*)(lb + rowb * RB + ((c ^ (rowb & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma_step(af[i], bfr[j], acc[i][j]);
      }
    }
  }

  // epilogue.  A lane holds C[row = 16i + 4fg + r][col = 16j + fr]: storing from there means 2/4-byte
  // stores in 32/64-byte pieces, and the store-issue tail then costs more than the whole K loop.
  // Instead each wave transposes its 64x64 tile through its own slice of the (now idle) staging LDS,
  // EP_ROWS rows at a time, and writes whole rows with 16 bytes per lane.
  __builtin_amdgcn_s_barrier();                       // every wave is done reading the last stage
  constexpr int EP_LD = 68;                           // floats per LDS row (64 + 4 pad, 16-B aligned rows)
  constexpr int EP_ROWS = (NST * STAGE_BYTES / NWAVES >= 32 * EP_LD * 4) ? 32 : 16;
  static_assert(NST * STAGE_BYTES / NWAVES >= EP_ROWS * EP_LD * 4, "epilogue slice does not fit");
  float* wl = (float*)(lds + uw * (EP_ROWS * EP_LD * 4));
  const int rbase = m0 + wr * 64, cbase = n0 + wc * 64;
#pragma unroll
  for (int part = 0; part < 64 / EP_ROWS; ++part) {
#pragma unroll
    for (int ii = 0; ii < EP_ROWS / 16; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          wl[(ii * 16 + fg * 4 + r) * EP_LD + j * 16 + fr] = acc[part * (EP_ROWS / 16) + ii][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS round trip: in order, no barrier needed
    if constexpr (sizeof(TOUT) == 2) {
      // 64 cols = 128 B per row -> 8 lanes per row, 8 rows per instruction
#pragma unroll
      for (int it = 0; it < EP_ROWS / 8; ++it) {
        const int lr = it * 8 + (lane >> 3), lc = (lane & 7) * 8;
        const f32x4 a = *(const f32x4*)(wl + lr * EP_LD + lc), b = *(const f32x4*)(wl + lr * EP_LD + lc + 4);
        const int row = rbase + part * EP_ROWS + lr, col = cbase + lc;
        if (row < M && col + 8 <= N)
          *(u32x4*)(C + (size_t)row * ldc + col) =
              u32x4{pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w)};
      }
    } else {
      // 64 cols = 256 B per row -> 16 lanes per row, 4 rows per instruction
#pragma unroll
      for (int it = 0; it < EP_ROWS / 4; ++it) {
        const int lr = it * 4 + (lane >> 4), lc = (lane & 15) * 4;
        f32x4 a = *(const f32x4*)(wl + lr * EP_LD + lc);
        const int row = rbase + part * EP_ROWS + lr, col = cbase + lc;
        if (row < M && col + 4 <= N) {
          f32x4* p = (f32x4*)(C + (size_t)row * ldc + col);
          if (ACCUM) a += *p;
          *p = a;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next part overwrites
  }
}

// gemm8.hip: the ping-pong kernel for the tall bf16 shapes (returns 0 when the shape is not its)
int mrmt3_gemm_nt8_try(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       int out_dtype, int accumulate, hipStream_t s);
static bool use_gemm8() { return MR_KNOB("MRMT3_GEMM8", 1) != 0; }        // tuning / A-B switch only

template <typename TIN, typename TOUT, bool ACCUM>
static int launch_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     hipStream_t s) {
  if constexpr (sizeof(TIN) == 2) {
    if (use_gemm8() && mrmt3_gemm_nt8_try(A, lda, B, ldb, C, ldc, M, N, K, sizeof(TOUT) == 2 ? MRMT3_BF16 : MRMT3_F32,
                                           ACCUM ? 1 : 0, s)) {
      MR_CHECK_LAUNCH("gemm_nt8");
      mrmt3_count(MRMT3_CNT_GEMM_NT8);
      return MRMT3_OK;
    }
  }
  if (N % 256 == 0 && M >= 2048 && (K * sizeof(TIN)) % NT_RB_BIG == 0) {
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL((gemm_nt_kernel<TIN, TOUT, ACCUM, 4, 4, NT_RB_BIG, NT_ST_BIG>), dim3((unsigned)(tiles_m * tiles_n)), dim3(1024), 0, s,
                       (const TIN*)A, lda, (const TIN*)B, ldb, (TOUT*)C, ldc, M, N, K, tiles_n);
  } else {
    const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 128);
    hipLaunchKernelGGL((gemm_nt_kernel<TIN, TOUT, ACCUM, 4, 2, 64, 3>), dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, s,
                       (const TIN*)A, lda, (const TIN*)B, ldb, (TOUT*)C, ldc, M, N, K, tiles_n);
  }
  MR_CHECK_LAUNCH("gemm_nt");
  mrmt3_count(MRMT3_CNT_GEMM_NT_TILE);
  return MRMT3_OK;
}

extern "C" int mrmt3_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                             int K, int in_dtype, int out_dtype, int accumulate, void* stream) {
  MR_CHECK_ARG(A && B && C, "gemm_nt: null pointer");
  MR_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_nt: bad sizes M=%d N=%d K=%d", M, N, K);
  const int esz = in_dtype == MRMT3_BF16 ? 2 : 4;
  MR_CHECK_ARG((K * esz) % 64 == 0, "gemm_nt: K*elem_size must be a multiple of 64 bytes (K=%d)", K);
  MR_CHECK_ARG((lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0, "gemm_nt: row strides must be 16-byte multiples");
  {
    const int osz = out_dtype == MRMT3_BF16 ? 2 : 4;
    MR_CHECK_ARG((ldc * osz) % 16 == 0 && (N * osz) % 16 == 0 && ((uintptr_t)C % 16) == 0,
                 "gemm_nt: C rows must be 16-byte aligned (ldc, N multiples of %d)", 16 / osz);
    MR_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_nt: A and B must be 16-byte aligned");
  }
  MR_CHECK_ARG(!(accumulate && out_dtype != MRMT3_F32), "gemm_nt: accumulate needs f32 output");
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == MRMT3_BF16) {
    if (out_dtype == MRMT3_BF16) return launch_nt<bf16_t, bf16_t, false>(A, lda, B, ldb, C, ldc, M, N, K, s);
    if (accumulate) return launch_nt<bf16_t, float, true>(A, lda, B, ldb, C, ldc, M, N, K, s);
    return launch_nt<bf16_t, float, false>(A, lda, B, ldb, C, ldc, M, N, K, s);
  } else if (in_dtype == MRMT3_F32) {
    MR_CHECK_ARG(out_dtype == MRMT3_F32, "gemm_nt: f32 inputs need f32 output");
    if (accumulate) return launch_nt<float, float, true>(A, lda, B, ldb, C, ldc, M, N, K, s);
    return launch_nt<float, float, false>(A, lda, B, ldb, C, ldc, M, N, K, s);
  }
  mrmt3_set_error("gemm_nt: unknown dtype %d", in_dtype);
  return MRMT3_ERR_INVALID_ARG;
}

int mrmt3_gemm_nt8_splitk_try(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                              int in_dtype, int out_dtype, int accumulate, void* workspace, size_t workspace_bytes,
                              hipStream_t s);

// mrmt3_gemm_nt with a caller-owned scratch buffer (mrmt3_gemm_nt_workspace_bytes): short inputs with a long K run
// split over K (gemm8.hip); everything else — and a NULL / too small workspace — is mrmt3_gemm_nt itself.
extern "C" int mrmt3_gemm_nt_ws(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                int in_dtype, int out_dtype, int accumulate, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (workspace && A && B && C && M > 0 && N > 0 && K > 0 && in_dtype == MRMT3_BF16 &&
      (out_dtype == MRMT3_BF16 || out_dtype == MRMT3_F32) && !(accumulate && out_dtype != MRMT3_F32) &&
      (lda * 2) % 16 == 0 && (ldb * 2) % 16 == 0 && (ldc * (out_dtype == MRMT3_BF16 ? 2 : 4)) % 16 == 0 &&
      ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0 && use_gemm8()) {
    if (mrmt3_gemm_nt8_splitk_try(A, lda, B, ldb, C, ldc, M, N, K, in_dtype, out_dtype, accumulate, workspace,
                                  workspace_bytes, (hipStream_t)stream)) {
      MR_CHECK_LAUNCH("gemm_nt (split K)");
      return MRMT3_OK;
    }
  }
  return mrmt3_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, in_dtype, out_dtype, accumulate, stream);
}

// ------------------------------------------------------------------------------------------------
// TN: C[N1,N2] (+)= A[M,N1]^T . B[M,N2]   (bf16 in, f32 out)
// ------------------------------------------------------------------------------------------------
#define TN_ROWS 64    // token rows per step
#define TN_STRIDE 256  // bytes per LDS row (128 bf16); 32-B units XOR-swizzled by (row & 7)

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];  // source of out-of-range rows

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb,
                                                         float* __restrict__ slab, int M, int N1, int N2,
                                                         int tiles_n2, int n_tiles, int n_splits, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][TN_ROWS * TN_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order: all output tiles of one token-range (split) run on the same XCD at the same time,
  // so the 128-row slices of dY and X they share are served by that XCD's L2 instead of being fetched
  // by all eight.  (blocks with equal blockIdx%8 share an XCD; speed-only assumption.)
  int tile, split;
  {
    // work items in split-major order (all tiles of token range 0, then of range 1, ...); XCD x (= blockIdx % 8) takes
    // the contiguous eighth [x*n/8, (x+1)*n/8) of them, in dispatch order.  The grid is padded to a multiple of 8.
    const int n = (int)gridDim.x, per = n >> 3;
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const int n_work = n_tiles * n_splits;
    if (w >= n_work) return;
    split = w / n_tiles;
    tile = w - split * n_tiles;
  }
  const int a0 = (tile / tiles_n2) * TILE, b0 = (tile % tiles_n2) * TILE;
  const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);

  // staging: a tile (64 rows x 256 B = 16 KiB) is 16 wave-instructions of 1 KiB (4 rows); wave w issues
  // 4w..4w+3.  Lane p fills LDS (row = 4t + p/16, c' = p%16) with global 16-B chunk
  // c = (((c'>>1) ^ (row&7)) << 1) | (c'&1)  (32-B units swizzled by row).
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  int srow[4], scol_a[4], scol_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = uw * 4 + i;
    srow[i] = t * 4 + (lane >> 4);
    const int cp = lane & 15;
    const int c = ((((cp >> 1) ^ (srow[i] & 7)) << 1) | (cp & 1));
    scol_a[i] = min(a0 + c * 8, N1 - 8);
    scol_b[i] = min(b0 + c * 8, N2 - 8);
  }
  auto stage = [&](int buf, int mrow0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mrow0 + srow[i];
      const bool ok = m < mend;
      const void* pa = ok ? (const void*)(A + (size_t)m * lda + scol_a[i]) : (const void*)(g_zero_page + (lane & 15) * 16);
      const void* pb = ok ? (const void*)(B + (size_t)m * ldb + scol_b[i]) : (const void*)(g_zero_page + (lane & 15) * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                       (__attribute__((address_space(3))) void*)(&lds[buf][0][(uw * 4 + i) * 1024]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb,
                                       (__attribute__((address_space(3))) void*)(&lds[buf][1][(uw * 4 + i) * 1024]), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: 16-lane group g, lane = 4*q+p inside the group supplies the
  // address of row (4g+q) [+16 for the second half of the k-slots], columns 4p..4p+3
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  const int nsteps = ceil_div(max(mend - mbeg, 0), TN_ROWS);
  if (nsteps > 0) stage(0, mbeg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) stage(cur ^ 1, mbeg + (st + 1) * TN_ROWS);
    const unsigned char* la = &lds[cur][0][0];
    const unsigned char* lb = &lds[cur][1][0];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int rbase = ks * 32 + fg * 4 + fq;
      bf16x8 af[4], bfr[4];
      // element column n = w*64 + i*16 + fp*4 -> chunk c = w*8 + i*2 + (fp>>1); the 32-B unit (c>>1)
      // is XOR-ed with (row & 7) (rbase and rbase+16 share row & 7)
      const int rx = rbase & 7, sub = (fp >> 1) * 16 + (fp & 1) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cola = (((wr * 4 + i) ^ rx) << 5) + sub, colb = (((wc * 4 + i) ^ rx) << 5) + sub;
        s16x4 a_lo = lds_tr16(la + rbase * TN_STRIDE + cola);
        s16x4 a_hi = lds_tr16(la + (rbase + 16) * TN_STRIDE + cola);
        s16x4 b_lo = lds_tr16(lb + rbase * TN_STRIDE + colb);
        s16x4 b_hi = lds_tr16(lb + (rbase + 16) * TN_STRIDE + colb);
        af[i] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
        bfr[i] = {b_lo.x, b_lo.y, b_lo.z, b_lo.w, b_hi.x, b_hi.y, b_hi.z, b_hi.w};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_step(af[i], bfr[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: transpose each wave's 64x64 tile through LDS (two 32-row halves) and store whole rows
  // with 16 bytes per lane (see gemm_nt_kernel).
  float* out = slab + (size_t)split * N1 * N2;
  const int fr = lane & 15;
  constexpr int EP_LD = 68;
  __syncthreads();
  float* wl = (float*)(&lds[0][0][0] + uw * (32 * EP_LD * 4));
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) wl[(ii * 16 + fg * 4 + r) * EP_LD + j * 16 + fr] = acc[half * 2 + ii][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int lr = it * 4 + (lane >> 4), lc = (lane & 15) * 4;
      const f32x4 a = *(const f32x4*)(wl + lr * EP_LD + lc);
      const int row = a0 + wr * 64 + half * 32 + lr, col = b0 + wc * 64 + lc;
      if (row < N1 && col + 4 <= N2) *(f32x4*)(out + (size_t)row * N2 + col) = a;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next half overwrites
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int ldc, int N1, int N2,
                                   int splits, int accumulate) {
  const size_t n = (size_t)N1 * N2;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (size_t)gridDim.x * blockDim.x * 4) {
    // eight slabs requested before the first is added (same summation order, so the result is unchanged): the kernel
    // is a few hundred small workgroups and lives on how many loads each thread keeps in flight
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(slab + (size_t)(k + u) * n + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += *(const f32x4*)(slab + (size_t)k * n + i);
    const int row = (int)(i / N2), col = (int)(i % N2);
    float* p = C + (size_t)row * ldc + col;
    if (accumulate) s += *(const f32x4*)p;
    *(f32x4*)p = s;
  }
}

// The slab sums of MANY weight gradients in one launch.  A training step has 45-50 weight-gradient GEMMs; summing each
// one's slabs right behind it cost a launch of a few hundred small workgroups apiece (19.5 us x 90 per step in round 1,
// 1.8 ms).  With every site keeping its slabs until a gradient bucket is due, one launch of ~45 K workgroups streams
// all of them (same per-element summation order as slab_reduce_kernel, so the result is bit-identical).
struct TnSite {             // = mrmt3_tn_site (include/mrmt3_hip.h)
  unsigned long long slabs, C;
  int N1, N2, ldc, splits, accumulate, block0, pad0, pad1;
};
__global__ __launch_bounds__(256) void slab_reduce_sites_kernel(const TnSite* __restrict__ sites, int n_sites) {
  // site of this workgroup: block0 is ascending.  (A scan of the table in global memory is n_sites DEPENDENT scalar
  // loads per workgroup — with 45 sites that was most of the kernel's 586 us; one wave-wide load + ballot instead.)
  __shared__ int s_site;
  if (threadIdx.x < 64) {
    int cnt = 0;
    for (int base = 0; base < n_sites; base += 64) {
      const int i = base + (int)threadIdx.x;
      const bool le = i < n_sites && sites[i].block0 <= (int)blockIdx.x;
      cnt += __popcll(__ballot(le));
    }
    if (threadIdx.x == 0) s_site = cnt - 1;
  }
  __syncthreads();
  const TnSite st = sites[s_site];
  const float* __restrict__ slab = (const float*)st.slabs;
  float* __restrict__ C = (float*)st.C;
  const size_t n = (size_t)st.N1 * st.N2;
  const size_t i = ((size_t)(blockIdx.x - st.block0) * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= st.splits; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(slab + (size_t)(k + u) * n + i));
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < st.splits; ++k) s += __builtin_nontemporal_load((const f32x4*)(slab + (size_t)k * n + i));
  const int row = (int)(i / st.N2), col = (int)(i % st.N2);
  float* p = C + (size_t)row * st.ldc + col;
  if (st.accumulate) s += *(const f32x4*)p;
  *(f32x4*)p = s;
}

// gemm_tn8.hip: the ping-pong kernel for the large weight gradients
int mrmt3_tn8_plan(int M, int N1, int N2, int* tiles, int* splits, int* rows_per_split);
int mrmt3_tn8_launch(const void* A, int lda, const void* B, int ldb, float* slab, int M, int N1, int N2, int tiles,
                     int splits, int rps, hipStream_t s);
static bool use_tn8(int M, int N1, int N2, int* tiles, int* splits, int* rps) {
  if (MR_KNOB("MRMT3_TN8", 1) == 0) return false;   // tuning / A-B switch only
  return mrmt3_tn8_plan(M, N1, N2, tiles, splits, rps) != 0;
}

static void tn_plan(int M, int N1, int N2, int* tiles, int* splits, int* rows_per_split) {
  if (use_tn8(M, N1, N2, tiles, splits, rows_per_split)) return;
  const int t = ceil_div(N1, TILE) * ceil_div(N2, TILE);
  const int steps = ceil_div(M, TN_ROWS);
  const int max_s = steps / 8 > 0 ? steps / 8 : 1;  // at least 8 steps (512 rows) per split
  // 512 workgroups are resident at once (2 per CU): a split count that fills exactly one such wave beats
  // every other choice measured (w_wo 650 -> 785 TF, w_wi 661 -> 698, w_o 394 -> 500), and exactly three
  // waves beat 1.5 (w_lm 554 -> 604).  Any count will do for the XCD mapping (contiguous eighths, see kernel).
  // (a multiple of 8 — whole token ranges per XCD — is preferred when one lies in the efficient range)
  for (int k = 1; k <= 3; ++k)
    for (int pass = 0; pass < 2; ++pass)
    for (int c = 512 * k / t; c >= 1 && c * t * 100 >= 512 * k * (k == 1 ? 93 : 99); --c)
      if ((pass == 1 || c <= 8 || c % 8 == 0) && c <= max_s && ceil_div(M, ceil_div(steps, c) * TN_ROWS) == c) {
        *tiles = t; *splits = c; *rows_per_split = ceil_div(steps, c) * TN_ROWS;
        return;
      }
  int s = ceil_div(768, t);               // otherwise ~3 workgroups per CU (each split costs a slab round trip)
  s = s < 1 ? 1 : s;
  if (s > max_s) s = max_s;
  int rps = ceil_div(steps, s) * TN_ROWS;
  s = ceil_div(M, rps);                   // keep the split count exact
  *tiles = t; *splits = s; *rows_per_split = rps;
}

extern "C" size_t mrmt3_gemm_tn_workspace_bytes(int M, int N1, int N2) {
  int t, s, r;
  tn_plan(M, N1, N2, &t, &s, &r);
  return (size_t)s * N1 * N2 * sizeof(float);
}

extern "C" int mrmt3_gemm_tn_splits(int M, int N1, int N2) {
  int t, s, r;
  tn_plan(M, N1, N2, &t, &s, &r);
  return s;
}

extern "C" int mrmt3_gemm_tn_partial(const void* A, int lda, const void* B, int ldb, int M, int N1, int N2,
                                     void* slabs, size_t slab_bytes, void* stream) {
  MR_CHECK_ARG(A && B && slabs, "gemm_tn_partial: null pointer");
  MR_CHECK_ARG(M > 0 && N1 >= 8 && N2 >= 8, "gemm_tn_partial: bad sizes");
  MR_CHECK_ARG(N1 % 8 == 0 && N2 % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0,
               "gemm_tn_partial: N1/lda/ldb must be multiples of 8, N2 of 4");
  int tiles, splits, rps;
  tn_plan(M, N1, N2, &tiles, &splits, &rps);
  MR_CHECK_ARG(slab_bytes >= (size_t)splits * N1 * N2 * sizeof(float), "gemm_tn_partial: slab buffer too small");
  {
    int t8, s8, r8;
    if (use_tn8(M, N1, N2, &t8, &s8, &r8)) {
      mrmt3_tn8_launch(A, lda, B, ldb, (float*)slabs, M, N1, N2, t8, s8, r8, (hipStream_t)stream);
      MR_CHECK_LAUNCH("gemm_tn8");
      mrmt3_count(MRMT3_CNT_TN8);
      return MRMT3_OK;
    }
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((tiles * splits + 7) & ~7)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (float*)slabs, M, N1, N2, ceil_div(N2, TILE), tiles,
                     splits, rps);
  MR_CHECK_LAUNCH("gemm_tn_partial");
  mrmt3_count(MRMT3_CNT_TN_TILE);
  return MRMT3_OK;
}

extern "C" int mrmt3_tn_reduce_sites(const void* sites_dev, int n_sites, int total_blocks, void* stream) {
  MR_CHECK_ARG(sites_dev && n_sites > 0 && total_blocks > 0, "tn_reduce_sites: bad arguments");
  static_assert(sizeof(TnSite) == sizeof(mrmt3_tn_site), "descriptor layout");
  hipLaunchKernelGGL(slab_reduce_sites_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const TnSite*)sites_dev, n_sites);
  MR_CHECK_LAUNCH("tn_reduce_sites");
  return MRMT3_OK;
}

extern "C" int mrmt3_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1,
                             int N2, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  MR_CHECK_ARG(A && B && C && workspace, "gemm_tn: null pointer");
  MR_CHECK_ARG(M > 0 && N1 >= 8 && N2 >= 8, "gemm_tn: bad sizes");
  MR_CHECK_ARG(N1 % 8 == 0 && N2 % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
               "gemm_tn: N1/lda/ldb must be multiples of 8, N2/ldc of 4");
  int tiles, splits, rps;
  tn_plan(M, N1, N2, &tiles, &splits, &rps);
  MR_CHECK_ARG(workspace_bytes >= (size_t)splits * N1 * N2 * sizeof(float), "gemm_tn: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int t8, s8, r8;
  if (use_tn8(M, N1, N2, &t8, &s8, &r8)) {
    mrmt3_tn8_launch(A, lda, B, ldb, (float*)workspace, M, N1, N2, t8, s8, r8, s);
    mrmt3_count(MRMT3_CNT_TN8);
  } else {
    mrmt3_count(MRMT3_CNT_TN_TILE);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((tiles * splits + 7) & ~7)), dim3(256), 0, s, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, (float*)workspace, M, N1, N2, ceil_div(N2, TILE), tiles, splits, rps);
  }
  MR_CHECK_LAUNCH("gemm_tn");
  const size_t n = (size_t)N1 * N2;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, C, ldc,
                     N1, N2, splits, accumulate);
  MR_CHECK_LAUNCH("gemm_tn reduce");
  return MRMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// exact-f32 TN for the fp32 training / parity path (the reference trains at `precision: 32`):
// C[N1,N2] (+)= A[M,N1]^T . B[M,N2], plain f32 FMAs in row order — one workgroup per 64x64 tile of C, every thread a
// 4x4 block, 16 token rows staged per step.  No split over M: one writer per element, a fixed summation order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                          int ldb, float* __restrict__ C, int ldc, int M, int N1, int N2,
                                                          int accumulate) {
  __shared__ float As[16][64], Bs[16][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n1_0 = blockIdx.y * 64, n2_0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;      // staging: row lr of 16, 4 consecutive columns
  for (int m0 = 0; m0 < M; m0 += 16) {
    const int m = m0 + lr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c1 = n1_0 + lc + e, c2 = n2_0 + lc + e;
      As[lr][lc + e] = (m < M && c1 < N1) ? A[(size_t)m * lda + c1] : 0.f;
      Bs[lr][lc + e] = (m < M && c2 < N2) ? B[(size_t)m * ldb + c2] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = n1_0 + ty * 4 + i;
    if (r >= N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = n2_0 + tx * 4 + j;
      if (c < N2) {
        float* p = C + (size_t)r * ldc + c;
        *p = accumulate ? *p + acc[i][j] : acc[i][j];
      }
    }
  }
}

extern "C" int mrmt3_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                                 int accumulate, void* stream) {
  MR_CHECK_ARG(A && B && C, "gemm_tn_f32: null pointer");
  MR_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0 && lda >= N1 && ldb >= N2 && ldc >= N2, "gemm_tn_f32: bad sizes");
  hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3((unsigned)ceil_div(N2, 64), (unsigned)ceil_div(N1, 64)), dim3(256), 0,
                     (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N1, N2, accumulate);
  MR_CHECK_LAUNCH("gemm_tn_f32");
  mrmt3_count(MRMT3_CNT_TN_F32);
  return MRMT3_OK;
}
