// NT GEMM, second generation: C[M,N] (+)= A[M,K] . B[N,K]^T for the tall shapes of the training step
// (M = 16 K .. 64 K token rows, N and K = 384 .. 2048), bf16 operands, f32 accumulation.
//
// Stands for the same nn.Linear(bias=False) layers as gemm.hip (HF T5Attention q/k/v/o, T5DenseGatedGeluDense
// wi_0/wi_1/wo, models/t5.py:51 proj, :72 lm_head, and their dgrad).  What round 1's kernel left on the table
// (profiles/r01_pmc_gemm_sq.txt: waves parked on s_waitcnt / s_barrier 39 %, MFMA pipe busy ~30 %) was its
// schedule — one barrier per K step, the fragment reads of a step starting only behind that barrier, two
// workgroup-wide stages — not its traffic.  This kernel is built around the schedule instead:
//
//   * 512 threads = 8 waves = 2 per SIMD; the two waves of a SIMD (wave w and w + 4: "group" 0 and 1) run the SAME
//     program one s_barrier apart, so while one of them issues its 16 MFMAs of a phase the other reads fragments
//     from LDS and issues the next LDS-DMA loads (ping-pong; MI355X_MICROARCH "two waves per SIMD").
//   * a 256 x 256 (or 128 x 256) tile, K step 64 (128-byte rows), two LDS buffers of four 16-KiB half-tiles each.
//     A half-tile is what ONE phase reads: HA0 / HA1 = the first / second half of every wave's rows, HB0 / HB1 the
//     first / second half of every wave's columns.  A K step is four phases of 16 (8) MFMAs per wave:
//         ph1 reads HA0, HB0   (r0 x c0)      ph2 reads HB1   (r0 x c1)
//         ph3 reads HA1        (r1 x c1)      ph4 reads none  (r1 x c0)
//     and every phase issues ONE half-tile of LDS-DMA (buffer_load ... lds) two K steps ahead, into the slot whose
//     last reader finished two phases ago:  ph1: HB1(u+1)  ph2: HA1(u+1)  ph3: HA0(u+2)  ph4: HB0(u+2).
//     Each load therefore has ~5 phases to land, and the only wait in the loop is a counted vmcnt that leaves the
//     four youngest half-tiles in flight (never 0), placed one phase before the half-tile is read.
//   * persistent workgroups: the K-step stream runs on across output tiles (the first K steps of the next tile are
//     in flight while the current one finishes); XCD x owns a contiguous range of (m-tile, n-tile) pairs.
//   * the MFMAs are issued transposed (weights as the A operand), with the weight rows of a wave's 64 columns
//     permuted on the LDS-DMA source address, so that a lane's accumulators hold runs of consecutive columns of one
//     C row and the four lanes of a row write one contiguous 64-byte segment per store instruction: the epilogue
//     stores straight from registers, 16 bytes per lane, no LDS transposition.
//   * N need only be a multiple of 128: the last column tile is shifted left to end at N and masks the columns the
//     tile before it owns.
#include <stdlib.h>

#include "common.h"

#define G8_OOB 0x7FFF0000
#define G8_HALF 16384
#define G8_BUF (4 * G8_HALF)

struct G8Params {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int lda, ldb, ldc, M, N, K;
  int tiles_n, n_tiles;
  int dbg;             // diagnostics (MRMT3_GEMM8_DBG): 1 no C stores, 2 nt stores, 4 every K step re-reads K step 0 (cache-hot),
                       // 8 no fragment reads, 16 loads switched off (zero fill, no traffic)
  int skew_ticks;      // start delay per (slot % 8), in 10-ns ticks of s_memrealtime (see the kernel)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g8_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

// (kept in a __device__ function: called from a lambda inside the kernel template, the builtin makes hipcc's host
// pass drop the kernel's launch stub without a diagnostic)
__device__ __forceinline__ void g8_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

#define G8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// RT = 16-row tiles per wave (8: 256-row workgroup tile, 4: 128-row).  NST = global stores (and loads, ACCUM) the
// epilogue of one tile issues per lane: they sit in the vmcnt queue behind the loads the next K step waits for.
template <typename TOUT, bool ACCUM, int RT>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(G8Params P) {
  constexpr int GROUP_ROWS = RT * 16;          // rows of one wave group
  constexpr int BM = 2 * GROUP_ROWS;
  constexpr int NA = RT / 4;                   // LDS-DMA instructions per thread and A half-tile (2 or 1)
  constexpr int A_HALF_ROWS = GROUP_ROWS;      // rows in HA0 (= RT*8 per group x 2 groups)
  constexpr int YOUNG = 2 * NA + 4;            // loads of the four youngest half-tiles (2 A halves, 2 B halves)
  constexpr int NST = (sizeof(TOUT) == 2 ? RT * 2 : RT * 4) * (ACCUM ? 2 : 1);
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * G8_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = w >> 2, wc = w & 3;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- this workgroup's tiles: XCD x (= blockIdx % 8) owns a contiguous range, its workgroups stride through it
  const int nx = (int)gridDim.x >> 3;                       // workgroups per XCD (grid is a multiple of 8)
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int q = P.n_tiles >> 3, r = P.n_tiles & 7;
  const int xstart = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xcount = q + (xcd < r ? 1 : 0);
  const int my_tiles = slot < xcount ? (xcount - slot + nx - 1) / nx : 0;
  if (my_tiles == 0) return;
  const int nk = P.K >> 6;
  // De-phase the workgroups.  Every tile costs the same, so without this all 256 workgroups reach their epilogues
  // together: 32 MB of C leave the chip in one burst at the HBM write rate (measured: 6.2 us per tile with the MFMA
  // pipes idle, against 11 us for the tile's K loop at K = 512), then HBM idles through the next K loop.  Spread over
  // a tile period the same bytes need less than half the write bandwidth and drain behind the next tile's MFMAs.
  if (P.skew_ticks > 0 && (slot & 7)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)(slot & 7) * P.skew_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }

  auto tile_origin = [&](int it, int& m0, int& n0, int& cmin) {
    const int t = xstart + slot + it * nx;
    const int mt = t / P.tiles_n, nt = t - mt * P.tiles_n;
    m0 = mt * BM;
    n0 = min(nt * 256, P.N - 256);
    cmin = nt * 256;                                        // columns below belong to the tile on the left
  };

  const __amdgpu_buffer_rsrc_t ra = g8_rsrc(P.A, ((size_t)(P.M - 1) * P.lda + P.K) * 2);
  const __amdgpu_buffer_rsrc_t rb = g8_rsrc(P.B, ((size_t)(P.N - 1) * P.ldb + P.K) * 2);

  // ---- LDS-DMA source offsets.  One wave-instruction = 1 KiB = 8 rows x 128 B of a half-tile; lane p fills
  // (row 8*piece + p/8, chunk p%8) with the row's 16-byte chunk (p%8) ^ (p/8) (bank swizzle on the SOURCE side).
  // A half-tile row rr <-> tile row (rr / (RT*8)) * GROUP_ROWS + rr % (RT*8)   [+ RT*8 for HA1]
  // B half-tile row rr <-> weight row 64*(rr/32) + colmap(ct, rr%16)   [+ 32 for HB1], see below
  unsigned voffA[NA], voffB[2];
  const int sw = (lane & 7) ^ (lane >> 3);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int rr = (i * 8 + w) * 8 + (lane >> 3);
    const int arow = (rr / (RT * 8)) * GROUP_ROWS + rr % (RT * 8);
    voffA[i] = (unsigned)(arow * P.lda * 2 + sw * 16);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = (i * 8 + w) * 8 + (lane >> 3);
    // column tile ct = 2*half + j (j = (rr/16)%2), index i = rr%16 of wave column block rr/32 <-> C column
    //   bf16 out: 32*(ct/2) + 8*(i/4) + 4*(ct%2) + i%4   one 16-byte store = 8 columns, the 4 lanes of a row 64 B apart... contiguous
    //   f32 out : 16*ct + i                              one 16-byte store = 4 columns of column tile ct
    // either way the four lanes that share a C row write one contiguous 64-byte segment per store instruction (the
    // first version left 16-byte holes between them: 64 separate 16-byte requests per instruction, and the epilogue
    // cost 5.5 us per tile against 11 us for its K loop)
    const int j = (rr >> 4) & 1, ii = rr & 15;
    const int cm = sizeof(TOUT) == 2 ? (ii >> 2) * 8 + j * 4 + (ii & 3) : j * 16 + ii;
    const int brow = (rr >> 5) * 64 + cm;
    voffB[i] = (unsigned)(brow * P.ldb * 2 + sw * 16);
  }
  const unsigned a1_delta = (unsigned)(RT * 8 * P.lda * 2), b1_delta = (unsigned)(32 * P.ldb * 2);
  const int piece0 = w * 1024;                              // LDS offset of this wave's piece inside a half-tile

  auto load_a = [&](int buf, int half, int soff) {          // half 0: HA0, 1: HA1
    if (P.dbg & 16) soff = G8_OOB;
    unsigned char* base = lds + buf * G8_BUF + half * G8_HALF + piece0;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      g8_dma16(ra, base + i * 8192, voffA[i] + (half ? a1_delta : 0u), soff);
  };
  auto load_b = [&](int buf, int half, int soff) {          // half 0: HB0, 1: HB1
    if (P.dbg & 16) soff = G8_OOB;
    unsigned char* base = lds + buf * G8_BUF + (2 + half) * G8_HALF + piece0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      g8_dma16(rb, base + i * 8192, voffB[i] + (half ? b1_delta : 0u), soff);
  };

  // ---- load cursor: the K step whose half-tiles are being requested (two ahead of the one being computed)
  int l_it = 0, l_k = 0, l_sa, l_sb;
  {
    int m0, n0, cm;
    tile_origin(0, m0, n0, cm);
    l_sa = m0 * P.lda * 2;
    l_sb = n0 * P.ldb * 2;
  }
  auto cursor_next = [&]() {
    ++l_k;
    if (l_k < nk) { if (!(P.dbg & 4)) { l_sa += 128; l_sb += 128; } return; }
    l_k = 0;
    ++l_it;
    if (l_it < my_tiles) {
      int m0, n0, cm;
      tile_origin(l_it, m0, n0, cm);
      l_sa = m0 * P.lda * 2;
      l_sb = n0 * P.ldb * 2;
    } else {
      l_sa = l_sb = G8_OOB;                                  // switched off: zero fill, no memory traffic
    }
  };

  // ---- fragment read offsets (bytes inside a buffer); ks = 1 is the same address ^ 64
  const int fsw = (fg ^ (fr & 7)) << 4;
  const int a_off0 = g * (RT * 8 * 128) + fr * 128 + fsw;    // + half * G8_HALF + (rt % (RT/2)) * 2048
  const int b_off0 = 2 * G8_HALF + wc * 4096 + fr * 128 + fsw;  // + half * G8_HALF + (ct & 1) * 2048
  const int a_off1 = a_off0 ^ 64, b_off1 = b_off0 ^ 64;

  constexpr int RH = RT / 2;                                 // row tiles per half
  bf16x8 af[RH][2], bf_[4][2];
  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto read_a = [&](int buf, int half) {
    if (P.dbg & 8) return;
    const unsigned char* p = lds + buf * G8_BUF + half * G8_HALF;
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      af[i][0] = *(const bf16x8*)(p + a_off0 + i * 2048);
      af[i][1] = *(const bf16x8*)(p + a_off1 + i * 2048);
    }
  };
  auto read_b = [&](int buf, int half) {
    if (P.dbg & 8) return;
    const unsigned char* p = lds + buf * G8_BUF + half * G8_HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf_[half * 2 + j][0] = *(const bf16x8*)(p + b_off0 + j * 2048);
      bf_[half * 2 + j][1] = *(const bf16x8*)(p + b_off1 + j * 2048);
    }
  };
  // 16 (8) MFMAs: row half rh x column half ch, both k-steps.  Weights are the A operand: D = W_frag . X_frag^T
  auto mma = [&](int rh, int ch, int ks) {
#pragma unroll
    for (int i = 0; i < RH; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[rh * RH + i][ch * 2 + j] =
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[ch * 2 + j][ks], af[i][ks], acc[rh * RH + i][ch * 2 + j], 0, 0, 0);
  };
  // (Tried: issuing a phase's LDS-DMA requests between its MFMAs instead of in the read segment.  A wave issues in
  // order and an LDS-DMA instruction takes 60+ cycles to issue, so the MFMAs behind it wait: barriers + MFMAs alone
  // went from 1.03 to 1.30 us per K step.  In the read segment the same issue time runs beside the partner's MFMAs.)
#define G8_MMA(rh, ch) \
  mma(rh, ch, 0);      \
  mma(rh, ch, 1)
#define G8_SEG_END()                            \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_barrier();                 \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_setprio(1)
#define G8_MMA_END()                            \
  __builtin_amdgcn_s_setprio(0);                \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_barrier();                 \
  __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: K steps 0 and (half of) 1
  load_a(0, 0, l_sa); load_b(0, 0, l_sb); load_b(0, 1, l_sb); load_a(0, 1, l_sa);
  cursor_next();
  load_a(1, 0, l_sa); load_b(1, 0, l_sb);
  if (YOUNG == 8) G8_VMCNT(8); else G8_VMCNT(6);            // HA0(0), HB0(0) landed; four half-tiles behind them
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one barrier behind group 0

  int c_it = 0, c_k = 0, c_m0, c_n0, c_cmin;
  tile_origin(0, c_m0, c_n0, c_cmin);
  bool after_epi = false;
  const int total = my_tiles * nk;
  for (int u = 0; u < total; ++u) {
    const int buf = u & 1;
    // the waits: everything older than the four youngest half-tiles (and, in the K step behind an epilogue, than the
    // epilogue's stores) has landed
#define G8_WAIT()                                                             \
  do {                                                                        \
    if (after_epi) {                                                          \
      if (YOUNG + NST == 24) G8_VMCNT(24); else if (YOUNG + NST == 40) G8_VMCNT(40);       \
      else if (YOUNG + NST == 14) G8_VMCNT(14); else if (YOUNG + NST == 22) G8_VMCNT(22);  \
      else if (YOUNG == 8) G8_VMCNT(8); else G8_VMCNT(6);                     \
    } else if (YOUNG == 8) G8_VMCNT(8); else G8_VMCNT(6);                     \
  } while (0)
    // ph1: r0 x c0
    read_a(buf, 0);
    read_b(buf, 0);
    load_b(buf ^ 1, 1, l_sb);
    G8_WAIT();
    G8_SEG_END();
    G8_MMA(0, 0);
    G8_MMA_END();
    // ph2: r0 x c1
    read_b(buf, 1);
    load_a(buf ^ 1, 1, l_sa);
    G8_WAIT();
    G8_SEG_END();
    G8_MMA(0, 1);
    G8_MMA_END();
    // ph3: r1 x c1
    read_a(buf, 1);
    cursor_next();
    load_a(buf, 0, l_sa);
    G8_SEG_END();
    G8_MMA(1, 1);
    G8_MMA_END();
    // ph4: r1 x c0
    load_b(buf, 0, l_sb);
    G8_WAIT();
    G8_SEG_END();
    G8_MMA(1, 0);
    G8_MMA_END();
    after_epi = false;
    if (++c_k == nk) {
      // ---- epilogue: lane holds C[row][16 consecutive columns] per row tile
      TOUT* C = (TOUT*)P.C;
      const int col = c_n0 + wc * 64 + (sizeof(TOUT) == 2 ? fg * 8 : fg * 4);
      const bool col_ok = c_n0 + wc * 64 >= c_cmin;
      const bool full = c_m0 + BM <= P.M && c_n0 >= c_cmin;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int row = c_m0 + g * GROUP_ROWS + i * 16 + fr;
        if (row < P.M && col_ok && !(P.dbg & 1) && !((P.dbg & 32) && i >= RT / 2)) {
          if constexpr (sizeof(TOUT) == 2) {
            u32x4 lo = {pack_bf2(acc[i][0][0], acc[i][0][1]), pack_bf2(acc[i][0][2], acc[i][0][3]),
                        pack_bf2(acc[i][1][0], acc[i][1][1]), pack_bf2(acc[i][1][2], acc[i][1][3])};
            u32x4 hi = {pack_bf2(acc[i][2][0], acc[i][2][1]), pack_bf2(acc[i][2][2], acc[i][2][3]),
                        pack_bf2(acc[i][3][0], acc[i][3][1]), pack_bf2(acc[i][3][2], acc[i][3][3])};
            bf16_t* p = (bf16_t*)C + (size_t)row * P.ldc + col;
            if (P.dbg & 2) {
              __builtin_nontemporal_store(lo, (u32x4*)p);
              __builtin_nontemporal_store(hi, (u32x4*)(p + 32));
            } else {
              *(u32x4*)p = lo;
              *(u32x4*)(p + 32) = hi;
            }
          } else {
            float* p = (float*)C + (size_t)row * P.ldc + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 v = acc[i][j];
              if (ACCUM) v += *(const f32x4*)(p + j * 16);
              *(f32x4*)(p + j * 16) = v;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      c_k = 0;
      ++c_it;
      if (c_it < my_tiles) tile_origin(c_it, c_m0, c_n0, c_cmin);
      after_epi = full;             // the allowance below counts the stores of a tile stored in full
    }
  }
  G8_VMCNT(0);
  if (g == 0) __builtin_amdgcn_s_barrier();                  // pairs with group 1's extra barrier at the start
}

static int g8_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// Returns 1 if the shape was launched on the second-generation kernel, 0 if the caller should use gemm.hip's.
int mrmt3_gemm_nt8_try(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       int out_dtype, int accumulate, hipStream_t s) {
  if (M < 4096 || N < 256 || N % 128 != 0 || K % 64 != 0 || K < 128) return 0;
  if (((size_t)M * lda + K) * 2 >= 0x7FFF0000ull || ((size_t)N * ldb + K) * 2 >= 0x7FFF0000ull) return 0;
  G8Params P;
  P.A = (const bf16_t*)A; P.B = (const bf16_t*)B; P.C = C;
  P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.M = M; P.N = N; P.K = K;
  P.tiles_n = ceil_div(N, 256);
  const int cus = g8_cus() & ~7;
  const int tiles256 = ceil_div(M, 256) * P.tiles_n;
  const bool small = tiles256 < cus;                         // not enough 256-row tiles to fill the chip: 128-row tiles
  P.n_tiles = small ? ceil_div(M, 128) * P.tiles_n : tiles256;
  {
    // one tile ~ nk x 1.4 us (0.7 for 128-row tiles) + the stores; eight start phases across that period
    { const char* e = getenv("MRMT3_GEMM8_DBG"); P.dbg = e ? atoi(e) : 0; }
    static int skew_pct = -1;
    if (skew_pct < 0) { const char* e = getenv("MRMT3_GEMM8_SKEW"); skew_pct = e ? atoi(e) : 0; }
    const double tile_us = (K / 64) * (small ? 0.7 : 1.4) + 1.5;
    const int tiles_per_wg = ceil_div(P.n_tiles, cus);
    P.skew_ticks = tiles_per_wg >= 2 ? (int)(tile_us * 100.0 / 8.0 * skew_pct / 100.0) : 0;
  }
  int grid = P.n_tiles < cus ? ((P.n_tiles + 7) & ~7) : cus;
  if (grid > P.n_tiles) grid = (P.n_tiles + 7) & ~7;
#define G8_LAUNCH(TOUT, ACC)                                                                                     \
  do {                                                                                                           \
    if (small) hipLaunchKernelGGL((gemm_nt8_kernel<TOUT, ACC, 4>), dim3((unsigned)grid), dim3(512), 0, s, P);   \
    else hipLaunchKernelGGL((gemm_nt8_kernel<TOUT, ACC, 8>), dim3((unsigned)grid), dim3(512), 0, s, P);         \
  } while (0)
  if (out_dtype == MRMT3_BF16) G8_LAUNCH(bf16_t, false);
  else if (accumulate) G8_LAUNCH(float, true);
  else G8_LAUNCH(float, false);
#undef G8_LAUNCH
  return 1;
}
