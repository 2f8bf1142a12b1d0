// TN GEMM (weight gradients), second generation: slab[split][N1][N2] = A[rows of the split, N1]^T . B[rows, N2],
// bf16 operands, f32 accumulation — the autograd backward of every nn.Linear(bias=False) of the T5 stack with respect
// to its weight (dW = dY^T X; reference: torch autograd through HF T5Attention / T5DenseGatedGeluDense, models/t5.py:51,72).
//
// Same schedule as gemm8.hip (read that header first): 512 threads, the two waves of a SIMD one barrier apart, a
// 256 x 256 tile of dW per workgroup, 64 token rows per K step, two LDS buffers of four 16-KiB half-tiles, one half-tile
// of LDS-DMA requested per phase two K steps ahead, counted vmcnt.  What differs:
//   * the reduction runs over the TOKEN rows, the slow index of both operands: a half-tile is stored
//     [64 tokens][128 features] (256-byte rows, 32-byte units XOR-swizzled by token & 7 on the LDS-DMA source address) and
//     the MFMA fragments are read with ds_read_b64_tr_b16 (hardware transpose), two reads per 16 x 32 fragment — the
//     layout and addressing of gemm.hip's first TN kernel, which this one replaces for the large shapes;
//   * one workgroup = one (tile, token range): a K loop of 16 .. 128 steps, then a 256 KiB f32 slab; the slabs of all
//     the weight gradients of a step are summed later in one launch (mrmt3_tn_reduce_sites), in split order (bitwise
//     reproducible, no atomics);
//   * ragged N1 / N2 (384, 768, 1152): the last tile is shifted to end at N and stores only what the tile before it
//     does not own.
#include <stdlib.h>

#include "common.h"

#define T8_OOB 0x7FFF0000
#define T8_HALF 16384
#define T8_BUF (4 * T8_HALF)
#define T8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

struct T8Params {
  const bf16_t* A;     // dY [M][lda]   (N1 features)
  const bf16_t* B;     // X  [M][ldb]   (N2 features)
  float* slab;         // [splits][N1][N2]
  int lda, ldb, M, N1, N2;
  int tiles_n2, n_tiles, n_splits, rows_per_split;
};

__device__ __forceinline__ void t8_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
// one 16 x 32 MFMA operand: features of one 32-byte unit, the lane group's 4 + 4 token rows (rows R and R + 16).
// The transposed reads are INLINE ASM on purpose: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 while an LDS-DMA load is outstanding (it cannot prove the two do not alias; plain
// ds_read_b128 of the same array are left alone), which drains the whole prefetch pipeline every phase.  The asm form is
// invisible to that pass; its results are waited for by the explicit lgkmcnt(0) + sched_barrier at the segment boundary.
template <int OFF>
__device__ __forceinline__ bf16x8 t8_frag(unsigned addr) {
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 16 * 256) : "memory");
  const u32x4 r = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(T8Params P) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * T8_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = w >> 2, wc = w & 3;
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;

  // work item: all tiles of one token range run on one XCD (they share the range's rows of dY and X in its L2)
  int tile, split;
  {
    const int n = (int)gridDim.x, per = n >> 3;
    const int wi = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wi >= P.n_tiles * P.n_splits) return;
    split = wi / P.n_tiles;
    tile = wi - split * P.n_tiles;
  }
  const int t1 = tile / P.tiles_n2, t2 = tile - t1 * P.tiles_n2;
  const int a0 = min(t1 * 256, P.N1 - 256), b0 = min(t2 * 256, P.N2 - 256);
  const int rmin = t1 * 256, cmin = t2 * 256;             // rows / columns below belong to the neighbouring tile
  const int nk = P.rows_per_split >> 6;                    // even (rows_per_split is a multiple of 128)

  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)(((size_t)(P.M - 1) * P.lda + P.N1) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)(((size_t)(P.M - 1) * P.ldb + P.N2) * 2), 0x00020000);

  // ---- LDS-DMA source offsets.  A half-tile = 64 token rows x 256 B; one wave-instruction = 1 KiB = 4 token rows;
  // lane p fills (token 4*piece + p/16, 16-byte chunk p%16) with the token row's chunk whose 32-byte unit is
  // (p%16 / 2) ^ (token & 7).  Virtual feature v = 8 * chunk of the half-tile:
  //   A role (dY):  HA0 = row tiles 0-3 of both wave groups: feature a0 + 128 (v / 64) + v % 64          [+ 64: HA1]
  //   B role (X) :  HB0 = column tiles 0,1 of the four wave columns: feature b0 + 64 (v / 32) + v % 32   [+ 32: HB1]
  unsigned voffA[2], voffB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int tk = (i * 8 + w) * 4 + (lane >> 4), cp = lane & 15;
    const int c = ((((cp >> 1) ^ (tk & 7)) << 1) | (cp & 1)), v = c * 8;
    voffA[i] = (unsigned)((tk * P.lda + (v >> 6) * 128 + (v & 63)) * 2);
    voffB[i] = (unsigned)((tk * P.ldb + (v >> 5) * 64 + (v & 31)) * 2);
  }
  const int piece0 = w * 1024;
  auto load_a = [&](int buf, int half, int soff) {
    unsigned char* base = lds + buf * T8_BUF + half * T8_HALF + piece0;
#pragma unroll
    for (int i = 0; i < 2; ++i) t8_dma16(ra, base + i * 8192, voffA[i] + (half ? 128u : 0u), soff);
  };
  auto load_b = [&](int buf, int half, int soff) {
    unsigned char* base = lds + buf * T8_BUF + (2 + half) * T8_HALF + piece0;
#pragma unroll
    for (int i = 0; i < 2; ++i) t8_dma16(rb, base + i * 8192, voffB[i] + (half ? 64u : 0u), soff);
  };
  int l_k = 0;
  int l_sa = (split * P.rows_per_split * P.lda + a0) * 2, l_sb = (split * P.rows_per_split * P.ldb + b0) * 2;
  const int step_a = 64 * P.lda * 2, step_b = 64 * P.ldb * 2;
  auto cursor_next = [&]() {
    ++l_k;
    if (l_k < nk) { l_sa += step_a; l_sb += step_b; }
    else { l_sa = l_sb = T8_OOB; }                          // past the token range: zero fill, no memory traffic
  };

  // ---- fragment addresses: token row ks*32 + 4 fg + fq (and + 16), unit U stored at U ^ (row & 7), 8-byte piece fp
  const int rrow = fg * 4 + fq, rx = rrow & 7;
  const int f_lane = rrow * 256 + (fp >> 1) * 16 + (fp & 1) * 8;
  const int a_addr = f_lane + (((g * 4) ^ rx) << 5);                        // ^ (i << 5) for row tile i, + ks * 8192
  const int b_addr = 2 * T8_HALF + f_lane + (((wc * 2) ^ rx) << 5);         // ^ (j << 5) for column tile j

  bf16x8 af[4][2], bf_[4][2];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
  auto read_a = [&](int buf, int half) {
    const unsigned p = lds0 + buf * T8_BUF + half * T8_HALF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = t8_frag<0>(p + (a_addr ^ (i << 5)));
      af[i][1] = t8_frag<8192>(p + (a_addr ^ (i << 5)));
    }
  };
  auto read_b = [&](int buf, int half) {
    const unsigned p = lds0 + buf * T8_BUF + half * T8_HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf_[half * 2 + j][0] = t8_frag<0>(p + (b_addr ^ (j << 5)));
      bf_[half * 2 + j][1] = t8_frag<8192>(p + (b_addr ^ (j << 5)));
    }
  };
  // X fragments are the MFMA's first operand: a lane ends up with 4 consecutive n2 columns of one n1 row per tile
  auto mma = [&](int rh, int ch) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[rh * 4 + i][ch * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[ch * 2 + j][ks], af[i][ks], acc[rh * 4 + i][ch * 2 + j], 0, 0, 0);
  };
#define T8_SEG_END()                            \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_barrier();                 \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_setprio(1)
#define T8_MMA_END()                            \
  __builtin_amdgcn_s_setprio(0);                \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_barrier();                 \
  __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: K steps 0 and (half of) 1
  load_a(0, 0, l_sa); load_b(0, 0, l_sb); load_b(0, 1, l_sb); load_a(0, 1, l_sa);
  cursor_next();
  load_a(1, 0, l_sa); load_b(1, 0, l_sb);
  T8_VMCNT(8);
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one barrier behind group 0

  auto kstep = [&](int buf) {
    read_a(buf, 0);
    read_b(buf, 0);
    load_b(buf ^ 1, 1, l_sb);
    T8_VMCNT(8);
    T8_SEG_END();
    mma(0, 0);
    T8_MMA_END();
    read_b(buf, 1);
    load_a(buf ^ 1, 1, l_sa);
    T8_VMCNT(8);
    T8_SEG_END();
    mma(0, 1);
    T8_MMA_END();
    read_a(buf, 1);
    cursor_next();
    load_a(buf, 0, l_sa);
    T8_SEG_END();
    mma(1, 1);
    T8_MMA_END();
    load_b(buf, 0, l_sb);
    T8_VMCNT(8);
    T8_SEG_END();
    mma(1, 0);
    T8_MMA_END();
  };
  for (int u = 0; u < nk; u += 2) {
    kstep(0);
    kstep(1);
  }
  T8_VMCNT(0);                                               // the switched-off requests of the last steps
  if (g == 0) __builtin_amdgcn_s_barrier();                  // pairs with group 1's extra barrier at the start

  // ---- epilogue: lane holds slab[n1 = row tile i, row fr][n2 = 16 ct + 4 fg + r].  The lanes of rows r and r + 8
  // exchange halves (DPP row rotate) so that a store instruction writes 8 rows x 128 contiguous bytes.
  if (b0 + wc * 64 < cmin || a0 + g * 128 < rmin) return;    // owned by the neighbouring tile (wave-uniform)
  float* out = P.slab + (size_t)split * P.N1 * P.N2;
  const bool up = fr >= 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row1 = a0 + g * 128 + i * 16 + (fr & 7), row2 = row1 + 8;
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      const f32x4 lo = acc[i][2 * cp], hi = acc[i][2 * cp + 1];
      const f32x4 send = up ? lo : hi;
      f32x4 recv;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        recv[e] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send[e]), 0x128, 0xf, 0xf, false));
      const f32x4 v1 = up ? recv : lo, v2 = up ? hi : recv;
      float* p = out + b0 + wc * 64 + cp * 32 + (up ? 16 : 0) + fg * 4;
      __builtin_nontemporal_store(v1, (f32x4*)(p + (size_t)row1 * P.N2));
      __builtin_nontemporal_store(v2, (f32x4*)(p + (size_t)row2 * P.N2));
    }
  }
}

// Plan shared with gemm.hip's mrmt3_gemm_tn*: returns 1 when this kernel takes the shape.
int mrmt3_tn8_plan(int M, int N1, int N2, int* tiles, int* splits, int* rows_per_split) {
  if (M < 8192 || N1 < 256 || N2 < 256 || N1 % 8 != 0 || N2 % 8 != 0) return 0;
  if ((size_t)M * (size_t)(N1 > N2 ? N1 : N2) * 2 >= 0x7FFF0000ull) return 0;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int t = ceil_div(N1, 256) * ceil_div(N2, 256);
  int s = cus / t;                                           // one workgroup per CU (128 KiB of LDS each)
  if (s < 1) s = 1;
  int rps = ceil_div(ceil_div(M, s), 128) * 128;
  if (rps < 1024) rps = 1024;                                // at least 16 K steps behind one 256 KiB slab
  const int sp = ceil_div(M, rps);
  // Shapes this kernel is measured to win on (profiles/r02_gemm_tn_ab.txt: w_wi x1.64, w_lm x1.72, w_wo x1.30,
  // w_qkv x1.22, e_wi x1.13): the chip is filled (>= 90 % of the CUs get a workgroup) and at most 15 % of the MFMAs go
  // to the overlap of a shifted last tile (N = 384 would redo a third).  MRMT3_TN8_ALL=1 (tuning) takes every shape.
  const char* force = getenv("MRMT3_TN8_ALL");
  if (!(force && force[0] == '1')) {
    if (t * sp * 10 < cus * 9) return 0;
    if ((double)(ceil_div(N1, 256) * 256) * (ceil_div(N2, 256) * 256) > 1.15 * (double)N1 * N2) return 0;
  }
  *tiles = t; *splits = sp; *rows_per_split = rps;
  return 1;
}

int mrmt3_tn8_launch(const void* A, int lda, const void* B, int ldb, float* slab, int M, int N1, int N2, int tiles,
                     int splits, int rps, hipStream_t s) {
  T8Params P;
  P.A = (const bf16_t*)A; P.B = (const bf16_t*)B; P.slab = slab;
  P.lda = lda; P.ldb = ldb; P.M = M; P.N1 = N1; P.N2 = N2;
  P.tiles_n2 = ceil_div(N2, 256); P.n_tiles = tiles; P.n_splits = splits; P.rows_per_split = rps;
  hipLaunchKernelGGL(gemm_tn8_kernel, dim3((unsigned)((tiles * splits + 7) & ~7)), dim3(512), 0, s, P);
  return 0;
}
