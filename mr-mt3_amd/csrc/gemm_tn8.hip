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
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

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
// One unit of work: a 256 x 256 tile of one weight gradient over one range of token rows.  Element (n1, n2) of the
// partial sum goes to out[n1 * ldo + n2] (n1, n2 are the gradient's own indices: `out` is pre-offset by the planner).
struct T8Item {
  const bf16_t* A;
  const bf16_t* B;
  float* out;
  int lda, ldb, ldo;
  int M, N1, N2;       // bounds of the operands (rows past M read as zeros)
  int a0, b0;          // tile origin (a shifted last tile starts at N - 256) ...
  int rmin, cmin;      // ... and the first n1 / n2 it owns
  int row0, nk;        // token range: rows row0 .. row0 + 64 nk (nk even)
  int sync_idx, sync_n; // grouped launch: the sync_n items of a shelf wait for each other before they start (0: no wait)
};
static_assert(sizeof(T8Item) == 80, "item layout (the planner writes these records on the host)");
// One gradient tile of a grouped launch: C[owned part of the tile] (+)= sum of its n_part partial tiles, in order.
struct T8RTile {
  const float* slab;   // n_part consecutive [256][256] f32 partial tiles
  float* C;            // the gradient [N1][ldc]
  int ldc, a0, b0, rmin, cmin, n_part, accumulate, pad;
};
static_assert(sizeof(T8RTile) == 48, "reduce record layout");

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

__device__ __forceinline__ void t8_run_item(const T8Item& P, unsigned char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = w >> 2, wc = w & 3;
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;
  const int a0 = P.a0, b0 = P.b0;
  const int rmin = P.rmin, cmin = P.cmin;                  // rows / columns below belong to the neighbouring tile
  const int nk = P.nk;                                     // even

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
  int l_sa = (P.row0 * P.lda + a0) * 2, l_sb = (P.row0 * P.ldb + b0) * 2;
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
  float* out = P.out;
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
      __builtin_nontemporal_store(v1, (f32x4*)(p + (size_t)row1 * P.ldo));
      __builtin_nontemporal_store(v2, (f32x4*)(p + (size_t)row2 * P.ldo));
    }
  }
}

__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(T8Params P) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * T8_BUF];
  // work item: all tiles of one token range run on one XCD (they share the range's rows of dY and X in its L2)
  const int n = (int)gridDim.x, per = n >> 3;
  const int wi = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (wi >= P.n_tiles * P.n_splits) return;
  const int split = wi / P.n_tiles, tile = wi - split * P.n_tiles;
  const int t1 = tile / P.tiles_n2, t2 = tile - t1 * P.tiles_n2;
  T8Item I;
  I.A = P.A; I.B = P.B; I.out = P.slab + (size_t)split * P.N1 * P.N2;
  I.lda = P.lda; I.ldb = P.ldb; I.ldo = P.N2; I.M = P.M; I.N1 = P.N1; I.N2 = P.N2;
  I.a0 = min(t1 * 256, P.N1 - 256); I.b0 = min(t2 * 256, P.N2 - 256);
  I.rmin = t1 * 256; I.cmin = t2 * 256;
  I.row0 = split * P.rows_per_split; I.nk = P.rows_per_split >> 6;     // (rows_per_split is a multiple of 128)
  I.sync_idx = 0; I.sync_n = 0;
  t8_run_item(I, lds);
}

// ---- grouped form: the weight gradients of SEVERAL linear layers in one launch ---------------------------------------
// A training step has 45-50 weight gradients; launched one by one, each has to cut its token rows into 16-32 ranges to
// occupy 256 CUs (a gradient is only 2-16 tiles), so every tile is written as 16-32 f32 partial tiles and read back by
// the reduce: 3 GB each way per step, and 16-64 K steps behind every 256-KiB epilogue.  The backward keeps the operands
// of a whole gradient bucket alive instead (memory is not the constraint on this part) and launches its gradients
// together: items (gradient, token range, tile) of nearly equal length, planned on the host (mrmt3_tn_group_plan) so that
// the item count is a whole number of rounds over the CUs — 2-8 partial tiles per gradient tile, 100-400 K steps per item.
// Workgroup l (logical index: XCD-major, so that the tiles of one token range share an L2) runs its own list of items
// (longest first; the planner deals the items, longest first, to the least loaded workgroup).
__global__ __launch_bounds__(512, 2) void gemm_tn8_group_kernel(const T8Item* __restrict__ items,
                                                                  const int* __restrict__ cta_start,
                                                                  const int* __restrict__ cta_items, int* sync) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * T8_BUF];
  const int n = (int)gridDim.x, per = n >> 3;
  const int l = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int q1 = cta_start[l + 1];
  for (int q = cta_start[l]; q < q1; ++q) {
    const T8Item I = items[cta_items[q]];
    if (I.sync_n > 1) {
      // The items of a shelf share their operand rows through the XCD's L2 only while they walk the rows together:
      // wait (bounded: this is a performance hint, never a correctness condition — a workgroup that is not resident
      // yet must not hang the others) until the shelf's other items have arrived.  Once in step they stay in step:
      // whoever runs ahead takes the L2 misses.
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync + I.sync_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(sync + I.sync_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < I.sync_n &&
               __builtin_amdgcn_s_memrealtime() - t0 < 3000ull)        // 30 us at 100 MHz
          __builtin_amdgcn_s_sleep(4);
      }
      __syncthreads();
    }
    t8_run_item(I, lds);
  }
}

// sum of the partial tiles of the grouped launch: 64 workgroups per gradient tile, fixed order (bitwise reproducible)
__global__ __launch_bounds__(256) void tn8_group_reduce_kernel(const T8RTile* __restrict__ tiles) {
  const T8RTile t = tiles[blockIdx.x >> 6];
  const int idx = ((int)(blockIdx.x & 63) << 8) + (int)threadIdx.x;    // float4 index inside the tile
  const int r = idx >> 6, c = (idx & 63) << 2;
  const int n1 = t.a0 + r, n2 = t.b0 + c;
  if (n1 < t.rmin || n2 < t.cmin) return;
  const float* __restrict__ sp = t.slab + (size_t)r * 256 + c;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 4 <= t.n_part; k += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(sp + (size_t)(k + u) * 65536));
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
  }
  for (; k < t.n_part; ++k) s += __builtin_nontemporal_load((const f32x4*)(sp + (size_t)k * 65536));
  float* p = t.C + (size_t)n1 * t.ldc + n2;
  if (t.accumulate) s += *(const f32x4*)p;
  *(f32x4*)p = s;
}

// Plan shared with gemm.hip's mrmt3_gemm_tn*: returns 1 when this kernel takes the shape.
int mrmt3_tn8_plan(int M, int N1, int N2, int* tiles, int* splits, int* rows_per_split) {
  // The shifted last tile (columns N - 256 .. N) is addressed in whole half-tiles of 128 features on the dY side and in
  // 64-column fragments on the X side: N1 must be a multiple of 128 and N2 of 64 (the grouped launch's rule,
  // mrmt3_tn_group_ok).  Other widths go to gemm_tn_kernel (found by tests/test_fuzz_gpu.py: N1 = 576 was admitted
  // with % 8 and came back wrong in rows 512..575).
  if (M < 8192 || N1 < 256 || N2 < 256 || N1 % 128 != 0 || N2 % 64 != 0) return 0;
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
  if (MR_KNOB("MRMT3_TN8_ALL", 0) != 1) {
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


// ---- grouped launch: host planner -----------------------------------------------------------------------------------
static int t8_cus() {
  static int hw = 0;
  if (hw == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) hw = prop.multiProcessorCount;
    if (hw <= 0) hw = 256;
  }
  int cus = hw;
  { const int e = MR_KNOB("MRMT3_TN_GROUP_CTAS", 0); if (e >= 8) cus = e; }   // tests / tuning
  return cus & ~7;
}

extern "C" int mrmt3_tn_group_ok(int M, int N1, int N2, int lda, int ldb, int ldc) {
  return M >= 1024 && N1 >= 256 && N2 >= 256 && N1 % 128 == 0 && N2 % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
         ldc % 4 == 0 && ((size_t)M + 128) * (size_t)(lda > ldb ? lda : ldb) * 2 < 0x7FFF0000ull;
}

// Units of 128 token rows (two K steps).  For every candidate item length the gradients are cut into R_s token ranges
// of equal length, the items dealt round-robin over the workgroups in launch order (longest first), and the cost is the
// busiest workgroup's units (+ a prologue/epilogue allowance per item) plus the partial tiles' traffic.
extern "C" int mrmt3_tn_group_plan(const mrmt3_tn_gsite* sites, int n_sites, void* slab_dev, void* table_host,
                                   size_t table_cap, mrmt3_tn_group_info* info) {
  MR_CHECK_ARG(sites && n_sites > 0 && info, "tn_group_plan: bad arguments");
  const int G = t8_cus();
  struct SP { int T1, T2, T, U, R, len; };
  std::vector<SP> sp(n_sites);
  double W = 0;
  for (int s = 0; s < n_sites; ++s) {
    const mrmt3_tn_gsite& S = sites[s];
    MR_CHECK_ARG(S.A && S.B && S.C && mrmt3_tn_group_ok(S.M, S.N1, S.N2, S.lda, S.ldb, S.ldc),
                 "tn_group_plan: site %d (M=%d N1=%d N2=%d) is outside the grouped kernel's shapes", s, S.M, S.N1, S.N2);
    sp[s].T1 = ceil_div(S.N1, 256); sp[s].T2 = ceil_div(S.N2, 256); sp[s].T = sp[s].T1 * sp[s].T2;
    sp[s].U = ceil_div(S.M, 128);
    W += (double)sp[s].T * sp[s].U;
  }
  const double OVH = 2.0, US_PER_UNIT = 2.9, US_PER_ITEM = 0.12;
  std::vector<int> bestR(n_sites, 1);
  // Cost of cutting gradient s into R[s] token ranges.  The tiles of one (gradient, token range) — a GROUP — read the
  // same rows of dY and X, and a 256 x 256 tile needs 128 flop per operand byte: 11 TB/s at full MFMA rate unless the
  // tiles of a group run on one XCD AT THE SAME TIME and share the rows through its L2 (measured with the groups dealt to
  // workgroups one by one, longest first: 17.7 GB fetched against 11 GB of operands; 13.3 GB with the groups kept
  // together).  So the unit of scheduling is the group and the machine is 8 XCDs x (G/8) lanes: a group, longest first,
  // goes to the XCD where as many lanes as it has tiles are free the earliest, and its items start together.
  const int LANES = G / 8;
  struct Grp { int s, r, t0, T, len, xcd; std::vector<int> lanes; };
  std::vector<Grp> groups;
  std::vector<int> placed_order;
  auto pack = [&](const std::vector<int>& R) -> double {
    groups.clear();
    long n_it = 0;
    for (int s = 0; s < n_sites; ++s) {
      const int len = ceil_div(sp[s].U, R[s]), R_s = ceil_div(sp[s].U, len);
      for (int r = 0; r < R_s; ++r)
        for (int t0 = 0; t0 < sp[s].T; t0 += LANES)           // (a gradient of more than G/8 tiles: several groups)
          groups.push_back({s, r, t0, std::min(LANES, sp[s].T - t0), std::min(len, sp[s].U - r * len), 0, {}});
      n_it += (long)R_s * sp[s].T;
    }
    placed_order.resize(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) placed_order[i] = (int)i;
    std::stable_sort(placed_order.begin(), placed_order.end(), [&](int a, int b) {
      return groups[a].len != groups[b].len ? groups[a].len > groups[b].len : groups[a].T > groups[b].T;
    });
    // list scheduling of rigid jobs: a group takes T lanes of ONE XCD from the moment T of its lanes are free
    std::vector<long> free_at((size_t)8 * LANES, 0);
    std::vector<std::pair<long, int>> lf(LANES);
    long mk = 0;
    for (int gi : placed_order) {
      Grp& g2 = groups[gi];
      int bx = 0;
      long bstart = -1, bwaste = 0;
      for (int x = 0; x < 8; ++x) {
        for (int l2 = 0; l2 < LANES; ++l2) lf[l2] = {free_at[(size_t)x * LANES + l2], l2};
        std::partial_sort(lf.begin(), lf.begin() + g2.T, lf.end());
        const long start = lf[g2.T - 1].first;
        long waste = 0;
        for (int l2 = 0; l2 < g2.T; ++l2) waste += start - lf[l2].first;
        if (bstart < 0 || start < bstart || (start == bstart && waste < bwaste)) { bstart = start; bwaste = waste; bx = x; }
      }
      for (int l2 = 0; l2 < LANES; ++l2) lf[l2] = {free_at[(size_t)bx * LANES + l2], l2};
      std::partial_sort(lf.begin(), lf.begin() + g2.T, lf.end());
      g2.xcd = bx;
      g2.lanes.resize(g2.T);
      for (int l2 = 0; l2 < g2.T; ++l2) {
        g2.lanes[l2] = lf[l2].second;
        free_at[(size_t)bx * LANES + lf[l2].second] = bstart + g2.len + (long)OVH;
      }
      mk = std::max(mk, bstart + g2.len + (long)OVH);
    }
    return (double)mk * US_PER_UNIT + (double)n_it * US_PER_ITEM;
  };
  auto evaluate = [&](const std::vector<int>& R) -> double { return pack(R); };
  // the search depends on the shapes only: remembered per (shapes, G) so that a re-plan for new addresses is cheap
  static std::mutex memo_mu;
  static std::map<std::vector<int>, std::vector<int>> memo;
  std::vector<int> memo_key;
  memo_key.push_back(G);
  for (int s = 0; s < n_sites; ++s) { memo_key.push_back(sites[s].M); memo_key.push_back(sites[s].N1); memo_key.push_back(sites[s].N2); }
  bool have = false;
  {
    std::lock_guard<std::mutex> lk(memo_mu);
    auto it = memo.find(memo_key);
    if (it != memo.end()) { bestR = it->second; have = true; }
  }
  if (!have) {
    double best = 1e30;
    auto clampR = [&](int s, int r) { const int rmax = sp[s].U / 8 > 0 ? sp[s].U / 8 : 1; return r < 1 ? 1 : (r > rmax ? rmax : r); };
    // (1) one target item length for every gradient ...
    for (int k = 1; k <= 12; ++k)
      for (int fi = 0; fi <= 40; ++fi) {
        const double Lt = W / ((double)k * G) * (0.70 + 0.015 * fi);
        if (Lt < 8.0 && !(k == 1 && fi == 40)) continue;     // at least 1024 rows behind a 256 KiB partial tile
        std::vector<int> R(n_sites);
        for (int s = 0; s < n_sites; ++s) {
          const int r = clampR(s, (int)((double)sp[s].U / (Lt < 8.0 ? 8.0 : Lt) + 0.5));
          R[s] = ceil_div(sp[s].U, ceil_div(sp[s].U, r));
        }
        const double cost = evaluate(R);
        if (cost < best) { best = cost; bestR = R; }
      }
    // (1b) ... or one number of ranges per token count (a step has two or three: decoder rows, encoder rows, memory rows):
    // every combination — this is what finds "decoder gradients in 4 ranges, encoder gradients whole" when that makes
    // all items the same length and the shelves come out even
    {
      std::vector<int> classes;
      for (int s = 0; s < n_sites; ++s)
        if (std::find(classes.begin(), classes.end(), sp[s].U) == classes.end()) classes.push_back(sp[s].U);
      if (classes.size() <= 3) {
        std::vector<int> rc(classes.size(), 1), R(n_sites);
        for (;;) {
          for (int s = 0; s < n_sites; ++s) {
            const int c = (int)(std::find(classes.begin(), classes.end(), sp[s].U) - classes.begin());
            const int r = clampR(s, rc[c]);
            R[s] = ceil_div(sp[s].U, ceil_div(sp[s].U, r));
          }
          const double cost = evaluate(R);
          if (cost < best) { best = cost; bestR = R; }
          size_t c = 0;
          for (; c < classes.size(); ++c) {
            if (++rc[c] <= std::min(16, std::max(1, classes[c] / 8))) break;
            rc[c] = 1;
          }
          if (c == classes.size()) break;
        }
      }
    }
    // (2) ... then single gradients take one range more or fewer while that shortens the busiest workgroup
    for (int pass = 0; pass < 40; ++pass) {
      int bs = -1, br = 0;
      double bc = best - 1e-9;
      for (int s = 0; s < n_sites; ++s)
        for (int d = -1; d <= 1; d += 2) {
          const int r = clampR(s, bestR[s] + d);
          if (r == bestR[s] || ceil_div(sp[s].U, ceil_div(sp[s].U, r)) != r) continue;
          const int keep = bestR[s];
          bestR[s] = r;
          const double c = evaluate(bestR);
          bestR[s] = keep;
          if (c < bc) { bc = c; bs = s; br = r; }
        }
      if (bs < 0) break;
      bestR[bs] = br;
      best = bc;
    }
    std::lock_guard<std::mutex> lk(memo_mu);
    if (memo.size() > 64) memo.clear();
    memo[memo_key] = bestR;
  }
  // ---- emit
  std::vector<int> order(n_sites), len(n_sites);
  long n_items = 0, n_rt = 0;
  for (int s = 0; s < n_sites; ++s) {
    sp[s].R = bestR[s];
    len[s] = sp[s].len = ceil_div(sp[s].U, sp[s].R);
    order[s] = s;
    n_items += (long)sp[s].T * sp[s].R;
    n_rt += sp[s].T;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });
  const size_t items_bytes = ((size_t)n_items * sizeof(T8Item) + 63) & ~(size_t)63;
  const size_t rt_bytes = ((size_t)n_rt * sizeof(T8RTile) + 63) & ~(size_t)63;
  info->n_ctas = G;
  info->n_items = (int)n_items;
  info->n_rtiles = (int)n_rt;
  info->rounds = (int)((n_items + G - 1) / G);
  info->rtile_offset = (uint64_t)items_bytes;
  info->list_offset = (uint64_t)(items_bytes + rt_bytes);            // int cta_start[G + 1], int cta_items[n_items]
  pack(bestR);                                               // the schedule of the chosen plan
  const size_t list_bytes = ((((size_t)(G + 1) + (size_t)n_items) * sizeof(int)) + 63) & ~(size_t)63;
  info->sync_offset = (uint64_t)(items_bytes + rt_bytes + list_bytes);   // int arrived[n_groups], zero in the host table
  info->table_bytes = (uint64_t)(items_bytes + rt_bytes + list_bytes + groups.size() * sizeof(int));
  info->slab_bytes = (uint64_t)n_items * 65536ull * sizeof(float);
  if (!table_host) return MRMT3_OK;                          // sizing pass
  MR_CHECK_ARG(slab_dev && table_cap >= info->table_bytes, "tn_group_plan: table buffer too small or no slab buffer");
  T8Item* items = (T8Item*)table_host;
  T8RTile* rts = (T8RTile*)((char*)table_host + items_bytes);
  memset(table_host, 0, (size_t)info->table_bytes);
  float* slab = (float*)slab_dev;
  int* cta_start = (int*)((char*)table_host + info->list_offset);
  int* cta_items = cta_start + G + 1;
  std::vector<std::vector<int>> lists(G);
  std::vector<long> item0(n_sites);                          // item (s, r, t) = item0[s] + r * T + t
  long j = 0, slab_idx = 0, rt = 0;
  for (int oi = 0; oi < n_sites; ++oi) {
    const int s = order[oi];
    const mrmt3_tn_gsite& S = sites[s];
    const SP& p = sp[s];
    const long base = slab_idx;                              // partial tile (t, r) of this gradient: base + t * R + r
    item0[s] = j;
    for (int r = 0; r < p.R; ++r) {
      const int bl = std::min(p.len, p.U - r * p.len);
      for (int t = 0; t < p.T; ++t, ++j) {
        const int t1 = t / p.T2, t2 = t - t1 * p.T2;
        T8Item& I = items[j];
        I.A = (const bf16_t*)S.A; I.B = (const bf16_t*)S.B;
        I.lda = S.lda; I.ldb = S.ldb; I.ldo = 256; I.M = S.M; I.N1 = S.N1; I.N2 = S.N2;
        I.a0 = std::min(t1 * 256, S.N1 - 256); I.b0 = std::min(t2 * 256, S.N2 - 256);
        I.rmin = t1 * 256; I.cmin = t2 * 256;
        I.row0 = r * p.len * 128; I.nk = bl * 2;
        I.out = slab + (base + (long)t * p.R + r) * 65536 - ((long)I.a0 * 256 + I.b0);
      }
    }
    for (int t = 0; t < p.T; ++t, ++rt) {
      const int t1 = t / p.T2, t2 = t - t1 * p.T2;
      T8RTile& Q = rts[rt];
      Q.slab = slab + (base + (long)t * p.R) * 65536;
      Q.C = S.C; Q.ldc = S.ldc;
      Q.a0 = std::min(t1 * 256, S.N1 - 256); Q.b0 = std::min(t2 * 256, S.N2 - 256);
      Q.rmin = t1 * 256; Q.cmin = t2 * 256;
      Q.n_part = p.R; Q.accumulate = S.accumulate;
    }
    slab_idx += (long)p.T * p.R;
  }
  for (int gi : placed_order) {                              // schedule -> per-workgroup item lists, in start order
    const Grp& g2 = groups[gi];
    for (int t = 0; t < g2.T; ++t) {
      const long it = item0[g2.s] + (long)g2.r * sp[g2.s].T + g2.t0 + t;
      lists[g2.xcd * LANES + g2.lanes[t]].push_back((int)it);
      items[it].sync_idx = gi;
      items[it].sync_n = g2.T;
    }
  }
  int q = 0, rounds = 0;
  for (int c = 0; c < G; ++c) {
    cta_start[c] = q;
    for (int it : lists[c]) cta_items[q++] = it;
    rounds = std::max(rounds, (int)lists[c].size());
  }
  cta_start[G] = q;
  info->rounds = rounds;
  return MRMT3_OK;
}

extern "C" int mrmt3_tn_group_run(void* table_dev, const void* table_host, const mrmt3_tn_group_info* info, void* stream) {
  MR_CHECK_ARG(table_dev && info && info->n_items > 0 && info->n_rtiles > 0 && info->n_ctas >= 8, "tn_group_run: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (table_host) {
    const hipError_t e = hipMemcpyAsync(table_dev, table_host, (size_t)info->table_bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
      mrmt3_set_error("tn_group_run: table copy failed: %s", hipGetErrorString(e));
      return MRMT3_ERR_HIP;
    }
  }
  const int* cta_start = (const int*)((const char*)table_dev + info->list_offset);
  hipLaunchKernelGGL(gemm_tn8_group_kernel, dim3((unsigned)info->n_ctas), dim3(512), 0, s, (const T8Item*)table_dev,
                     cta_start, cta_start + info->n_ctas + 1, (int*)((char*)table_dev + info->sync_offset));
  MR_CHECK_LAUNCH("tn_group_run");
  hipLaunchKernelGGL(tn8_group_reduce_kernel, dim3((unsigned)info->n_rtiles * 64u), dim3(256), 0, s,
                     (const T8RTile*)((const char*)table_dev + info->rtile_offset));
  MR_CHECK_LAUNCH("tn_group_run reduce");
  mrmt3_count(MRMT3_CNT_TN_GROUP);
  return MRMT3_OK;
}
