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
//   * (tried, not kept: all 160 KiB of LDS — a third B buffer, B requested three K steps ahead, 7 instead of 4 half-tiles
//     in flight per CU.  Same results, 917 against 945 TF over the step's shapes: the K loop is not waiting for bytes in
//     flight.  A side lesson of that variant: when the K-step body grew a second lambda, hipcc stopped inlining it, the
//     accumulators it captures by reference went to scratch memory and the kernel ran 50x slower — if a lambda here is
//     ever not inlined, force it with __attribute__((always_inline)).)
//   * N need only be a multiple of 128: the last column tile is shifted left to end at N and masks the columns the
//     tile before it owns.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#define G8_OOB 0x7FFF0000
// kernel knock-outs (MRMT3_GEMM8_DBG) and the start skew exist in the diagnostics build only (common.h: MR_DIAG)
#define G8_DBG(bit) MR_DIAG(P.dbg & (bit))
#define G8_HALF 16384
#define G8_BUF (4 * G8_HALF)

struct G8Params {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int lda, ldb, ldc, M, N, K;
  int tiles_n, n_tiles;
  // EPI = 1 (gated GELU, the wi projection): N = 2 dff, C = h [M][2 dff] as always, C2 = g [M][dff] =
  // dropout(gelu_new(h0) * h1) computed in the epilogue from the bf16-rounded h (what mrmt3_geglu_fwd reads back)
  void* C2;
  int ldc2, dff;
  DropCfg drop;
  // split K (short inputs: too few tiles for the chip and a long K): `ksplit` partial products, each over `K` (the
  // per-split depth) of `kfull` columns, written as f32 to a slab [ksplit][mpad][N] (C = the slab, ldc = N); row tile
  // indices run over ksplit * mpad VIRTUAL rows.  ksplit = 1: kfull = K, mpad unused.
  int ksplit, mpad, kfull;
  int nt_c;            // 1: streaming (non-temporal) C stores; bit 1 (EPI = 1): also for g.  See g8_store_mode().
  int dbg;             // diagnostics (MRMT3_GEMM8_DBG): 1 no C stores, 2 plain instead of streaming C stores (bf16), 4 every K step re-reads K step 0 (cache-hot),
                       // 8 no fragment reads, 16 loads switched off (zero fill, no traffic); EPI = 1 only: 64 no h stores, 128 no g stores
  int skew_ticks;      // start delay per (slot % 8), in 10-ns ticks of s_memrealtime (see the kernel)
};

// C store mode (G8Params::nt_c), MRMT3_GEMM8_NT: bit 0 streaming C, bit 1 streaming g (fused wi + GEGLU launch).
// Default 2: plain C stores.  Same-box, three alternations at 64 segments: 3 (both streaming, the round-2 state) 25.80 ms,
// 1 25.77, 2 25.61, 0 25.63 — the consumer of C is the next kernel and finds it in the Infinity Cache.
static int g8_store_mode() { return MR_KNOB("MRMT3_GEMM8_NT", 2) & 3; }

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
template <typename TOUT, bool ACCUM, int RT, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(G8Params P) {
  static_assert(EPI == 0 || (sizeof(TOUT) == 2 && !ACCUM), "the GEGLU epilogue writes bf16");
  constexpr int GROUP_ROWS = RT * 16;          // rows of one wave group
  constexpr int BM = 2 * GROUP_ROWS;
  constexpr int NA = RT / 4;                   // LDS-DMA instructions per thread and A half-tile (2 or 1)
  constexpr int A_HALF_ROWS = GROUP_ROWS;      // rows in HA0 (= RT*8 per group x 2 groups)
  constexpr int YOUNG = 2 * NA + 4;            // loads of the four youngest half-tiles (2 A halves, 2 B halves)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * G8_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = w >> 2, wc = w & 3;
  const int fr = lane & 15, fg = lane >> 4;
  DropCfg dc = P.drop;
  if (EPI == 1) DROP_STEP(dc);

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
  if (MR_DIAG(P.skew_ticks > 0) && (slot & 7)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)(slot & 7) * P.skew_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }

  auto tile_origin = [&](int it, int& m0, int& n0, int& cmin) {
    const int t = xstart + slot + it * nx;
    const int mt = t / P.tiles_n, nt = t - mt * P.tiles_n;
    m0 = mt * BM;
    if (EPI == 1) {                                         // a tile = 128 features: their h0 AND h1 columns
      n0 = nt * 128;
      cmin = 0;
      return;
    }
    n0 = min(nt * 256, P.N - 256);
    cmin = nt * 256;                                        // columns below belong to the tile on the left
  };

  // split K (f32, non-accumulating instantiations only: the others compile exactly as before): tile rows are virtual,
  // row block z = m0 / mpad is K range [z * K, (z + 1) * K) of the operands and slab z of the output
  constexpr bool SPLITK = sizeof(TOUT) == 4 && !ACCUM && EPI == 0;
  auto src_off = [&](int m0, int n0, int& sa, int& sb) {
    if constexpr (SPLITK) {
      if (P.ksplit > 1) {
        const int z = m0 / P.mpad;
        sa = (m0 - z * P.mpad) * P.lda * 2 + z * P.K * 2;
        sb = n0 * P.ldb * 2 + z * P.K * 2;
        return;
      }
    }
    sa = m0 * P.lda * 2;
    sb = n0 * P.ldb * 2;
  };
  auto row_limit = [&](int m0) -> int {                       // first row index past the rows a tile at m0 may store
    if constexpr (SPLITK) {
      if (P.ksplit > 1) return (m0 / P.mpad) * P.mpad + P.M;
    }
    return P.M;
  };
  const int kcols = SPLITK ? P.kfull : P.K;                   // (columns of the operands; P.K is the depth of a tile)
  const __amdgpu_buffer_rsrc_t ra = g8_rsrc(P.A, ((size_t)(P.M - 1) * P.lda + kcols) * 2);
  const __amdgpu_buffer_rsrc_t rb = g8_rsrc(P.B, ((size_t)(P.N - 1) * P.ldb + kcols) * 2);

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
    // EPI 1: the wave's 64 columns are 32 features x (h0 | h1): HB0 = the wi_0 rows of the features, HB1 = their wi_1
    // rows (dff rows further down), so a lane holds h0 and h1 of the same 8 features
    const int brow = EPI == 1 ? (rr >> 5) * 32 + cm : (rr >> 5) * 64 + cm;
    voffB[i] = (unsigned)(brow * P.ldb * 2 + sw * 16);
  }
  const unsigned a1_delta = (unsigned)(RT * 8 * P.lda * 2);
  const unsigned b1_delta = EPI == 1 ? (unsigned)(P.dff * P.ldb * 2) : (unsigned)(32 * P.ldb * 2);
  const int piece0 = w * 1024;                              // LDS offset of this wave's piece inside a half-tile

  auto load_a = [&](int buf, int half, int soff) {          // half 0: HA0, 1: HA1
    if (G8_DBG(16)) soff = G8_OOB;
    unsigned char* base = lds + buf * G8_BUF + half * G8_HALF + piece0;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      g8_dma16(ra, base + i * 8192, voffA[i] + (half ? a1_delta : 0u), soff);
  };
  auto load_b = [&](int buf, int half, int soff) {          // half 0: HB0, 1: HB1
    if (G8_DBG(16)) soff = G8_OOB;
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
    src_off(m0, n0, l_sa, l_sb);
  }
  auto cursor_next = [&]() {
    ++l_k;
    if (l_k < nk) { if (!(G8_DBG(4))) { l_sa += 128; l_sb += 128; } return; }
    l_k = 0;
    ++l_it;
    if (l_it < my_tiles) {
      int m0, n0, cm;
      tile_origin(l_it, m0, n0, cm);
      src_off(m0, n0, l_sa, l_sb);
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
    if (G8_DBG(8)) return;
    const unsigned char* p = lds + buf * G8_BUF + half * G8_HALF;
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      af[i][0] = *(const bf16x8*)(p + a_off0 + i * 2048);
      af[i][1] = *(const bf16x8*)(p + a_off1 + i * 2048);
    }
  };
  auto read_b = [&](int buf, int half) {
    if (G8_DBG(8)) return;
    const unsigned char* p = lds + buf * G8_BUF + half * G8_HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf_[half * 2 + j][0] = *(const bf16x8*)(p + b_off0 + j * 2048);
      bf_[half * 2 + j][1] = *(const bf16x8*)(p + b_off1 + j * 2048);
    }
  };
  // 16 (8) MFMAs: row half rh x column half ch, both k-steps.  Weights are the A operand: D = W_frag . X_frag^T.
  // FIRST (the first K step of an output tile): the accumulators start from zero instead of being cleared.
  auto mma = [&](int rh, int ch, bool first) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < RH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 c = (first && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[rh * RH + i][ch * 2 + j];
          acc[rh * RH + i][ch * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[ch * 2 + j][ks], af[i][ks], c, 0, 0, 0);
        }
  };
  // (Tried: issuing a phase's LDS-DMA requests between its MFMAs instead of in the read segment.  A wave issues in
  // order and an LDS-DMA instruction takes 60+ cycles to issue, so the MFMAs behind it wait: barriers + MFMAs alone
  // went from 1.03 to 1.30 us per K step.  In the read segment the same issue time runs beside the partner's MFMAs.)
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

  // ---- the epilogue, two row tiles at a time.  A tile's 128 KiB of C issued in one go cost 5-7 us per tile (the K
  // loop of a K = 512 tile: 11 us) — not HBM bandwidth (de-phasing the workgroups did not help) but the store path of
  // the CU: HALF the stores cost 1.3 us, the second half 5.7, and what an instruction costs goes with the number of
  // cache lines it touches.  So (a) the lanes of rows r and r + 8 (lane ^ 8: one DPP row rotate) exchange halves, which
  // makes a store instruction 8 rows x 128 contiguous bytes (whole lines) instead of 16 rows x 64; (b) the tile leaves in
  // four batches, in the read segments of four consecutive phases:
  //     row tiles 0,1 (final after ph2 of the tile's last K step) -> its ph3      row tiles 2,3 -> its ph4
  //     row tiles of r1 (final after ph4)                         -> ph1, ph2 of the NEXT tile (overwritten in its ph3)
  // Stores share the in-order vmcnt queue with the loads: the counted waits of these K steps allow for the batches
  // that are younger than the load they wait for.
  auto store_rows = [&](int i0, int m0, int n0, int cmin, bool count) __attribute__((always_inline)) {
    if (G8_DBG(1)) return;
    if constexpr (EPI == 1) {
      // gated GELU: three 16-byte stores per row tile (h0, h1, g), rows on fr, 8 features per lane
      const int f0 = n0 + wc * 32 + fg * 8;
#pragma unroll
      for (int ii = 0; ii < RH / 2; ++ii) {
        const int i = i0 + ii;
        // (the lane's row base goes through an opaque move: hipcc otherwise keeps `base | 16 i` for every i in a register of
        // its own from the prologue on — at 256 VGPRs those were the values it spilled and reloaded behind a vmcnt(0) in
        // every tile's epilogue)
        int rbase = g * GROUP_ROWS + fr;
        asm volatile("" : "+v"(rbase));
        const int row = m0 + rbase + i * 16;
        const unsigned h0p[4] = {pack_bf2(acc[i][0][0], acc[i][0][1]), pack_bf2(acc[i][0][2], acc[i][0][3]),
                                 pack_bf2(acc[i][1][0], acc[i][1][1]), pack_bf2(acc[i][1][2], acc[i][1][3])};
        const unsigned h1p[4] = {pack_bf2(acc[i][2][0], acc[i][2][1]), pack_bf2(acc[i][2][2], acc[i][2][3]),
                                 pack_bf2(acc[i][3][0], acc[i][3][1]), pack_bf2(acc[i][3][2], acc[i][3][3])};
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a_lo = __uint_as_float(h0p[e] << 16), a_hi = __uint_as_float(h0p[e] & 0xFFFF0000u);
          const float b_lo = __uint_as_float(h1p[e] << 16), b_hi = __uint_as_float(h1p[e] & 0xFFFF0000u);
          o[2 * e] = gelu_new_f(a_lo) * b_lo;
          o[2 * e + 1] = gelu_new_f(a_hi) * b_hi;
        }
        if (dc.thresh) {
          float mk[8];
          const unsigned long long e8 = ((unsigned long long)row * P.dff + f0) >> 2;
          drop_mask4(dc, e8, mk);
          drop_mask4(dc, e8 + 1, mk + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] *= mk[e];
        }
        if (row < P.M) {
          bf16_t* hp = (bf16_t*)P.C + (size_t)row * P.ldc + f0;
          if (!(G8_DBG(64))) {
            __builtin_nontemporal_store((u32x4{h0p[0], h0p[1], h0p[2], h0p[3]}), (u32x4*)hp);
            __builtin_nontemporal_store((u32x4{h1p[0], h1p[1], h1p[2], h1p[3]}), (u32x4*)(hp + P.dff));
          }
          const u32x4 gv = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
          u32x4* gp = (u32x4*)((bf16_t*)P.C2 + (size_t)row * P.ldc2 + f0);
          if (G8_DBG(128)) {}
          else if (P.nt_c & 2) __builtin_nontemporal_store(gv, gp);      // (h: only the backward reads it again; g: the next kernel)
          else *gp = gv;
        }
      }
      (void)count;
      return;
    }
    if (n0 + wc * 64 < cmin) return;                          // columns owned by the tile on the left (wave-uniform)
    TOUT* C = (TOUT*)P.C;
    const bool up = fr >= 8;
    const int mlim = row_limit(m0);
    (void)mlim;
#pragma unroll
    for (int ii = 0; ii < RH / 2; ++ii) {
      const int i = i0 + ii;
      const int row1 = m0 + g * GROUP_ROWS + i * 16 + (fr & 7), row2 = row1 + 8;
      if constexpr (sizeof(TOUT) == 2) {
        // keep = the half this lane stores for its own row, send = the half its partner lane stores (selected before
        // the bf16 packing: fewer live registers than packing both halves and selecting afterwards)
        const f32x4 k0 = up ? acc[i][2] : acc[i][0], k1 = up ? acc[i][3] : acc[i][1];
        const f32x4 s0 = up ? acc[i][0] : acc[i][2], s1 = up ? acc[i][1] : acc[i][3];
        const u32x4 keep = {pack_bf2(k0[0], k0[1]), pack_bf2(k0[2], k0[3]), pack_bf2(k1[0], k1[1]), pack_bf2(k1[2], k1[3])};
        const u32x4 send = {pack_bf2(s0[0], s0[1]), pack_bf2(s0[2], s0[3]), pack_bf2(s1[0], s1[1]), pack_bf2(s1[2], s1[3])};
        u32x4 recv;
#pragma unroll
        for (int e = 0; e < 4; ++e) recv[e] = (unsigned)__builtin_amdgcn_mov_dpp((int)send[e], 0x128, 0xf, 0xf, false);   // row_ror:8
        const u32x4 v1 = up ? recv : keep, v2 = up ? keep : recv;
        bf16_t* p = (bf16_t*)C + n0 + wc * 64 + (up ? 32 : 0) + fg * 8;
        // C is written once and not read by this kernel: streaming stores keep it from flushing the operand panels out
        // of the XCD's L2 (a round of 32 tiles is 4 MB, the whole L2)
        // The output of a product is read by the very next kernel (a norm, the attention, the GEGLU): plain stores leave
        // it in the Infinity Cache.  Streaming stores made THIS kernel faster (the operand panels stay in the XCD's L2:
        // a round of 32 tiles is 4 MB, the whole L2) but the step slower: 25.79 -> 25.58 ms with plain stores, three
        // same-box alternations (MRMT3_GEMM8_NT=1 brings the streaming stores back).
        if (!(P.nt_c & 1)) {
          if (row1 < P.M) *(u32x4*)(p + (size_t)row1 * P.ldc) = v1;
          if (row2 < P.M) *(u32x4*)(p + (size_t)row2 * P.ldc) = v2;
        } else {
          if (row1 < P.M) __builtin_nontemporal_store(v1, (u32x4*)(p + (size_t)row1 * P.ldc));
          if (row2 < P.M) __builtin_nontemporal_store(v2, (u32x4*)(p + (size_t)row2 * P.ldc));
        }
      } else {
        // f32: column tiles (0,1) and (2,3) each make one 128-byte line per row
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const f32x4 lo = acc[i][2 * cp], hi = acc[i][2 * cp + 1];
          const f32x4 send = up ? lo : hi;
          f32x4 recv;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            recv[e] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send[e]), 0x128, 0xf, 0xf, false));
          f32x4 v1 = up ? recv : lo, v2 = up ? hi : recv;
          float* p = (float*)C + n0 + wc * 64 + cp * 32 + (up ? 16 : 0) + fg * 4;
          if (row1 < mlim) { float* q1 = p + (size_t)row1 * P.ldc; if (ACCUM) v1 += *(const f32x4*)q1; *(f32x4*)q1 = v1; }
          if (row2 < mlim) { float* q2 = p + (size_t)row2 * P.ldc; if (ACCUM) v2 += *(const f32x4*)q2; *(f32x4*)q2 = v2; }
        }
      }
    }
    (void)count;
  };
  // counted wait: everything but the YOUNG youngest loads and `extra` batches of interleaved stores has completed.
  // (The immediates must be compile-time constants and a chain of scalar branches per wait is not free: computing the
  // allowance from a running request counter cost 0.45 us per K step.  The five cases below cover the schedule.)
  constexpr int SQ = ACCUM ? 0 : (RH / 2) * (EPI == 1 ? 3 : (sizeof(TOUT) == 2 ? 2 : 4));   // stores per lane and batch (0: not counted)
  auto wait_young = [&](int extra) {
    if (SQ == 0 || extra == 0) { if (YOUNG == 8) G8_VMCNT(8); else G8_VMCNT(6); return; }
#define G8_W(k)                                                                               \
  do {                                                                                        \
    constexpr int n_ = YOUNG + (k) * SQ;                                                      \
    if (n_ == 8) G8_VMCNT(8); else if (n_ == 10) G8_VMCNT(10); else if (n_ == 12) G8_VMCNT(12);           \
    else if (n_ == 14) G8_VMCNT(14); else if (n_ == 16) G8_VMCNT(16); else if (n_ == 18) G8_VMCNT(18);     \
    else if (n_ == 20) G8_VMCNT(20); else if (n_ == 22) G8_VMCNT(22); else if (n_ == 24) G8_VMCNT(24);     \
    else if (n_ == 32) G8_VMCNT(32); else if (n_ == 40) G8_VMCNT(40);                                      \
    else if (n_ > 40) G8_VMCNT(40); else if (n_ > 32) G8_VMCNT(32); else if (n_ > 24) G8_VMCNT(24);        \
    else G8_VMCNT(6);                                                                         \
  } while (0)
    if (extra == 1) G8_W(1);
    else if (extra == 2) G8_W(2);
    else if (extra == 3) G8_W(3);
    else G8_W(4);
#undef G8_W
  };
  auto ld_a = [&](int buf, int half, int soff) { load_a(buf, half, soff); };
  auto ld_b = [&](int buf, int half, int soff) { load_b(buf, half, soff); };

  // ---- prologue: K steps 0 and (half of) 1
  ld_a(0, 0, l_sa); ld_b(0, 0, l_sb); ld_b(0, 1, l_sb); ld_a(0, 1, l_sa);
  cursor_next();
  ld_a(1, 0, l_sa); ld_b(1, 0, l_sb);
  wait_young(0);                                             // HA0(0), HB0(0) landed; four half-tiles behind them
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one barrier behind group 0

  int c_it = 0, c_m0, c_n0, c_cmin;
  tile_origin(0, c_m0, c_n0, c_cmin);
  int p_m0 = 0, p_n0 = 0, p_cmin = 0;                        // the tile whose r1 rows are still to be stored
  bool p_pending = false, p_full = false, p_second_full = false;
  const int total = my_tiles * nk;
  // One K step (buf is a compile-time constant so that seq[][] stays in registers).  `first`: first K step of its tile
  // (compile time: zero-initialised MFMAs; the r1 rows of the previous tile leave in ph1, ph2); `last`: last K step of
  // its tile (run time: the r0 rows leave in ph3, ph4).  A tile that is not stored in full (M edge, masked columns) may
  // issue fewer stores than a count would assume: its stores are not counted, which only makes the waits stricter.
  // Store batches T1 (ph3), T2 (ph4) of a tile's last K step u, T3 (ph1), T4 (ph2) of the next tile's first K step v.
  // Batches younger than the load a wait is for:  u: ph4 +2   v: ph1 +3, ph2 +4, ph4 +2   v+1: ph1 +1.
  auto kstep = [&](auto buf_tag, auto first_tag, bool second, bool last) {
    constexpr int buf = decltype(buf_tag)::value;
    constexpr bool first = decltype(first_tag)::value;
    const bool full = c_m0 + BM <= row_limit(c_m0) && c_n0 >= c_cmin && !ACCUM && !(G8_DBG(1));
    const bool pf = p_pending && p_full;                      // the previous tile's batches were issued in full
    const int x4 = (last && full ? 2 : 0) + (first && pf ? 2 : 0);
    // ph1: r0 x c0
    read_a(buf, 0);
    read_b(buf, 0);
    if (first && p_pending) store_rows(RH, p_m0, p_n0, p_cmin, p_full);
    ld_b(buf ^ 1, 1, l_sb);
    wait_young(first ? (pf ? 3 : 0) : (second && p_second_full ? 1 : 0));     // HB1 of this K step (read in ph2)
    G8_SEG_END();
    mma(0, 0, first);
    G8_MMA_END();
    // ph2: r0 x c1
    read_b(buf, 1);
    if (first && p_pending) store_rows(RH + RH / 2, p_m0, p_n0, p_cmin, p_full);
    ld_a(buf ^ 1, 1, l_sa);
    wait_young(first && pf ? 4 : 0);                                          // HA1 of this K step (read in ph3)
    G8_SEG_END();
    mma(0, 1, first);
    G8_MMA_END();
    // ph3: r1 x c1
    read_a(buf, 1);
    if (last) store_rows(0, c_m0, c_n0, c_cmin, full);
    cursor_next();
    ld_a(buf, 0, l_sa);
    G8_SEG_END();
    mma(1, 1, first);
    G8_MMA_END();
    // ph4: r1 x c0
    if (last) store_rows(RH / 2, c_m0, c_n0, c_cmin, full);
    ld_b(buf, 0, l_sb);
    wait_young(x4);                                                           // HA0, HB0 of the next K step
    G8_SEG_END();
    mma(1, 0, first);
    G8_MMA_END();
    if (first) { p_second_full = pf; p_pending = false; }
    if (last) {
      p_m0 = c_m0; p_n0 = c_n0; p_cmin = c_cmin; p_pending = true; p_full = full;
      ++c_it;
      if (c_it < my_tiles) tile_origin(c_it, c_m0, c_n0, c_cmin);
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  // nk is even for every shape the launcher admits, so a tile always starts on buffer 0
  for (int t = 0; t < my_tiles; ++t) {
    kstep(B0{}, std::true_type{}, false, false);
    kstep(B1{}, std::false_type{}, true, nk == 2);
    for (int k = 2; k < nk; k += 2) {
      kstep(B0{}, std::false_type{}, false, false);
      kstep(B1{}, std::false_type{}, false, k + 2 == nk);
    }
  }
  if (p_pending) { store_rows(RH, p_m0, p_n0, p_cmin, false); store_rows(RH + RH / 2, p_m0, p_n0, p_cmin, false); }
  G8_VMCNT(0);
  if (g == 0) __builtin_amdgcn_s_barrier();                  // pairs with group 1's extra barrier at the start
}

// Returns 1 if the shape was launched on the second-generation kernel, 0 if the caller should use gemm.hip's.
int mrmt3_gemm_nt8_try(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       int out_dtype, int accumulate, hipStream_t s) {
  int min_m = 2048;
  { const int e = MR_KNOB("MRMT3_GEMM8_MIN_M", 0); if (e >= 128) min_m = e; }      // tuning only
  if (M < min_m || N < 256 || N % 128 != 0 || K % 128 != 0 || K < 128) return 0;      // (an even number of K steps)
  // Which shapes: measured COLD (profiles/r03_gemm_ab_cold.txt: 512 MiB written before every launch, as inside the
  // training step — the back-to-back table of round 2 re-read operands out of the Infinity Cache and flattered round
  // 1's kernel on the N % 256 == 128 shapes).  This kernel wins wherever its tiles come in whole waves of workgroups:
  // at most one wave (short inputs: 12 segments per GPU, the encoder), or a tile count that fills >= 80 % of the
  // last wave's slots.  A last column tile that is half overlap (N = 384 / 1152: 11-33 % of the MFMAs redone) still
  // wins cold (qkv 115 -> 96 us).  Not: the accumulate form (one launch per decoder layer).
  {
    if (MR_KNOB("MRMT3_GEMM8_ALL", 0) != 1) {                   // (1: tuning / tests, take every admissible shape)
      if (accumulate) return 0;
      const int cu = g8_cus() & ~7, tn = ceil_div(N, 256), t256 = ceil_div(M, 256) * tn;
      const int nt = t256 < cu ? ceil_div(M, 128) * tn : t256;
      const int waves = ceil_div(nt, cu);
      if (nt > cu && nt * 5 < waves * cu * 4) return 0;        // a last wave of workgroups less than 80 % full
    }
  }
  if (((size_t)M * lda + K) * 2 >= 0x7FFF0000ull || ((size_t)N * ldb + K) * 2 >= 0x7FFF0000ull) return 0;
  G8Params P;
  P.A = (const bf16_t*)A; P.B = (const bf16_t*)B; P.C = C;
  P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.M = M; P.N = N; P.K = K;
  P.ksplit = 1; P.mpad = 0; P.kfull = K;
  P.dbg = 0; P.skew_ticks = 0;
  P.nt_c = g8_store_mode();
  P.tiles_n = ceil_div(N, 256);
  const int cus = g8_cus() & ~7;
  const int tiles256 = ceil_div(M, 256) * P.tiles_n;
  const bool small = tiles256 < cus;                         // not enough 256-row tiles to fill the chip: 128-row tiles
  P.n_tiles = small ? ceil_div(M, 128) * P.tiles_n : tiles256;
  int grid = P.n_tiles < cus ? ((P.n_tiles + 7) & ~7) : cus;
  if (grid > P.n_tiles) grid = (P.n_tiles + 7) & ~7;
#ifdef MRMT3_DIAG
  {
    // (experiment, closed: start skew) one tile ~ nk x 1.4 us (0.7 for 128-row tiles) + the stores; eight start phases across that period
    P.dbg = mrmt3_diag_env("MRMT3_GEMM8_DBG");
    const int skew_pct = MR_KNOB("MRMT3_GEMM8_SKEW", 0);
    const double tile_us = (K / 64) * (small ? 0.7 : 1.4) + 1.5;
    const int tiles_per_wg = ceil_div(P.n_tiles, cus);
    P.skew_ticks = tiles_per_wg >= 2 ? (int)(tile_us * 100.0 / 8.0 * skew_pct / 100.0) : 0;
    const int g = MR_KNOB("MRMT3_GEMM8_GRID", 0);
    if (g > 0 && g < grid) grid = g & ~7;
  }
#endif
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


// ---- split K for short inputs ------------------------------------------------------------------------------------------
// 12 segments per GPU leave the encoder 3072 rows: 24-48 tiles of 128 x 256 for 256 CUs, each walking the whole K
// (e_dwi, K = 2048: 35 us at 184 TFLOP/s; the cross-attention K|V gradient, K = 6144: 95 us).  With a workspace the
// product is cut into `ksplit` K ranges that run as separate tiles of the same launch (f32 partial sums in a slab
// [ksplit][mpad][N]) and one reduce pass sums the slabs in split order into C (bf16 or f32, += for the accumulate form):
// a fixed summation order, no atomics.  mrmt3_gemm_nt_workspace_bytes() > 0 says when this pays.
static int g8_splitk_plan(int M, int N, int K, int in_dtype) {
  if (MR_KNOB("MRMT3_GEMM8_SPLITK", 1) == 0) return 1;     // tuning / A-B switch only
  // The second launch and the slab round trip cost ~10 us: measured cold at 3072 rows (profiles/r03_gemm_splitk.txt)
  // K = 6144 98.8 -> 48.0 us and K = 2048 35.8 -> 29.6 us, but K = 1024 / 1152 22.8 -> 28.6 / 24.6 -> 27.9 us: only
  // from K = 2048, with >= 512 per split.
  if (in_dtype != MRMT3_BF16 || M < 1024 || N < 256 || N % 128 != 0 || K < 2048 || K % 128 != 0) return 1;
  const int cus = g8_cus() & ~7;
  const int tiles = ceil_div(M, 128) * ceil_div(N, 256);
  if (tiles * 2 > cus) return 1;                             // the chip is at least half full without a split
  for (int sp = 4; sp >= 2; --sp)                            // at most one wave of workgroups, >= 512 deep per split
    if (tiles * sp <= cus && K % sp == 0 && (K / sp) % 128 == 0 && K / sp >= 512) return sp;
  return 1;
}

extern "C" size_t mrmt3_gemm_nt_workspace_bytes(int M, int N, int K, int in_dtype) {
  const int sp = g8_splitk_plan(M, N, K, in_dtype);
  return sp > 1 ? (size_t)sp * (size_t)(ceil_div(M, 128) * 128) * (size_t)N * sizeof(float) : 0;
}

template <typename TOUT, bool ACCUM>
__global__ __launch_bounds__(256) void g8_splitk_reduce_kernel(const float* __restrict__ slab, TOUT* __restrict__ C, int ldc,
                                                               int M, int N, int mpad, int ksplit) {
  const int n4 = N >> 2;
  const size_t total = (size_t)M * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / n4), c = (int)(i - (size_t)m * n4) * 4;
    f32x4 a = *(const f32x4*)(slab + (size_t)m * N + c);
    for (int z = 1; z < ksplit; ++z) a += *(const f32x4*)(slab + ((size_t)z * mpad + m) * N + c);
    if constexpr (sizeof(TOUT) == 2) {
      *(u32x2*)((bf16_t*)C + (size_t)m * ldc + c) = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
    } else {
      float* q = (float*)C + (size_t)m * ldc + c;
      if (ACCUM) a += *(const f32x4*)q;
      *(f32x4*)q = a;
    }
  }
}

// 1 = launched (kernel + reduce), 0 = not a split-K shape or the workspace is too small: use the plain path.
int mrmt3_gemm_nt8_splitk_try(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                              int in_dtype, int out_dtype, int accumulate, void* workspace, size_t workspace_bytes,
                              hipStream_t s) {
  const int sp = g8_splitk_plan(M, N, K, in_dtype);
  if (sp <= 1 || !workspace) return 0;
  const int mpad = ceil_div(M, 128) * 128;
  if (workspace_bytes < (size_t)sp * mpad * N * sizeof(float) || ((uintptr_t)workspace & 15)) return 0;
  if (((size_t)M * lda + K) * 2 >= 0x7FFF0000ull || ((size_t)N * ldb + K) * 2 >= 0x7FFF0000ull) return 0;
  G8Params P;
  memset(&P, 0, sizeof(P));
  P.A = (const bf16_t*)A; P.B = (const bf16_t*)B; P.C = workspace;
  P.lda = lda; P.ldb = ldb; P.ldc = N; P.M = M; P.N = N;
  P.K = K / sp; P.kfull = K; P.ksplit = sp; P.mpad = mpad;
  P.tiles_n = ceil_div(N, 256);
  P.n_tiles = sp * (mpad / 128) * P.tiles_n;
#ifdef MRMT3_DIAG
  P.dbg = mrmt3_diag_env("MRMT3_GEMM8_DBG");
#endif
  const int grid = (P.n_tiles + 7) & ~7;
  hipLaunchKernelGGL((gemm_nt8_kernel<float, false, 4>), dim3((unsigned)grid), dim3(512), 0, s, P);
  const size_t total = (size_t)M * (N >> 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  const float* slab = (const float*)workspace;
  if (out_dtype == MRMT3_BF16)
    hipLaunchKernelGGL((g8_splitk_reduce_kernel<bf16_t, false>), dim3((unsigned)blocks), dim3(256), 0, s, slab, (bf16_t*)C, ldc, M, N, mpad, sp);
  else if (accumulate)
    hipLaunchKernelGGL((g8_splitk_reduce_kernel<float, true>), dim3((unsigned)blocks), dim3(256), 0, s, slab, (float*)C, ldc, M, N, mpad, sp);
  else
    hipLaunchKernelGGL((g8_splitk_reduce_kernel<float, false>), dim3((unsigned)blocks), dim3(256), 0, s, slab, (float*)C, ldc, M, N, mpad, sp);
  mrmt3_count(MRMT3_CNT_GEMM_NT_SPLITK);
  return 1;
}


// ---- K2+K7 in one launch: h = x . wi^T and g = dropout(gelu_new(h[:, :dff]) * h[:, dff:]) from the same accumulators.
// Saves the GEGLU kernel's read of h (rows x 2 dff bf16) and one launch per feed-forward block; h is still written
// (the backward needs it).  Shapes the fused kernel does not take run as the two separate kernels: same results bit for
// bit either way (the epilogue rounds h to bf16 first and applies the element-wise kernel's arithmetic and mask).
extern "C" int mrmt3_gemm_nt_geglu(const void* x, int ldx, const void* wi, int ldw, void* h, int ldh, void* g, int ldg,
                                   int rows, int dff, int K, float p_drop, uint64_t seed, const int32_t* step_dev,
                                   uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(x && wi && h && g, "gemm_nt_geglu: null pointer");
  MR_CHECK_ARG(rows > 0 && dff > 0 && K > 0 && dff % 8 == 0, "gemm_nt_geglu: bad sizes rows=%d dff=%d K=%d", rows, dff, K);
  MR_CHECK_ARG(ldh >= 2 * dff && ldg >= dff, "gemm_nt_geglu: ldh >= 2 dff and ldg >= dff");
  hipStream_t s = (hipStream_t)stream;
  // (from 2048 rows since round 6 — 4096 before: 12 segments per GPU leave the encoder 3072 rows = 192 tiles of 128 rows, the
  // same tile count as the unfused product, and the fused launch saves the geglu kernel behind it: 7.60-7.63 -> 7.56-7.59 ms per
  // 12-segment step, profiles/r06_geglu_fused_min_rows_ab.txt; MRMT3_GEGLU_FUSED_MIN_ROWS moves the line for A/B runs)
  const int min_rows = MR_KNOB("MRMT3_GEGLU_FUSED_MIN_ROWS", 2048);
  bool fused = rows >= min_rows && dff % 128 == 0 && K % 128 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldh % 8 == 0 &&
               ldg % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wi % 16) == 0 && ((uintptr_t)h % 16) == 0 &&
               ((uintptr_t)g % 16) == 0 && ((size_t)rows * ldx + K) * 2 < 0x7FFF0000ull &&
               ((size_t)2 * dff * ldw + K) * 2 < 0x7FFF0000ull;
  if (MR_KNOB("MRMT3_GEGLU_FUSED", 1) == 0) fused = false;
  if (!fused) {
    int rc = mrmt3_gemm_nt(x, ldx, wi, ldw, h, ldh, rows, 2 * dff, K, MRMT3_BF16, MRMT3_BF16, 0, stream);
    if (rc != MRMT3_OK) return rc;
    MR_CHECK_ARG(ldh == 2 * dff && ldg == dff, "gemm_nt_geglu: the unfused path needs dense h and g");
    return mrmt3_geglu_fwd(h, g, rows, dff, MRMT3_BF16, p_drop, seed, step_dev, stream_id, stream);
  }
  G8Params P;
  P.A = (const bf16_t*)x; P.B = (const bf16_t*)wi; P.C = h; P.C2 = g;
  P.lda = ldx; P.ldb = ldw; P.ldc = ldh; P.ldc2 = ldg; P.dff = dff;
  P.M = rows; P.N = 2 * dff; P.K = K;
  P.ksplit = 1; P.mpad = 0; P.kfull = K;
  P.nt_c = g8_store_mode();
  P.drop = make_drop(p_drop, seed, stream_id, step_dev);
  P.tiles_n = dff / 128;
  const int cus = g8_cus() & ~7;
  const int tiles256 = ceil_div(rows, 256) * P.tiles_n, tiles128 = ceil_div(rows, 128) * P.tiles_n;
  // 128-row tiles when there are too few 256-row tiles for the chip, or when they come in a badly filled last wave and
  // the 128-row tiling does not (12 segments per GPU: 384 tiles = 1.5 waves against 768 = 3 waves)
  auto fill = [&](int nt) { return (double)nt / ((double)ceil_div(nt, cus) * cus); };
  const bool small = tiles256 < cus || (tiles256 < 4 * cus && fill(tiles128) > fill(tiles256) + 0.15);
  P.n_tiles = small ? tiles128 : tiles256;
  P.dbg = 0;
#ifdef MRMT3_DIAG
  P.dbg = mrmt3_diag_env("MRMT3_GEMM8_DBG");
#endif
  P.skew_ticks = 0;
  int grid = P.n_tiles < cus ? ((P.n_tiles + 7) & ~7) : cus;
  if (small) hipLaunchKernelGGL((gemm_nt8_kernel<bf16_t, false, 4, 1>), dim3((unsigned)grid), dim3(512), 0, s, P);
  else hipLaunchKernelGGL((gemm_nt8_kernel<bf16_t, false, 8, 1>), dim3((unsigned)grid), dim3(512), 0, s, P);
  MR_CHECK_LAUNCH("gemm_nt_geglu");
  mrmt3_count(MRMT3_CNT_GEMM_NT_GEGLU);
  return MRMT3_OK;
}
