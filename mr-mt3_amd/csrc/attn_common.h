// Helpers shared by the attention kernels (attention.hip: forward, two-pass backward, exact-f32 kernels;
// attention_onepass.hip: the one-pass backward): fragment reads, the swizzled LDS tile image and its LDS-DMA fill, the
// attention-probability dropout generator, the launch parameter block.
#pragma once
#include "common.h"

#define HD 64          // head dim (d_kv)

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
  return bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  const u32x4 r = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
  return __builtin_bit_cast(bf16x8, r);      // four v_cvt_pk_bf16_f32 (pairs), nothing to join
}

// attention-probability dropout.  Element (q,k) of head-matrix bh takes byte (k & 3) of the 32-bit word
// W(bh, q, k >> 2) = mix24(seed + bh*CB + q*CQ + (k >> 2)*CK): one mix per query row and group of four consecutive keys.
// Forward and dQ hold exactly such a group per lane and accumulator tile (query on the lane, keys 4g..4g+3 in the four
// registers): ONE mix per 4 elements, the lane's part of the argument is loop-invariant and the key tile's part a scalar.
// dK/dV hold the transposed group (key on the lane, four consecutive queries): the four lanes of a quad hold the same
// four queries for keys 4g..4g+3, so each computes the word of ONE query and reads the other three through DPP
// quad-permutes (a v_and with the lane's byte mask as the DPP instruction, then a compare with the threshold shifted into
// that byte).  Round 2 used 2x2 groups — two mixes per 4 elements in every layout — and the mask was 42-53 % of the
// kernels' vector instructions (profiles/r03_attn_isa_mix.txt).  The drop probability is quantised to thresh8/256
// (p=0.1 -> 26/256) and the keep scale is 256/(256-thresh8), so forward and backward stay exactly consistent and
// unbiased.  The kernels only ZERO the dropped probabilities in the loop; the constant keep scale is applied once to
// the accumulated O / dV (and inside the fused multiply-add that forms dS).  oracle/dropout_ref.attn_keep_mask restates
// the generator; tests compare the kernels' kept sets with it bit for bit and check its statistics on the CPU.
struct AttnDrop {
  const int* step;   // nullable device step counter (see DropCfg::step)
  unsigned seed;
  unsigned thresh8;  // 0 = off
  float scale;
};
#define DROP_CQ 0x9E3779B1u
#define DROP_CK 0x85EBCA6Bu
#define DROP_CB 0xC2B2AE35u
__host__ inline AttnDrop make_attn_drop(float p, unsigned long long seed, unsigned stream, const int* step) {
  AttnDrop d;
  d.step = p > 0.f ? step : nullptr;
  d.seed = ((unsigned)seed ^ (unsigned)(seed >> 32)) + stream * 0x27D4EB2Fu;
  if (p <= 0.f) { d.thresh8 = 0; d.scale = 1.f; }
  else {
    d.thresh8 = (unsigned)(p * 256.0f + 0.5f);
    if (d.thresh8 > 255) d.thresh8 = 255;
    d.scale = 256.0f / (256.0f - (float)d.thresh8);
  }
  return d;
}
// VALU and MFMA do not overlap on a gfx950 SIMD (profiles/tools/coissue_probe.hip), so every instruction of
// the mask costs kernel time.  32-bit integer multiplies are quarter rate; v_mul_u32_u24 is full rate, and
// with the xor-shifts folding the high bits down first the two 24-bit multiplies mix just as well here
// (byte histograms, keep rate and neighbour correlations checked against the 32-bit finaliser).
__device__ __forceinline__ unsigned mix24(unsigned x) {
  x ^= x >> 16; x = __umul24(x, 0x7feb35u); x ^= x >> 15; x = __umul24(x, 0x6ca68bu);
  return x;
}
// byte `byte` (0..3, a compile-time constant: an SDWA byte select on the compare) of the group's word decides element `byte`
__device__ __forceinline__ float drop_sel(const AttnDrop& d, unsigned g, int byte, float v) {
  return ((g >> (8 * byte)) & 0xFFu) >= d.thresh8 ? v : 0.f;
}
// dK/dV layout: word of query r of the lane's quad, fetched from the lane that computed it (quad_perm [r,r,r,r])
template <int R>
__device__ __forceinline__ unsigned quad_word(unsigned w) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)w, R * 0x55, 0xF, 0xF, true);
}
// max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48): gfx950's v_permlane16/32_swap are
// plain VALU moves, so the reduction has no LDS (ds_bpermute) round trip in the softmax's dependency chain
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows_max(float v) {
  unsigned u = __float_as_uint(v);
  u32x2_t r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(r.x), __uint_as_float(r.y)));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
// Epilogue rows, widened (the guide's T21: per-lane 8-byte row stores are issue-bound, 16 quarter-lines per instruction).
// A lane holds, of ONE output row, the four 8-byte chunks c[dt] = features 16 dt + 4 fg .. + 3 (the MFMA accumulator
// layout with the row on the lane).  Two v_permlane16_swap per chunk pair trade chunks between the lane groups fg and
// fg ^ 1, after which a lane owns 16 contiguous bytes twice: out[p] = features 32 p + widen_off(fg) .. + 7.  A store
// instruction then writes 64 contiguous bytes per row instead of 32, and a row takes two instructions instead of four.
__device__ __forceinline__ void widen_rows(const u32x2 c[4], u32x4 out[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const u32x2_t r0 = __builtin_amdgcn_permlane16_swap(c[2 * p].x, c[2 * p + 1].x, false, false);
    const u32x2_t r1 = __builtin_amdgcn_permlane16_swap(c[2 * p].y, c[2 * p + 1].y, false, false);
    out[p] = u32x4{r0.x, r1.x, r0.y, r1.y};
  }
}
__device__ __forceinline__ int widen_off(int fg) { return (fg & 1) * 16 + (fg >> 1) * 8; }
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

struct AttnParams {
  const bf16_t *q, *k, *v, *o, *d_o;
  // low half of the f32 attention output, bf16(O - bf16(O)): written by the forward next to `out` when the caller
  // keeps the tape, read back by the dQ kernel so that delta = rowsum(dO * O) sees O to ~16 bits.  With delta taken
  // from the bf16-rounded O alone, dP - delta loses the exact cancellation of whatever the value rows have in common
  // (dS_ij = P_ij dO_i.(V_j - O_i)): q/k weight gradients were 3x further from the fp32 gradient than the
  // reference's own bf16-autocast run (tests/golden/bf16_bound.npz), the v/o gradients were not.
  const bf16_t* o_lo_in;
  bf16_t* o_lo_out;
  bf16_t *out, *dq, *dk, *dv;
  float* lse;
  float* delta;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, H, Lq, Lk, causal;
  int tile_mode;     // (tuning) attn_tile: low nibble = flip shift (0 = none), bit 4 = zigzag order inside a group
  AttnDrop drop;
};

// ---- LDS tiles: [rows][64] bf16 = 128-B rows, the 16-B chunk index XOR-ed with (row & 7).  Tiles are
// filled by buffer_load ... lds (memory -> LDS, no VGPR round trip, see blds_rows8 below): one wave-instruction
// covers 8 rows x 128 B linearly, so the swizzle is applied to each lane's SOURCE chunk.  The same image
// serves ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads, both conflict-free.
// Rows past the end of the tensor are zero-filled by the buffer bounds check (their scores are masked).
__device__ __forceinline__ bf16x8 lds_row8(const unsigned char* tile, int row, int c) {
  return *(const bf16x8*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
}
// transposed fragment: rows {row..row+3 via the lane's fq} and +16, chunk c, 8-byte half `sub`
__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* tile, int row, int c, int sub) {
  const unsigned char* a = tile + row * 128 + ((c ^ (row & 7)) << 4) + sub;
  return cat8(lds_tr16(a), lds_tr16(a + 16 * 128));
}

// Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8) and each XCD has its own L2.  All tiles
// of one (batch, head) stream the same K/V (or Q/dO) rows, so they are made to run on ONE XCD, back to back:
// XCD x takes the contiguous range [x*n/8, (x+1)*n/8) of the (b, h, tile) space, tile fastest.  Without this the
// 8 query tiles of a head land on 8 different XCDs and every one of them pulls K/V through the fabric again.
__device__ __forceinline__ void attn_tile(int& tile, int& h, int& b, int mode) {
  const int nt = gridDim.x, H = gridDim.y;
  const int n = nt * H * (int)gridDim.z;
  int w = blockIdx.x + nt * (blockIdx.y + H * blockIdx.z);
  if ((n & 7) == 0) w = (w & 7) * (n >> 3) + (w >> 3);
  tile = w % nt;
  const int bh = w / nt;
  // A small causal launch (12 segments: 576 unpaired workgroups for 768 slots) is resident all at once, and the
  // dispatcher deals an XCD's workgroups over its 32 CUs in rounds: with tile = j % 8 a CU gets the SAME tile index in
  // every round — 2-3 x 16 key tiles on some CUs, 2-3 x 2 on others.  Every other round of 32 (mode & 15 = 5) takes its
  // tiles in descending order instead, so a CU's workgroups are (t, nt-1-t).  Measured on the decoder self-attention,
  // forward / backward: 8 segments 35.8 -> 28.1 / 91.9 -> 76.8 us, 12 segments 47.5 -> 39.5 / 117.4 -> 106.4, 16
  // segments 49.2 -> 41.2 / 135.4 -> 120.0; from 22 segments the tiles run paired (equal work) and nothing changes.
  // Other round lengths (8, 16, 64) and a zigzag order inside a group were no better (MRMT3_ATTN_TILE_MODE, tuning).
  // (whole (batch, head) groups flip — the position of the group's first tile in its XCD's share decides, a function of
  // bh alone — so the map stays a bijection for any nt, H, B)
  if (mode & 16) tile = (tile & 1) ? nt - 1 - (tile >> 1) : (tile >> 1);
  if (mode & 32) tile = nt - 1 - tile;
  if ((mode & 15) && ((((bh * nt) % ((n & 7) == 0 ? (n >> 3) : n)) >> (mode & 15)) & 1) != 0) tile = nt - 1 - tile;
  h = bh % H;
  b = bh / H;
}
// causal launches pair the 128-row tiles (see the kernels): ceil(n/2) workgroups along x — when that still leaves a few
// workgroups per CU.  A small batch (12 segments x 6 heads: 288 pairs for 512-768 resident workgroup slots) runs the
// tiles unpaired instead: twice the workgroups, unequal (2..16 key tiles each) but all resident at once.
static inline bool attn_paired(int L, int causal, int H, int B) {
  const int n = ceil_div(L, 128);
  return causal && n >= 2 && ((n + 1) / 2) * H * B >= 512;
}
// Small launches (12 segments per GPU: 72 (batch, head) pairs) do not fill the chip with 128-row tiles — the encoder's
// self-attention is 144 workgroups, the decoder's 576 unequal ones for 768 slots: such launches run 64-row tiles
// (16 rows per wave, RT = 1 instantiations) instead, twice the workgroups.  MRMT3_ATTN_FINE=0 / 1 forces it (tuning).
static inline bool attn_fine(int L, bool paired, int causal, int H, int B) {
  if (paired || L <= 64) return false;
  const int e = MR_KNOB("MRMT3_ATTN_FINE", -1);             // (tests switch it: mrmt3_set_knob)
  if (e == 0 || e == 1) return e == 1;
  // causal launches that are too small to pair (up to 21 segments) have unequal tiles: the finer grain balances them
  // (20 segments: forward 54.8 -> 50.0 us, backward 148.4 -> 131.2 us); equal tiles only while the chip is underfilled
  // (cross-attention forward at 16-32 segments is 7-10 % slower with 64-row tiles)
  return causal || (long long)ceil_div(L, 128) * H * B < 768;
}
static inline int attn_grid_x(int L, bool paired) {
  const int n = ceil_div(L, 128);
  return paired ? (n + 1) / 2 : n;
}
// K/V (Q/dO) tiles are staged with buffer_load ... lds: the per-lane byte offset inside an 8-row group is constant
// for the whole kernel, the tile offset is a scalar, and rows past the end of the tensor (or a whole tile that is
// switched off by an out-of-range scalar offset) come back as zeros without touching memory — so the loop needs no
// address arithmetic, no row clamp and no branch around the prefetch.
#define BUF_OOB 0x7FFF0000
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const bf16_t* base, int nrows, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, ((nrows - 1) * ld + HD) * 2, 0x00020000);
}
__device__ __forceinline__ unsigned rows8_lane_off(int ld, int lane) {     // row lane>>3, swizzled 16-B chunk
  return (unsigned)(((lane >> 3) * ld + (((lane & 7) ^ (lane >> 3)) << 3)) * 2);
}
__device__ __forceinline__ void blds_rows8(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int tile_byte_off, unsigned char* dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, lane_off, tile_byte_off, 0, 0);
}
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// attention_onepass.hip: the backward in one pass when a workgroup can own all keys of a (batch, head); 1 = launched
int mrmt3_attn_bwd_onepass_try(const AttnParams& P, hipStream_t s);

// attention_general.hip: exact-f32 arithmetic on f32 / bf16 operands, optional additive bias and its gradient
int mrmt3_attn_general_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* bias,
                           long long bias_bs, void* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int causal,
                           int dtype, const AttnDrop& drop, hipStream_t s);
int mrmt3_attn_general_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                           const void* d_o, int lddo, const float* lse, float* delta, const float* bias, long long bias_bs,
                           void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, float* dbias, int B, int H, int Lq,
                           int Lk, int causal, int dtype, const AttnDrop& drop, hipStream_t s);
