// K5 backward in ONE pass for the attention shapes whose keys fit one workgroup: 256 keys, not causal — the decoder's
// cross-attention over the 256 encoder frames (Lq = 1024) and the encoder's self-attention (Lq = 256) of MT3Net, i.e. 16 of
// the 24 attention sites of a training step (HF T5Attention autograd via models/t5.py:636-648) — and, round 6 (NT = 3), 256 < Lk
// <= 320 keys: the cross-attention of MR-MT3's own model over 256 frames + 64 memory slots (models/t5_segmem_v2_with_prev.py:
// 125-128), 20 key tiles dealt 3 + 2 over the two waves of a SIMD; profiles/r06_onepass320.txt holds what it took to fit 256
// registers, its time (190.8 us per site in the step, two-pass 242) and the two re-balancings that lost.
//
// The two-pass backward (attention.hip) recomputes S, exp, the dropout mask and dP in both of its kernels: 7 matrix products
// and two passes of vector work per (query, key) pair, and both kernels are bound by vector-instruction issue.  Here a
// workgroup owns ALL keys of one (batch, head): 8 waves x 32 keys, K and V fragments in registers for the whole kernel,
// and walks the queries in blocks of 32:
//   phase A  S = Q K^T, dP = dO V^T (key on the lane), P = exp2(S - lse), mask, dS = P (keep * scale * dP - delta);
//            dV^T += dO^T P, dK^T += Q^T dS in registers (one writer per key row at the end: no atomics);
//            dS (bf16) goes to an LDS exchange buffer [key][32 queries];
//   phase C  (one block later, behind the block's only barrier) dQ^T[64 x 32] = K^T dS^T over all 256 keys: eight
//            16 x 16 tiles, one per wave, both operands read transposed (ds_read_b64_tr_b16) from the K image and the
//            exchange buffer; the block's dQ rows are complete and stored once.
// delta = rowsum(dO * (O + O_lo)) of the NEXT block is formed from the staged dO / O / O_lo tiles, four rows per wave.
// 5 products, one exp and one mask per pair; the same bits in dK / dV as the two-pass kernels' arithmetic up to the
// order of the query blocks (identical: ascending), dQ summed over keys in one chain instead of per 64-key tile.
// (Round 5, measured and closed: the two waves of a SIMD — w and w + 4 — taking phase A and the tail (delta of the next block,
// dQ of the previous one) in OPPOSITE order inside a barrier interval, the cheapest form of a half-block stagger: 155 -> 176-181 us
// per cross-attention site, +0.2 ms per step, profiles/r05_onepass_stagger_ab.txt.  MFMA and VALU instructions do not overlap on
// a gfx950 SIMD, so there is nothing for a stagger to overlap; what it adds is 20 VGPRs and a second copy of the loop.)
#include <type_traits>

#include "attn_common.h"

#define OP_STAGES 4
#define OP_STAGE_BYTES (4 * 4096 + 256)        // Q | dO | O | O_lo tiles of 32 rows x 128 B, then lse[32] (twice)
// key capacity of a workgroup: NT = 2 key tiles of 16 per wave -> 256 keys; NT = 3 -> 320 keys = 20 tiles, dealt 3 + 2 over the
// two waves of a SIMD (waves w and w + 4), so every SIMD carries 5 tiles (MR-MT3's own model: 256 encoder frames + 64 memory slots,
// models/t5_segmem_v2_with_prev.py:125-128)
#define OP_LKMAX(NT) ((NT) == 2 ? 256 : 320)
#define OP_K_BYTES(NT) (OP_LKMAX(NT) * 128)
#define OP_X_BYTES(NT) (OP_LKMAX(NT) * 64)     // dS exchange: [key][32 queries] bf16, 8-byte chunks swizzled by the key
#define OP_LDS_BYTES(NT) (OP_K_BYTES(NT) + 2 * OP_X_BYTES(NT) + OP_STAGES * OP_STAGE_BYTES + 2 * 32 * 4)

// exchange buffer swizzle: 8-byte chunk c (4 queries) of key row r sits at chunk c ^ xsw(r).  Rows r, r+4, r+8, r+12 —
// one bank group apart at the 64-byte pitch — get four different chunk shifts, so the 8-byte writes of 16 consecutive
// keys and the transposed reads of rows 4g..4g+3 are both conflict-free.
__device__ __forceinline__ int xsw(int row) { return (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) << 1); }

template <bool DROP, int NT = 2>
__global__ __launch_bounds__(512, 2) void attn_bwd_onepass_kernel(AttnParams P) {
  constexpr int LKMAX = OP_LKMAX(NT);
  __shared__ __attribute__((aligned(16))) unsigned char lds[OP_LDS_BYTES(NT)];
  unsigned char* const ldsK = lds;
  unsigned char* const ldsX = lds + OP_K_BYTES(NT);
  unsigned char* const ldsS = ldsX + 2 * OP_X_BYTES(NT);
  float* const ldsD = (float*)(ldsS + OP_STAGES * OP_STAGE_BYTES);      // delta[2][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;
  const int h = blockIdx.x, b = blockIdx.y;
  const bf16_t* qb = P.q + (size_t)b * P.Lq * P.ldq + h * HD;
  const bf16_t* dob = P.d_o + (size_t)b * P.Lq * P.lddo + h * HD;
  const bf16_t* ob = P.o + (size_t)b * P.Lq * P.ldo + h * HD;
  const bf16_t* olb = P.o_lo_in ? P.o_lo_in + (size_t)b * P.Lq * P.ldo + h * HD : ob;
  const bf16_t* kb = P.k + (size_t)b * P.Lk * P.ldk + h * HD;
  const bf16_t* vb = P.v + (size_t)b * P.Lk * P.ldv + h * HD;
  const float* lse = P.lse + ((size_t)b * P.H + h) * P.Lq;
  const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * P.H + h) * DROP_CB;
  const __amdgpu_buffer_rsrc_t qres = rows_rsrc(qb, P.Lq, P.ldq), dores = rows_rsrc(dob, P.Lq, P.lddo);
  const __amdgpu_buffer_rsrc_t ores = rows_rsrc(ob, P.Lq, P.ldo), olres = rows_rsrc(olb, P.Lq, P.ldo);
  const __amdgpu_buffer_rsrc_t kres = rows_rsrc(kb, P.Lk, P.ldk);        // rows past Lk read as zeros
  const unsigned q_lane = rows8_lane_off(P.ldq, lane), do_lane = rows8_lane_off(P.lddo, lane);
  const unsigned o_lane = rows8_lane_off(P.ldo, lane), k_lane = rows8_lane_off(P.ldk, lane);
  const bool have_lo = P.o_lo_in != nullptr;

  // K image of the whole head -> LDS (phase C's A operand), LKMAX / 8 rows per wave
  constexpr int KPIECES = LKMAX / 64;                       // 8-row pieces per wave
#pragma unroll
  for (int i = 0; i < KPIECES; ++i) blds_rows8(kres, k_lane, ((uw * KPIECES + i) * 8) * P.ldk * 2, ldsK + (uw * KPIECES + i) * 1024);

  // this wave's key tiles: K and V fragments (B operands of S = Q K^T and dP = dO V^T) for the whole kernel.
  // NT = 2: tiles 2w, 2w + 1.  NT = 3: waves 0-3 own tiles 5w .. 5w + 2, waves 4-7 tiles 5(w - 4) + 3, + 4 (their third slot is idle:
  // every piece of work on tile slot 2 sits behind the wave-uniform test `nt < my_nt`).
  const int my_nt = NT == 2 ? 2 : (uw < 4 ? 3 : 2);
  const int tile0 = NT == 2 ? 2 * uw : (uw < 4 ? 5 * uw : 5 * (uw - 4) + 3);
  bf16x8 kf[NT][2], vf[NT][2];
  int key[NT];
  unsigned drop_k[NT];
  const unsigned drop_bmask = 0xFFu << (8 * fp), drop_bthr = P.drop.thresh8 << (8 * fp);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    key[nt] = (tile0 + nt) * 16 + fr;
    drop_k[nt] = drop_bh + ((unsigned)key[nt] >> 2) * DROP_CK + (unsigned)(fg * 4 + fp) * DROP_CQ;
    const bool live = NT == 2 || (nt < my_nt && key[nt] < P.Lk);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (NT == 2) kf[nt][ks] = *(const bf16x8*)(kb + (size_t)key[nt] * P.ldk + ks * 32 + fg * 8);     // (NT = 3: from the LDS image)
      vf[nt][ks] = live ? *(const bf16x8*)(vb + (size_t)key[nt] * P.ldv + ks * 32 + fg * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  f32x4 dkT[NT][4], dvT[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dkT[nt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[nt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nblk = ceil_div(P.Lq, 32);
  // a stage = the four 32-row tiles of a query block + its lse.  Every wave issues three loads per stage (waves 0-3 a
  // slice of Q and of dO, waves 4-7 of O and of O_lo; all of them the 128 bytes of lse), so one vmcnt value fits all.
  auto stage = [&](int buf, int qb0, bool on) {
    unsigned char* base = ldsS + buf * OP_STAGE_BYTES;
    if (uw < 4) {
      blds_rows8(qres, q_lane, on ? (qb0 + uw * 8) * P.ldq * 2 : BUF_OOB, base + uw * 1024);
      blds_rows8(dores, do_lane, on ? (qb0 + uw * 8) * P.lddo * 2 : BUF_OOB, base + 4096 + uw * 1024);
    } else {
      blds_rows8(ores, o_lane, on ? (qb0 + (uw - 4) * 8) * P.ldo * 2 : BUF_OOB, base + 8192 + (uw - 4) * 1024);
      blds_rows8(olres, o_lane, (on && have_lo) ? (qb0 + (uw - 4) * 8) * P.ldo * 2 : BUF_OOB, base + 12288 + (uw - 4) * 1024);
    }
    const float* sp = lse + min(qb0 + (lane & 31), P.Lq - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                     (__attribute__((address_space(3))) void*)(base + 16384), 4, 0, 0);
  };
  // delta of the 32 queries staged in `buf`: wave w takes rows 4w..4w+3, 16 lanes per row, 4 of the 64 values per lane
  auto delta_of = [&](int buf, int slot) {
    const unsigned char* base = ldsS + buf * OP_STAGE_BYTES;
    const int row = uw * 4 + fg, c16 = fr;
    const int off = row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8;
    const u32x2 d2 = *(const u32x2*)(base + 4096 + off), o2 = *(const u32x2*)(base + 8192 + off);
    const u32x2 l2 = *(const u32x2*)(base + 12288 + off);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float d0 = __uint_as_float(d2[e] << 16), d1 = __uint_as_float(d2[e] & 0xFFFF0000u);
      part = fmaf(d0, __uint_as_float(o2[e] << 16), part);
      part = fmaf(d1, __uint_as_float(o2[e] & 0xFFFF0000u), part);
      part = fmaf(d0, __uint_as_float(l2[e] << 16), part);
      part = fmaf(d1, __uint_as_float(l2[e] & 0xFFFF0000u), part);
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    part += __shfl_xor(part, 8, 64);
    if (fr == 0) ldsD[slot * 32 + row] = part;
  };
  // phase C: dQ^T tile (dt = wave & 3, qt = wave >> 2) of the block whose dS sits in exchange buffer `xb`
  const int c_dt = uw & 3, c_qt = uw >> 2;
  auto dq_of = [&](int xb, int qb0) {
    const unsigned char* X = ldsX + xb * OP_X_BYTES(NT);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // (NT = 3: the lane's coordinates from an opaque copy of the lane id — the ten per-step read addresses below are otherwise
    // hoisted out of the block loop and held in ten registers of their own, which the 320-key form does not have: re-derived
    // here they are ONE base each plus immediate offsets)
    int lane_c = lane;
    if (NT == 3) asm volatile("" : "+v"(lane_c));
    const int fr = lane_c & 15, fg = lane_c >> 4, fq = fr >> 2, fp = lane_c & 3;
#pragma unroll
    for (int ks = 0; ks < LKMAX / 32; ++ks) {
      const bf16x8 kt_ = lds_tr8(ldsK, ks * 32 + fg * 4 + fq, c_dt * 2 + (fp >> 1), (fp & 1) * 8);
      const int row = ks * 32 + fg * 4 + fq;
      const unsigned char* a = X + row * 64 + (((c_qt * 4 + fp) ^ xsw(row)) << 3);
      const bf16x8 ds_ = cat8(lds_tr16(a), lds_tr16(a + 16 * 64));
      acc = mfma16(kt_, ds_, acc);
    }
    const int q = qb0 + c_qt * 16 + fr;
    if (q < P.Lq)
      *(u32x2*)(P.dq + ((size_t)b * P.Lq + q) * P.lddq + h * HD + c_dt * 16 + fg * 4) =
          u32x2{pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
  };

  for (int i = 0; i < 3; ++i) stage(i, i * 32, i < nblk);
  VMCNT(6);                                   // the K image, the K / V fragments and stage 0 have landed
  __builtin_amdgcn_s_barrier();
  delta_of(0, 0);
  int cur = 0;

  for (int it = 0; it < nblk; ++it) {
    const int qb0 = it * 32;
    VMCNT(3);                                  // stage it+1 landed too (its delta is formed in this iteration)
    __builtin_amdgcn_s_barrier();              // ... for every wave; delta[it & 1] and exchange buffer (it-1) & 1 are complete
    stage(cur == 0 ? 3 : cur - 1, qb0 + 96, it + 3 < nblk);
    const unsigned char* lq = ldsS + cur * OP_STAGE_BYTES;
    const unsigned char* ldo_ = lq + 4096;
    const float* lstat = (const float*)(lq + 16384);
    const float* ldlt = ldsD + (it & 1) * 32;
    const int nxt = cur == OP_STAGES - 1 ? 0 : cur + 1;
    const bool need_mask = qb0 + 32 > P.Lq;
    unsigned char* X = ldsX + (it & 1) * OP_X_BYTES(NT);
    // phase A for the key-tile slots [N0, N0 + CNT) of this wave.  NT = 2: one call for both tiles (the kernel of round 3).
    // NT = 3: slots 0-1, then — waves 0-3 only, a wave-uniform branch — slot 2 in a pass of its own, so that the temporaries of
    // one pass are dead before the other starts: with all three tiles in one pass the kernel needs ~300 registers and spills 143.
    // The K fragments of the NT = 3 form come from the K image in LDS (it is there for phase C anyway), not from registers.
    auto phase_a = [&](auto n0_tag, auto cnt_tag) __attribute__((always_inline)) {
      constexpr int N0 = decltype(n0_tag)::value, CNT = decltype(cnt_tag)::value;
      bf16x8 pdB[CNT], dsB[CNT];
      f32x4 pd[2][CNT], ds[2][CNT];  // [qt][nt]
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const bf16x8 qa0 = lds_row8(lq, qt * 16 + fr, fg), qa1 = lds_row8(lq, qt * 16 + fr, 4 + fg);
        const bf16x8 da0 = lds_row8(ldo_, qt * 16 + fr, fg), da1 = lds_row8(ldo_, qt * 16 + fr, 4 + fg);
        float lrow[4], drow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lrow[r] = lstat[qt * 16 + fg * 4 + r] * LOG2E;
          drow[r] = ldlt[qt * 16 + fg * 4 + r];
        }
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          constexpr int dummy = 0;
          (void)dummy;
          const int nt = N0 + c;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if constexpr (NT == 2) {
            s = mfma16(qa0, kf[nt][0], s);
            s = mfma16(qa1, kf[nt][1], s);
          } else {
            s = mfma16(qa0, lds_row8(ldsK, key[nt], fg), s);
            s = mfma16(qa1, lds_row8(ldsK, key[nt], 4 + fg), s);
          }
          dp = mfma16(da0, vf[nt][0], dp);
          dp = mfma16(da1, vf[nt][1], dp);
          float pv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -lrow[r]));
          if (__builtin_expect(need_mask, 0)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (qb0 + qt * 16 + fg * 4 + r >= P.Lq) pv[r] = 0.f;
          }
          if (NT == 3 && key[nt] >= P.Lk) {                  // 256 < Lk < 320: the keys past the end
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = 0.f;
          }
          float pk[4] = {pv[0], pv[1], pv[2], pv[3]}, dk_[4] = {dp[0], dp[1], dp[2], dp[3]};
          if (DROP && P.drop.thresh8) {
            const unsigned w = mix24(drop_k[nt] + (unsigned)(qb0 + qt * 16) * DROP_CQ);
            const bool k0_ = (quad_word<0>(w) & drop_bmask) >= drop_bthr, k1_ = (quad_word<1>(w) & drop_bmask) >= drop_bthr;
            const bool k2_ = (quad_word<2>(w) & drop_bmask) >= drop_bthr, k3_ = (quad_word<3>(w) & drop_bmask) >= drop_bthr;
            pk[0] = k0_ ? pk[0] : 0.f; dk_[0] = k0_ ? dk_[0] : 0.f;
            pk[1] = k1_ ? pk[1] : 0.f; dk_[1] = k1_ ? dk_[1] : 0.f;
            pk[2] = k2_ ? pk[2] : 0.f; dk_[2] = k2_ ? dk_[2] : 0.f;
            pk[3] = k3_ ? pk[3] : 0.f; dk_[3] = k3_ ? dk_[3] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pd[qt][c][r] = pk[r];
            ds[qt][c][r] = pv[r] * fmaf(dk_[r], P.drop.scale, -drow[r]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CNT; ++c) {
        pdB[c] = pack8(pd[0][c], pd[1][c]);  // k-slot (g,j) <-> q = 16*(j>>2) + 4g + (j&3)
        dsB[c] = pack8(ds[0][c], ds[1][c]);
        // dS of key (wave, nt, fr) for the queries 4g..4g+3 of both 16-query tiles: two 8-byte chunks of its row
        const int row = key[N0 + c], sw = xsw(row);
        const u32x4 dsw = __builtin_bit_cast(u32x4, dsB[c]);
        *(u32x2*)(X + row * 64 + ((fg ^ sw) << 3)) = u32x2{dsw[0], dsw[1]};
        *(u32x2*)(X + row * 64 + (((4 + fg) ^ sw) << 3)) = u32x2{dsw[2], dsw[3]};
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 dot_ = lds_tr8(ldo_, fg * 4 + fq, dt * 2 + (fp >> 1), (fp & 1) * 8);
        const bf16x8 qt_ = lds_tr8(lq, fg * 4 + fq, dt * 2 + (fp >> 1), (fp & 1) * 8);
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          dvT[N0 + c][dt] = mfma16(dot_, pdB[c], dvT[N0 + c][dt]);
          dkT[N0 + c][dt] = mfma16(qt_, dsB[c], dkT[N0 + c][dt]);
        }
      }
    };
    phase_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    if constexpr (NT == 3) {
      if (my_nt == 3) {
        __builtin_amdgcn_sched_barrier(0);                  // the two passes do not interleave (registers)
        phase_a(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
      } else {
        // the idle slot's rows of the exchange buffer are never written by anyone: phase C would read garbage there — waves 4-7
        // own tiles 5(w-4)+3, +4 only, and the tiles 5w+2 ARE written by waves 0-3: every row of the buffer has exactly one writer
      }
    }
    if (it + 1 < nblk) delta_of(nxt, (it + 1) & 1);
    if (it > 0) dq_of((it - 1) & 1, qb0 - 32);
    cur = nxt;
  }
  VMCNT(0);
  __builtin_amdgcn_s_barrier();                // the last block's dS is complete
  dq_of((nblk - 1) & 1, (nblk - 1) * 32);

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (NT == 3 && (nt >= my_nt || key[nt] >= P.Lk)) continue;
    bf16_t* dkrow = P.dk + ((size_t)b * P.Lk + key[nt]) * P.lddk + h * HD;
    bf16_t* dvrow = P.dv + ((size_t)b * P.Lk + key[nt]) * P.lddv + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 a = dkT[nt][dt], c = dvT[nt][dt] * P.drop.scale;
      *(u32x2*)(dkrow + dt * 16 + fg * 4) = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
      *(u32x2*)(dvrow + dt * 16 + fg * 4) = u32x2{pack_bf2(c[0], c[1]), pack_bf2(c[2], c[3])};
    }
  }
}

// 1 = launched.  Takes: not causal, exactly 256 keys, and enough (batch, head) pairs to give most CUs a workgroup —
// below that the two-pass kernels' 128-row tiles fill the chip better.
int mrmt3_attn_bwd_onepass_try(const AttnParams& P, hipStream_t s) {
  int min_bh = 96;
  const bool enabled = MR_KNOB("MRMT3_ATTN_ONEPASS", 1) != 0;     // A/B switch (tuning / tests only)
  const int m = MR_KNOB("MRMT3_ATTN_ONEPASS_MIN_BH", 0);
  if (m > 0) min_bh = m;
  if (!enabled || P.causal || P.B * P.H < min_bh || P.Lq < 32) return 0;
  const dim3 grid((unsigned)P.H, (unsigned)P.B);
  if (P.Lk == 256) {
    if (P.drop.thresh8) hipLaunchKernelGGL((attn_bwd_onepass_kernel<true, 2>), grid, dim3(512), 0, s, P);
    else hipLaunchKernelGGL((attn_bwd_onepass_kernel<false, 2>), grid, dim3(512), 0, s, P);
    return 1;
  }
  // 256 < Lk <= 320 (MR-MT3's own model: 256 encoder frames + 64 memory slots): the 3 + 2 tile form, key tail masked
  if (P.Lk > 256 && P.Lk <= 320 && MR_KNOB("MRMT3_ATTN_ONEPASS_320", 1) != 0) {
    if (P.drop.thresh8) hipLaunchKernelGGL((attn_bwd_onepass_kernel<true, 3>), grid, dim3(512), 0, s, P);
    else hipLaunchKernelGGL((attn_bwd_onepass_kernel<false, 3>), grid, dim3(512), 0, s, P);
    return 1;
  }
  return 0;
}
