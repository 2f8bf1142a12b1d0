// K5 — attention core of the T5 block for gfx950 (head_dim 64, any head count — T5-small has 6).
//
// Stands for HF T5Attention as the reference instantiates it (models/t5.py:487-490: no relative
// position bias; scores = q.k^T UNSCALED, + causal mask for the decoder's self-attention
// (4.18 get_extended_attention_mask, models/t5.py:567-568), softmax in fp32, dropout on the
// probabilities, then p.v), and for its autograd backward.
//
// bf16 kernels are flash-style (never materialise [B,H,Lq,Lk]); fp32 softmax statistics.
//   forward : workgroup = 128 query rows of one (batch, head); 4 waves x 32 rows.  K/V tiles of 64
//             keys go memory->LDS directly (buffer_load ... lds, 3 stages, two tiles in flight across a raw
//             s_barrier with counted vmcnt); 128-B LDS rows with XOR-swizzled 16-B chunks are
//             conflict-free for both ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads.
//             The scores are computed TRANSPOSED (S^T = K.Q^T, query on the MFMA lane) so that the
//             softmax row statistics are per-lane scalars and the exponentiated tile is already the
//             B operand of O^T = V^T.P^T (accumulator-as-operand, no LDS round trip for P).
//   backward: (1) dQ: workgroup = 128 queries, query on the lane, loops over 64-key tiles; it also derives
//             delta = rowsum(dO*O) from operands it holds anyway and stores it; (2) dK/dV: workgroup = 128
//             keys, key on the lane, loops over 32-query blocks.  P is recomputed from the saved log-sum-exp.  No atomics: every output
//             element has exactly one writer, results are bitwise reproducible.
//   launch  : every tile of one (batch, head) runs on ONE XCD (attn_tile), causal launches give each workgroup the
//             tile pair (n-1-t, t) so all workgroups are equal, and VALU work is kept to a minimum because VALU and
//             MFMA instructions do not overlap on a gfx950 SIMD.
// The f32 kernel is a plain exact-f32 implementation used for the fp32 parity / greedy-decode path.
#include "common.h"

#include "attn_common.h"


// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
#define KV_STAGES 3
#define KV_STAGE_BYTES 16384   // K tile 8 KiB + V tile 8 KiB

static int attn_tile_mode() { return MR_KNOB("MRMT3_ATTN_TILE_MODE", 5); }      // tuning only

template <bool PAIR, bool DROP, int RT = 2>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnParams P) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[KV_STAGES * KV_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;
  int tile_, h, b;
  attn_tile(tile_, h, b, P.tile_mode);
  const bf16_t* qb = P.q + (size_t)b * P.Lq * P.ldq + h * HD;
  const bf16_t* kb = P.k + (size_t)b * P.Lk * P.ldk + h * HD;
  const bf16_t* vb = P.v + (size_t)b * P.Lk * P.ldv + h * HD;
  const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * P.H + h) * DROP_CB;
  const int n_qt = ceil_div(P.Lq, 64 * RT);
  const __amdgpu_buffer_rsrc_t kres = rows_rsrc(kb, P.Lk, P.ldk), vres = rows_rsrc(vb, P.Lk, P.ldv);
  const unsigned k_lane = rows8_lane_off(P.ldk, lane), v_lane = rows8_lane_off(P.ldv, lane);
  // 16-byte output rows need 16-byte aligned rows (kernel-uniform)
  const bool wide_rows = (P.ldo & 7) == 0 && (((uintptr_t)P.out | (uintptr_t)P.o_lo_out) & 15) == 0;

  // causal: query tile t needs 2(t+1) key tiles, so a workgroup takes the PAIR (n_qt-1-t, t) — every workgroup
  // of the launch then does the same amount of work and the launch has no tail of heavy tiles
#pragma nounroll
  for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
  int q_tile = tile_;
  if (PAIR) {
    q_tile = pass == 0 ? n_qt - 1 - tile_ : tile_;
    if (pass == 1 && 2 * tile_ == n_qt - 1) break;
    if (pass == 1) __syncthreads();      // every wave is done reading the previous tile's LDS stages
  }
  const int q0 = q_tile * (64 * RT);

  // Q fragments (B operand): lane holds Q[q = qrow(qt)][d = 32ks + 8g .. +7]
  bf16x8 qf[RT][2];
  int qrow[RT];
  unsigned drop_q[RT];     // the lane's part of the mask argument: head, query row, key group 4g..4g+3 inside a 16-key tile
#pragma unroll
  for (int qt = 0; qt < RT; ++qt) {
    qrow[qt] = q0 + uw * (16 * RT) + qt * 16 + fr;
    drop_q[qt] = drop_bh + (unsigned)qrow[qt] * DROP_CQ + (unsigned)fg * DROP_CK;
    const int r = min(qrow[qt], P.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *(const bf16x8*)(qb + (size_t)r * P.ldq + ks * 32 + fg * 8);
  }
  f32x4 oT[RT][4];
  float m_run[RT], l_run[RT];
#pragma unroll
  for (int qt = 0; qt < RT; ++qt) {
    m_run[qt] = -INFINITY;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oT[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  int n_kv = ceil_div(P.Lk, 64);
  if (P.causal) n_kv = min(n_kv, (min(q0 + 64 * RT - 1, P.Lq - 1)) / 64 + 1);
  // each wave stages 16 rows of K and of V per tile (2 + 2 wave-instructions); `on` = false turns the tile into
  // four zero-fills that never leave the CU, which keeps the vmcnt arithmetic of the loop uniform
  auto stage = [&](int buf, int kv0, bool on) {
    unsigned char* base = lds + buf * KV_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = uw * 2 + i;
      blds_rows8(kres, k_lane, on ? (kv0 + t * 8) * P.ldk * 2 : BUF_OOB, base + t * 1024);
      blds_rows8(vres, v_lane, on ? (kv0 + t * 8) * P.ldv * 2 : BUF_OOB, base + 8192 + t * 1024);
    }
  };
  stage(0, 0, true);
  stage(1, 64, n_kv > 1);
  int cur = 0;

  for (int j = 0; j < n_kv; ++j) {
    const int kv0 = j * 64;
    VMCNT(4);                                       // tile j landed (tile j+1 may still be in flight)
    __builtin_amdgcn_s_barrier();
    stage(cur == 0 ? 2 : cur - 1, kv0 + 128, j + 2 < n_kv);
    const unsigned char* lk = lds + cur * KV_STAGE_BYTES;
    const unsigned char* lv = lk + 8192;
    cur = cur == KV_STAGES - 1 ? 0 : cur + 1;

    // (causal: the last key tile of a query tile lies entirely above the 32 rows of waves 0 and 1.  They run it anyway,
    // fully masked: a per-wave skip made the accumulators loop-carried through two paths and hipcc copied all of them
    // at every back-edge — 72 v_mov per tile, 16 % of the loop's vector instructions — while the skipped waves only
    // waited at the next barrier.)
    // S^T = K . Q^T : sT[qt][kt] holds S^T[key = kt*16 + 4g + r][q = fr]
    f32x4 sT[RT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      bf16x8 ka0 = lds_row8(lk, kt * 16 + fr, fg);
      bf16x8 ka1 = lds_row8(lk, kt * 16 + fr, 4 + fg);
#pragma unroll
      for (int qt = 0; qt < RT; ++qt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16(ka0, qf[qt][0], acc);
        acc = mfma16(ka1, qf[qt][1], acc);
        sT[qt][kt] = acc;
      }
    }
    const bool need_mask = (kv0 + 64 > P.Lk) || (P.causal && kv0 + 63 > q0 + uw * (16 * RT));
#pragma unroll
    for (int qt = 0; qt < RT; ++qt) {
      if (__builtin_expect(need_mask, 0)) {
        // one limit per query row (the last key it may see) and ONE compare per element against it: written as
        // `key >= Lk || (causal && key > q)` the sixteen `key >= Lk` tests are common to both query tiles, hipcc kept their
        // results in SGPR pairs across the tiles and spilled 68 SGPRs to VGPR lanes in <true, true, 2>
        const int lim = P.causal ? min(P.Lk - 1, qrow[qt]) : P.Lk - 1;
        const int key0 = kv0 + fg * 4;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + kt * 16 + r > lim) sT[qt][kt][r] = -INFINITY;
      }
      float mloc = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sT[qt][kt][r]);
      mloc = rows_max(mloc);
      // running max kept in the exp2 domain (m2 = max * log2 e): p = exp2(s*log2e - m2).  The reference point is the
      // exact running max (a row's dominant probability is then exactly 1.0 in bf16, which the gradients of peaked
      // attention rows are sensitive to; a lazily updated reference point was measured: same speed, 6% more error)
      const float m_new = fmaxf(m_run[qt], mloc * LOG2E);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_use);
      m_run[qt] = m_new;
      float lsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sT[qt][kt][r], LOG2E, -m_use));
          lsum += p;
          sT[qt][kt][r] = p;
        }
      l_run[qt] = l_run[qt] * alpha + lsum;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oT[qt][dt] *= alpha;
      // DROP only removes the block for p = 0.  With dropout on, the test stays a run-time one on purpose: as its own
      // basic block the mask code keeps its registers to itself (164 VGPRs, no spill); merged into the exp loop by the
      // scheduler the kernel spills and is 12% slower.
      if (DROP && P.drop.thresh8) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const unsigned g = mix24(drop_q[qt] + (unsigned)((kv0 >> 2) + kt * 4) * DROP_CK);   // scalar tile part
#pragma unroll
          for (int r = 0; r < 4; ++r) sT[qt][kt][r] = drop_sel(P.drop, g, r, sT[qt][kt][r]);
        }
      }
    }
    // O^T += V^T . P^T : k-slot (g, j) of a 32-key step <-> key = 32*ks + 16*(j>>2) + 4g + (j&3)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pb[RT];
#pragma unroll
      for (int qt = 0; qt < RT; ++qt) pb[qt] = pack8(sT[qt][2 * ks], sT[qt][2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x8 vt = lds_tr8(lv, ks * 32 + fg * 4 + fq, dt * 2 + (fp >> 1), (fp & 1) * 8);
#pragma unroll
        for (int qt = 0; qt < RT; ++qt) oT[qt][dt] = mfma16(vt, pb[qt], oT[qt][dt]);
      }
    }
  }
  VMCNT(0);      // the switched-off prefetches of the last two iterations still write their zeros

#pragma unroll
  for (int qt = 0; qt < RT; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const bool row_ok = qrow[qt] < P.Lq;
    const float inv = l > 0.f ? P.drop.scale / l : 0.f;    // the dropout keep scale is applied here, once
    const size_t ooff = ((size_t)b * P.Lq + qrow[qt]) * P.ldo + h * HD;
    bf16_t* orow = P.out + ooff;
    u32x2 ch[4], cl[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 v = oT[qt][dt] * inv;
      const unsigned h01 = pack_bf2(v[0], v[1]), h23 = pack_bf2(v[2], v[3]);
      ch[dt] = u32x2{h01, h23};
      if (P.o_lo_out) {
        const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xFFFF0000u);
        const float r2 = v[2] - __uint_as_float(h23 << 16), r3 = v[3] - __uint_as_float(h23 & 0xFFFF0000u);
        cl[dt] = u32x2{pack_bf2(r0, r1), pack_bf2(r2, r3)};
      } else {
        cl[dt] = u32x2{0u, 0u};
      }
    }
    if (wide_rows) {                        // 16-byte stores (see widen_rows); every lane of the wave takes part in the swaps
      u32x4 w[2];
      widen_rows(ch, w);
      if (row_ok) {
        *(u32x4*)(orow + widen_off(fg)) = w[0];
        *(u32x4*)(orow + 32 + widen_off(fg)) = w[1];
      }
      if (P.o_lo_out) {
        widen_rows(cl, w);
        if (row_ok) {
          *(u32x4*)(P.o_lo_out + ooff + widen_off(fg)) = w[0];
          *(u32x4*)(P.o_lo_out + ooff + 32 + widen_off(fg)) = w[1];
        }
      }
    } else if (row_ok) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(u32x2*)(orow + dt * 16 + fg * 4) = ch[dt];
        if (P.o_lo_out) *(u32x2*)(P.o_lo_out + ooff + dt * 16 + fg * 4) = cl[dt];
      }
    }
    if (row_ok && fg == 0 && P.lse) P.lse[((size_t)b * P.H + h) * P.Lq + qrow[qt]] = m_run[qt] * LN2 + __logf(l);
  }
  }  // pass
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV.  workgroup = 128 keys (wave = 32 keys, key on the lane), loop over 32-query
// blocks staged 3 blocks ahead (4 LDS stages of Q | dO | lse,delta)
// ------------------------------------------------------------------------------------------------
#define QD_STAGES 4
#define QD_STAGE_BYTES 8448   // Q 4 KiB + dO 4 KiB + 64 floats

template <bool PAIR, bool DROP, int RT = 2>
__global__ __launch_bounds__(256, RT == 1 ? 3 : 2) void attn_bwd_dkdv_kernel(AttnParams P) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[QD_STAGES * QD_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;
  int tile_, h, b;
  attn_tile(tile_, h, b, P.tile_mode);
  const bf16_t* qb = P.q + (size_t)b * P.Lq * P.ldq + h * HD;
  const bf16_t* dob = P.d_o + (size_t)b * P.Lq * P.lddo + h * HD;
  const bf16_t* kb = P.k + (size_t)b * P.Lk * P.ldk + h * HD;
  const bf16_t* vb = P.v + (size_t)b * P.Lk * P.ldv + h * HD;
  const float* lse = P.lse + ((size_t)b * P.H + h) * P.Lq;
  const float* dlt = P.delta + ((size_t)b * P.H + h) * P.Lq;
  const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * P.H + h) * DROP_CB;
  const int n_kt = ceil_div(P.Lk, 64 * RT);
  const __amdgpu_buffer_rsrc_t qres = rows_rsrc(qb, P.Lq, P.ldq), dores = rows_rsrc(dob, P.Lq, P.lddo);
  const unsigned q_lane = rows8_lane_off(P.ldq, lane), do_lane = rows8_lane_off(P.lddo, lane);

  // causal: key tile t is seen by the queries from 128 t on, so the pair (t, n_kt-1-t) balances the launch
#pragma nounroll
  for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
  int k_tile = tile_;
  if (PAIR) {
    k_tile = pass == 0 ? tile_ : n_kt - 1 - tile_;
    if (pass == 1 && 2 * tile_ == n_kt - 1) break;
    if (pass == 1) __syncthreads();
  }
  const int k0 = k_tile * (64 * RT);

  bf16x8 kf[RT][2], vf[RT][2];
  int key[RT];
  // mask: the lane computes the word of query (fg*4 + fp) of each 16-query tile for its key's group of four; its own
  // element sits in byte (key & 3) = fp of every word of the quad
  unsigned drop_k[RT];
  const unsigned drop_bmask = 0xFFu << (8 * fp), drop_bthr = P.drop.thresh8 << (8 * fp);
#pragma unroll
  for (int nt = 0; nt < RT; ++nt) {
    key[nt] = k0 + uw * (16 * RT) + nt * 16 + fr;
    drop_k[nt] = drop_bh + ((unsigned)key[nt] >> 2) * DROP_CK + (unsigned)(fg * 4 + fp) * DROP_CQ;
    const int r = min(key[nt], P.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[nt][ks] = *(const bf16x8*)(kb + (size_t)r * P.ldk + ks * 32 + fg * 8);
      vf[nt][ks] = *(const bf16x8*)(vb + (size_t)r * P.ldv + ks * 32 + fg * 8);
    }
  }
  f32x4 dkT[RT][4], dvT[RT][4];
#pragma unroll
  for (int nt = 0; nt < RT; ++nt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dkT[nt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[nt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int qstart = P.causal ? (k0 / 32) * 32 : 0;
  const int nblk = qstart < P.Lq ? ceil_div(P.Lq - qstart, 32) : 0;
  // per tile every wave issues: 8 rows of Q, 8 rows of dO, and the 64 row statistics (the same 256 bytes from all
  // four waves — identical data, keeps the vmcnt arithmetic uniform).  `on` = false: zero-fills that stay on the CU
  // (the statistics are simply read again), so the loop below prefetches and waits without a branch.
  auto stage = [&](int buf, int qb0, bool on) {
    unsigned char* base = lds + buf * QD_STAGE_BYTES;
    blds_rows8(qres, q_lane, on ? (qb0 + uw * 8) * P.ldq * 2 : BUF_OOB, base + uw * 1024);
    blds_rows8(dores, do_lane, on ? (qb0 + uw * 8) * P.lddo * 2 : BUF_OOB, base + 4096 + uw * 1024);
    const float* sp = (lane < 32 ? lse : dlt) + min(qb0 + (lane & 31), P.Lq - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                     (__attribute__((address_space(3))) void*)(base + 8192), 4, 0, 0);
  };
  for (int i = 0; i < 3; ++i) stage(i, qstart + i * 32, i < nblk);
  int cur = 0;

  for (int it = 0; it < nblk; ++it) {
    const int qb0 = qstart + it * 32;
    VMCNT(6);                                  // two younger tiles (3 loads each) may stay in flight
    __builtin_amdgcn_s_barrier();
    stage(cur == 0 ? 3 : cur - 1, qb0 + 96, it + 3 < nblk);
    const unsigned char* lq = lds + cur * QD_STAGE_BYTES;
    const unsigned char* ldo_ = lq + 4096;
    const float* lstat = (const float*)(lq + 8192);   // [0..31] lse, [32..63] delta
    cur = cur == QD_STAGES - 1 ? 0 : cur + 1;
    // causal: a query block entirely above this wave's 32 keys contributes nothing to them
    // ... and a wave whose 32 keys all lie past the end of the sequence (ragged last key tile, e.g. 256 encoder
    // frames + 64 memory slots = 320 keys) has nothing to compute at all
    const bool wave_active = (k0 + uw * (16 * RT) < P.Lk) && !(P.causal && qb0 + 31 < k0 + uw * (16 * RT));
    if (wave_active) {
    const bool need_mask = (qb0 + 32 > P.Lq) || (k0 + uw * (16 * RT) + 16 * RT > P.Lk) || (P.causal && qb0 < k0 + uw * (16 * RT) + 16 * RT - 1);
    bf16x8 pdB[RT], dsB[RT];  // per key tile: B operands built from both query tiles
    f32x4 pd[2][RT], ds[2][RT];  // [qt][nt]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      bf16x8 qa0 = lds_row8(lq, qt * 16 + fr, fg);
      bf16x8 qa1 = lds_row8(lq, qt * 16 + fr, 4 + fg);
      bf16x8 da0 = lds_row8(ldo_, qt * 16 + fr, fg);
      bf16x8 da1 = lds_row8(ldo_, qt * 16 + fr, 4 + fg);
      float lrow[4], drow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lrow[r] = lstat[qt * 16 + fg * 4 + r] * LOG2E;
        drow[r] = lstat[32 + qt * 16 + fg * 4 + r];
      }
#pragma unroll
      for (int nt = 0; nt < RT; ++nt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = mfma16(qa0, kf[nt][0], s);
        s = mfma16(qa1, kf[nt][1], s);
        dp = mfma16(da0, vf[nt][0], dp);
        dp = mfma16(da1, vf[nt][1], dp);
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -lrow[r]));
        if (__builtin_expect(need_mask, 0)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qb0 + qt * 16 + fg * 4 + r;
            const bool valid = q < P.Lq && key[nt] < P.Lk && !(P.causal && key[nt] > q);
            if (!valid) pv[r] = 0.f;
          }
        }
        float pk[4] = {pv[0], pv[1], pv[2], pv[3]}, dk_[4] = {dp[0], dp[1], dp[2], dp[3]};
        if (DROP && P.drop.thresh8) {    // run-time test on purpose, see the forward kernel
          // this lane's word: query (fp-th of its quad's four), key group key >> 2; the quad's other three by DPP
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
          pd[qt][nt][r] = pk[r];                                           // keep scale: applied to dV at the end
          ds[qt][nt][r] = pv[r] * fmaf(dk_[r], P.drop.scale, -drow[r]);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < RT; ++nt) {
      pdB[nt] = pack8(pd[0][nt], pd[1][nt]);  // k-slot (g,j) <-> q = 16*(j>>2) + 4g + (j&3)
      dsB[nt] = pack8(ds[0][nt], ds[1][nt]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x8 dot_ = lds_tr8(ldo_, fg * 4 + fq, dt * 2 + (fp >> 1), (fp & 1) * 8);
      bf16x8 qt_ = lds_tr8(lq, fg * 4 + fq, dt * 2 + (fp >> 1), (fp & 1) * 8);
#pragma unroll
      for (int nt = 0; nt < RT; ++nt) {
        dvT[nt][dt] = mfma16(dot_, pdB[nt], dvT[nt][dt]);
        dkT[nt][dt] = mfma16(qt_, dsB[nt], dkT[nt][dt]);
      }
    }
    }  // wave_active
  }
  VMCNT(0);
#pragma unroll
  for (int nt = 0; nt < RT; ++nt) {
    if (key[nt] >= P.Lk) continue;
    bf16_t* dkrow = P.dk + ((size_t)b * P.Lk + key[nt]) * P.lddk + h * HD;
    bf16_t* dvrow = P.dv + ((size_t)b * P.Lk + key[nt]) * P.lddv + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 a = dkT[nt][dt], c = dvT[nt][dt] * P.drop.scale;
      *(u32x2*)(dkrow + dt * 16 + fg * 4) = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
      *(u32x2*)(dvrow + dt * 16 + fg * 4) = u32x2{pack_bf2(c[0], c[1]), pack_bf2(c[2], c[3])};
    }
  }
  }  // pass
}

// ------------------------------------------------------------------------------------------------
// backward: dQ.  workgroup = 128 queries (wave = 32, query on the lane), loop over 64-key tiles
// ------------------------------------------------------------------------------------------------
// Three workgroups per CU (<= 168 VGPRs; the causal + dropout instantiation spills 4 registers outside the loop): the
// kernel waits on dependent LDS-read -> MFMA -> exp chains more than it issues, and a third wave per SIMD measured
// -4 % on the decoder's self-attention backward (profiles/r03_attn_micro.txt).  The dK/dV kernel stays at two: at 168
// registers it spills 39 and runs 1.6x slower.
template <bool PAIR, bool DROP, int RT = 2>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnParams P) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[KV_STAGES * KV_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;
  int tile_, h, b;
  attn_tile(tile_, h, b, P.tile_mode);
  const bf16_t* qb = P.q + (size_t)b * P.Lq * P.ldq + h * HD;
  const bf16_t* dob = P.d_o + (size_t)b * P.Lq * P.lddo + h * HD;
  const bf16_t* kb = P.k + (size_t)b * P.Lk * P.ldk + h * HD;
  const bf16_t* vb = P.v + (size_t)b * P.Lk * P.ldv + h * HD;
  const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * P.H + h) * DROP_CB;
  const int n_qt = ceil_div(P.Lq, 64 * RT);
  const __amdgpu_buffer_rsrc_t kres = rows_rsrc(kb, P.Lk, P.ldk), vres = rows_rsrc(vb, P.Lk, P.ldv);
  const unsigned k_lane = rows8_lane_off(P.ldk, lane), v_lane = rows8_lane_off(P.ldv, lane);

  // causal: the pair of query tiles (n_qt-1-t, t), as in the forward kernel
#pragma nounroll
  for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
  int q_tile = tile_;
  if (PAIR) {
    q_tile = pass == 0 ? n_qt - 1 - tile_ : tile_;
    if (pass == 1 && 2 * tile_ == n_qt - 1) break;
    if (pass == 1) __syncthreads();
  }
  const int q0 = q_tile * (64 * RT);
  // (per pass, the lane's coordinates are re-derived from an opaque copy of the lane id: everything that depends only on the
  // lane is otherwise hoisted out of the pass loop and held across both passes in registers of its own — at the 168 registers
  // three workgroups per CU allow, hipcc spilled four of them: 20 B of scratch in <true, true, 2>)
  int lane_p = lane;
  asm volatile("" : "+v"(lane_p));
  const int fr = lane_p & 15, fg = lane_p >> 4, fq = fr >> 2, fp = lane_p & 3;

  bf16x8 qf[RT][2], dof[RT][2];
  int qrow[RT];
  unsigned drop_q[RT];
  float lse_q[RT], dlt_q[RT];
#pragma unroll
  for (int qt = 0; qt < RT; ++qt) {
    qrow[qt] = q0 + uw * (16 * RT) + qt * 16 + fr;
    drop_q[qt] = drop_bh + (unsigned)qrow[qt] * DROP_CQ + (unsigned)fg * DROP_CK;
    const int r = min(qrow[qt], P.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[qt][ks] = *(const bf16x8*)(qb + (size_t)r * P.ldq + ks * 32 + fg * 8);
      dof[qt][ks] = *(const bf16x8*)(dob + (size_t)r * P.lddo + ks * 32 + fg * 8);
    }
    lse_q[qt] = P.lse[((size_t)b * P.H + h) * P.Lq + r] * LOG2E;
    // delta = rowsum(dO * O): this lane has 16 of the row's 64 dO values already; the same 16 of O come in
    // two 16-byte loads, the four lane groups of a row are summed with two shuffles.  The value is also
    // written out for the dK/dV kernel, which runs after this one (no separate delta pass over dO and O).
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const size_t ooff = ((size_t)b * P.Lq + r) * P.ldo + h * HD + ks * 32 + fg * 8;
      const bf16x8 of = *(const bf16x8*)(P.o + ooff);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(bf2f((bf16_t)dof[qt][ks][e]), bf2f((bf16_t)of[e]), part);
      if (P.o_lo_in) {
        const bf16x8 ol = *(const bf16x8*)(P.o_lo_in + ooff);
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf(bf2f((bf16_t)dof[qt][ks][e]), bf2f((bf16_t)ol[e]), part);
      }
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    dlt_q[qt] = part;
    if (fg == 0 && qrow[qt] < P.Lq) P.delta[((size_t)b * P.H + h) * P.Lq + r] = part;
  }
  f32x4 dqT[RT][4];
#pragma unroll
  for (int qt = 0; qt < RT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqT[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int n_kv = ceil_div(P.Lk, 64);
  if (P.causal) n_kv = min(n_kv, (min(q0 + 64 * RT - 1, P.Lq - 1)) / 64 + 1);
  auto stage = [&](int buf, int kv0, bool on) {      // as in the forward kernel
    unsigned char* base = lds + buf * KV_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = uw * 2 + i;
      blds_rows8(kres, k_lane, on ? (kv0 + t * 8) * P.ldk * 2 : BUF_OOB, base + t * 1024);
      blds_rows8(vres, v_lane, on ? (kv0 + t * 8) * P.ldv * 2 : BUF_OOB, base + 8192 + t * 1024);
    }
  };
  stage(0, 0, true);
  stage(1, 64, n_kv > 1);
  int cur = 0;

  for (int j = 0; j < n_kv; ++j) {
    const int kv0 = j * 64;
    VMCNT(4);
    __builtin_amdgcn_s_barrier();
    stage(cur == 0 ? 2 : cur - 1, kv0 + 128, j + 2 < n_kv);
    const unsigned char* lk = lds + cur * KV_STAGE_BYTES;
    const unsigned char* lv = lk + 8192;
    cur = cur == KV_STAGES - 1 ? 0 : cur + 1;
    // (no per-wave skip of the fully masked last causal tile: see the forward kernel)
    const bool need_mask = (kv0 + 64 > P.Lk) || (P.causal && kv0 + 63 > q0 + uw * (16 * RT));
    f32x4 dsT[RT][4];  // [qt][kt] : dS^T[key = kt*16+4g+r][q = fr]
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      bf16x8 ka0 = lds_row8(lk, kt * 16 + fr, fg);
      bf16x8 ka1 = lds_row8(lk, kt * 16 + fr, 4 + fg);
      bf16x8 va0 = lds_row8(lv, kt * 16 + fr, fg);
      bf16x8 va1 = lds_row8(lv, kt * 16 + fr, 4 + fg);
#pragma unroll
      for (int qt = 0; qt < RT; ++qt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = mfma16(ka0, qf[qt][0], s);
        s = mfma16(ka1, qf[qt][1], s);
        dp = mfma16(va0, dof[qt][0], dp);
        dp = mfma16(va1, dof[qt][1], dp);
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -lse_q[qt]));
        if (__builtin_expect(need_mask, 0)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kv0 + kt * 16 + fg * 4 + r;
            if (key >= P.Lk || (P.causal && key > qrow[qt])) pv[r] = 0.f;
          }
        }
        float dk_[4] = {dp[0], dp[1], dp[2], dp[3]};
        if (DROP && P.drop.thresh8) {    // run-time test on purpose, see the forward kernel
          const unsigned g = mix24(drop_q[qt] + (unsigned)((kv0 >> 2) + kt * 4) * DROP_CK);
#pragma unroll
          for (int r = 0; r < 4; ++r) dk_[r] = drop_sel(P.drop, g, r, dk_[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dsT[qt][kt][r] = pv[r] * fmaf(dk_[r], P.drop.scale, -dlt_q[qt]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 db[RT];
#pragma unroll
      for (int qt = 0; qt < RT; ++qt) db[qt] = pack8(dsT[qt][2 * ks], dsT[qt][2 * ks + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x8 kt_ = lds_tr8(lk, ks * 32 + fg * 4 + fq, dt * 2 + (fp >> 1), (fp & 1) * 8);
#pragma unroll
        for (int qt = 0; qt < RT; ++qt) dqT[qt][dt] = mfma16(kt_, db[qt], dqT[qt][dt]);
      }
    }
  }
  VMCNT(0);
  // (the lane's first row goes through an opaque move and the rows are re-derived from it: hipcc otherwise keeps the
  // prologue's qrow[] / row pointers alive across the key loop in registers of their own — at the 168 registers three
  // workgroups per CU allow, those were what it spilled: 4 VGPRs / 20 B of scratch in <true, true, 2>)
  int row0 = q0 + uw * (16 * RT) + fr;
  asm volatile("" : "+v"(row0));
#pragma unroll
  for (int qt = 0; qt < RT; ++qt) {
    const int qr = row0 + qt * 16;
    if (qr >= P.Lq) continue;
    bf16_t* row = P.dq + ((size_t)b * P.Lq + qr) * P.lddq + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 a = dqT[qt][dt];
      *(u32x2*)(row + dt * 16 + fg * 4) = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
    }
  }
  }  // pass
}

// exact-f32 attention (the reference's `precision: 32`) lives in attention_general.hip: one workgroup per query / key row,
// f32 arithmetic, the same masks; these two entry points launch it without a bias.
extern "C" int mrmt3_attn_bwd_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                  int ldo, const float* d_o, int lddo, const float* lse, float* delta, float* dq, int lddq,
                                  float* dk, int lddk, float* dv, int lddv, int B, int H, int Lq, int Lk, int causal,
                                  float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attn_bwd_f32: null pointer");
  MR_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_bwd_f32: bad sizes");
  return mrmt3_attn_general_bwd(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse, delta, nullptr, 0, dq, lddq, dk, lddk, dv,
                                lddv, nullptr, B, H, Lq, Lk, causal, MRMT3_F32,
                                make_attn_drop(p_drop, seed, stream_id, step_dev), (hipStream_t)stream);
}

extern "C" int mrmt3_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o,
                              int ldo, void* o_lo, float* lse, int B, int H, int Lq, int Lk, int causal, int dtype, float p_drop,
                              uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(q && k && v && o, "attn_fwd: null pointer");
  MR_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_fwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MRMT3_F32)
    return mrmt3_attn_general_fwd(q, ldq, k, ldk, v, ldv, nullptr, 0, o, ldo, lse, B, H, Lq, Lk, causal, MRMT3_F32,
                                  make_attn_drop(p_drop, seed, stream_id, step_dev), s);
  MR_CHECK_ARG(dtype == MRMT3_BF16, "attn_fwd: unknown dtype");
  MR_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd: bf16 strides must be multiples of 8");
  AttnParams P;
  memset(&P, 0, sizeof(P));
  P.q = (const bf16_t*)q; P.k = (const bf16_t*)k; P.v = (const bf16_t*)v; P.out = (bf16_t*)o; P.lse = lse;
  P.o_lo_out = (bf16_t*)o_lo;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo;
  P.B = B; P.H = H; P.Lq = Lq; P.Lk = Lk; P.causal = causal;
  P.drop = make_attn_drop(p_drop, seed, stream_id, step_dev);
  P.tile_mode = attn_tile_mode();
  const bool pair = attn_paired(Lq, causal, H, B);
  if (attn_fine(Lq, pair, causal, H, B)) {                          // small launch: 64-row tiles
    const dim3 gf(ceil_div(Lq, 64), H, B);
    if (P.drop.thresh8) hipLaunchKernelGGL((attn_fwd_kernel<false, true, 1>), gf, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false, 1>), gf, dim3(256), 0, s, P);
    MR_CHECK_LAUNCH("attn_fwd");
    mrmt3_count(MRMT3_CNT_ATTN_FWD);
    return MRMT3_OK;
  }
  const dim3 grid(attn_grid_x(Lq, pair), H, B);
  if (pair && P.drop.thresh8) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, dim3(256), 0, s, P);
  else if (pair) hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, dim3(256), 0, s, P);
  else if (P.drop.thresh8) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(256), 0, s, P);
  MR_CHECK_LAUNCH("attn_fwd");
  mrmt3_count(MRMT3_CNT_ATTN_FWD);
  return MRMT3_OK;
}

extern "C" int mrmt3_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                              int ldo, const void* o_lo, const void* d_o, int lddo, const float* lse, float* delta, void* dq, int lddq,
                              void* dk, int lddk, void* dv, int lddv, int B, int H, int Lq, int Lk, int causal,
                              float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attn_bwd: null pointer");
  MR_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_bwd: bad sizes");
  MR_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && ldo % 8 == 0 && lddq % 4 == 0 &&
                   lddk % 4 == 0 && lddv % 4 == 0, "attn_bwd: strides must be multiples of 8 (inputs) / 4 (outputs)");
  AttnParams P;
  memset(&P, 0, sizeof(P));
  P.q = (const bf16_t*)q; P.k = (const bf16_t*)k; P.v = (const bf16_t*)v; P.o = (const bf16_t*)o;
  P.d_o = (const bf16_t*)d_o; P.lse = (float*)lse; P.delta = delta;
  P.o_lo_in = (const bf16_t*)o_lo;
  P.dq = (bf16_t*)dq; P.dk = (bf16_t*)dk; P.dv = (bf16_t*)dv;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo; P.lddo = lddo; P.lddq = lddq; P.lddk = lddk; P.lddv = lddv;
  P.B = B; P.H = H; P.Lq = Lq; P.Lk = Lk; P.causal = causal;
  P.drop = make_attn_drop(p_drop, seed, stream_id, step_dev);
  P.tile_mode = attn_tile_mode();
  hipStream_t s = (hipStream_t)stream;
  if (mrmt3_attn_bwd_onepass_try(P, s)) {      // all keys of a (batch, head) in one workgroup: dQ, dK, dV in one pass
    MR_CHECK_LAUNCH("attn_bwd onepass");
    mrmt3_count(MRMT3_CNT_ATTN_BWD_ONEPASS);
    return MRMT3_OK;
  }
  // dQ first: it derives delta = rowsum(dO * O) from operands it loads anyway and leaves it for dK/dV
  const bool pair_q = attn_paired(Lq, causal, H, B), pair_k = attn_paired(Lk, causal, H, B);
  const dim3 gq(attn_grid_x(Lq, pair_q), H, B), gk(attn_grid_x(Lk, pair_k), H, B);
  const bool drop = P.drop.thresh8 != 0;
  const bool fine_q = attn_fine(Lq, pair_q, causal, H, B), fine_k = attn_fine(Lk, pair_k, causal, H, B);
  if (fine_q) {
    const dim3 gf(ceil_div(Lq, 64), H, B);
    if (drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true, 1>), gf, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false, 1>), gf, dim3(256), 0, s, P);
  } else
  if (pair_q && drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true>), gq, dim3(256), 0, s, P);
  else if (pair_q) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), gq, dim3(256), 0, s, P);
  else if (drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), gq, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), gq, dim3(256), 0, s, P);
  MR_CHECK_LAUNCH("attn_bwd dq");
  if (fine_k) {
    const dim3 gf(ceil_div(Lk, 64), H, B);
    if (drop) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, true, 1>), gf, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, false, 1>), gf, dim3(256), 0, s, P);
  } else
  if (pair_k && drop) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, true>), gk, dim3(256), 0, s, P);
  else if (pair_k) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, false>), gk, dim3(256), 0, s, P);
  else if (drop) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, true>), gk, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, false>), gk, dim3(256), 0, s, P);
  MR_CHECK_LAUNCH("attn_bwd dkdv");
  mrmt3_count(MRMT3_CNT_ATTN_BWD);
  return MRMT3_OK;
}
