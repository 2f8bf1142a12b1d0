// K5, forward, for SHORT key sets: every key of a (batch, head) resident in LDS.
//
// The decoder's cross-attention (HF T5LayerCrossAttention as called at models/t5.py:636-648) attends 1024 queries to the
// encoder's 256 frames — 320 with MR-MT3's 64 memory slots (models/t5_segmem_v2_with_prev.py:125-128).  The streaming
// forward kernel (attention.hip) runs that as 8 workgroups of 128 queries per (batch, head), each of which stages the same
// 4-5 key tiles through a three-deep LDS pipeline: per 128 queries a prologue, one barrier and one counted wait per tile, and
// the whole K | V of the head pulled from the L2 again.  A 50 us launch spends about 30 % of its time outside the tile loop
// (DESIGN 0d).  Here a workgroup loads K | V of the head ONCE — 64 or 80 KiB, LDS-DMA, one barrier — and then walks
// 256 queries (two 128-query tiles) against it with no synchronisation at all: the waves only read the LDS, the next tile's
// Q fragments are requested before the current tile's arithmetic.  Two workgroups per CU.
//
// The tile arithmetic — scores transposed, online softmax in the exp2 domain, dropout keyed by absolute (query, key), P as
// the MFMA operand of O^T = V^T P^T, the epilogue with the low half of O — is attention.hip's, statement for statement:
// the two kernels give the same bits (tests/test_kernels_gpu.py::test_attn_fwd_resident_equals_the_streaming_kernel), so
// the backward kernels, which recompute P from the saved log-sum-exp, do not care which one ran.
//
// NOT PART OF THE PRODUCT BUILD (round 5: built, bit-identical to the streaming kernel, 0.02-0.10 ms slower in the step;
// profiles/r05_attn_resident_ab.txt).  To rebuild the experiment: copy into mr-mt3_amd/csrc/, add to SRCS in the Makefile,
// declare mrmt3_attn_fwd_resident_try in attn_common.h and call it at the top of mrmt3_attn_fwd's bf16 branch.
#include "common.h"

#include "attn_common.h"

#define AR_TILE_BYTES 16384            // K tile 8 KiB + V tile 8 KiB (the streaming kernel's stage layout)
#define AR_Q_PER_WG 256                // queries per workgroup: two passes of 128

template <bool DROP, int NT>
__global__ __launch_bounds__(256, 2) void attn_fwd_resident_kernel(AttnParams P) {
  constexpr int RT = 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NT * AR_TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & 15, fg = lane >> 4, fq = fr >> 2, fp = lane & 3;
  int tile_, h, b;
  attn_tile(tile_, h, b, 0);                                // (every workgroup of a (batch, head) on one XCD)
  const bf16_t* qb = P.q + (size_t)b * P.Lq * P.ldq + h * HD;
  const bf16_t* kb = P.k + (size_t)b * P.Lk * P.ldk + h * HD;
  const bf16_t* vb = P.v + (size_t)b * P.Lk * P.ldv + h * HD;
  const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * P.H + h) * DROP_CB;
  const bool wide_rows = (P.ldo & 7) == 0 && (((uintptr_t)P.out | (uintptr_t)P.o_lo_out) & 15) == 0;

  // ---- K | V of the head: NT tiles of 64 keys, every wave 16 rows of K and of V per tile; rows past Lk come back as zeros
  {
    const __amdgpu_buffer_rsrc_t kres = rows_rsrc(kb, P.Lk, P.ldk), vres = rows_rsrc(vb, P.Lk, P.ldv);
    const unsigned k_lane = rows8_lane_off(P.ldk, lane), v_lane = rows8_lane_off(P.ldv, lane);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = uw * 2 + i;
        blds_rows8(kres, k_lane, (j * 64 + t * 8) * P.ldk * 2, lds + j * AR_TILE_BYTES + t * 1024);
        blds_rows8(vres, v_lane, (j * 64 + t * 8) * P.ldv * 2, lds + j * AR_TILE_BYTES + 8192 + t * 1024);
      }
  }
  const int n_kv = ceil_div(P.Lk, 64);                       // <= NT
  const int q_first = tile_ * AR_Q_PER_WG;

  // Q fragments of a 128-query pass (B operand): lane holds Q[q = qrow(qt)][d = 32 ks + 8 g .. + 7]
  auto load_q = [&](int q0, bf16x8 qf[RT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < RT; ++qt) {
      const int r = min(q0 + uw * (16 * RT) + qt * 16 + fr, P.Lq - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *(const bf16x8*)(qb + (size_t)r * P.ldq + ks * 32 + fg * 8);
    }
  };
  bf16x8 qf[RT][2], qn[RT][2];
  load_q(q_first, qf);
  VMCNT(0);
  __builtin_amdgcn_s_barrier();                              // the only one: from here on the LDS is read-only

#pragma nounroll
  for (int pass = 0; pass < AR_Q_PER_WG / 128; ++pass) {
    const int q0 = q_first + pass * 128;
    if (q0 >= P.Lq) break;
    if (pass + 1 < AR_Q_PER_WG / 128) load_q(q0 + 128, qn);      // the next pass's fragments (rows clamped), behind this pass's arithmetic
    int qrow[RT];
    unsigned drop_q[RT];
#pragma unroll
    for (int qt = 0; qt < RT; ++qt) {
      qrow[qt] = q0 + uw * (16 * RT) + qt * 16 + fr;
      drop_q[qt] = drop_bh + (unsigned)qrow[qt] * DROP_CQ + (unsigned)fg * DROP_CK;
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

    for (int j = 0; j < n_kv; ++j) {
      const int kv0 = j * 64;
      const unsigned char* lk = lds + j * AR_TILE_BYTES;
      const unsigned char* lv = lk + 8192;
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
      const bool need_mask = kv0 + 64 > P.Lk;
#pragma unroll
      for (int qt = 0; qt < RT; ++qt) {
        if (__builtin_expect(need_mask, 0)) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kv0 + kt * 16 + fg * 4 + r;
              if (key >= P.Lk) sT[qt][kt][r] = -INFINITY;
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sT[qt][kt][r]);
        mloc = rows_max(mloc);
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
        if (DROP && P.drop.thresh8) {                          // (its own basic block on purpose: see attention.hip)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            const unsigned g = mix24(drop_q[qt] + (unsigned)((kv0 >> 2) + kt * 4) * DROP_CK);
#pragma unroll
            for (int r = 0; r < 4; ++r) sT[qt][kt][r] = drop_sel(P.drop, g, r, sT[qt][kt][r]);
          }
        }
      }
      // O^T += V^T . P^T
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

#pragma unroll
    for (int qt = 0; qt < RT; ++qt) {
      float l = l_run[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const bool row_ok = qrow[qt] < P.Lq;
      const float inv = l > 0.f ? P.drop.scale / l : 0.f;
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
      if (wide_rows) {
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
      if (row_ok && fg == 0 && P.lse) P.lse[((size_t)b * P.H + h) * P.Lq + qrow[qt]] = fmaf(m_run[qt], LN2, __logf(l));
    }
    if (pass + 1 < AR_Q_PER_WG / 128) {
#pragma unroll
      for (int qt = 0; qt < RT; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = qn[qt][ks];
    }
  }
}

// 1 = launched.  Takes: not causal, at most 320 keys (5 tiles: 80 KiB, two workgroups per CU), at least 256 queries, and
// enough workgroups of 256 queries to fill the chip's 512 slots — below that the streaming kernel's 128- or 64-query
// workgroups spread the launch better.  Knob MRMT3_ATTN_RESIDENT = 0 switches it off (A/B, parity tests).
int mrmt3_attn_fwd_resident_try(const AttnParams& P, hipStream_t s) {
  if (MR_KNOB("MRMT3_ATTN_RESIDENT", 1) == 0) return 0;
  const int min_wg = MR_KNOB("MRMT3_ATTN_RESIDENT_MIN_WG", 512);
  const int nx = ceil_div(P.Lq, AR_Q_PER_WG);
  if (P.causal || P.Lk > 320 || P.Lq < AR_Q_PER_WG || (long long)nx * P.H * P.B < min_wg) return 0;
  const dim3 grid((unsigned)nx, (unsigned)P.H, (unsigned)P.B);
  const bool drop = P.drop.thresh8 != 0;
  if (P.Lk <= 256) {
    if (drop) hipLaunchKernelGGL((attn_fwd_resident_kernel<true, 4>), grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((attn_fwd_resident_kernel<false, 4>), grid, dim3(256), 0, s, P);
  } else {
    if (drop) hipLaunchKernelGGL((attn_fwd_resident_kernel<true, 5>), grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((attn_fwd_resident_kernel<false, 5>), grid, dim3(256), 0, s, P);
  }
  return 1;
}
