// The general attention path: exact-f32 arithmetic on f32 OR bf16 operands, with the optional ADDITIVE BIAS of HF
// T5Attention (`scores += position_bias`, models/t5.py:636-648 — the relative-position bias of stock T5; MR-MT3 itself
// passes zeros, models/t5.py:487-490, which is why the MFMA flash kernels of attention.hip carry no bias operand) and
// its gradient.  One workgroup per query row (forward; delta, dS, dQ, dBias) and per key row (dK, dV); every sum runs
// in a fixed order, no atomics.  This is the parity / fp32-training / bias path, not the throughput path:
//   * mrmt3_attn_fwd(dtype = MRMT3_F32) and mrmt3_attn_bwd_f32 launch it with bias = NULL (`precision: 32`);
//   * mrmt3_attn_fwd_bias / mrmt3_attn_bwd_bias are the SURVEY §8b `attn(q, k, v, bias_or_null, ...)` boundary.
// The dropout masks are those of the bf16 kernels (attn_common.h), one element at a time.
#include "attn_common.h"

namespace {

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
  const u32x2 w = *(const u32x2*)p;
  return f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16),
               __uint_as_float(w.y & 0xFFFF0000u)};
}
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// element (q, key) of head-matrix bh kept?  (the generator of the bf16 kernels, one element at a time)
__device__ __forceinline__ bool attn_keep1(const AttnDrop& d, unsigned drop_bh, int q, int key) {
  const unsigned w = mix24(drop_bh + (unsigned)q * DROP_CQ + ((unsigned)key >> 2) * DROP_CK);
  return ((w >> (8 * (key & 3))) & 0xFFu) >= d.thresh8;
}

struct GenP {
  const void *q, *k, *v, *o, *d_o;
  void *out, *dq, *dk, *dv;
  const float* bias;   // nullable: [H][Lq][Lk] (bias_bs = 0, shared by the batch) or [B][H][Lq][Lk] (bias_bs = H*Lq*Lk)
  float* dbias;        // nullable, same layout as bias
  long long bias_bs;
  float *lse, *delta;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, H, Lq, Lk, causal;
  int bloop;           // batches one dQ workgroup walks (B when it sums dBias over the batch, else 1)
  AttnDrop drop;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_gen_fwd_kernel(GenP P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // scores[Lk] | q[64] | red[8] | part[4][64]
  const int Lk = P.Lk, Lq = P.Lq, H = P.H;
  float* sc = sm;
  float* qs = sm + Lk;
  float* red = qs + 64;
  float* part = red + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const T* qp = (const T*)P.q + ((size_t)b * Lq + qi) * P.ldq + h * HD;
  const T* kb = (const T*)P.k + (size_t)b * Lk * P.ldk + h * HD;
  const T* vb = (const T*)P.v + (size_t)b * Lk * P.ldv + h * HD;
  const float* bias = P.bias ? P.bias + (size_t)b * P.bias_bs + ((size_t)h * Lq + qi) * Lk : nullptr;
  if (tid < 64) qs[tid] = ld1(qp + tid);
  __syncthreads();
  const int nk = P.causal ? min(Lk, qi + 1) : Lk;
  float mx = -INFINITY;
  for (int key = tid; key < nk; key += 256) {
    const T* kr = kb + (size_t)key * P.ldk;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 kv = ld4(kr + d);
      s = fmaf(qs[d], kv.x, s); s = fmaf(qs[d + 1], kv.y, s); s = fmaf(qs[d + 2], kv.z, s); s = fmaf(qs[d + 3], kv.w, s);
    }
    if (bias) s += bias[key];
    sc[key] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int key = tid; key < nk; key += 256) {
    const float p = expf(sc[key] - mx);
    sc[key] = p;
    se += p;
  }
  se = wave_sum(se);
  if (lane == 0) red[4 + wave] = se;
  __syncthreads();
  se = red[4] + red[5] + red[6] + red[7];
  if (P.drop.thresh8) {      // the normaliser is the sum of ALL probabilities; dropped ones leave the product only
    const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * H + h) * DROP_CB;
    for (int key = tid; key < nk; key += 256)
      if (!attn_keep1(P.drop, drop_bh, qi, key)) sc[key] = 0.f;
    __syncthreads();
  }
  float acc = 0.f;
  for (int key = wave; key < nk; key += 4) acc = fmaf(sc[key], ld1(vb + (size_t)key * P.ldv + lane), acc);
  part[wave * 64 + lane] = acc;
  __syncthreads();
  if (tid < 64) {
    const float r = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
    st1((T*)P.out + ((size_t)b * Lq + qi) * P.ldo + h * HD + tid, r * P.drop.scale / se);
  }
  if (tid == 0 && P.lse) P.lse[((size_t)b * H + h) * Lq + qi] = mx + logf(se);
}

// (1) one workgroup per query row: delta = rowsum(dO * O), dS = P (keep * scale * dP - delta), dQ = dS . K,
//     dBias = dS (summed over the `bloop` batches this workgroup walks, in batch order)
template <typename T>
__global__ __launch_bounds__(256) void attn_gen_bwd_dq_kernel(GenP P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // ds[Lk] | q[64] | dO[64] | red[4] | part[4][64] | dbias[Lk]
  const int Lk = P.Lk, Lq = P.Lq, H = P.H;
  float* sc = sm;
  float* qs = sm + Lk;
  float* dos = qs + 64;
  float* red = dos + 64;
  float* part = red + 4;
  float* dbs = part + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = blockIdx.x, h = blockIdx.y;
  const int nk = P.causal ? min(Lk, qi + 1) : Lk;
  if (P.dbias)
    for (int key = tid; key < Lk; key += 256) dbs[key] = 0.f;
  for (int b = blockIdx.z * P.bloop; b < (int)(blockIdx.z + 1) * P.bloop; ++b) {
    const size_t qrow = (size_t)b * Lq + qi;
    const T* kb = (const T*)P.k + (size_t)b * Lk * P.ldk + h * HD;
    const T* vb = (const T*)P.v + (size_t)b * Lk * P.ldv + h * HD;
    const float* bias = P.bias ? P.bias + (size_t)b * P.bias_bs + ((size_t)h * Lq + qi) * Lk : nullptr;
    const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * H + h) * DROP_CB;
    __syncthreads();           // (the previous batch's readers of qs / dos / sc / part are done)
    if (tid < 64) {
      qs[tid] = ld1((const T*)P.q + qrow * P.ldq + h * HD + tid);
      dos[tid] = ld1((const T*)P.d_o + qrow * P.lddo + h * HD + tid);
    }
    __syncthreads();
    float dl = 0.f;
    if (wave == 0) {
      dl = wave_sum(dos[lane] * ld1((const T*)P.o + qrow * P.ldo + h * HD + lane));
      if (lane == 0) { red[0] = dl; P.delta[((size_t)b * H + h) * Lq + qi] = dl; }
    }
    __syncthreads();
    dl = red[0];
    const float l = P.lse[((size_t)b * H + h) * Lq + qi];
    for (int key = tid; key < nk; key += 256) {
      const T* kr = kb + (size_t)key * P.ldk;
      const T* vr = vb + (size_t)key * P.ldv;
      float sv = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const f32x4 kv = ld4(kr + d), vv = ld4(vr + d);
        sv = fmaf(qs[d], kv.x, sv); sv = fmaf(qs[d + 1], kv.y, sv); sv = fmaf(qs[d + 2], kv.z, sv); sv = fmaf(qs[d + 3], kv.w, sv);
        dp = fmaf(dos[d], vv.x, dp); dp = fmaf(dos[d + 1], vv.y, dp); dp = fmaf(dos[d + 2], vv.z, dp); dp = fmaf(dos[d + 3], vv.w, dp);
      }
      if (bias) sv += bias[key];
      const float pr = expf(sv - l);
      if (P.drop.thresh8) dp = attn_keep1(P.drop, drop_bh, qi, key) ? dp * P.drop.scale : 0.f;
      const float ds = pr * (dp - dl);
      sc[key] = ds;
      if (P.dbias) dbs[key] += ds;     // (each key is owned by one thread: fixed order over the batches)
    }
    __syncthreads();
    float acc = 0.f;
    for (int key = wave; key < nk; key += 4) acc = fmaf(sc[key], ld1(kb + (size_t)key * P.ldk + lane), acc);
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (tid < 64)
      st1((T*)P.dq + qrow * P.lddq + h * HD + tid, (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]));
  }
  if (P.dbias) {
    float* out = P.dbias + (size_t)blockIdx.z * P.bias_bs + ((size_t)h * Lq + qi) * Lk;
    for (int key = tid; key < Lk; key += 256) out[key] = dbs[key];     // (own keys only: no barrier needed; masked keys 0)
  }
}

// (2) one workgroup per key row: dV = Pd^T dO, dK = dS^T Q
template <typename T>
__global__ __launch_bounds__(256) void attn_gen_bwd_dkdv_kernel(GenP P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // pd[Lq] | ds[Lq] | k[64] | v[64] | part[2][4][64]
  const int Lk = P.Lk, Lq = P.Lq, H = P.H;
  float* pd = sm;
  float* ds = sm + Lq;
  float* ks = ds + Lq;
  float* vs = ks + 64;
  float* part = vs + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int key = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const size_t krow = (size_t)b * Lk + key;
  const T* qb = (const T*)P.q + (size_t)b * Lq * P.ldq + h * HD;
  const T* dob = (const T*)P.d_o + (size_t)b * Lq * P.lddo + h * HD;
  const float* lseb = P.lse + ((size_t)b * H + h) * Lq;
  const float* dltb = P.delta + ((size_t)b * H + h) * Lq;
  const float* bias = P.bias ? P.bias + (size_t)b * P.bias_bs + (size_t)h * Lq * Lk + key : nullptr;
  const unsigned drop_bh = P.drop.seed + step_salt(P.drop.step) + (unsigned)(b * H + h) * DROP_CB;
  if (tid < 64) {
    ks[tid] = ld1((const T*)P.k + krow * P.ldk + h * HD + tid);
    vs[tid] = ld1((const T*)P.v + krow * P.ldv + h * HD + tid);
  }
  __syncthreads();
  const int q_lo = P.causal ? key : 0;
  for (int qi = q_lo + tid; qi < Lq; qi += 256) {
    const T* qr = qb + (size_t)qi * P.ldq;
    const T* dr = dob + (size_t)qi * P.lddo;
    float sv = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 qv = ld4(qr + d), dv4 = ld4(dr + d);
      sv = fmaf(qv.x, ks[d], sv); sv = fmaf(qv.y, ks[d + 1], sv); sv = fmaf(qv.z, ks[d + 2], sv); sv = fmaf(qv.w, ks[d + 3], sv);
      dp = fmaf(dv4.x, vs[d], dp); dp = fmaf(dv4.y, vs[d + 1], dp); dp = fmaf(dv4.z, vs[d + 2], dp); dp = fmaf(dv4.w, vs[d + 3], dp);
    }
    if (bias) sv += bias[(size_t)qi * Lk];
    const float pr = expf(sv - lseb[qi]);
    float keep = 1.f;
    if (P.drop.thresh8) keep = attn_keep1(P.drop, drop_bh, qi, key) ? P.drop.scale : 0.f;
    pd[qi] = pr * keep;
    ds[qi] = pr * (dp * keep - dltb[qi]);
  }
  __syncthreads();
  float av = 0.f, ak = 0.f;
  for (int qi = q_lo + wave; qi < Lq; qi += 4) {
    av = fmaf(pd[qi], ld1(dob + (size_t)qi * P.lddo + lane), av);
    ak = fmaf(ds[qi], ld1(qb + (size_t)qi * P.ldq + lane), ak);
  }
  part[wave * 64 + lane] = av;
  part[256 + wave * 64 + lane] = ak;
  __syncthreads();
  if (tid < 64) {
    st1((T*)P.dv + krow * P.lddv + h * HD + tid, (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]));
    st1((T*)P.dk + krow * P.lddk + h * HD + tid, (part[256 + tid] + part[320 + tid]) + (part[384 + tid] + part[448 + tid]));
  }
}

int check_bias(const float* bias, long long bias_bs, int H, int Lq, int Lk, const char* who) {
  if (bias && bias_bs != 0 && bias_bs != (long long)H * Lq * Lk) {
    mrmt3_set_error("%s: bias batch stride must be 0 (one [H][Lq][Lk] bias for the batch) or H*Lq*Lk", who);
    return MRMT3_ERR_INVALID_ARG;
  }
  return MRMT3_OK;
}

}  // namespace

int mrmt3_attn_general_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* bias,
                           long long bias_bs, void* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int causal,
                           int dtype, const AttnDrop& drop, hipStream_t s) {
  MR_CHECK_ARG(dtype == MRMT3_F32 || dtype == MRMT3_BF16, "attn (general path): unknown dtype");
  MR_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "attn (general path): strides must be multiples of 4");
  if (int rc = check_bias(bias, bias_bs, H, Lq, Lk, "attn_fwd_bias")) return rc;
  const size_t shm = (size_t)(Lk + 64 + 8 + 256) * sizeof(float);
  MR_CHECK_ARG(shm <= 160 * 1024, "attn (general path): Lk too large");
  GenP P;
  memset(&P, 0, sizeof(P));
  P.q = q; P.k = k; P.v = v; P.out = o; P.lse = lse; P.bias = bias; P.bias_bs = bias ? bias_bs : 0;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo;
  P.B = B; P.H = H; P.Lq = Lq; P.Lk = Lk; P.causal = causal; P.bloop = 1; P.drop = drop;
  if (dtype == MRMT3_F32) hipLaunchKernelGGL(attn_gen_fwd_kernel<float>, dim3(Lq, H, B), dim3(256), shm, s, P);
  else hipLaunchKernelGGL(attn_gen_fwd_kernel<bf16_t>, dim3(Lq, H, B), dim3(256), shm, s, P);
  MR_CHECK_LAUNCH("attn_fwd (general path)");
  mrmt3_count(MRMT3_CNT_ATTN_F32);
  return MRMT3_OK;
}

int mrmt3_attn_general_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                           const void* d_o, int lddo, const float* lse, float* delta, const float* bias, long long bias_bs,
                           void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, float* dbias, int B, int H, int Lq,
                           int Lk, int causal, int dtype, const AttnDrop& drop, hipStream_t s) {
  MR_CHECK_ARG(dtype == MRMT3_F32 || dtype == MRMT3_BF16, "attn_bwd (general path): unknown dtype");
  MR_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && lddo % 4 == 0,
               "attn_bwd (general path): strides must be multiples of 4");
  MR_CHECK_ARG(!dbias || bias, "attn_bwd_bias: dbias without bias (its layout is the bias's)");
  if (int rc = check_bias(bias, bias_bs, H, Lq, Lk, "attn_bwd_bias")) return rc;
  const size_t shm_q = (size_t)(2 * Lk + 64 + 64 + 4 + 256) * sizeof(float), shm_k = (size_t)(2 * Lq + 128 + 512) * sizeof(float);
  MR_CHECK_ARG(shm_q <= 160 * 1024 && shm_k <= 160 * 1024, "attn_bwd (general path): sequence too long for the parity kernel");
  GenP P;
  memset(&P, 0, sizeof(P));
  P.q = q; P.k = k; P.v = v; P.o = o; P.d_o = d_o; P.lse = (float*)lse; P.delta = delta;
  P.bias = bias; P.dbias = dbias; P.bias_bs = bias ? bias_bs : 0;
  P.dq = dq; P.dk = dk; P.dv = dv;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo; P.lddo = lddo; P.lddq = lddq; P.lddk = lddk; P.lddv = lddv;
  P.B = B; P.H = H; P.Lq = Lq; P.Lk = Lk; P.causal = causal; P.drop = drop;
  P.bloop = (dbias && P.bias_bs == 0) ? B : 1;      // a bias shared by the batch: its gradient sums over the batch
  const dim3 gq(Lq, H, B / P.bloop), gk(Lk, H, B);
  if (dtype == MRMT3_F32) {
    hipLaunchKernelGGL(attn_gen_bwd_dq_kernel<float>, gq, dim3(256), shm_q, s, P);
    MR_CHECK_LAUNCH("attn_bwd dq (general path)");
    hipLaunchKernelGGL(attn_gen_bwd_dkdv_kernel<float>, gk, dim3(256), shm_k, s, P);
  } else {
    hipLaunchKernelGGL(attn_gen_bwd_dq_kernel<bf16_t>, gq, dim3(256), shm_q, s, P);
    MR_CHECK_LAUNCH("attn_bwd dq (general path)");
    hipLaunchKernelGGL(attn_gen_bwd_dkdv_kernel<bf16_t>, gk, dim3(256), shm_k, s, P);
  }
  MR_CHECK_LAUNCH("attn_bwd dkdv (general path)");
  mrmt3_count(MRMT3_CNT_ATTN_F32);
  return MRMT3_OK;
}

extern "C" int mrmt3_attn_fwd_bias(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                                   const float* bias, long long bias_batch_stride, void* o, int ldo, float* lse, int B,
                                   int H, int Lq, int Lk, int causal, int dtype, float p_drop, uint64_t seed,
                                   const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(q && k && v && o, "attn_fwd_bias: null pointer");
  MR_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_fwd_bias: bad sizes");
  return mrmt3_attn_general_fwd(q, ldq, k, ldk, v, ldv, bias, bias_batch_stride, o, ldo, lse, B, H, Lq, Lk, causal, dtype,
                                make_attn_drop(p_drop, seed, stream_id, step_dev), (hipStream_t)stream);
}

extern "C" int mrmt3_attn_bwd_bias(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                                   int ldo, const void* d_o, int lddo, const float* lse, float* delta, const float* bias,
                                   long long bias_batch_stride, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                                   float* dbias, int B, int H, int Lq, int Lk, int causal, int dtype, float p_drop,
                                   uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream) {
  MR_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attn_bwd_bias: null pointer");
  MR_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_bwd_bias: bad sizes");
  return mrmt3_attn_general_bwd(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse, delta, bias, bias_batch_stride, dq, lddq, dk,
                                lddk, dv, lddv, dbias, B, H, Lq, Lk, causal, dtype,
                                make_attn_drop(p_drop, seed, stream_id, step_dev), (hipStream_t)stream);
}
