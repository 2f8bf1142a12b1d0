// K1 — log-mel frontend for gfx950: framed STFT magnitude -> sparse HTK mel filterbank -> safe log
// -> clip/scale, one 256-thread workgroup per 2048-sample frame.
//
// Replaces (reference file:line):
//   contrib/spectrograms.py:92-98    pad_end            (zero pad folded into the frame loader)
//   contrib/spectrograms.py:128-141  torchaudio MelSpectrogram(n_fft=2048, hop=128, n_mels=512,
//                                    f_min=20, f_max=7600, power=1, center=False)
//   contrib/spectrograms.py:100-103  safe_log
//   dataset/dataset_2_random.py:288-289 == inference.py:115-117   clamp(-12,5), (x+12)/17
//   inference.py:125-126             zeroing of frames beyond the real audio (valid_frames)
//
// Layout in HBM: audio [B][n_samples] f32 (one segment per row); out [B][n_frames][n_mels] f32 or
// bf16.  A frame is read as 2048 consecutive floats (coalesced 8 KiB; the 16x overlap between
// neighbouring frames is served by L2), windowed into LDS, transformed by a 1024-point complex
// radix-4 Stockham FFT (the 2048 real samples packed as 1024 complex), unpacked to the 1025
// one-sided bins, and reduced to 512 mel bins through the filterbank's <=max_taps non-zeros per
// filter (the reference multiplies by the dense 1025x512 matrix, 99.6 % zeros).
// Roofline: HBM — 131072 B in + 524288 B out per 256-frame segment.
#include <stdlib.h>

#include "common.h"

#define FFT_N 2048
#define CFFT_N 1024
#define LM_THREADS 256

struct c2 { float x, y; };
__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return {a.x - b.x, a.y - b.y}; }

// tw[k] = exp(-2*pi*i*k/2048), k in [0,1024); indices in [1024,2048) by symmetry
__device__ __forceinline__ c2 twiddle(const c2* tw, int idx) {
  c2 t = tw[idx & 1023];
  if (idx & 1024) { t.x = -t.x; t.y = -t.y; }
  return t;
}

template <bool OUT_BF16>
__global__ __launch_bounds__(LM_THREADS) void logmel_kernel(
    const float* __restrict__ audio, int n_samples, int n_frames, int hop,
    const float* __restrict__ window,      // [2048]
    const float* __restrict__ twid,        // [1024][2]
    const int* __restrict__ fb_start,      // [n_mels]
    const int* __restrict__ fb_cnt,        // [n_mels]
    const float* __restrict__ fb_w,        // [n_mels][max_taps]
    int n_mels, int max_taps,
    const int* __restrict__ valid_frames,  // [B] or null
    const long long* __restrict__ seg_start,  // [B] sample offsets into one long recording, or null
    long long total_samples,
    float lo, float hi, int normalize, void* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) c2 buf0[CFFT_N];
  __shared__ __attribute__((aligned(16))) c2 buf1[CFFT_N];
  __shared__ __attribute__((aligned(16))) c2 tw[CFFT_N];
  const int tid = threadIdx.x;
  const int frame = blockIdx.x % n_frames;
  const int seg = blockIdx.x / n_frames;
  const size_t out_off = ((size_t)seg * n_frames + frame) * n_mels;

  if (valid_frames != nullptr && frame >= valid_frames[seg]) {  // F7: padded frame -> zeros
    for (int m = tid; m < n_mels; m += LM_THREADS) {
      if (OUT_BF16) ((bf16_t*)out)[out_off + m] = 0;
      else ((float*)out)[out_off + m] = 0.f;
    }
    return;
  }

  // twiddles -> LDS (8 KiB, L2-resident)
  for (int i = tid; i < CFFT_N; i += LM_THREADS) tw[i] = ((const c2*)twid)[i];
  // windowed frame -> LDS as 1024 complex (even sample = re, odd = im); zero beyond n_samples
  // batched rows, or crops gathered straight out of one recording: a crop sees zeros past its own end
  // (its valid frames / the end of the recording), never its neighbour's samples
  const float* src = audio + (size_t)seg * n_samples;
  if (seg_start != nullptr) {
    const long long st = seg_start[seg];
    src = audio + st;
    long long lim = total_samples - st;
    if (lim > n_samples) lim = n_samples;
    if (valid_frames != nullptr && (long long)valid_frames[seg] * hop < lim) lim = (long long)valid_frames[seg] * hop;
    n_samples = lim > 0 ? (int)lim : 0;
  }
  const int base = frame * hop;
#pragma unroll
  for (int r = 0; r < CFFT_N / LM_THREADS; ++r) {
    int c = tid + r * LM_THREADS;  // complex index
    int s = base + 2 * c;
    float2 w = *(const float2*)(window + 2 * c);
    float x0 = (s < n_samples) ? src[s] : 0.f;
    float x1 = (s + 1 < n_samples) ? src[s + 1] : 0.f;
    buf0[c] = {x0 * w.x, x1 * w.y};
  }
  __syncthreads();

  // 1024-point complex FFT: 5 radix-4 Stockham passes, one butterfly per thread per pass
  c2* in = buf0;
  c2* outb = buf1;
  const int t = CFFT_N / 4;
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int p = 1 << (2 * pass);
    const int k = tid & (p - 1);
    const int j = ((tid - k) << 2) + k;
    const int twm = (512 >> (2 * pass)) * k;  // twiddle index step for this butterfly
    c2 u0 = in[tid];
    c2 u1 = cmul(in[tid + t], twiddle(tw, twm));
    c2 u2 = cmul(in[tid + 2 * t], twiddle(tw, 2 * twm));
    c2 u3 = cmul(in[tid + 3 * t], twiddle(tw, 3 * twm));
    c2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3);
    c2 d = csub(u1, u3);
    c2 v3 = {d.y, -d.x};  // -i * (u1 - u3)
    outb[j] = cadd(v0, v2);
    outb[j + p] = cadd(v1, v3);
    outb[j + 2 * p] = csub(v0, v2);
    outb[j + 3 * p] = csub(v1, v3);
    __syncthreads();
    c2* tmp = in; in = outb; outb = tmp;
  }
  // `in` now holds Z = FFT1024(z).  Unpack to the real-input spectrum and take magnitudes:
  //   X[k] = (Z[k] + conj(Z[N-k]))/2 - i/2 * e^{-2 pi i k/2048} * (Z[k] - conj(Z[N-k])), k=0..1024
  float* mag = (float*)outb;  // 1025 floats fit in the idle ping-pong buffer (8 KiB)
  for (int k = tid; k <= CFFT_N; k += LM_THREADS) {
    c2 zk = in[k & (CFFT_N - 1)];
    c2 zn = in[(CFFT_N - k) & (CFFT_N - 1)];
    c2 a = {0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
    c2 b = {0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y)};
    c2 w = (k == CFFT_N) ? c2{-1.f, 0.f} : tw[k];
    c2 wb = cmul(w, b);          // e^{..} * (Z[k]-conj(Z[N-k]))/2
    c2 X = {a.x + wb.y, a.y - wb.x};  // a - i*wb
    mag[k] = sqrtf(X.x * X.x + X.y * X.y);
  }
  __syncthreads();

  // sparse mel + safe_log (+ clip/scale)
  for (int m = tid; m < n_mels; m += LM_THREADS) {
    const int s = fb_start[m], cnt = fb_cnt[m];
    const float* w = fb_w + (size_t)m * max_taps;
    float acc = 0.f;
    for (int q = 0; q < cnt; ++q) acc += mag[s + q] * w[q];
    float v = logf(acc <= 0.f ? 1e-5f : acc);
    if (normalize) {
      v = fminf(fmaxf(v, lo), hi);
      v = (v - lo) / (hi - lo);
    }
    if (OUT_BF16) ((bf16_t*)out)[out_off + m] = f2bf(v);
    else ((float*)out)[out_off + m] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: one WAVE per frame, 64 frames per workgroup, the whole CU's LDS.
//
// The kernel above spends its time on things that are not the transform: every one of 16 384 workgroups refills the same
// 8 KiB twiddle table (134 MB of L2 reads for 25 MB of payload), reads its frame with its 16-fold overlap out of L2, runs
// five barrier-separated radix-4 passes at one butterfly per thread, and walks the filterbank's weights in global memory:
// 140 us for 64 segments, 0.022 of the HBM roof.  Here
//   * a workgroup (8 waves) takes 64 consecutive frames of a segment: their 10 112 samples are staged ONCE in LDS (40 KB;
//     the frames' overlap is then LDS traffic), the unpack twiddles (8 KB) and the filterbank weights (transposed,
//     [tap][mel], 20 KB) once per workgroup;
//   * a WAVE transforms a frame on its own: 1024 complex points = 16 per lane, as 16 x 16 x 4 — a 16-point FFT in
//     registers over n1 (x[64 n1 + lane]), twiddle, a transpose through the wave's own 9-KB LDS buffer, a second 16-point
//     FFT in registers, twiddle, the same transpose back in place, and a last radix-4 from 32-byte LDS rows.  No
//     workgroup barrier anywhere in the frame loop: a wave's LDS instructions execute in order, and the compiler is held
//     to program order by wave barriers.  Window and both twiddle sets live in registers (loaded once per wave);
//   * the unpack to the 1025 real-input bins, the magnitudes, the sparse mel sums (8 consecutive mel bins per lane, the
//     taps in the old kernel's order: same mel arithmetic), log and scaling follow in the same wave, and a frame leaves
//     as one 16-byte store per lane (bf16).
// The workgroup's static LDS is the CU's whole 160 KB: profiles/r04_one_process_two_stream_soak.txt shows the old kernel
// coming out wrong (one launch in ~8) whenever workgroups of the flash-attention / round-1 tile kernels shared its CU — from
// another stream of the SAME process as much as from another process.  With all of the LDS taken no such kernel can be its
// co-tenant.  (Later in round 4 the fault itself was traced to packed f32 VALU instructions, which the library no longer
// contains — profiles/r04_lds_read_fault.txt, csrc/Makefile — so the old kernel is clean too; the exclusive request stays
// as a second line of defence and costs nothing: one 8-wave workgroup per CU is this kernel's shape anyway.)
#define LMW_WAVES 8
#define LMW_THREADS (LMW_WAVES * 64)
#define LMW_STRIP_FLOATS 10112                 // (64 - 1) * 128 + 2048
#define LMW_WBUF_BYTES 9216                    // per wave: 16 x 68 complex (transposes) / 1150 complex (padded natural order)
#define LMW_MAX_TAPS 12
#define LMW_LDS_BYTES 163840

__device__ __forceinline__ c2 mul_mi(c2 a) { return {a.y, -a.x}; }   // -i * a
__device__ __forceinline__ c2 mul_pi(c2 a) { return {-a.y, a.x}; }   // +i * a
// forward 16-point FFT in registers, natural order in and out: radix 4 x 4 (decimation in time)
__device__ __forceinline__ void fft16(c2 x[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
  // W_16^m = exp(-2 pi i m / 16) for the products n2 * k1 that occur
  const c2 W1 = {C1, -S1}, W2 = {C2, -C2}, W3 = {S1, -C1}, W4 = {0.f, -1.f}, W6 = {-C2, -C2}, W9 = {-C1, S1};
  c2 a[4][4];   // a[k1][n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    const c2 p = x[n2], q = x[4 + n2], r = x[8 + n2], s = x[12 + n2];
    const c2 pr = cadd(p, r), mr = csub(p, r), qs = cadd(q, s), ms = csub(q, s);
    a[0][n2] = cadd(pr, qs);
    a[1][n2] = cadd(mr, mul_mi(ms));
    a[2][n2] = csub(pr, qs);
    a[3][n2] = cadd(mr, mul_pi(ms));
  }
  a[1][1] = cmul(a[1][1], W1); a[1][2] = cmul(a[1][2], W2); a[1][3] = cmul(a[1][3], W3);
  a[2][1] = cmul(a[2][1], W2); a[2][2] = cmul(a[2][2], W4); a[2][3] = cmul(a[2][3], W6);
  a[3][1] = cmul(a[3][1], W3); a[3][2] = cmul(a[3][2], W6); a[3][3] = cmul(a[3][3], W9);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    const c2 p = a[k1][0], q = a[k1][1], r = a[k1][2], s = a[k1][3];
    const c2 pr = cadd(p, r), mr = csub(p, r), qs = cadd(q, s), ms = csub(q, s);
    x[k1] = cadd(pr, qs);
    x[k1 + 4] = cadd(mr, mul_mi(ms));
    x[k1 + 8] = csub(pr, qs);
    x[k1 + 12] = cadd(mr, mul_pi(ms));
  }
}
#define LMW_WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

template <bool OUT_BF16>
__global__ __launch_bounds__(LMW_THREADS, 2) void logmel_wave_kernel(
    const float* __restrict__ audio, int n_samples, int n_frames, int hop, int fpw /* frames per workgroup */,
    const float* __restrict__ window, const float* __restrict__ twid, const int* __restrict__ fb_start,
    const int* __restrict__ fb_cnt, const float* __restrict__ fb_w, int max_taps,
    const int* __restrict__ valid_frames, const long long* __restrict__ seg_start, long long total_samples,
    float lo, float hi, int normalize, void* __restrict__ out) {
  __shared__ __attribute__((aligned(32))) unsigned char lds[LMW_LDS_BYTES];
  float* const strip = (float*)lds;                                              // [LMW_STRIP_FLOATS]
  c2* const twl = (c2*)(lds + LMW_STRIP_FLOATS * 4);                             // [1024]
  float* const fbt = (float*)(lds + LMW_STRIP_FLOATS * 4 + 8192);                // [max_taps][512]
  unsigned char* const wbase = lds + LMW_STRIP_FLOATS * 4 + 8192 + LMW_MAX_TAPS * 2048;
  static_assert(LMW_STRIP_FLOATS * 4 + 8192 + LMW_MAX_TAPS * 2048 + LMW_WAVES * LMW_WBUF_BYTES <= LMW_LDS_BYTES, "LDS budget");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int groups = (n_frames + fpw - 1) / fpw;
  const int seg = blockIdx.x / groups, f0 = (blockIdx.x - seg * groups) * fpw;
  const int nf = min(fpw, n_frames - f0);

  // a crop sees zeros past its own end (its valid frames / the end of the recording), never its neighbour's samples
  const float* src = audio + (size_t)seg * n_samples;
  if (seg_start != nullptr) {
    const long long st = seg_start[seg];
    src = audio + st;
    long long lim = total_samples - st;
    if (lim > n_samples) lim = n_samples;
    if (valid_frames != nullptr && (long long)valid_frames[seg] * hop < lim) lim = (long long)valid_frames[seg] * hop;
    n_samples = lim > 0 ? (int)lim : 0;
  }
  const int vf = valid_frames != nullptr ? valid_frames[seg] : n_frames;
  // ---- once per workgroup: the frames' samples, the unpack twiddles, the filterbank weights [tap][mel]
  {
    const int s0 = f0 * hop, ns = (nf - 1) * hop + FFT_N;
    for (int i = tid; i < ns; i += LMW_THREADS) strip[i] = (s0 + i < n_samples) ? src[s0 + i] : 0.f;
    for (int i = tid; i < CFFT_N; i += LMW_THREADS) twl[i] = ((const c2*)twid)[i];
    for (int i = tid; i < max_taps * 512; i += LMW_THREADS) {     // coalesced read of [mel][tap], transposed into LDS
      const int m = i / max_taps, q = i - m * max_taps;
      // taps at and beyond fb_cnt[m] count as zero whatever the caller left there (the header's contract: sum over
      // q < fb_cnt[m]); masked here, once per workgroup, so the mel loop below can run all max_taps taps unconditionally
      fbt[q * 512 + m] = q < fb_cnt[m] ? fb_w[i] : 0.f;
    }
  }
  __syncthreads();
  // ---- once per wave, in registers: window, both twiddle sets, the lane's filter windows
  const int k1s = lane >> 2, qq = lane & 3;
  float2 win[16];
  c2 twA[16], twB[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    win[n1] = *(const float2*)(window + 2 * (64 * n1 + lane));
    // (gathered out of the LDS copy of the table: as 32 scattered global loads per lane these were most of a small launch)
    twA[n1] = twiddle(twl, 2 * (lane * n1));                       // W_1024^(lane n1) = exp(-2 pi i lane n1 / 1024)
    twB[n1] = twiddle(twl, 32 * (qq * n1));                        // W_64^(q r)
  }
  int fs[8];
  {
    const u32x4 s0 = *(const u32x4*)(fb_start + 8 * lane), s1 = *(const u32x4*)(fb_start + 8 * lane + 4);
    fs[0] = (int)s0.x; fs[1] = (int)s0.y; fs[2] = (int)s0.z; fs[3] = (int)s0.w;
    fs[4] = (int)s1.x; fs[5] = (int)s1.y; fs[6] = (int)s1.z; fs[7] = (int)s1.w;
    // a (zero-weight) padded tap reads mag[fb_start + q], q < max_taps <= LMW_MAX_TAPS: bins 0..1024 and the 63 zeros
    // behind them.  A start outside [0, 1024] is not a filter of a 1025-bin spectrum: clamped, so that no LDS read leaves
    // the wave's buffer whatever the table holds.
#pragma unroll
    for (int e = 0; e < 8; ++e) fs[e] = min(max(fs[e], 0), CFFT_N);
  }

  c2* const wb = (c2*)(wbase + w * LMW_WBUF_BYTES);
  for (int fl = w; fl < nf; fl += LMW_WAVES) {
    const int frame = f0 + fl;
    const size_t out_off = ((size_t)seg * n_frames + frame) * 512 + 8 * lane;
    if (frame >= vf) {                                              // F7: padded frame -> zeros
      if (OUT_BF16) *(u32x4*)((bf16_t*)out + out_off) = u32x4{0u, 0u, 0u, 0u};
      else { *(f32x4*)((float*)out + out_off) = f32x4{0.f, 0.f, 0.f, 0.f}; *(f32x4*)((float*)out + out_off + 4) = f32x4{0.f, 0.f, 0.f, 0.f}; }
      continue;
    }
    // (1) x[64 n1 + lane], windowed; 16-point FFT over n1; twiddle W_1024^(lane k1)
    c2 x[16];
    const float* fr = strip + fl * hop + 2 * lane;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const float2 v = *(const float2*)(fr + 128 * n1);
      x[n1] = {v.x * win[n1].x, v.y * win[n1].y};
    }
    fft16(x);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = cmul(x[k1], twA[k1]);
    LMW_WAVE_SYNC();                                                // (the previous frame's reads of this buffer are done)
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) wb[k1 * 68 + lane] = x[k1];
    LMW_WAVE_SYNC();
    // (2) lane (k1, q): the 16 points n2 = 4 m + q of row k1; 16-point FFT over m; twiddle W_64^(q r); back in place
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = wb[k1s * 68 + 4 * m + qq];
    fft16(x);
#pragma unroll
    for (int r = 1; r < 16; ++r) x[r] = cmul(x[r], twB[r]);
    LMW_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; ++r) wb[k1s * 68 + 4 * r + qq] = x[r];
    LMW_WAVE_SYNC();
    // (3) the last radix 4 over q: group g = (k1, r) -> Z[k1 + 16 r + 256 s], s = 0..3
    c2 z[4][4];
    f32x4 lo4s[4], hi4s[4];                                          // (all eight reads requested before the first is used)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int g = lane + 64 * t, k1 = g >> 4, r = g & 15;
      lo4s[t] = *(const f32x4*)(wb + k1 * 68 + 4 * r);
      hi4s[t] = *(const f32x4*)(wb + k1 * 68 + 4 * r + 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 lo4 = lo4s[t], hi4 = hi4s[t];
      const c2 v0 = {lo4.x, lo4.y}, v1 = {lo4.z, lo4.w}, v2 = {hi4.x, hi4.y}, v3 = {hi4.z, hi4.w};
      const c2 pr = cadd(v0, v2), mr = csub(v0, v2), qs = cadd(v1, v3), ms = csub(v1, v3);
      z[t][0] = cadd(pr, qs);
      z[t][1] = cadd(mr, mul_mi(ms));
      z[t][2] = csub(pr, qs);
      z[t][3] = cadd(mr, mul_pi(ms));
    }
    LMW_WAVE_SYNC();
    // natural order, padded by 2 complex per 16 (conflict-free writes): slot(k) = k + 2 (k >> 4)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int g = lane + 64 * t, k1 = g >> 4, r = g & 15;
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        const int k = k1 + 16 * r + 256 * sx;
        wb[k + 2 * (k >> 4)] = z[t][sx];
      }
    }
    LMW_WAVE_SYNC();
    // (4) real-input unpack and magnitudes: X[k] = (Z[k] + conj Z[N-k]) / 2 - i/2 e^(-2 pi i k / 2048) (Z[k] - conj Z[N-k])
    float mg[16], mg_ny = 0.f;
    c2 zks[16], zns[16], tws[16];                                    // (48 reads in flight, one wait)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = lane + 64 * i, kn = (CFFT_N - k) & (CFFT_N - 1);
      zks[i] = wb[k + 2 * (k >> 4)];
      zns[i] = wb[kn + 2 * (kn >> 4)];
      tws[i] = twl[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const c2 zk = zks[i], zn = zns[i];
      const c2 a = {0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
      const c2 b = {0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y)};
      const c2 wbv = cmul(tws[i], b);
      const c2 X = {a.x + wbv.y, a.y - wbv.x};
      mg[i] = __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y);           // (v_sqrt_f32, 1 ulp: far inside the 1e-4 of the log-mel)
      if (i == 0) {                                                  // lane 0 holds Z[0]: bin 1024 = Re Z[0] - Im Z[0]
        const c2 wn = cmul(c2{-1.f, 0.f}, b);
        const c2 Xn = {a.x + wn.y, a.y - wn.x};
        mg_ny = __builtin_amdgcn_sqrtf(Xn.x * Xn.x + Xn.y * Xn.y);
      }
    }
    LMW_WAVE_SYNC();
    float* mag = (float*)wb;                                          // 1025 floats over the wave's buffer (+ zeros behind)
#pragma unroll
    for (int i = 0; i < 16; ++i) mag[lane + 64 * i] = mg[i];
    mag[CFFT_N + lane] = lane == 0 ? mg_ny : 0.f;                     // bin 1024, then zeros: a filter's padded taps read them
    LMW_WAVE_SYNC();
    // (5) sparse mel, safe log, clip / scale: mel bins 8 lane .. 8 lane + 7; tap q of filter e at magp[e][q] (constant offsets)
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < LMW_MAX_TAPS; ++q) {
      if (q < max_taps) {
        const f32x4 w0 = *(const f32x4*)(fbt + q * 512 + 8 * lane), w1 = *(const f32x4*)(fbt + q * 512 + 8 * lane + 4);
        const float wq[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += mag[fs[e] + q] * wq[e];
      }
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = __builtin_amdgcn_logf(acc[e] <= 0.f ? 1e-5f : acc[e]) * 0.6931471805599453f;   // v_log_f32 (log2) x ln 2
      if (normalize) {
        v[e] = fminf(fmaxf(v[e], lo), hi);
        v[e] = (v[e] - lo) / (hi - lo);
      }
    }
    if (OUT_BF16) {
      *(u32x4*)((bf16_t*)out + out_off) = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    } else {
      *(f32x4*)((float*)out + out_off) = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)((float*)out + out_off + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

static int logmel_launch(const float* audio, int batch, int n_samples, int hop, const float* window,
                         const float* twiddle, const int* fb_start, const int* fb_cnt, const float* fb_w, int n_mels,
                         int max_taps, const int* valid_frames, const long long* seg_start, long long total_samples,
                         int normalize, int out_bf16, void* out, void* stream) {
  const int n_frames = ceil_div(n_samples, hop);  // pad_end: ceil(n/hop) full windows
  const float lo = -12.f, hi = 5.f;  // MIN_LOG_MEL / MAX_LOG_MEL
  hipStream_t s = (hipStream_t)stream;
  {
    // the wave-per-frame kernel: the model's 512 mel bins, filters of at most LMW_MAX_TAPS taps, a hop that leaves room
    // for at least one frame's samples in the strip (MRMT3_LOGMEL=0: the round-1 kernel, A/B and parity)
    // ... and tables it may read as vectors: fb_start in 16-byte pieces, the window as float2, 16-byte output rows
    const bool fast = MR_KNOB("MRMT3_LOGMEL", 1) != 0 && n_mels == 512 && max_taps <= LMW_MAX_TAPS && hop % 2 == 0 && hop >= 2 &&
                      ((uintptr_t)fb_start % 16) == 0 && ((uintptr_t)window % 8) == 0 && ((uintptr_t)out % 16) == 0;
    if (fast) {
      // frames per workgroup: up to 64 (8 per wave), fewer for small batches so that the grid still covers the CUs —
      // a workgroup's waves walk their frames one after the other (12 segments: 16 frames each, 1 segment: 8)
      int fpw = (LMW_STRIP_FLOATS - FFT_N) / hop + 1;
      if (fpw > 64) fpw = 64;
      {
        static int cus = 0;
        if (cus == 0) {
          int dev = 0;
          hipDeviceProp_t prop;
          cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        const long long total = (long long)batch * n_frames;
        while (fpw > LMW_WAVES && total / fpw < cus) fpw >>= 1;
      }
      if (fpw > n_frames) fpw = n_frames;
      const dim3 grid((unsigned)(batch * ceil_div(n_frames, fpw)));
      if (out_bf16)
        hipLaunchKernelGGL(logmel_wave_kernel<true>, grid, dim3(LMW_THREADS), 0, s, audio, n_samples, n_frames, hop, fpw, window,
                           twiddle, fb_start, fb_cnt, fb_w, max_taps, valid_frames, seg_start, total_samples, lo, hi, normalize, out);
      else
        hipLaunchKernelGGL(logmel_wave_kernel<false>, grid, dim3(LMW_THREADS), 0, s, audio, n_samples, n_frames, hop, fpw, window,
                           twiddle, fb_start, fb_cnt, fb_w, max_taps, valid_frames, seg_start, total_samples, lo, hi, normalize, out);
      MR_CHECK_LAUNCH("logmel_fwd");
      return MRMT3_OK;
    }
  }
  dim3 grid((unsigned)(batch * n_frames)), block(LM_THREADS);
  if (out_bf16)
    hipLaunchKernelGGL(logmel_kernel<true>, grid, block, 0, s, audio, n_samples, n_frames, hop, window, twiddle,
                       fb_start, fb_cnt, fb_w, n_mels, max_taps, valid_frames, seg_start, total_samples, lo, hi, normalize,
                       out);
  else
    hipLaunchKernelGGL(logmel_kernel<false>, grid, block, 0, s, audio, n_samples, n_frames, hop, window, twiddle,
                       fb_start, fb_cnt, fb_w, n_mels, max_taps, valid_frames, seg_start, total_samples, lo, hi, normalize,
                       out);
  MR_CHECK_LAUNCH("logmel_fwd");
  return MRMT3_OK;
}

extern "C" int mrmt3_logmel_fwd(const float* audio, int batch, int n_samples, int hop,
                                const float* window, const float* twiddle, const int* fb_start,
                                const int* fb_cnt, const float* fb_w, int n_mels, int max_taps,
                                const int* valid_frames, int normalize, int out_bf16, void* out,
                                void* stream) {
  MR_CHECK_ARG(audio && window && twiddle && fb_start && fb_cnt && fb_w && out, "logmel_fwd: null pointer");
  MR_CHECK_ARG(batch > 0 && n_samples > 0 && hop > 0 && n_mels > 0 && max_taps > 0, "logmel_fwd: bad sizes");
  return logmel_launch(audio, batch, n_samples, hop, window, twiddle, fb_start, fb_cnt, fb_w, n_mels, max_taps,
                       valid_frames, nullptr, 0, normalize, out_bf16, out, stream);
}

extern "C" int mrmt3_logmel_crops_fwd(const float* audio, long long total_samples, const long long* seg_start,
                                      int batch, int n_samples, int hop, const float* window,
                                      const float* twiddle, const int* fb_start, const int* fb_cnt,
                                      const float* fb_w, int n_mels, int max_taps, const int* valid_frames,
                                      int normalize, int out_bf16, void* out, void* stream) {
  MR_CHECK_ARG(audio && seg_start && window && twiddle && fb_start && fb_cnt && fb_w && out,
               "logmel_crops_fwd: null pointer");
  MR_CHECK_ARG(total_samples > 0 && batch > 0 && n_samples > 0 && hop > 0 && n_mels > 0 && max_taps > 0,
               "logmel_crops_fwd: bad sizes");
  return logmel_launch(audio, batch, n_samples, hop, window, twiddle, fb_start, fb_cnt, fb_w, n_mels, max_taps,
                       valid_frames, seg_start, total_samples, normalize, out_bf16, out, stream);
}
