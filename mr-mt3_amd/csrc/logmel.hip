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

static int logmel_launch(const float* audio, int batch, int n_samples, int hop, const float* window,
                         const float* twiddle, const int* fb_start, const int* fb_cnt, const float* fb_w, int n_mels,
                         int max_taps, const int* valid_frames, const long long* seg_start, long long total_samples,
                         int normalize, int out_bf16, void* out, void* stream) {
  const int n_frames = ceil_div(n_samples, hop);  // pad_end: ceil(n/hop) full windows
  dim3 grid((unsigned)(batch * n_frames)), block(LM_THREADS);
  const float lo = -12.f, hi = 5.f;  // MIN_LOG_MEL / MAX_LOG_MEL
  hipStream_t s = (hipStream_t)stream;
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
