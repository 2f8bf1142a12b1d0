// A canary for LDS corruption by a co-resident workgroup of another process (profiles/r03_two_process_soak.txt): every
// workgroup fills its LDS with a pattern, waits, and checks it; mismatches are recorded with their offset and value.
//   hipcc --offload-arch=gfx950 -O2 -o lds_canary lds_canary.hip && ./lds_canary <seconds> <lds KiB per workgroup>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <chrono>
#include <map>

struct Rec { unsigned off, got, want, wg; };

__global__ __launch_bounds__(256) void canary(unsigned words, unsigned iter, unsigned* n_bad, Rec* recs, int max_recs, int spin) {
  extern __shared__ unsigned lds[];
  const unsigned key = 0xA5000000u ^ (iter << 12);
  for (unsigned i = threadIdx.x; i < words; i += 256) lds[i] = key ^ i;
  __syncthreads();
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < words; i += 256) {
    const unsigned v = lds[i];
    if (v != (key ^ i)) {
      const unsigned k = atomicAdd(n_bad, 1u);
      if ((int)k < max_recs) recs[k] = Rec{i * 4, v, key ^ i, blockIdx.x};
    }
  }
}

// mode 1: data handed from wave to wave through LDS behind __syncthreads(), 64 rounds per launch (what an LDS FFT does):
// thread t writes word t of the round's row, reads the word of thread (t + 64 r + 1) % 256 after the barrier
__global__ __launch_bounds__(256) void exchange(unsigned iter, unsigned* n_bad, Rec* recs, int max_recs) {
  __shared__ unsigned buf[2][256];
  const unsigned t = threadIdx.x;
  unsigned bad = 0, got0 = 0, want0 = 0, r0 = 0;
  for (unsigned r = 0; r < 64; ++r) {
    const unsigned key = (iter << 16) ^ (blockIdx.x << 6) ^ r;
    buf[r & 1][t] = key * 2654435761u + t;
    __syncthreads();
    const unsigned src = (t + 64 * (r & 3) + 1) & 255;
    const unsigned v = buf[r & 1][src], want = key * 2654435761u + src;
    if (v != want && !bad) { bad = 1; got0 = v; want0 = want; r0 = r; }
    // (the other buffer is written next round: everyone is past this round's barrier, so its reads of two rounds ago are done)
  }
  if (bad) {
    const unsigned k = atomicAdd(n_bad, 1u);
    if ((int)k < max_recs) recs[k] = Rec{r0, got0, want0, blockIdx.x};
  }
}

// mode 4: the FFT's DATA MOVEMENT without its arithmetic: per pass every thread reads four 8-byte words at tid + m*256 and
// writes four at the radix-4 Stockham output positions (strided: bank conflicts), one barrier per pass, 5 passes; the
// payload of a word is (origin index, pass) so every read can be checked exactly.
__global__ __launch_bounds__(256) void move_canary(unsigned iter, unsigned* n_bad, Rec* recs, int max_recs) {
  __shared__ __attribute__((aligned(16))) uint2 buf0[1024];
  __shared__ __attribute__((aligned(16))) uint2 buf1[1024];
  const int tid = threadIdx.x;
  const unsigned key = (iter << 12) ^ (blockIdx.x * 977u);
  for (int r = 0; r < 4; ++r) { const int c = tid + r * 256; buf0[c] = uint2{(unsigned)c ^ key, 0u}; }
  __syncthreads();
  uint2* in = buf0; uint2* outb = buf1;
  unsigned bad = 0, got0 = 0, want0 = 0, where = 0;
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int p = 1 << (2 * pass);
    const int k = tid & (p - 1);
    const int j = ((tid - k) << 2) + k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint2 v = in[tid + m * 256];
      // what sits at position q of the input of pass `pass`: written in pass-1 as outb[j' + m' p'] = (position, pass)
      const unsigned want = (unsigned)(tid + m * 256) ^ key;
      if ((v.x != want || v.y != (unsigned)pass) && !bad) { bad = 1; got0 = v.x; want0 = want; where = (pass << 16) | (tid + m * 256); }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) outb[j + m * p] = uint2{(unsigned)(j + m * p) ^ key, (unsigned)(pass + 1)};
    __syncthreads();
    uint2* tmp = in; in = outb; outb = tmp;
  }
  if (bad) {
    const unsigned kk = atomicAdd(n_bad, 1u);
    if ((int)kk < max_recs) recs[kk] = Rec{where, got0, want0, blockIdx.x};
  }
}

// mode 5: the FFT's TWIDDLE reads alone: a 1024 x 8-byte table in LDS (index-derived values), read with the FFT's
// per-pass index patterns (pass 0: every lane the same word; later passes 4, 16, 64, 256 distinct words per wave)
__global__ __launch_bounds__(256) void twiddle_canary(unsigned iter, unsigned* n_bad, Rec* recs, int max_recs) {
  __shared__ __attribute__((aligned(16))) uint2 tw[1024];
  const int tid = threadIdx.x;
  const unsigned key = iter * 40503u;
  for (int i = tid; i < 1024; i += 256) tw[i] = uint2{(unsigned)i ^ key, ~((unsigned)i ^ key)};
  __syncthreads();
  unsigned bad = 0, got0 = 0, want0 = 0, where = 0;
  for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
      const int p = 1 << (2 * pass);
      const int k = tid & (p - 1);
      const int twm = (512 >> (2 * pass)) * k;
#pragma unroll
      for (int m = 1; m < 4; ++m) {
        const int idx = (m * twm) & 1023;
        const uint2 v = tw[idx];
        if ((v.x != ((unsigned)idx ^ key) || v.y != ~((unsigned)idx ^ key)) && !bad) { bad = 1; got0 = v.x; want0 = (unsigned)idx ^ key; where = (pass << 16) | idx; }
      }
    }
    __syncthreads();
  }
  if (bad) {
    const unsigned kk = atomicAdd(n_bad, 1u);
    if ((int)kk < max_recs) recs[kk] = Rec{where, got0, want0, blockIdx.x};
  }
}

// mode 6 / 7: arithmetic only, no LDS, no loads: 512 dependent multiply-adds per thread on values derived from the thread
// index, as PACKED f32 instructions (mode 6: v_pk_fma_f32, what hipcc makes of the FFT's complex arithmetic) or as
// scalar v_fma_f32 (mode 7); every workgroup computes the same 256 results, compared with a reference launch.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PACKED>
__global__ __launch_bounds__(256) void alu_canary(float* __restrict__ ref, int make_ref, unsigned* n_bad, Rec* recs, int max_recs) {
  const int tid = threadIdx.x;
  f2 x = {1.0f + tid * 0.001f, 0.5f - tid * 0.002f};
  const f2 a = {0.99991f, -0.99987f}, b = {0.0003f * (tid & 7), 0.0001f * (tid & 15)};
  for (int i = 0; i < 512; ++i) {
    if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(a.x), "v"(b.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.y) : "v"(a.y), "v"(b.y)); }
  }
  if (make_ref) { if (blockIdx.x == 0) { ref[2 * tid] = x.x; ref[2 * tid + 1] = x.y; } }
  else if (__float_as_uint(x.x) != __float_as_uint(ref[2 * tid]) || __float_as_uint(x.y) != __float_as_uint(ref[2 * tid + 1])) {
    const unsigned k = atomicAdd(n_bad, 1u);
    if ((int)k < max_recs) recs[k] = Rec{(unsigned)tid, __float_as_uint(x.x), __float_as_uint(ref[2 * tid]), blockIdx.x};
  }
}

// mode 3: nothing but 8-byte global loads of a small read-only table (what the FFT canary does first): every word is
// checked in registers (rec.wg = 0x1000 | wg), then after a trip through LDS with 8-byte ds ops (rec.wg = 0x2000 | wg)
__global__ __launch_bounds__(256) void load_canary(const unsigned* __restrict__ tab, unsigned* n_bad, Rec* recs, int max_recs) {
  __shared__ __attribute__((aligned(16))) uint2 l[1024];
  const int tid = threadIdx.x;
  for (int r = 0; r < 4; ++r) {
    const int c = tid + r * 256;
    const uint2 v = ((const uint2*)tab)[c];
    l[c] = v;
    if (v.x != (unsigned)(2 * c) * 2654435761u || v.y != (unsigned)(2 * c + 1) * 2654435761u) {
      const unsigned k = atomicAdd(n_bad, 1u);
      if ((int)k < max_recs) recs[k] = Rec{(unsigned)c, v.x, (unsigned)(2 * c) * 2654435761u, 0x1000u | blockIdx.x};
    }
  }
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int c = (tid * 4 + r + 517) & 1023;
    const uint2 v = l[c];
    if (v.x != (unsigned)(2 * c) * 2654435761u || v.y != (unsigned)(2 * c + 1) * 2654435761u) {
      const unsigned k = atomicAdd(n_bad, 1u);
      if ((int)k < max_recs) recs[k] = Rec{(unsigned)c, v.x, (unsigned)(2 * c) * 2654435761u, 0x2000u | blockIdx.x};
    }
  }
}

// mode 2: the log-mel kernel's LDS FFT (csrc/logmel.hip: 1024-point complex radix-4 Stockham, twiddles staged from a global
// table, 8-byte LDS accesses, one barrier per pass) on the same input in every workgroup; the output is compared word
// for word with a reference copy made by the first launch.  stage: 0 = FFT only, 1 = + unpack to magnitudes (sqrtf)
struct c2 { float x, y; };
__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c2 twiddle(const c2* tw, int idx) {
  c2 t = tw[idx & 1023];
  if (idx & 1024) { t.x = -t.x; t.y = -t.y; }
  return t;
}
template <int VAR>
__global__ __launch_bounds__(256) void fft_canary(const float* __restrict__ in_g, const float* __restrict__ twid, float* __restrict__ ref,
                                                  int make_ref, unsigned* n_bad, Rec* recs, int max_recs) {
  __shared__ __attribute__((aligned(16))) c2 buf0[1024];
  __shared__ __attribute__((aligned(16))) c2 buf1[1024];
  __shared__ __attribute__((aligned(16))) c2 tw[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) tw[i] = ((const c2*)twid)[i];
  for (int r = 0; r < 4; ++r) { const int c = tid + r * 256; buf0[c] = {in_g[2 * c], in_g[2 * c + 1]}; }
  __syncthreads();
  c2* in = buf0; c2* outb = buf1;
  const int t = 256;
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int p = 1 << (2 * pass);
    const int k = tid & (p - 1);
    const int j = ((tid - k) << 2) + k;
    const int twm = (512 >> (2 * pass)) * k;
    // VAR 1: no twiddle multiplies (adds / subtracts only); VAR 2: twiddles applied but read ONCE before the loop from
    // the global table (no LDS table); VAR 3: the first two passes only; VAR 4: multiplies by a constant instead
    if (VAR == 3 && pass >= 2) break;
    c2 u0 = in[tid];
    c2 u1 = in[tid + t], u2 = in[tid + 2 * t], u3 = in[tid + 3 * t];
    if (VAR == 0 || VAR == 3) { u1 = cmul(u1, twiddle(tw, twm)); u2 = cmul(u2, twiddle(tw, 2 * twm)); u3 = cmul(u3, twiddle(tw, 3 * twm)); }
    if (VAR == 2) { u1 = cmul(u1, ((const c2*)twid)[twm & 1023]); u2 = cmul(u2, ((const c2*)twid)[(2 * twm) & 1023]); u3 = cmul(u3, ((const c2*)twid)[(3 * twm) & 1023]); }
    if (VAR == 4) { const c2 k1 = {0.7071f, -0.7071f}; u1 = cmul(u1, k1); u2 = cmul(u2, k1); u3 = cmul(u3, k1); }
    c2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3);
    c2 d = csub(u1, u3);
    c2 v3 = {d.y, -d.x};
    outb[j] = cadd(v0, v2); outb[j + p] = cadd(v1, v3); outb[j + 2 * p] = csub(v0, v2); outb[j + 3 * p] = csub(v1, v3);
    __syncthreads();
    c2* tmp = in; in = outb; outb = tmp;
  }
  for (int r = 0; r < 4; ++r) {
    const int c = tid + r * 256;
    const c2 z = in[c];
    const float m = sqrtf(z.x * z.x + z.y * z.y);
    const unsigned v0 = __float_as_uint(z.x), v1 = __float_as_uint(m);
    if (make_ref) { if (blockIdx.x == 0) { ref[2 * c] = z.x; ref[2 * c + 1] = m; } }
    else if (v0 != __float_as_uint(ref[2 * c]) || v1 != __float_as_uint(ref[2 * c + 1])) {
      const unsigned k = atomicAdd(n_bad, 1u);
      if ((int)k < max_recs) recs[k] = Rec{(unsigned)c, v0, __float_as_uint(ref[2 * c]), blockIdx.x};
    }
  }
}

#define FFT_LAUNCH(G, MK) do { switch (var) { case 1: hipLaunchKernelGGL(fft_canary<1>, dim3(G), dim3(256), 0, 0, in_g, tw_g, ref_g, MK, n_bad, recs, max_recs); break; \
      case 3: hipLaunchKernelGGL(fft_canary<3>, dim3(G), dim3(256), 0, 0, in_g, tw_g, ref_g, MK, n_bad, recs, max_recs); break; \
      case 4: hipLaunchKernelGGL(fft_canary<4>, dim3(G), dim3(256), 0, 0, in_g, tw_g, ref_g, MK, n_bad, recs, max_recs); break; \
      default: hipLaunchKernelGGL(fft_canary<0>, dim3(G), dim3(256), 0, 0, in_g, tw_g, ref_g, MK, n_bad, recs, max_recs); } } while (0)

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const int kib = argc > 2 ? atoi(argv[2]) : 24;
  const int spin = argc > 3 ? atoi(argv[3]) : 40;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const int var = argc > 5 ? atoi(argv[5]) : 0;
  const unsigned words = kib * 256;
  unsigned* n_bad; Rec* recs; const int max_recs = 4096;
  hipMalloc(&n_bad, 4); hipMalloc(&recs, sizeof(Rec) * max_recs);
  hipMemset(n_bad, 0, 4);
  hipFuncSetAttribute((const void*)canary, hipFuncAttributeMaxDynamicSharedMemorySize, kib * 1024);
  float *in_g = nullptr, *tw_g = nullptr, *ref_g = nullptr;
  if (mode == 2) {
    float* h = (float*)malloc(2048 * 4); float* ht = (float*)malloc(2048 * 4);
    unsigned st = 12345;
    for (int i = 0; i < 2048; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (int i = 0; i < 1024; ++i) { ht[2 * i] = (float)cos(-2 * M_PI * i / 2048); ht[2 * i + 1] = (float)sin(-2 * M_PI * i / 2048); }
    hipMalloc(&in_g, 8192); hipMalloc(&tw_g, 8192); hipMalloc(&ref_g, 8192);
    hipMemcpy(in_g, h, 8192, hipMemcpyHostToDevice); hipMemcpy(tw_g, ht, 8192, hipMemcpyHostToDevice);
    FFT_LAUNCH(1, 1);
    hipDeviceSynchronize();
    // the reference itself may have been taken beside the aggressor: take it three times and insist they agree
    float r1[2048], r2[2048];
    hipMemcpy(r1, ref_g, 8192, hipMemcpyDeviceToHost);
    for (int k = 0; k < 2; ++k) {
      FFT_LAUNCH(1, 1);
      hipDeviceSynchronize();
      hipMemcpy(r2, ref_g, 8192, hipMemcpyDeviceToHost);
      if (memcmp(r1, r2, 8192)) printf("fft canary: two reference launches already differ\n");
    }
  }
  float* aref = nullptr;
  if (mode == 6 || mode == 7) {
    hipMalloc(&aref, 2048);
    float r1[512], r2[512];
    for (int k = 0; k < 3; ++k) {
      if (mode == 6) hipLaunchKernelGGL(alu_canary<1>, dim3(1), dim3(256), 0, 0, aref, 1, n_bad, recs, max_recs);
      else hipLaunchKernelGGL(alu_canary<0>, dim3(1), dim3(256), 0, 0, aref, 1, n_bad, recs, max_recs);
      hipDeviceSynchronize();
      hipMemcpy(k ? r2 : r1, aref, 2048, hipMemcpyDeviceToHost);
      if (k && memcmp(r1, r2, 2048)) printf("alu canary: two reference launches already differ\n");
    }
  }
  unsigned* tab_g = nullptr;
  if (mode == 3) {
    unsigned* h = (unsigned*)malloc(8192);
    for (unsigned i = 0; i < 2048; ++i) h[i] = i * 2654435761u;
    hipMalloc(&tab_g, 8192);
    hipMemcpy(tab_g, h, 8192, hipMemcpyHostToDevice);
  }
  auto t0 = std::chrono::steady_clock::now();
  unsigned iter = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 50; ++k) {
      if (mode == 6) { hipLaunchKernelGGL(alu_canary<1>, dim3(2048), dim3(256), 0, 0, aref, 0, n_bad, recs, max_recs); iter++; }
      else if (mode == 7) { hipLaunchKernelGGL(alu_canary<0>, dim3(2048), dim3(256), 0, 0, aref, 0, n_bad, recs, max_recs); iter++; }
      else if (mode == 5) { hipLaunchKernelGGL(twiddle_canary, dim3(512), dim3(256), 0, 0, iter, n_bad, recs, max_recs); iter++; }
      else if (mode == 4) { hipLaunchKernelGGL(move_canary, dim3(512), dim3(256), 0, 0, iter, n_bad, recs, max_recs); iter++; }
      else if (mode == 3) { hipLaunchKernelGGL(load_canary, dim3(512), dim3(256), 0, 0, tab_g, n_bad, recs, max_recs); iter++; }
      else if (mode == 2) { FFT_LAUNCH(512, 0); iter++; }
      else if (mode == 1) hipLaunchKernelGGL(exchange, dim3(512), dim3(256), 0, 0, iter++, n_bad, recs, max_recs);
      else hipLaunchKernelGGL(canary, dim3(512), dim3(256), kib * 1024, 0, words, iter++, n_bad, recs, max_recs, spin);
    }
    hipDeviceSynchronize();
  }
  unsigned nb; hipMemcpy(&nb, n_bad, 4, hipMemcpyDeviceToHost);
  if (mode == 6 || mode == 7) printf("alu canary (%s): %u launches of 2048 workgroups x 256 threads x 512 multiply-adds, %u threads with a wrong result\n", mode == 6 ? "v_pk_fma_f32" : "v_fma_f32", iter, nb);
  else if (mode == 5) printf("twiddle canary: %u launches of 512 workgroups, %u threads read a wrong table word\n", iter, nb);
  else if (mode == 4) printf("move canary: %u launches of 512 workgroups x 5 passes, %u threads read a wrong word\n", iter, nb);
  else if (mode == 3) printf("load canary: %u launches of 512 workgroups x 2048 words, %u wrong words (wg 0x1000|n: straight from the global load; 0x2000|n: after LDS)\n", iter, nb);
  else if (mode == 2) printf("fft canary (variant %d): %u launches of 512 workgroups, %u output words differ from the reference\n", var, iter, nb);
  else if (mode == 1) printf("exchange: %u launches of 512 workgroups x 64 barrier rounds, %u threads read a stale or foreign word\n", iter, nb);
  else printf("canary: %u launches of 512 workgroups x %d KiB LDS, %u corrupted words\n", iter, kib, nb);
  if (nb) {
    const int n = nb < (unsigned)max_recs ? nb : max_recs;
    Rec* h = (Rec*)malloc(sizeof(Rec) * n);
    hipMemcpy(h, recs, sizeof(Rec) * n, hipMemcpyDeviceToHost);
    std::map<unsigned, int> by_chunk, zeros;
    int nz = 0;
    for (int i = 0; i < n; ++i) { by_chunk[h[i].off / 1024]++; nz += h[i].got == 0; }
    printf("of the first %d records: %d read back ZERO; corrupted 1-KiB chunks of the workgroup's LDS (chunk: words):", n, nz);
    for (auto& kv : by_chunk) printf(" %u:%d", kv.first, kv.second);
    printf("\nexamples:");
    for (int i = 0; i < n && i < 12; ++i) printf(" [wg %u off %u got %08x want %08x]", h[i].wg, h[i].off, h[i].got, h[i].want);
    printf("\n");
  }
  return 0;
}
