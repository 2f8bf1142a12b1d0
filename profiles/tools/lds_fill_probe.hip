// lds_fill_probe.hip — how fast can one CU pull L2-resident data into LDS?
//   path A: global_load_lds_dwordx4 (LDS-direct, what the GEMM / attention kernels use)
//   path B: global_load_dwordx4 into VGPRs + ds_write_b128
// Every workgroup (1024 threads = the 256x256 GEMM's shape, one per CU) re-reads its own 64 KiB slice `iters` times,
// so after the first pass the data comes from the XCD's L2.  Reports GB/s per CU and for the chip.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_fill_probe lds_fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define SLICE 65536

template <int PATH, int DEPTH>
__global__ __launch_bounds__(1024) void fill(const unsigned char* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SLICE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned char* base = src + (size_t)blockIdx.x * SLICE;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    unsigned char* dst = lds[it & 1];
    if (PATH == 0) {
      // 64 wave-instructions of 1 KiB per slice; 16 waves -> 4 each
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + q * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // one older slice may stay in flight
    } else {
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *(const uint4*)(base + (wave * 4 + i) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) *(uint4*)(dst + (wave * 4 + i) * 1024 + lane * 16) = v[i];
    }
    __syncthreads();
    acc += *(const float*)(dst + ((tid * 68) & (SLICE - 4)));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 1024 + tid] = acc;
}

// GEMM-shaped access: a "K step" is 512 rows x 128 B taken at column offset kt*128 from a row-major [512 x ROWB]
// panel (ROWB = K * 2 bytes): the rows of one step differ only in address bits >= log2(ROWB).  ROT: workgroup w starts
// its K loop at step w % nk.
template <int ROWB, bool ROT, int SHARE>
__global__ __launch_bounds__(1024) void fill_panel(const unsigned char* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SLICE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // SHARE workgroups read the same 512-row panel (SHARE = 1: private panels, 128+ MB in total, i.e. Infinity Cache /
  // HBM traffic; SHARE = 256: one panel for the whole chip, L2-resident in every XCD after the first pass)
  const unsigned char* base = src + (size_t)(blockIdx.x / SHARE) * 512 * ROWB;
  constexpr int nk = ROWB / 128;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    unsigned char* dst = lds[it & 1];
    const int kt = ((ROT ? (int)blockIdx.x : 0) + it) % nk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wave * 4 + i;                                         // 8 rows x 128 B per wave-instruction
      const int row = q * 8 + (lane >> 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)row * ROWB + kt * 128 + (lane & 7) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *(const float*)(dst + ((tid * 68) & (SLICE - 4)));
  }
  out[blockIdx.x * 1024 + tid] = acc;
}
template <int ROWB, bool ROT, int SHARE>
static void run_panel(const char* name, const unsigned char* src, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  fill_panel<ROWB, ROT, SHARE><<<256, 1024>>>(src, out, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  fill_panel<ROWB, ROT, SHARE><<<256, 1024>>>(src, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_cu = (double)SLICE * iters / (ms * 1e-3) / 1e9;
  printf("%-58s %7.1f GB/s per CU  %6.2f TB/s chip  (%.2f us per 64 KiB)\n", name, per_cu, per_cu * 256 / 1e3, ms * 1e3 / iters);
}

template <int PATH, int DEPTH>
static void run(const char* name, const unsigned char* src, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  fill<PATH, DEPTH><<<256, 1024>>>(src, out, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  fill<PATH, DEPTH><<<256, 1024>>>(src, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_cu = (double)SLICE * iters / (ms * 1e-3) / 1e9;
  printf("%-58s %7.1f GB/s per CU  %6.2f TB/s chip  (%.2f us per 64 KiB)\n", name, per_cu, per_cu * 256 / 1e3, ms * 1e3 / iters);
}

int main() {
  unsigned char* src; float* out;
  const size_t src_bytes = (size_t)512 * 2304 * 256;       // largest panel set below (K = 1152)
  CHECK(hipMalloc(&src, src_bytes)); CHECK(hipMemset(src, 1, src_bytes));
  CHECK(hipMalloc(&out, 4 * 1024 * 256));
  run<0, 1>("global_load_lds b128, wait for each slice", src, out);
  run<0, 2>("global_load_lds b128, one slice kept in flight", src, out);
  run<1, 1>("global_load b128 -> VGPR -> ds_write_b128", src, out);
  run_panel<1024, false, 1>("K step of [512 x 512 bf16] panels, one per workgroup (128 MB)", src, out);
  run_panel<1024, true, 1>("  same, workgroup w starts at K step w % nk", src, out);
  run_panel<1024, false, 8>("  panels shared by 8 workgroups (16 MB in total)", src, out);
  run_panel<1024, false, 32>("  panels shared by 32 workgroups (4 MB in total)", src, out);
  run_panel<1024, false, 256>("  one panel for all workgroups (L2-resident)", src, out);
  run_panel<1024, true, 256>("  one panel, rotated start", src, out);
  run_panel<1152, false, 256>("  one panel, 1152-B row stride", src, out);
  run_panel<2048, false, 256>("  one [512 x 1024 bf16] panel (2048-B rows)", src, out);
  run_panel<768, false, 256>("  one [512 x 384 bf16] panel (768-B rows)", src, out);
  return 0;
}
