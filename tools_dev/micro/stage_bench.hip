// Microbenchmark 2: register-staged global->LDS copy with the loads in flight under an MFMA loop
// (T14 split: issue early, ds_write late) vs LDS-DMA issued before the same MFMA loop.
// One 256-thread workgroup per CU; per stage each thread moves VEC x 16 B; the MFMA loop is NM MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE, int VEC, int NM>
__global__ __launch_bounds__(256) void k(const char* src, size_t src_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a = *(const bf16x8*)(src + tid * 16), b = *(const bf16x8*)(src + 4096 + tid * 16);
  size_t base = ((size_t)blockIdx.x * 977 * 4096) % (src_bytes - (size_t)VEC * 4096 * 2);
  for (int it = 0; it < iters; ++it) {
    const char* s = src + (base + (size_t)it * VEC * 4096) % (src_bytes - (size_t)VEC * 4096 * 2);
    uint4 st[VEC];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * 4 + wave) * 1024 + lane * 16), (lds_void_t*)(smem + (j * 4 + wave) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) st[j] = *(const uint4*)(s + (j * 4 + wave) * 1024 + lane * 16);
    }
    // "compute": NM dependent-free MFMAs
#pragma unroll 1
    for (int m = 0; m < NM / 8; ++m)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) *(uint4*)(smem + (j * 4 + wave) * 1024 + lane * 16) = st[j];
    }
    __syncthreads();
    a = *(const bf16x8*)(smem + tid * 16);
    __syncthreads();
  }
  float t = 0;
  for (int i = 0; i < 8; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
}

template <int MODE, int VEC, int NM>
void run(const char* name, const char* d, size_t bytes, float* sink) {
  const int iters = 200, grid = 256;
  hipFuncSetAttribute((const void*)k<MODE, VEC, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, VEC, NM><<<grid, 256, VEC * 4096, 0>>>(d, bytes, 10, sink);
  hipEventRecord(a);
  k<MODE, VEC, NM><<<grid, 256, VEC * 4096, 0>>>(d, bytes, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double us = ms * 1e3 / iters;
  printf("%-6s src %7.1f MB  %2d KiB/stage  %4d MFMA/wave: %6.2f us/stage  (MFMA alone %.2f us @2.1GHz)  %5.1f GB/s per CU\n", name, bytes / 1e6,
         VEC * 4, NM, us, NM * 16 / 2100.0, VEC * 4096.0 / (us * 1e-6) / 1e9);
}

int main() {
  size_t big = 1ull << 30;
  char* d; hipMalloc(&d, big); hipMemset(d, 0, big);
  float* sink; hipMalloc(&sink, 4);
  for (int pass = 0; pass < 2; ++pass) {
    size_t bytes = pass == 0 ? (size_t)2 << 20 : big;   // L2-resident (2 MB) vs HBM stream
    run<0, 12, 0>("dma", d, bytes, sink);
    run<1, 12, 0>("regs", d, bytes, sink);
    run<0, 12, 216>("dma", d, bytes, sink);
    run<1, 12, 216>("regs", d, bytes, sink);
    run<0, 16, 216>("dma", d, bytes, sink);
    run<1, 16, 216>("regs", d, bytes, sink);
    run<1, 24, 216>("regs", d, bytes, sink);
    run<0, 8, 168>("dma", d, bytes, sink);
    run<1, 8, 168>("regs", d, bytes, sink);
  }
  return 0;
}
