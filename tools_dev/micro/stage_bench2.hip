// Microbenchmark 4: LDS-DMA staging against a v_mfma_f32_32x32x16_bf16 loop in the same waves.
//   MODE 0: the stage's VEC glds instructions per wave go out as one burst before the MFMA loop
//   MODE 2: one glds every NM/VEC MFMAs (rate-matched interleave)
// WGS = workgroups per CU (1 or 2, each 256 threads).  Source is L2-resident (2 MB) or an HBM stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int VEC, int NM>
__global__ __launch_bounds__(256) void k(const char* src, size_t src_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NA = 6;
  f32x16 acc[NA];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a = *(const bf16x8*)(src + tid * 16), b = *(const bf16x8*)(src + 4096 + tid * 16);
  constexpr int V = VEC > 0 ? VEC : 1;
  size_t base = src_bytes < (1u << 20) ? 0 : ((size_t)blockIdx.x * 977 * 4096) % (src_bytes - (size_t)V * 4096 * 2);
  for (int it = 0; it < iters; ++it) {
    const char* s = src + (base + (size_t)it * V * 4096) % (src_bytes - (size_t)V * 4096 * 2);
    char* dst = smem + (it & 1) * V * 4096;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * 4 + wave) * 1024 + lane * 16), (lds_void_t*)(dst + (j * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll 1
      for (int m = 0; m < NM / NA; ++m)
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    } else {
      constexpr int PER = VEC > 0 ? NM / V : NM;   // MFMAs between two glds
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (VEC > 0)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * 4 + wave) * 1024 + lane * 16), (lds_void_t*)(dst + (j * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i % NA] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NA], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (VEC > 0) a = *(const bf16x8*)(dst + tid * 16);
  }
  float t = 0;
  for (int i = 0; i < NA; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
}

template <int MODE, int VEC, int NM>
void run(const char* name, const char* d, size_t bytes, float* sink, int wgs) {
  const int iters = 400, grid = 256 * wgs;
  hipFuncSetAttribute((const void*)k<MODE, VEC, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t lds = (VEC > 0 ? VEC : 1) * 4096 * 2;
  k<MODE, VEC, NM><<<grid, 256, lds, 0>>>(d, bytes, 10, sink);
  hipEventRecord(a);
  k<MODE, VEC, NM><<<grid, 256, lds, 0>>>(d, bytes, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double us = ms * 1e3 / iters;
  printf("%-10s %d WG/CU src %7.1f MB  %2d KiB/stage/WG  %4d MFMA32/wave: %6.2f us/stage  (MFMA alone %.2f us @1.92GHz, x%d WGs)  %5.1f B/clk/CU\n", name, wgs, bytes / 1e6,
         VEC * 4, NM, us, NM * 32 / 1920.0, wgs, wgs * VEC * 4096.0 / (us * 1e-6) / 1.92e9);
}

int main() {
  size_t big = 1ull << 30;
  char* d; hipMalloc(&d, big); hipMemset(d, 0x3c, big);
  float* sink; hipMalloc(&sink, 4);
  // all workgroups stream the SAME small buffer (the weight-chunk pattern of the convolution)
  run<0, 7, 0>("same-src", d, 172032 + 2 * 7 * 4096, sink, 1);
  run<0, 7, 81>("same-src", d, 172032 + 2 * 7 * 4096, sink, 1);
  run<0, 11, 0>("same-src", d, 172032 + 2 * 11 * 4096, sink, 1);
  run<0, 11, 81>("same-src", d, 172032 + 2 * 11 * 4096, sink, 1);
  for (int pass = 0; pass < 2; ++pass) {
    size_t bytes = pass == 0 ? (size_t)2 << 20 : big;   // L2-resident (2 MB) vs HBM stream
    run<0, 0, 108>("mfma-only", d, bytes, sink, 1);
    run<0, 12, 0>("dma-only", d, bytes, sink, 1);
    run<0, 12, 108>("burst", d, bytes, sink, 1);
    run<2, 12, 108>("interleave", d, bytes, sink, 1);
    run<2, 6, 108>("interleave", d, bytes, sink, 1);
    run<0, 6, 108>("burst", d, bytes, sink, 1);
    run<0, 6, 54>("burst", d, bytes, sink, 2);
    run<2, 6, 54>("interleave", d, bytes, sink, 2);
    run<0, 0, 54>("mfma-only", d, bytes, sink, 2);
  }
  return 0;
}
