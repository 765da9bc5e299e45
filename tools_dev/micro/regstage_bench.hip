// Microbenchmark 7: how should the dedicated producer waves of the producer/consumer convolution move a stage's
// 28 KiB into LDS -- LDS-DMA (global_load_lds_dwordx4) or register staging (global_load_dwordx4 -> VGPRs ->
// ds_write_b128)?  512 threads: waves 4-7 produce, waves 0-3 run the convolution's tap loop (6 fragment reads per
// 9 32x32x16 MFMAs).  Source either a 1 MiB L2-resident buffer or a 2 GiB buffer walked with a large stride (HBM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// STAGE 0: LDS-DMA.  1: register staged, all loads issued, then all writes.  2: no producer traffic.
template <int STAGE, int VEC, int NM>
__global__ __launch_bounds__(512, 2) void k(const char* src, size_t src_bytes, int chunk_bytes, int iters, float* sink,
                                            unsigned long long* cyc) {
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave_all = tid >> 6, lane = tid & 63, wave = wave_all & 3;
  const bool producer = wave_all >= 4;
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a = *(const bf16x8*)(src + tid * 16), b = *(const bf16x8*)(src + 8192 + tid * 16);
  size_t pos = ((size_t)blockIdx.x * 7919 * 4096) % (src_bytes - chunk_bytes);
  for (int it = 0; it < iters; ++it) {
    if (producer) {
      const char* s = src + pos;
      pos = (pos + (size_t)chunk_bytes * 257) % (src_bytes - chunk_bytes);
      char* dst = smem + (it & 1) * 32768;
      if (STAGE == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * 4 + wave) * 1024 + lane * 16), (lds_void_t*)(dst + (j * 4 + wave) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (STAGE == 1) {
        u32x4 r[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[j]) : "v"(s + (j * 4 + wave) * 1024 + lane * 16) : "memory");
        const uint32_t la = (uint32_t)(size_t)dst + wave * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (j == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VEC - 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VEC - 1 - (j < VEC ? j : 0)) : "memory");
          asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(la), "v"(r[j]), "n"(j * 4096) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
      const uint32_t la = (uint32_t)(size_t)smem + 65536 + (tid & 255) * 16;
#pragma unroll 1
      for (int m = 0; m < NM / 9; ++m) {
        bf16x8 f[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(la), "n"(0));
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % 6], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(f[i]));
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  float t = 0;
  for (int i = 0; i < 6; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = __builtin_amdgcn_s_memtime() - t_begin;
}

template <int STAGE, int VEC, int NM>
void run(const char* name, const char* d, size_t src_bytes, float* sink) {
  const int iters = 600, chunk_bytes = VEC * 4096;
  hipFuncSetAttribute((const void*)k<STAGE, VEC, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  static unsigned long long* cyc = nullptr; if (!cyc) hipMalloc(&cyc, 8);
  k<STAGE, VEC, NM><<<256, 512, 128 * 1024>>>(d, src_bytes, chunk_bytes, 10, sink, cyc);
  hipEventRecord(a);
  k<STAGE, VEC, NM><<<256, 512, 128 * 1024>>>(d, src_bytes, chunk_bytes, iters, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / iters;
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.2f us/stage  %6.0f ticks/stage  %5.2f TB/s chip\n", name, us, (double)c / iters,
         STAGE == 2 ? 0.0 : chunk_bytes * 256.0 / (us * 1e-6) / 1e12);
}

int main() {
  const size_t big = (size_t)2 << 30;
  char* d; if (hipMalloc(&d, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0x3c, big);
  float* sink; hipMalloc(&sink, 4);
  for (int pass = 0; pass < 2; ++pass) {
    const size_t n = pass == 0 ? ((size_t)1 << 20) : big;
    printf("--- source: %s\n", pass == 0 ? "1 MiB (L2 resident)" : "2 GiB strided (HBM)");
    run<2, 7, 54>("no producer traffic, 54 MFMA + reads", d, n, sink);
    run<0, 7, 54>("LDS-DMA 28 KiB + 54 MFMA + reads", d, n, sink);
    run<1, 7, 54>("register staged 28 KiB + 54 MFMA + reads", d, n, sink);
    run<2, 7, 81>("no producer traffic, 81 MFMA + reads", d, n, sink);
    run<0, 7, 81>("LDS-DMA 28 KiB + 81 MFMA + reads", d, n, sink);
    run<1, 7, 81>("register staged 28 KiB + 81 MFMA + reads", d, n, sink);
    run<0, 11, 81>("LDS-DMA 44 KiB + 81 MFMA + reads", d, n, sink);
    run<1, 11, 81>("register staged 44 KiB + 81 MFMA + reads", d, n, sink);
    run<0, 7, 0>("LDS-DMA 28 KiB alone", d, n, sink);
    run<1, 7, 0>("register staged 28 KiB alone", d, n, sink);
  }
  return 0;
}
