// Microbenchmark 9: what a memory wave costs the MFMA wave next to it, and how to make it cheaper.
//   LAYOUT 0: 512 threads, waves 0-3 consumers (NM MFMAs + 6 ds_read_b128 per 9 MFMAs per stage), waves 4-7
//             producers (ND LDS-DMA instructions of 1 KiB each per wave and stage), one barrier per stage.
//   LAYOUT 1: 256 threads, waves 0-2 consumers, wave 3 (alone on its SIMD) issues all 4*ND DMAs of the stage.
//   LAYOUT 2: 512 threads, consumers ALSO issue NS buffer stores of 1 KiB per stage (consumer-side retire).
//   SLEEP  n: the producer executes s_sleep n (64*n cycles) after every DMA instruction (pacing instead of a burst).
//   PRIO    : s_setprio value of the producers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int N> __device__ __forceinline__ void nap() {
  if constexpr (N > 0) __builtin_amdgcn_s_sleep(N);
}

template <int LAYOUT, int ND, int NM, int SLEEP, int PRIO, int NS, int SPREAD = 0>
__global__ __launch_bounds__(LAYOUT == 1 ? 256 : 512, LAYOUT == 1 ? 1 : 2) void k(const char* src, char* dst_g, int chunk_bytes, int nchunks, int iters,
                                                                              float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool producer = LAYOUT == 1 ? wave_all == 3 : wave_all >= 4;
  const int wave = wave_all & 3;
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a = *(const bf16x8*)(src + tid * 16), b = *(const bf16x8*)(src + 8192 + tid * 16);
  unsigned long long t_loop = 0, t_all = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  if (producer && PRIO) __builtin_amdgcn_s_setprio(PRIO);
  for (int it = 0; it < iters; ++it) {
    if (producer) {
      const char* s = src + (size_t)((it + blockIdx.x) % nchunks) * chunk_bytes;
      char* dst = smem + (it & 1) * 49152;
      constexpr int CNT = LAYOUT == 1 ? 4 * ND : ND;
#pragma unroll
      for (int j = 0; j < CNT; ++j) {
        const int piece = LAYOUT == 1 ? j : j * 4 + wave;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + piece * 1024 + lane * 16), (lds_void_t*)(dst + piece * 1024), 16, 0, 0);
        nap<SLEEP>();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      const uint32_t la = (uint32_t)(size_t)smem + 98304 + (tid & 255) * 16;
#pragma unroll 1
      for (int m = 0; m < NM / 9; ++m) {
        bf16x8 f[6];
        if constexpr (SPREAD == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(la), "n"(0));
#pragma unroll
          for (int i = 0; i < 9; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % 6]) : "v"(a), "v"(b));
        } else if constexpr (SPREAD == 1) {   // one read per MFMA gap
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % 6]) : "v"(a), "v"(b));
            if (i < 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(la), "n"(0));
          }
        } else {   // two reads per gap in three gaps
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % 6]) : "v"(a), "v"(b));
            if (i < 3) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[2 * i]) : "v"(la), "n"(0));
                         asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[2 * i + 1]) : "v"(la), "n"(0)); }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(f[i]));
      }
      t_loop += __builtin_amdgcn_s_memtime() - t0;
      if constexpr (NS > 0) {
        char* o = dst_g + ((size_t)blockIdx.x * 8 + wave_all) * (NS * 1024) + lane * 16;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          u32x4 v; v.x = __builtin_bit_cast(unsigned, acc[j % 6][0]); v.y = __builtin_bit_cast(unsigned, acc[j % 6][1]);
          v.z = __builtin_bit_cast(unsigned, acc[j % 6][2]); v.w = __builtin_bit_cast(unsigned, acc[j % 6][3]);
          *reinterpret_cast<u32x4*>(o + j * 1024) = v;
        }
      }
      t_all += __builtin_amdgcn_s_memtime() - t0;
    }
    __builtin_amdgcn_s_barrier();
  }
  float t = 0;
  for (int i = 0; i < 6; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
  if (blockIdx.x == 0 && lane == 0 && wave_all == 0) { cyc[0] = __builtin_amdgcn_s_memtime() - t_begin; cyc[1] = t_loop; cyc[2] = t_all; }
}

template <int LAYOUT, int ND, int NM, int SLEEP, int PRIO, int NS, int SPREAD = 0>
void run(const char* name, const char* d, char* dst_g, float* sink) {
  const int iters = 600;
  const int chunk_bytes = 4 * ND * 1024, nchunks = (1 << 20) / chunk_bytes;
  hipFuncSetAttribute((const void*)k<LAYOUT, ND, NM, SLEEP, PRIO, NS, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int threads = LAYOUT == 1 ? 256 : 512;
  static unsigned long long* cyc = nullptr; if (!cyc) hipMalloc(&cyc, 32);
  k<LAYOUT, ND, NM, SLEEP, PRIO, NS, SPREAD><<<256, threads, 150 * 1024>>>(d, dst_g, chunk_bytes, nchunks, 200, sink, cyc);
  hipEventRecord(a);
  k<LAYOUT, ND, NM, SLEEP, PRIO, NS, SPREAD><<<256, threads, 150 * 1024>>>(d, dst_g, chunk_bytes, nchunks, iters, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double us = ms * 1e3 / iters;
  unsigned long long c[3]; hipMemcpy(c, cyc, 24, hipMemcpyDeviceToHost);
  const int nwc = LAYOUT == 1 ? 3 : 4;
  const double tf = 256.0 * nwc * NM * 32.0 * 32 * 16 * 2 / (us * 1e-6) / 1e12;
  printf("%-58s %6.2f us/stage %6.0f ticks/stage  loop %5.1f cyc/MFMA  stage %5.1f cyc/MFMA  clk %.2f GHz  %6.0f TFLOP/s\n", name, us,
         (double)c[0] / iters, (double)c[1] / iters / NM, (double)c[0] / iters / NM, (double)c[0] / iters / (us * 1e3), tf);
}

int main() {
  char* d; hipMalloc(&d, 1 << 20);
  char* dst_g; hipMalloc(&dst_g, 64 << 20);
  unsigned short* h = (unsigned short*)malloc(1 << 20);
  for (int i = 0; i < (1 << 19); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  float* sink; hipMalloc(&sink, 4);
  run<0, 0, 81, 0, 0, 0>("4+4, no DMA, burst reads", d, dst_g, sink);
  run<0, 0, 81, 0, 0, 0, 1>("4+4, no DMA, 1 read per gap", d, dst_g, sink);
  run<0, 0, 81, 0, 0, 0, 2>("4+4, no DMA, 2 reads per gap", d, dst_g, sink);
  run<0, 7, 81, 0, 3, 0>("4+4, 7 DMA/wave burst prio3, burst reads", d, dst_g, sink);
  run<0, 7, 81, 0, 3, 0, 1>("4+4, 7 DMA/wave burst prio3, 1 read per gap", d, dst_g, sink);
  run<0, 7, 81, 0, 3, 0, 2>("4+4, 7 DMA/wave burst prio3, 2 reads per gap", d, dst_g, sink);
  run<0, 7, 81, 2, 0, 0, 1>("4+4, 7 DMA/wave sleep2 prio0, 1 read per gap", d, dst_g, sink);
  run<0, 14, 81, 0, 3, 0>("4+4, 14 DMA/wave burst prio3, burst reads", d, dst_g, sink);
  run<0, 14, 81, 0, 3, 0, 1>("4+4, 14 DMA/wave burst prio3, 1 read per gap", d, dst_g, sink);
  run<0, 14, 81, 0, 0, 0, 1>("4+4, 14 DMA/wave burst prio0, 1 read per gap", d, dst_g, sink);
  run<0, 14, 81, 1, 0, 0, 1>("4+4, 14 DMA/wave sleep1 prio0, 1 read per gap", d, dst_g, sink);
  run<0, 14, 81, 2, 0, 0, 1>("4+4, 14 DMA/wave sleep2 prio0, 1 read per gap", d, dst_g, sink);
  run<1, 0, 81, 0, 0, 0, 1>("3+1, no DMA, 1 read per gap", d, dst_g, sink);
  run<1, 7, 108, 0, 0, 0, 1>("3+1, 28 DMA on wave 3, 108 MFMA, 1 read per gap", d, dst_g, sink);
  run<1, 14, 108, 0, 0, 0, 1>("3+1, 56 DMA on wave 3, 108 MFMA, 1 read per gap", d, dst_g, sink);
  run<0, 4, 81, 0, 3, 3, 1>("4+4, 4 DMA/wave + consumers store 3 KiB each, 1 read per gap", d, dst_g, sink);
  run<0, 7, 81, 0, 3, 3, 1>("4+4, 7 DMA/wave + consumers store 3 KiB each, 1 read per gap", d, dst_g, sink);
  return 0;
}
