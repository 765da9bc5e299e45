// Microbenchmark 6: LDS-DMA issued by dedicated producer waves (4-7) of a 512-thread workgroup while
// waves 0-3 idle at the barrier or run an MFMA stream; same 28 KiB/stage weight-chunk pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 3: as 2, and the consumers read 6 fragments (ds_read_b128) per 9 MFMAs like the convolution's tap loop.
// MODE 0: 256 threads, every wave issues.  MODE 1: 512 threads, waves 4-7 issue, 0-3 barrier only.
// MODE 2: as 1, waves 0-3 run NM MFMAs per stage.  LOOPED: issue through a runtime loop with an EXEC mask.
template <int MODE, int VEC, int NM, int LOOPED>
__global__ __launch_bounds__(MODE == 0 ? 256 : MODE == 5 ? 1024 : 512, MODE == 0 ? 1 : MODE == 5 ? 4 : 2) void k(const char* src, int chunk_bytes, int nchunks, int iters, float* sink, unsigned long long* cyc) {
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave_all = tid >> 6, lane = tid & 63, wave = wave_all & 3, ptid = tid & 255;
  const bool producer = MODE == 0 || MODE >= 4 || wave_all >= 4;
  constexpr int NW = MODE == 4 ? 8 : MODE == 5 ? 16 : 4;   // waves sharing the stage's DMA
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a = *(const bf16x8*)(src + tid * 16), b = *(const bf16x8*)(src + 8192 + tid * 16);
  for (int it = 0; it < iters; ++it) {
    if (producer) {
      const char* s = src + (size_t)((it + blockIdx.x) % nchunks) * chunk_bytes;
      char* dst = smem + (it & 1) * 32768;
      if (LOOPED) {
        for (int o = 0; o < chunk_bytes; o += 4096) {
          const int mine = o + ptid * 16;
          if (mine < chunk_bytes) __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + mine), (lds_void_t*)(dst + o + wave * 1024), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC * 4 / NW; ++j) {
          const int w = MODE >= 4 ? wave_all : wave;
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * NW + w) * 1024 + lane * 16), (lds_void_t*)(dst + (j * NW + w) * 1024), 16, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 3) {
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
    } else if (MODE == 2) {
#pragma unroll 1
      for (int m = 0; m < NM / 6; ++m)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
  }
  float t = 0;
  for (int i = 0; i < 6; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = __builtin_amdgcn_s_memtime() - t_begin;
}

template <int MODE, int VEC, int NM, int LOOPED>
void run(const char* name, const char* d, int chunk_bytes, int nchunks, float* sink) {
  const int iters = 600;
  hipFuncSetAttribute((const void*)k<MODE, VEC, NM, LOOPED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int threads = MODE == 0 ? 256 : MODE == 5 ? 1024 : 512;
  static unsigned long long* cyc = nullptr; if (!cyc) hipMalloc(&cyc, 8);
  k<MODE, VEC, NM, LOOPED><<<256, threads, 128 * 1024>>>(d, chunk_bytes, nchunks, 10, sink, cyc);
  hipEventRecord(a);
  k<MODE, VEC, NM, LOOPED><<<256, threads, 128 * 1024>>>(d, chunk_bytes, nchunks, iters, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double us = ms * 1e3 / iters;
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-46s %6.2f us/stage  %5.1f GB/s per CU  %6.0f ticks/stage  shader clock %.2f GHz\n", name, us, chunk_bytes / (us * 1e-6) / 1e9,
         (double)c / iters, (double)c / iters / (us * 1e3));
}

int main() {
  char* d; hipMalloc(&d, 1 << 20);
  unsigned short* h = (unsigned short*)malloc(1 << 20);
  for (int i = 0; i < (1 << 19); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  float* sink; hipMalloc(&sink, 4);
  run<0, 7, 0, 0>("256 thr, all waves issue, unrolled, 28 KiB", d, 28672, 6, sink);
  run<0, 7, 0, 1>("256 thr, all waves issue, looped, 27 KiB", d, 27648, 6, sink);
  run<1, 7, 0, 0>("512 thr, waves 4-7 issue, unrolled", d, 28672, 6, sink);
  run<1, 7, 0, 1>("512 thr, waves 4-7 issue, looped, 27 KiB", d, 27648, 6, sink);
  run<0, 16, 0, 0>("256 thr,  4 waves issue 64 KiB", d, 65536, 3, sink);
  run<4, 16, 0, 0>("512 thr,  8 waves issue 64 KiB", d, 65536, 3, sink);
  run<5, 16, 0, 0>("1024 thr, 16 waves issue 64 KiB", d, 65536, 3, sink);
  run<2, 0, 54, 0>("512 thr, 54 MFMA consumers, idle producers", d, 28672, 6, sink);
  run<2, 7, 54, 1>("512 thr, producers looped + 54 MFMA consumers", d, 27648, 6, sink);
  run<2, 7, 54, 0>("512 thr, producers unrolled + 54 MFMA consumers", d, 28672, 6, sink);
  run<3, 7, 54, 0>("512 thr, producers + 54 MFMA + LDS-read consumers", d, 28672, 6, sink);
  run<3, 7, 81, 0>("512 thr, producers + 81 MFMA + LDS-read consumers", d, 28672, 6, sink);
  run<3, 11, 81, 0>("512 thr, 44 KiB + 81 MFMA + LDS-read consumers", d, 45056, 3, sink);
  run<2, 11, 81, 0>("512 thr, 44 KiB + 81 MFMA consumers (no reads)", d, 45056, 3, sink);
  return 0;
}
