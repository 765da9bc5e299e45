// Microbenchmark 5: the convolution's inner loop in isolation -- per tap 6 ds_read_b128 fragment reads
// (3 weight rows, 3 pixel columns) and 9 v_mfma_f32_32x32x16_bf16, reads one tap ahead -- with and
// without a DMA burst per 9 taps.  Reports cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OFF> __device__ __forceinline__ void rd(bf16x8& d, uint32_t a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
__device__ __forceinline__ void mf(f32x16& c, const bf16x8& a, const bf16x8& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void landed(bf16x8* a, bf16x8* b) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = 0; i < 3; ++i) { asm volatile("" : "+v"(a[i])); asm volatile("" : "+v"(b[i])); }
}

// MODE bit0: LDS fragment reads, bit1: DMA burst of VEC KiB-rows per wave before each chunk, bit2: B reads with a 2-way bank conflict
template <int MODE, int VEC>
__global__ __launch_bounds__(256, 1) void k(const char* src, int chunks, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, r = lane & 31;
  for (int i = tid; i < 100 * 1024 / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)src)[i];
  __syncthreads();
  f32x16 acc[9];
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const uint32_t wbase = (uint32_t)(size_t)smem + (half * 96 + r) * 16;                  // weight rows: 96-row block
  uint32_t xb[3];
  for (int n = 0; n < 3; ++n) {
    const int p = (wave * 3 + n) * 32 + r;
    const int slot = (MODE & 4) ? (p / 32) * 64 + (p % 32) * 2 : (p / 48) * 50 + p % 48;
    xb[n] = (uint32_t)(size_t)smem + 60 * 1024 + half * 8192 + slot * 16;
  }
  bf16x8 a0[3], b0[3], a1[3], b1[3];
  for (int i = 0; i < 3; ++i) { a0[i] = a1[i] = *(const bf16x8*)(smem + (i * 64 + lane) * 16); b0[i] = b1[i] = *(const bf16x8*)(smem + 8192 + (i * 64 + lane) * 16); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < chunks; ++c) {
    if (MODE & 2) {
      const char* s = src + ((size_t)c * VEC * 4096) % (160 * 1024);
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(s + (j * 4 + wave) * 1024 + lane * 16), (lds_void_t*)(smem + 100 * 1024 + (j * 4 + wave) * 1024), 16, 0, 0);
    }
    const uint32_t wa = wbase + (c & 1) * 27648;
#define TAP(T, CA, CB, NA, NB)                                                                  \
    mf(acc[0], CA[0], CB[0]); mf(acc[1], CA[1], CB[0]); mf(acc[2], CA[2], CB[0]);               \
    if (MODE & 1) { rd<(T) * 3072>(NA[0], wa); rd<(T) * 3072 + 512>(NA[1], wa); rd<(T) * 3072 + 1024>(NA[2], wa); \
                    rd<((T) % 3) * 16>(NB[0], xb[0] + ((T) / 3 % 3) * 800); rd<((T) % 3) * 16>(NB[1], xb[1] + ((T) / 3 % 3) * 800); rd<((T) % 3) * 16>(NB[2], xb[2] + ((T) / 3 % 3) * 800); } \
    mf(acc[3], CA[0], CB[1]); mf(acc[4], CA[1], CB[1]); mf(acc[5], CA[2], CB[1]);               \
    mf(acc[6], CA[0], CB[2]); mf(acc[7], CA[1], CB[2]); mf(acc[8], CA[2], CB[2]);               \
    if (MODE & 1) landed(NA, NB);
    TAP(1, a0, b0, a1, b1) TAP(2, a1, b1, a0, b0) TAP(3, a0, b0, a1, b1) TAP(4, a1, b1, a0, b0) TAP(5, a0, b0, a1, b1)
    TAP(6, a1, b1, a0, b0) TAP(7, a0, b0, a1, b1) TAP(8, a1, b1, a0, b0) TAP(0, a0, b0, a1, b1)
    for (int i = 0; i < 3; ++i) { a0[i] = a1[i]; b0[i] = b1[i]; }
    if (MODE & 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < 9; ++i) t += acc[i][0] + acc[i][7];
  if (t == 123.456f) sink[0] = t;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int VEC>
void run(const char* name, const char* d, float* sink, unsigned long long* cyc) {
  const int chunks = 400;
  hipFuncSetAttribute((const void*)k<MODE, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, VEC><<<256, 256, 160 * 1024>>>(d, 10, sink, cyc);
  hipEventRecord(a);
  k<MODE, VEC><<<256, 256, 160 * 1024>>>(d, chunks, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-36s %7.2f us/chunk (81 MFMA)  %5.1f ticks/MFMA  %5.1f ns/MFMA\n", name, ms * 1e3 / chunks, (double)c / (chunks * 81.0), ms * 1e6 / (chunks * 81.0));
}

int main() {
  char* d; hipMalloc(&d, 1 << 20);
  unsigned short* h = (unsigned short*)malloc(1 << 20);
  for (int i = 0; i < (1 << 19); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  float* sink; hipMalloc(&sink, 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  run<0, 0>("mfma only", d, sink, cyc);
  run<1, 0>("mfma + fragment reads", d, sink, cyc);
  run<5, 0>("mfma + reads, 2-way conflicts", d, sink, cyc);
  run<3, 7>("mfma + reads + 28 KiB DMA/chunk", d, sink, cyc);
  run<3, 11>("mfma + reads + 44 KiB DMA/chunk", d, sink, cyc);
  run<2, 11>("mfma + 44 KiB DMA/chunk (no reads)", d, sink, cyc);
  return 0;
}
