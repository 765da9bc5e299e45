// Microbenchmark 3: sustained v_mfma_f32_16x16x32_bf16 rate, one or two waves per SIMD on every CU,
// random vs zero operands (DVFS), to calibrate "cycles per MFMA" for the conv kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(256) void k(const bf16x8* src, int iters, float* sink, unsigned long long* cyc) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a = src[threadIdx.x], b = src[256 + threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < NACC; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k32(const bf16x8* src, int iters, float* sink, unsigned long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a = src[threadIdx.x], b = src[256 + threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < NACC; ++i) t += acc[i][0];
  if (t == 123.456f) sink[0] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
void run32(const char* name, const bf16x8* d, int wgs_per_cu, float* sink, unsigned long long* cyc) {
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k32<NACC><<<256 * wgs_per_cu, 256>>>(d, 100, sink, cyc);
  hipEventRecord(a);
  k32<NACC><<<256 * wgs_per_cu, 256>>>(d, iters, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)iters * NACC;
  double tf = nm * 32 * 32 * 16 * 2 * 4 * wgs_per_cu * 256 / (ms * 1e-3) / 1e12;
  printf("%-8s 32x32x16 %d wave(s)/SIMD, %2d accumulators: %6.1f ns per MFMA per wave, ticks/MFMA %.1f, %7.1f TFLOP/s chip\n", name, wgs_per_cu, NACC,
         ms * 1e6 / nm, (double)c / nm, tf);
}
template <int NACC>
void run(const char* name, const bf16x8* d, int wgs_per_cu, float* sink, unsigned long long* cyc) {
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<NACC><<<256 * wgs_per_cu, 256>>>(d, 100, sink, cyc);
  hipEventRecord(a);
  k<NACC><<<256 * wgs_per_cu, 256>>>(d, iters, sink, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)iters * NACC;                  // MFMAs per wave
  double tf = nm * 16 * 16 * 32 * 2 * 4 * wgs_per_cu * 256 / (ms * 1e-3) / 1e12;
  printf("%-8s %d wave(s)/SIMD, %2d accumulators: %6.1f ns per MFMA per wave, s_memtime ticks/MFMA %.1f, %7.1f TFLOP/s chip\n", name, wgs_per_cu, NACC,
         ms * 1e6 / nm, (double)c / nm, tf);
}
int main() {
  bf16x8* d; hipMalloc(&d, 16384);
  float* sink; hipMalloc(&sink, 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  unsigned short h[4096];
  for (int z = 0; z < 2; ++z) {
    for (int i = 0; i < 4096; ++i) h[i] = z ? 0 : (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
    const char* nm = z ? "zeros" : "random";
    run<4>(nm, d, 1, sink, cyc); run<12>(nm, d, 1, sink, cyc); run<24>(nm, d, 1, sink, cyc);
    run<12>(nm, d, 2, sink, cyc); run<24>(nm, d, 2, sink, cyc); run<12>(nm, d, 4, sink, cyc);
    run32<2>(nm, d, 1, sink, cyc); run32<6>(nm, d, 1, sink, cyc); run32<6>(nm, d, 2, sink, cyc);
  }
  return 0;
}
