// Microbenchmark: how fast can one 256-thread workgroup per CU pull an L2-resident (or HBM) stream
// into LDS, via LDS-DMA (global_load_lds_dwordx4) vs register staging?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int MODE, int PER_BARRIER>   // MODE 0: glds, 1: reg staging ; PER_BARRIER: KiB per wave between barriers
__global__ __launch_bounds__(256) void k(const char* src, size_t src_bytes, int iters, int wg_stride_bytes, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  size_t base = ((size_t)blockIdx.x * wg_stride_bytes) % src_bytes;
  float acc = 0;
  for (int it = 0; it < iters; ++it) {
    // each wave moves PER_BARRIER KiB: PER_BARRIER glds of 1 KiB
#pragma unroll
    for (int j = 0; j < PER_BARRIER; ++j) {
      const size_t off = (base + ((size_t)(it * PER_BARRIER + j) * 4 + wave) * 1024 + lane * 16) % src_bytes;
      char* l = smem + ((j * 4 + wave) * 1024);
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + off), (lds_void_t*)l, 16, 0, 0);
      } else {
        uint4 v = *(const uint4*)(src + off);
        *(uint4*)(l + lane * 16) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *(float*)(smem + tid * 4);
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int PB>
void run(const char* name, const char* d, size_t bytes, int stride, float* sink) {
  const int iters = 200, grid = 256;
  hipFuncSetAttribute((const void*)k<MODE, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, PB><<<grid, 256, PB * 4096, 0>>>(d, bytes, 10, stride, sink);
  hipEventRecord(a);
  k<MODE, PB><<<grid, 256, PB * 4096, 0>>>(d, bytes, iters, stride, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double total = (double)grid * iters * PB * 4096;
  printf("%-28s src %7.1f MB  %2d KiB/wave/barrier: %7.1f us/iter  %6.1f GB/s per CU  %6.2f TB/s chip\n", name, bytes / 1e6, PB,
         ms * 1e3 / iters, total / grid / (ms * 1e-3) / 1e9, total / (ms * 1e-3) / 1e12);
}

int main() {
  size_t big = 1ull << 30;
  char* d; hipMalloc(&d, big); hipMemset(d, 1, big);
  float* sink; hipMalloc(&sink, 4);
  for (int pass = 0; pass < 2; ++pass) {
    size_t bytes = pass == 0 ? (size_t)166 * 1024 : big;     // L2-resident weights vs HBM stream
    int stride = pass == 0 ? 0 : 4 << 20;
    run<0, 4>("glds", d, bytes, stride, sink);
    run<0, 8>("glds", d, bytes, stride, sink);
    run<0, 16>("glds", d, bytes, stride, sink);
    run<0, 32>("glds", d, bytes, stride, sink);
    run<1, 4>("regs", d, bytes, stride, sink);
    run<1, 8>("regs", d, bytes, stride, sink);
    run<1, 16>("regs", d, bytes, stride, sink);
  }
  return 0;
}
