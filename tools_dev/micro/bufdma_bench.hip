// Microbenchmark 8: cost of the producers' address arithmetic.  The convolution's producer waves issue one LDS-DMA per
// (halo position group, plane): (A) global_load_lds_dwordx4 with a per-lane 64-bit address = select(in-image, tensor +
// offset + plane stride, zero page); (B) buffer_load_dwordx4 ... lds with a per-item 32-bit VGPR offset, the chunk /
// plane offset in an SGPR, and out-of-image lanes pointed past num_records (the hardware returns zeros).
// Also checks that (B) writes the same LDS image as (A), zeros included.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int MODE, int NPOS, int NPL>
__global__ __launch_bounds__(256) void k(const char* in, const char* zero16, unsigned tensor_bytes, int HW, int nchunks, int items,
                                         unsigned* check, unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-resource type exists in device compilation only
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, tensor_bytes, 0x00020000);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < items; ++it) {
    // per item: halo positions of this lane (some outside the image)
    size_t xoff[NPOS];
    unsigned voff[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      // a 14 x 34 halo of a 48 x 48 image (tile 12 x 32), rows contiguous in memory; the tile walks over 16 images
      const int idx = i * 256 + tid, hy = idx / 34, hx = idx - hy * 34;
      const int tile = it * 256 + blockIdx.x, img = tile % 16, ty = (tile / 16) % 4, tx = (tile / 64) % 2;
      const int iy = ty * 12 - 1 + hy, ix = tx * 32 - 1 + hx;
      const bool ok = hy < 14 && iy >= 0 && iy < 48 && ix >= 0 && ix < 48;
      const int pix = img * 2304 + iy * 48 + ix;
      xoff[i] = ok ? (size_t)pix * 16 : ~(size_t)0;
      voff[i] = ok ? (unsigned)pix * 16u : 0xfffffff0u;
    }
    for (int c = 0; c < nchunks; ++c) {
      char* dst = smem + (c & 1) * (NPOS * NPL * 4096);
      if (MODE == 0) {
        const char* inb = in + (size_t)c * NPL * HW * 16;
#pragma unroll
        for (int i = 0; i < NPOS; ++i)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            const char* src = xoff[i] != ~(size_t)0 ? inb + xoff[i] + (size_t)pl * HW * 16 : zero16;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + (pl * NPOS + i) * 4096 + wave * 1024), 16, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < NPOS; ++i)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            const unsigned soff = (unsigned)((c * NPL + pl) * HW * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + (pl * NPOS + i) * 4096 + wave * 1024), 16, voff[i], soff, 0, 0);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0) {
    unsigned h = 0;
    for (int j = tid; j < NPOS * NPL * 1024 * 2; j += 256) h = h * 31 + ((unsigned*)smem)[j] * (j + 1);
    check[tid] = h;
    if (tid == 0) cyc[0] = t1 - t0;
  }
#endif
}

template <int MODE, int NPOS, int NPL>
void run(const char* name, const char* in, const char* zero, unsigned bytes, int HW, unsigned* hcheck) {
  static unsigned* check = nullptr; static unsigned long long* cyc = nullptr;
  if (!check) { hipMalloc(&check, 1024); hipMalloc(&cyc, 8); }
  const int nchunks = 6, items = 40;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, NPOS, NPL><<<256, 256, 2 * NPOS * NPL * 4096>>>(in, zero, bytes, HW, nchunks, 2, check, cyc);
  hipEventRecord(a);
  k<MODE, NPOS, NPL><<<256, 256, 2 * NPOS * NPL * 4096>>>(in, zero, bytes, HW, nchunks, items, check, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(hcheck, check, 1024, hipMemcpyDeviceToHost);
  const double ndma = (double)items * nchunks * NPOS * NPL;
  printf("%-44s %7.1f us  %6.0f ticks per DMA instruction (per wave)  %5.2f TB/s chip\n", name, ms * 1e3, (double)c / ndma,
         ndma * 4096.0 * 256 / (ms * 1e-3) / 1e12);
}

int main() {
  const int HW = 48 * 48 * 16;                 // pixels of a plane over 16 images
  const unsigned bytes = (unsigned)(12 * (size_t)HW * 16);
  char* in; hipMalloc(&in, bytes);
  char* zero; hipMalloc(&zero, 4096); hipMemset(zero, 0, 4096);
  unsigned short* h = (unsigned short*)malloc(bytes);
  for (size_t i = 0; i < bytes / 2; ++i) h[i] = (unsigned short)(rand() & 0xffff);
  hipMemcpy(in, h, bytes, hipMemcpyHostToDevice);
  unsigned ca[256], cb[256];
  run<0, 2, 2>("global_load_lds, 64-bit select, 2 pos x 2 pl", in, zero, bytes, HW, ca);
  run<1, 2, 2>("buffer_load lds, voffset + soffset", in, zero, bytes, HW, cb);
  int same = 1; for (int i = 0; i < 256; ++i) same &= ca[i] == cb[i];
  printf("LDS images identical (zeros for out-of-image lanes included): %s\n", same ? "yes" : "NO");
  run<0, 2, 1>("global_load_lds, 2 pos x 1 pl", in, zero, bytes, HW, ca);
  run<1, 2, 1>("buffer_load lds,  2 pos x 1 pl", in, zero, bytes, HW, cb);
  same = 1; for (int i = 0; i < 256; ++i) same &= ca[i] == cb[i];
  printf("LDS images identical: %s\n", same ? "yes" : "NO");
  return 0;
}
