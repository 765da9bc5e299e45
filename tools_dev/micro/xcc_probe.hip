// Which XCD does workgroup b of a 256-workgroup launch run on?  (HW_REG_XCC_ID against blockIdx.x & 7)
//   hipcc --offload-arch=gfx950 -O2 tools_dev/micro/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* o) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) o[blockIdx.x] = v;
}
int main() {
  unsigned* d; unsigned h[512];
  hipMalloc(&d, sizeof(h));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(512), dim3(512), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0; unsigned raw_or = 0;
    for (int b = 0; b < 512; ++b) { same += (int)((h[b] & 7u) == (unsigned)(b & 7)); raw_or |= h[b]; }
    printf("launch %d: XCC_ID & 7 == blockIdx & 7 for %d of 512 workgroups; OR of raw values 0x%x; first 16:", rep, same, raw_or);
    for (int b = 0; b < 16; ++b) printf(" %x", h[b]);
    printf("\n");
  }
  return 0;
}
