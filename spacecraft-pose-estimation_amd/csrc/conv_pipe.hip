// Host glue of the pipelined convolution: zero page, persistent-grid sizing, dtype dispatch.
// The kernel template lives in conv_pipe_kernel.h and is instantiated per storage type in
// conv_pipe_bf16.hip / conv_pipe_f16.hip (two translation units so the build parallelises).
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "common.h"

namespace scpose {

int32_t conv_pipe_dispatch_bf16(int ks, int stride, int mrep, int nrep, int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st);
int32_t conv_pipe_dispatch_f16(int ks, int stride, int mrep, int nrep, int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st);

int32_t conv_stag_dispatch_bf16(int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st);
int32_t conv_stag_dispatch_f16(int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st);

static void* g_zero_page[16] = {nullptr};
static unsigned long long* g_dbg_buf = nullptr;
static int g_dbg_grid = 0;

const void* conv_zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_zero_page[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    g_zero_page[dev] = p;
  }
  return g_zero_page[dev];
}

// development only: phase-cycle dump buffer of the LAST launch, printed by scpose_dbg_dump()
unsigned long long* conv_dbg_buffer(hipStream_t stream) {
  static unsigned long long* buf = nullptr;
  if (!buf) (void)hipMalloc(&buf, 2048 * 8 * 6 * 8);
  (void)hipMemsetAsync(buf, 0, 2048 * 8 * 6 * 8, stream);
  g_dbg_buf = buf; g_dbg_grid = 0;
  return buf;
}
void conv_dbg_set_grid(int grid) { g_dbg_grid = grid; }

int conv_device_cus() {
  static int cus[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

size_t conv_pipe_lds_bytes(const PackedConv& pc, int plane_stride, int groups) {
  const bool resident = pc.nchunks == 1 && pc.n_mblk == 1;
  const size_t lds_w = (size_t)pc.ksteps_full * 4 * pc.mt * 16;
  const size_t lds_bias = (((size_t)pc.n_mblk * pc.mt * 4) + 511) & ~(size_t)511;
  if (groups == 2) return 512 + lds_bias + 3 * lds_w + 4 * (size_t)pc.cp * plane_stride;   // staggered two-group schedule
  return 512 + lds_bias + (resident ? 1 : 2) * lds_w + 2 * (size_t)pc.cp * plane_stride;
}

int32_t conv_launch_pipe(const PackedConv& pc, ConvLaunch& L, int nrep, int nt, int occ, int groups, hipStream_t stream) {
  L.lds_w = pc.ksteps_full * 4 * pc.mt * 16;
  L.lds_x = pc.cp * L.plane_stride;
  L.lds_bias = ((pc.n_mblk * pc.mt * 4) + 511) & ~511;
  L.nbuf_w = (pc.nchunks == 1 && pc.n_mblk == 1) ? 1 : 2;
  L.nbuf_x = 2;
  const size_t lds = conv_pipe_lds_bytes(pc, L.plane_stride, groups);
  L.groups = groups;
  SCP_REQUIRE(lds <= 160 * 1024, "conv: LDS image of %d->%d k%d s%d (%zu B) does not fit 160 KiB", pc.cin, pc.cout, pc.ks, pc.stride, lds);
  L.zero16 = conv_zero_page();
  SCP_REQUIRE(L.zero16, "conv: cannot allocate the zero page");
  L.tiles_total = L.N * L.tiles_x * L.tiles_y;
  L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y); L.fd_tiles_x = make_fastdiv(L.tiles_x);
  L.fd_nmblk = make_fastdiv(pc.n_mblk);
  L.nt = nt;
  L.items_total = ((L.tiles_total + nt * groups - 1) / (nt * groups)) * pc.n_mblk;
  { static const char* e = dev_env("SCPOSE_DBG"); L.dbg = (kDevBuild && e) ? atoi(e) : 0; }
  L.dbg_buf = nullptr;
  if (L.dbg & 8) L.dbg_buf = conv_dbg_buffer(stream);
  const int cus = conv_device_cus();
  int per_cu = (int)((160 * 1024) / lds);
  { static const char* e = dev_env("SCPOSE_K1_WGS"); const int cap = (pc.ks == 1 && e) ? atoi(e) : 2; per_cu = per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu); }
  if (occ < 2 || groups > 1) per_cu = 1;   // the variant's register budget assumes one workgroup per CU
  int grid = cus * per_cu;
  if (grid > L.items_total) grid = L.items_total;
  L.items_per_wg = (L.items_total + grid - 1) / grid;
  L.grid = (L.items_total + L.items_per_wg - 1) / L.items_per_wg;
  g_dbg_grid = L.grid;
  if (groups == 2) {
    if (pc.dtype == SCPOSE_DT_BF16) return conv_stag_dispatch_bf16(pc.mrep, nrep, L, lds, stream);
    return conv_stag_dispatch_f16(pc.mrep, nrep, L, lds, stream);
  }
  if (pc.dtype == SCPOSE_DT_BF16) return conv_pipe_dispatch_bf16(pc.ks, pc.stride, pc.mrep, nrep, nt, occ, L, lds, stream);
  return conv_pipe_dispatch_f16(pc.ks, pc.stride, pc.mrep, nrep, nt, occ, L, lds, stream);
}

}  // namespace scpose

// development helper (not part of include/scpose.h): mean cycles per phase of the last launch
extern "C" void scpose_dbg_dump(void) {
  using namespace scpose;
  if (!g_dbg_buf || g_dbg_grid <= 0) return;
  (void)hipDeviceSynchronize();
  const int n = g_dbg_grid * 8 * 6;   // up to 8 waves per workgroup (unused slots stay zero)
  std::vector<unsigned long long> h(n);
  (void)hipMemcpy(h.data(), g_dbg_buf, n * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"setup+residual+DMA issue", "MFMA loop", "vmcnt wait", "finalize", "barrier", "stores"};
  double tot = 0, sum[6] = {0};
  int nw = 0;
  for (int i = 0; i < g_dbg_grid * 8; ++i) { double t = 0; for (int k = 0; k < 6; ++k) { sum[k] += (double)h[i * 6 + k]; t += (double)h[i * 6 + k]; } nw += t > 0; }
  for (int k = 0; k < 6; ++k) tot += sum[k];
  printf("  (%d workgroups, %d waves with stamps)\n", g_dbg_grid, nw);
  for (int k = 0; k < 6; ++k) printf("  %-26s %10.0f cycles/wave  %5.1f%%\n", names[k], sum[k] / (nw ? nw : 1), 100 * sum[k] / tot);
  // role-split kernels: waves 0-3 and 4-7 of each workgroup separately (slot meaning differs per role)
  for (int role = 0; role < 2; ++role) {
    double rs[6] = {0}; int rn = 0;
    for (int b = 0; b < g_dbg_grid; ++b)
      for (int w = role * 4; w < role * 4 + 4; ++w) { double t = 0; for (int k = 0; k < 6; ++k) { rs[k] += (double)h[(b * 8 + w) * 6 + k]; t += (double)h[(b * 8 + w) * 6 + k]; } rn += t > 0; }
    if (rn) { printf("  waves %d-%d:", role * 4, role * 4 + 3); for (int k = 0; k < 6; ++k) printf(" %9.0f", rs[k] / rn); printf("\n"); }
  }
}
