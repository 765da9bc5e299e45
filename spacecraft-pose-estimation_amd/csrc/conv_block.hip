// Host side of the fused BasicBlock kernel (conv_block_kernel.h): eligibility and launch.
#include <stdlib.h>
#include <type_traits>

#include "conv_block2_kernel.h"

namespace scpose {

bool block_fusable(const PackedConv& c1, const PackedConv& c2) {
  static const char* e = dev_env("SCPOSE_FUSE_BLOCK");
  if (e && atoi(e) == 0) return false;
  const int C = c1.cin;
  return c1.variant == 0 && c2.variant == 0 && c1.ks == 3 && c2.ks == 3 && c1.stride == 1 && c2.stride == 1 &&
         c1.cout == C && c2.cin == C && c2.cout == C && (C == 48 || C == 32) && c1.mt == C && c2.mt == C &&
         c1.nchunks == 1 && c2.nchunks == 1 && c1.n_mblk == 1 && c2.n_mblk == 1 && c1.dtype == c2.dtype &&
         c1.ksteps_full == block_ksteps(C / 16) && block_lds_bytes(C / 16) <= 160 * 1024;
}

int32_t block_launch(const PackedConv& c1, const PackedConv& c2, const void* in, int N, int H, int W, void* out,
                     hipStream_t stream) {
  SCP_REQUIRE(block_fusable(c1, c2), "block: the two convolutions are not a fusable BasicBlock");
  SCP_REQUIRE(N > 0 && H > 0 && W > 0, "block: bad shape N=%d H=%d W=%d", N, H, W);
  BlockLaunch L;
  L.in = in; L.w1 = c1.d_w; L.w2 = c2.d_w; L.b1 = c1.d_bias; L.b2 = c2.d_bias; L.out = out;
  L.zero16 = conv_zero_page();
  SCP_REQUIRE(L.zero16, "block: cannot allocate the zero page");
  {
    static const char* e = dev_env("SCPOSE_M32_BUF");
    const size_t bytes = (size_t)N * (c1.cin / 8) * H * W * 16;
    L.bytes = (bytes < 0xfffffff0ull && !(e && atoi(e) == 0)) ? (uint32_t)bytes : 0;
  }
  L.N = N; L.H = H; L.W = W;
  L.tiles_x = (W + kBlockTile - 1) / kBlockTile;
  L.tiles_y = (H + kBlockTile - 1) / kBlockTile;
  L.tiles_total = N * L.tiles_x * L.tiles_y;
  L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y);
  L.fd_tiles_x = make_fastdiv(L.tiles_x);
  int grid = conv_device_cus();
  if (grid > L.tiles_total) grid = L.tiles_total;
  L.tiles_per_wg = (L.tiles_total + grid - 1) / grid;
  L.grid = (L.tiles_total + L.tiles_per_wg - 1) / L.tiles_per_wg;
  { static const char* e = dev_env("SCPOSE_DBG"); const int dbg = (kDevBuild && e) ? atoi(e) : 0;
    L.dbg_buf = (dbg & 8) ? conv_dbg_buffer(stream) : nullptr;
    if (dbg & 8) conv_dbg_set_grid(L.grid); }
  const int mrep = c1.cin / 16;
  // second form (conv_block2_kernel.h: one layer per wave, weights in registers) unless the tensor needs 64-bit addressing
  static const char* v1 = dev_env("SCPOSE_BLOCK_V1");
  if (L.bytes != 0 && !(v1 && atoi(v1))) {
    if (c1.dtype == SCPOSE_DT_BF16) return mrep == 3 ? block2_launch_one<0, 3>(L, stream) : block2_launch_one<0, 2>(L, stream);
    return mrep == 3 ? block2_launch_one<1, 3>(L, stream) : block2_launch_one<1, 2>(L, stream);
  }
  if (c1.dtype == SCPOSE_DT_BF16) {
    if (mrep == 3) return block_launch_one<0, 3>(L, stream);
    return block_launch_one<0, 2>(L, stream);
  }
  if (mrep == 3) return block_launch_one<1, 3>(L, stream);
  return block_launch_one<1, 2>(L, stream);
}

}  // namespace scpose
