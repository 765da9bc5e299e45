// 3x3 stride-2 convolution (+ folded BN, optional ReLU) with the weights held in REGISTERS, on
// v_mfma_f32_16x16x32_{bf16,f16}: the fuse-layer down paths (pose_hrnet.py:211-239) and the transition convolutions
// that open a new branch (:355-369) for Cin = 32 / 48 / 64.
//
// Why a third kernel family for these layers: they read four input pixels per output pixel and have little K
// (Cin = 48: 14 k-steps), so on the producer/consumer kernel (conv_m32p_kernel.h) a stage holds 27-54 MFMAs against
// ~1 000 cycles of fixed stage cost (barrier, pipeline restart, weight-chunk staging): 2-3 TB/s and 0.15-0.28 MFMA
// utilisation (profiles/round1_final_*).  Here the whole K of a wave's output channels sits in its registers
// (Cin = 48: 3 blocks x 14 k-steps x 4 VGPRs = 168), LDS holds nothing but the input tile (double-buffered, filled
// by LDS-DMA), and a tile costs ONE workgroup barrier.
//
// One 512-thread workgroup per CU, persistent over tiles of 8 x 16 output pixels.  Waves split G ways over the
// output channels (16 * NBLK each) and 8 / G ways over the tile's rows; a wave therefore computes NBLK blocks x G rows.
// Input tile: [Cin / 8 planes][17 x 33 pixels][16 B]; 561 slots per plane is odd, so the stride-2 fragment reads of
// two k-groups that share a ds_read_b128 lane group never collide (MI355X_MICROARCH.md, LDS).
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // dma16_buf, make_buf, BUF_OOB

namespace scpose {

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
constexpr int kTH = 8, kTW = 16;
// input pixels a tile needs: stride 2: 17 x 33 = 561, stride 1: 10 x 18 = 180; plane pitch in 16-byte slots kept odd
constexpr int tile_mh(int s) { return s * (kTH - 1) + 3; }
constexpr int tile_mw(int s) { return s * (kTW - 1) + 3; }
constexpr int tile_pitch(int s) { return (tile_mh(s) * tile_mw(s)) | 1; }

struct S2rLaunch {
  const void* in;
  const void* w;         // [k-step][cout block][4 k-groups][16 rows][8]
  const float* bias;     // MFMA row order
  void* out;
  uint32_t in_bytes, out_bytes;
  int32_t N, H, W, Ho, Wo, cout_planes, relu;
  int32_t tiles_x, tiles_y, tiles_total, tiles_per_wg, grid;
  FastDiv fd_tiles_img, fd_tiles_x;   // tile decode by exact multiply-shift (two runtime divisions per decode were ~10 % of a 1.5 us tile)
};

inline int s2r_row_channel(int row) {
  const int q = row >> 2, reg = row & 3;
  return (q & 1) * 8 + (q >> 1) * 4 + reg;
}

__device__ __forceinline__ int s2r_fdiv(int n, const FastDiv& f) {   // exact n / d for a launch constant d (common.h)
  return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n * f.add) >> f.shift);
}

}  // namespace

template <int DT, int PLANES, int NBLK, int G, int S>
__global__ __launch_bounds__(512, 2) void conv_s2r_kernel(const S2rLaunch p) {
  constexpr int kMW = tile_mw(S), kMPix = tile_mh(S) * tile_mw(S), kMS = tile_pitch(S) * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int NPT = 9 * PLANES;              // (tap, plane) pairs
  constexpr int KS = (NPT + 3) / 4;            // 32-deep k-steps
  constexpr int NCOL = G;                      // tile rows per wave
  constexpr int XB = PLANES * kMS;             // bytes of one input-tile buffer
  constexpr int NSLOT = (kMPix + 63) / 64;     // 64-pixel DMA pieces per plane

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int g = wave % G, rset = wave / G;     // output-channel group, row set
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // ---- this wave's weights: NBLK blocks x KS k-steps ----
  frag_t wf[KS][NBLK];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb)
      wf[s][mb] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w) + ((((size_t)s * (NBLK * G) + g * NBLK + mb) * 4 + q) * 16 + r) * 16);
  float4 bs[NBLK];
#pragma unroll
  for (int mb = 0; mb < NBLK; ++mb) bs[mb] = *reinterpret_cast<const float4*>(p.bias + (g * NBLK + mb) * 16 + q * 4);
  // operand addressing: k-step s, k-group q -> pair t = 4 s + q = (tap, plane); padding pairs read slot 0 (finite, weight 0)
  int koff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int t = 4 * s + q, tap = t / PLANES, plane = t - tap * PLANES, ky = tap / 3, kx = tap - 3 * ky;
    koff[s] = t < NPT ? plane * kMS + (ky * kMW + kx) * 16 : 0;
  }

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(p.tiles_total, t_begin + p.tiles_per_wg);
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = s2r_fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = s2r_fdiv(rem, p.fd_tiles_x);
    oy0 = ty * kTH; ox0 = (rem - ty * p.tiles_x) * kTW;
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.in_bytes), rs_out = make_buf(p.out, p.out_bytes);
  // LDS-DMA of tile t into buffer b: piece = 64 pixel slots of one plane; wave w takes pieces w, w + 8, ...
  auto issue_x = [&](int t, int b) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    char* xl = smem + b * XB;
#pragma unroll 1
    for (int piece = wave; piece < NSLOT; piece += 8) {
      const int slot = piece * 64 + lane;
      const int my = slot / kMW, mx = slot - my * kMW;
      const int iy = S * oy0 - 1 + my, ix = S * ox0 - 1 + mx;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const uint32_t voff = ok ? (uint32_t)(img * PLANES * HW + iy * p.W + ix) * 16u : BUF_OOB;   // padding: read as zeros
      if (slot < kMPix) {   // lanes past the plane's last slot stay inactive: their LDS write would land in the next plane
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) dma16_buf(rs_in, voff, (uint32_t)(pl * HW) * 16u, xl + pl * kMS + piece * 1024);
      }
    }
  };

  if (t_begin < t_end) issue_x(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const size_t plane_sz = (size_t)p.Ho * p.Wo;
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    if (t + 1 < t_end) issue_x(t + 1, buf ^ 1);            // streams in under this tile's MFMAs
    const char* xl = smem + buf * XB;

    // columns (tile rows) of this wave: rset * NCOL + c; processed two at a time (one when NCOL == 1)
    constexpr int CSTEP = NCOL >= 2 ? 2 : 1;
#pragma unroll 1
    for (int c0 = 0; c0 < NCOL; c0 += CSTEP) {
      f32x4 acc[NBLK][CSTEP];
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb)
#pragma unroll
        for (int c = 0; c < CSTEP; ++c) acc[mb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int py0 = rset * NCOL + c0;
      const char* bcol = xl + ((S * py0) * kMW + S * r) * 16;
      // fragments two k-steps ahead of the MFMAs that use them (left alone, the compiler issues each read right before its
      // use and waits for it); the scheduling barriers pin that order
      frag_t bf[3][CSTEP];
#pragma unroll
      for (int s0 = 0; s0 < 2 && s0 < KS; ++s0)
#pragma unroll
        for (int c = 0; c < CSTEP; ++c) bf[s0][c] = *reinterpret_cast<const frag_t*>(bcol + c * (S * kMW * 16) + koff[s0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 2 < KS) {
#pragma unroll
          for (int c = 0; c < CSTEP; ++c) bf[(s + 2) % 3][c] = *reinterpret_cast<const frag_t*>(bcol + c * (S * kMW * 16) + koff[s + 2]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < NBLK; ++mb)
#pragma unroll
          for (int c = 0; c < CSTEP; ++c) acc[mb][c] = mfma16<T>(wf[s][mb], bf[s % 3][c], acc[mb][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: v_permlane32_swap gives the lower half-wave the 8 channels (plane 2 * block + psel) of column 0's
      // pixel and the upper half-wave those of column 1's (one column: the lower half-wave stores, the upper idles)
      const uint32_t relu_floor = p.relu ? 0u : 0x80008000u;
      const int oy = oy0 + py0 + (CSTEP == 2 ? half : 0), ox = ox0 + r;
      const bool store_ok = oy < p.Ho && ox < p.Wo && (CSTEP == 2 || half == 0);
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb) {
        const float4 b4 = bs[mb];
        uint32_t a[4], b[4];
        a[0] = __float_as_uint(acc[mb][0][0] + b4.x); a[1] = __float_as_uint(acc[mb][0][1] + b4.y);
        a[2] = __float_as_uint(acc[mb][0][2] + b4.z); a[3] = __float_as_uint(acc[mb][0][3] + b4.w);
        if constexpr (CSTEP == 2) {
          b[0] = __float_as_uint(acc[mb][1][0] + b4.x); b[1] = __float_as_uint(acc[mb][1][1] + b4.y);
          b[2] = __float_as_uint(acc[mb][1][2] + b4.z); b[3] = __float_as_uint(acc[mb][1][3] + b4.w);
        } else {
          b[0] = b[1] = b[2] = b[3] = 0u;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
          a[jj] = sw[0]; b[jj] = sw[1];
        }
        u32x4_t ov;
        ov[0] = relu2_16(pack2<T>(__uint_as_float(a[0]), __uint_as_float(a[1])), relu_floor);
        ov[1] = relu2_16(pack2<T>(__uint_as_float(a[2]), __uint_as_float(a[3])), relu_floor);
        ov[2] = relu2_16(pack2<T>(__uint_as_float(b[0]), __uint_as_float(b[1])), relu_floor);
        ov[3] = relu2_16(pack2<T>(__uint_as_float(b[2]), __uint_as_float(b[3])), relu_floor);
        // buffer store: every wave issues exactly one store instruction per block (pixels outside the map get an
        // out-of-range offset and are dropped by the hardware), which the counted wait below relies on
        const int plane = 2 * (g * NBLK + mb) + psel;
        const uint32_t voff = store_ok ? (uint32_t)(((size_t)img * p.cout_planes + plane) * plane_sz + (size_t)oy * p.Wo + ox) * 16u : BUF_OOB;
        store16_buf(rs_out, voff, 0u, __builtin_bit_cast(u32x4, ov));
      }
    }
    // next tile landed (this wave's pieces) -- vmcnt(0): stores may retire before older loads, so a counted wait that
    // skips this tile's output stores would not guarantee the DMA has landed
    // (one asm statement with a memory clobber: the builtin barrier is IntrNoMem to the compiler, which may then hoist the next
    // tile's ordinary LDS loads above it, into a buffer other waves' LDS-DMA is still filling -- conv_block2_kernel.h)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                          // ... and everybody is done with this buffer (LDS reads are consumed)
  }
}


// ---- Cin = 96 (12 input planes): 768-thread workgroups ---------------------------------------------------------------------
// The 3x3 stride-2 layers with 96 input channels (fuse down paths of branch 1: 96 -> 96 / 192 / 384) have K = 864: one
// 16-channel output block needs 27 k-steps x 4 VGPRs = 108 registers of weights, so a wave owns ONE block and a workgroup of
// TWELVE waves (three per SIMD, 168 registers each) covers 192 output channels of an 8 x 8-pixel tile.  12 planes = 3 k-steps
// per tap exactly, so the fragment address of k-step s is a per-lane base (k-group q -> plane q) plus a compile-time
// immediate (planes 4 (s % 3) and tap s / 3).  LDS holds only the double-buffered input tile ([12 planes][17 x 17 pixels]:
// 2 x 55.5 KB; wave w fills plane w), and a tile costs one barrier.  Cout = 96: the twelve waves are 6 blocks x 2 row sets;
// Cout = 384: two passes, each on its own half of the workgroups (weights stay in registers for a workgroup's whole life).
constexpr int k12T = 8, k12MW = 17, k12MPix = 17 * 17, k12MS = k12MPix * 16, k12XB = 12 * k12MS;

struct S2r12Launch {
  const void* in;
  const void* w;         // [27 k-steps][Cout / 16 blocks][4 k-groups][16 rows][8]  (conv_s2r_pack)
  const float* bias;     // MFMA row order
  void* out;
  uint32_t in_bytes, out_bytes;
  int32_t N, H, W, Ho, Wo, cout_planes, relu;
  int32_t tiles_x, tiles_y, tiles_total, npass, wgs_per_pass, tiles_per_wg;
  FastDiv fd_tiles_img, fd_tiles_x;
};

template <int DT, int G>   // G: output-channel blocks per workgroup pass (12, or 6 with two row sets)
__global__ __launch_bounds__(768, 3) void conv_s2r12_kernel(const S2r12Launch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int KS = 27, RSETS = 12 / G, NCOL = 4 / RSETS;     // a wave's columns: pairs of tile rows (16 pixels each)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int g = wave % G, rset = wave / G;
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int pass = blockIdx.x / p.wgs_per_pass, wgp = blockIdx.x - pass * p.wgs_per_pass;
  const int nblocks = p.cout_planes >> 1, blk = pass * G + g;

  frag_t wf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    wf[s] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w) + ((((size_t)s * nblocks + blk) * 4 + q) * 16 + r) * 16);
  const float4 bs = *reinterpret_cast<const float4*>(p.bias + blk * 16 + q * 4);

  const int wg = xcd_remap(wgp, p.wgs_per_pass);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(p.tiles_total, t_begin + p.tiles_per_wg);
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = s2r_fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = s2r_fdiv(rem, p.fd_tiles_x);
    oy0 = ty * k12T; ox0 = (rem - ty * p.tiles_x) * k12T;
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.in_bytes), rs_out = make_buf(p.out, p.out_bytes);
  // LDS-DMA of tile t into buffer b: wave w fills plane w, five 64-pixel pieces (the last one 33 pixels)
  auto issue_x = [&](int t, int b) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    char* xl = smem + b * k12XB + wave * k12MS;
#pragma unroll
    for (int piece = 0; piece < (k12MPix + 63) / 64; ++piece) {
      const int slot = piece * 64 + lane;
      const int my = slot / k12MW, mx = slot - my * k12MW;
      const int iy = 2 * oy0 - 1 + my, ix = 2 * ox0 - 1 + mx;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const uint32_t voff = ok ? (uint32_t)(img * 12 * HW + iy * p.W + ix) * 16u : BUF_OOB;   // padding: read as zeros
      if (slot < k12MPix) dma16_buf(rs_in, voff, (uint32_t)(wave * HW) * 16u, xl + piece * 1024);   // lanes past the plane stay inactive
    }
  };

  if (t_begin < t_end) issue_x(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // this lane's pixel inside a column: the column's two tile rows x 8 pixels
  const int lrow = r >> 3, lpx = (r - lrow) & 7;   // second row rotated by one pixel: every ds_read_b128 lane group then hits 16 distinct banks
  const int lbase = ((2 * lrow) * k12MW + 2 * lpx) * 16 + q * k12MS;
  const size_t plane_sz = (size_t)p.Ho * p.Wo;
  const uint32_t relu_floor = p.relu ? 0u : 0x80008000u;
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    if (t + 1 < t_end) issue_x(t + 1, buf ^ 1);            // streams in under this tile's MFMAs
    const char* xl = smem + buf * k12XB + lbase;
#pragma unroll 1
    for (int c0 = 0; c0 < NCOL; c0 += 2) {
      const int col0 = rset * NCOL + c0;                   // columns col0, col0 + 1: tile rows 2 col0 .. 2 col0 + 3
      const char* bcol = xl + col0 * (4 * k12MW * 16);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      auto koff = [](int s) { const int tap = s / 3, ky = tap / 3, kx = tap - 3 * ky; return 4 * (s % 3) * k12MS + (ky * k12MW + kx) * 16; };
      frag_t bf[3][2];                                     // fragments two k-steps ahead (see conv_s2r_kernel)
#pragma unroll
      for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[s0][c] = *reinterpret_cast<const frag_t*>(bcol + c * (4 * k12MW * 16) + koff(s0));
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 2 < KS) {
#pragma unroll
          for (int c = 0; c < 2; ++c) bf[(s + 2) % 3][c] = *reinterpret_cast<const frag_t*>(bcol + c * (4 * k12MW * 16) + koff(s + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma16<T>(wf[s], bf[s % 3][0], acc[0]);
        acc[1] = mfma16<T>(wf[s], bf[s % 3][1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: after the lane exchange the lower half-wave holds the 8 channels (plane 2 * block + psel) of column col0's
      // pixel, the upper half-wave those of column col0 + 1's
      uint32_t a[4], b[4];
      a[0] = __float_as_uint(acc[0][0] + bs.x); a[1] = __float_as_uint(acc[0][1] + bs.y);
      a[2] = __float_as_uint(acc[0][2] + bs.z); a[3] = __float_as_uint(acc[0][3] + bs.w);
      b[0] = __float_as_uint(acc[1][0] + bs.x); b[1] = __float_as_uint(acc[1][1] + bs.y);
      b[2] = __float_as_uint(acc[1][2] + bs.z); b[3] = __float_as_uint(acc[1][3] + bs.w);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
        a[jj] = sw[0]; b[jj] = sw[1];
      }
      u32x4_t ov;
      ov[0] = relu2_16(pack2<T>(__uint_as_float(a[0]), __uint_as_float(a[1])), relu_floor);
      ov[1] = relu2_16(pack2<T>(__uint_as_float(a[2]), __uint_as_float(a[3])), relu_floor);
      ov[2] = relu2_16(pack2<T>(__uint_as_float(b[0]), __uint_as_float(b[1])), relu_floor);
      ov[3] = relu2_16(pack2<T>(__uint_as_float(b[2]), __uint_as_float(b[3])), relu_floor);
      const int oy = oy0 + 2 * (col0 + half) + lrow, ox = ox0 + lpx;
      const bool store_ok = oy < p.Ho && ox < p.Wo;
      const int plane = 2 * blk + psel;
      const uint32_t voff = store_ok ? (uint32_t)(((size_t)img * p.cout_planes + plane) * plane_sz + (size_t)oy * p.Wo + ox) * 16u : BUF_OOB;
      store16_buf(rs_out, voff, 0u, __builtin_bit_cast(u32x4, ov));
    }
    // next tile landed (this wave's plane) -- vmcnt(0), not a counted wait: stores may retire before older loads
    // (one asm statement with a memory clobber: the builtin barrier is IntrNoMem to the compiler, which may then hoist the next
    // tile's ordinary LDS loads above it, into a buffer other waves' LDS-DMA is still filling -- conv_block2_kernel.h)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                          // ... and everybody is done with this buffer
  }
}

// ---- host ----
bool conv_s2r_config(int cin, int cout, int stride, int* planes, int* nblk, int* g) {
  if (stride == 1 && !(cin == 64 && cout == 64)) return false;   // stride 1: layer1's 3x3 (Bottleneck conv2, no residual)
  if (cin == 96) {                              // conv_s2r12_kernel: one block per wave, 6 or 12 blocks per workgroup pass
    // a pass of the kernel covers 16 * g output channels: 96 (g = 6) or whole multiples of 192 (g = 12).  Any other width
    // (288, 480, ...) would leave its last 96 channels unwritten -- those layers stay on the producer/consumer kernel.
    if (stride != 2 || !(cout == 96 || cout % 192 == 0)) return false;
    *planes = 12; *nblk = 1; *g = cout == 96 ? 6 : 12;
    return true;
  }
  if (cin != 32 && cin != 48 && cin != 64) return false;
  const int nb = cin == 48 ? 3 : 2;            // 16-channel blocks per wave: 48 / 32 output channels per group
  if (cout % (16 * nb) != 0) return false;
  const int groups = cout / (16 * nb);
  if (groups != 1 && groups != 2 && groups != 4) return false;
  *planes = cin / 8; *nblk = nb; *g = groups;
  return true;
}

size_t conv_s2r_pack(const float* w, int cout, int cin, int dtype, uint16_t* dst) {
  const int planes = cin / 8, npt = 9 * planes, ks = (npt + 3) / 4, nblocks = cout / 16;
  const size_t total = (size_t)ks * nblocks * 4 * 16 * 8;
  if (!dst) return total * 2;
  memset(dst, 0, total * 2);
  for (int s = 0; s < ks; ++s)
    for (int cb = 0; cb < nblocks; ++cb)
      for (int q = 0; q < 4; ++q) {
        const int t = 4 * s + q;
        if (t >= npt) continue;
        const int tap = t / planes, plane = t % planes, ky = tap / 3, kx = tap % 3;
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * cb + s2r_row_channel(r);
          uint16_t* d = dst + ((((size_t)s * nblocks + cb) * 4 + q) * 16 + r) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(w[((size_t)(co * cin + plane * 8 + j) * 3 + ky) * 3 + kx], dtype);
        }
      }
  return total * 2;
}

void conv_s2r_pack_bias(const float* bias, int cout, float* dst) {
  for (int pos = 0; pos < cout; ++pos) dst[pos] = bias ? bias[(pos & ~15) + s2r_row_channel(pos & 15)] : 0.f;
}

template <int DT, int PLANES, int NBLK, int G, int S = 2>
static int32_t s2r_launch_one(const S2rLaunch& L, hipStream_t st) {
  auto kern = conv_s2r_kernel<DT, PLANES, NBLK, G, S>;
  const size_t lds = 2 * (size_t)PLANES * tile_pitch(S) * 16;
  static LdsOptIn big;
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), (int)lds, &big); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(512), lds, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

template <int DT>
static int32_t s2r_dispatch(int planes, int g, int stride, const S2rLaunch& L, hipStream_t st) {
  if (stride == 1) return s2r_launch_one<DT, 8, 2, 2, 1>(L, st);
  if (planes == 6) { if (g == 1) return s2r_launch_one<DT, 6, 3, 1>(L, st); if (g == 2) return s2r_launch_one<DT, 6, 3, 2>(L, st); return s2r_launch_one<DT, 6, 3, 4>(L, st); }
  if (planes == 4) { if (g == 1) return s2r_launch_one<DT, 4, 2, 1>(L, st); if (g == 2) return s2r_launch_one<DT, 4, 2, 2>(L, st); return s2r_launch_one<DT, 4, 2, 4>(L, st); }
  if (g == 1) return s2r_launch_one<DT, 8, 2, 1>(L, st);
  if (g == 2) return s2r_launch_one<DT, 8, 2, 2>(L, st);
  return s2r_launch_one<DT, 8, 2, 4>(L, st);
}

template <int DT, int G>
static int32_t s2r12_launch_one(const S2r12Launch& L, int grid, hipStream_t st) {
  auto kern = conv_s2r12_kernel<DT, G>;
  static LdsOptIn big;
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 2 * k12XB, &big); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), 2 * k12XB, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

static int32_t conv_s2r12_launch(const PackedConv& pc, int g, const void* in, int N, int H, int W, int relu, void* out, hipStream_t stream) {
  S2r12Launch L{};
  L.in = in; L.w = pc.d_ws2; L.bias = pc.d_bs2; L.out = out;
  L.N = N; L.H = H; L.W = W; L.Ho = (H - 1) / 2 + 1; L.Wo = (W - 1) / 2 + 1;
  L.in_bytes = (uint32_t)((size_t)N * 12 * H * W * 16);
  L.out_bytes = (uint32_t)((size_t)N * (pc.cout / 8) * L.Ho * L.Wo * 16);
  L.cout_planes = pc.cout / 8; L.relu = relu;
  L.tiles_x = (L.Wo + k12T - 1) / k12T; L.tiles_y = (L.Ho + k12T - 1) / k12T;
  L.tiles_total = N * L.tiles_x * L.tiles_y;
  L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y); L.fd_tiles_x = make_fastdiv(L.tiles_x);
  L.npass = pc.cout / (16 * g);
  SCP_REQUIRE(L.npass * g * 16 == pc.cout, "conv s2r12: %d output channels are not a whole number of %d-channel passes", pc.cout, 16 * g);
  int per_pass = conv_device_cus() / L.npass;             // every pass gets its own share of the workgroups
  if (per_pass < 1) per_pass = 1;
  if (per_pass > L.tiles_total) per_pass = L.tiles_total;
  L.tiles_per_wg = (L.tiles_total + per_pass - 1) / per_pass;
  L.wgs_per_pass = (L.tiles_total + L.tiles_per_wg - 1) / L.tiles_per_wg;
  const int grid = L.wgs_per_pass * L.npass;
  if (pc.dtype == SCPOSE_DT_BF16) return g == 6 ? s2r12_launch_one<0, 6>(L, grid, stream) : s2r12_launch_one<0, 12>(L, grid, stream);
  return g == 6 ? s2r12_launch_one<1, 6>(L, grid, stream) : s2r12_launch_one<1, 12>(L, grid, stream);
}

int32_t conv_s2r_launch(const PackedConv& pc, const void* in, int N, int H, int W, int relu, void* out, hipStream_t stream) {
  int planes, nblk, g;
  SCP_REQUIRE(conv_s2r_config(pc.cin, pc.cout, pc.stride, &planes, &nblk, &g) && pc.d_ws2, "conv s2r: %d->%d not eligible", pc.cin, pc.cout);
  if (planes == 12) return conv_s2r12_launch(pc, g, in, N, H, W, relu, out, stream);
  S2rLaunch L{};
  L.in = in; L.w = pc.d_ws2; L.bias = pc.d_bs2; L.out = out;
  L.in_bytes = (uint32_t)((size_t)N * planes * H * W * 16);
  L.out_bytes = (uint32_t)((size_t)N * (pc.cout / 8) * ((H - 1) / pc.stride + 1) * ((W - 1) / pc.stride + 1) * 16);
  L.N = N; L.H = H; L.W = W; L.Ho = (H - 1) / pc.stride + 1; L.Wo = (W - 1) / pc.stride + 1;
  L.cout_planes = pc.cout / 8; L.relu = relu;
  L.tiles_x = (L.Wo + kTW - 1) / kTW; L.tiles_y = (L.Ho + kTH - 1) / kTH;
  L.tiles_total = N * L.tiles_x * L.tiles_y;
  L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y); L.fd_tiles_x = make_fastdiv(L.tiles_x);
  int grid = conv_device_cus();
  if (grid > L.tiles_total) grid = L.tiles_total;
  L.tiles_per_wg = (L.tiles_total + grid - 1) / grid;
  L.grid = (L.tiles_total + L.tiles_per_wg - 1) / L.tiles_per_wg;
  if (pc.dtype == SCPOSE_DT_BF16) return s2r_dispatch<0>(planes, g, pc.stride, L, stream);
  return s2r_dispatch<1>(planes, g, pc.stride, L, stream);
}

}  // namespace scpose
