// Fused Bottleneck of layer1:  y = ReLU( bn3(conv3( ReLU(bn2(conv2( ReLU(bn1(conv1(x))) ))) )) + x ),
// conv1 1x1 256 -> 64, conv2 3x3 64 -> 64, conv3 1x1 64 -> 256, all on v_mfma_f32_16x16x32_{bf16,f16}
// (landmark_regression/lib/models/pose_hrnet.py:60-98, the three identity-residual blocks of layer1, :374-391).
//
// Why: run as three layers a Bottleneck moves 4.8 GB at batch 256 / 96x96 -- the 256-channel tensor is read by conv1,
// read again as the residual and written by conv3, and two 64-channel tensors are written and re-read in between --
// for 1.0 ms at 5.0-5.4 TB/s (12 % of the forward for 4 % of its FLOPs).  Fused, HBM sees x once (plus the halo
// overlap of neighbouring tiles, mostly served by L2) and y once.
//
// One 512-thread workgroup per CU, persistent over tiles of 8 x 8 output pixels.  LDS (156.6 KB):
//   x tile   2 x [32 planes][10 x 10 halo pixels][16 B]   double-buffered, LDS-DMA: also the residual of phase C
//   t1 tile  [8 planes][10 x 10][16 B]                    conv1 output on the halo (zero outside the image)
//   t2 tile  [8 planes][8 x 8][16 B]                      conv2 output
//   W1       [8 k-steps][4 blocks][4 k-groups][16 rows][16 B] = 32 KB (conv1's weights; conv2's and conv3's live in registers)
// Per tile, three phases separated by workgroup barriers:
//   A  conv1 on the 100 halo pixels (7 columns of 16): a wave owns 32 output channels and two columns;
//      bias + ReLU + 16-bit rounding -> t1 (the value the unfused path stores)
//   B  conv2 (3x3) on the 64 output pixels: a wave owns 32 output channels x one column (two tile rows), its
//      2 x 18 weight fragments in 144 VGPRs; bias + ReLU + rounding -> t2
//   C  conv3 on the 64 pixels: a wave owns 32 of the 256 output channels x all four columns; bias, lane exchange,
//      + residual (centre of the x tile), ReLU, 16-byte stores.
// Rounding points are exactly those of the unfused layers, so the oracle's storage model is unchanged.
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // dma16_buf, store16_buf, make_buf, BUF_OOB, u32x4

namespace scpose {

namespace {

constexpr int kT = 8;                          // output tile edge
constexpr int kHW = kT + 2, kHPix = kHW * kHW; // 10 x 10 halo pixels
constexpr int kXS = (kHPix | 1) * 16;          // bytes of one x / t1 plane (101 slots: odd pitch)
constexpr int kT2S = (kT * kT | 1) * 16;       // bytes of one t2 plane (65 slots)
constexpr int kXBuf = 32 * kXS;                // one x tile
constexpr int kW1Bytes = 8 * 4 * 4 * 16 * 16;  // 32 KB
constexpr int kLds = 2 * kXBuf + 8 * kXS + 8 * kT2S + kW1Bytes + (64 + 64 + 256) * 4;

struct BneckLaunch {
  const void* in;
  const void* w1;   // [8][4][4][16][8]
  const void* w2;   // [18][4][4][16][8]   (tap, plane) pairs t = 4 s + q: tap = t >> 3, plane = t & 7
  const void* w3;   // [2][16][4][16][8]
  const float* b1;  // MFMA row order
  const float* b2;
  const float* b3;
  void* out;
  uint32_t bytes;   // size of in (= out)
  int32_t N, H, W;
  int32_t tiles_x, tiles_y, tiles_total, tiles_per_wg, grid;
};

inline int bneck_row_channel(int row) {
  const int q = row >> 2, reg = row & 3;
  return (q & 1) * 8 + (q >> 1) * 4 + reg;
}

}  // namespace

template <int DT>
__global__ __launch_bounds__(512, 2) void bottleneck_kernel(const BneckLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  char* xl0 = smem;
  char* t1l = smem + 2 * kXBuf;
  char* t2l = t1l + 8 * kXS;
  char* w1l = t2l + 8 * kT2S;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int ch = wave & 1, wq = wave >> 1;   // phases A / B: output-channel half (blocks 2ch, 2ch + 1) and column group
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // ---- weights ----
  for (int o = tid * 16; o < kW1Bytes; o += 512 * 16)
    *reinterpret_cast<u32x4*>(w1l + o) = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.w1) + o);
  frag_t w2f[18][2], w3f[2][2];
#pragma unroll
  for (int s = 0; s < 18; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      w2f[s][mb] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w2) + ((((size_t)s * 4 + 2 * ch + mb) * 4 + q) * 16 + r) * 16);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      w3f[s][mb] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w3) + ((((size_t)s * 16 + 2 * wave + mb) * 4 + q) * 16 + r) * 16);
  // biases stay in LDS (behind W1) and are re-read per phase: 24 VGPRs fewer, which is what keeps conv2's weights in registers
  float* bl = reinterpret_cast<float*>(w1l + kW1Bytes);
  for (int e = tid; e < 64 + 64 + 256; e += 512) bl[e] = e < 64 ? p.b1[e] : e < 128 ? p.b2[e - 64] : p.b3[e - 128];
  // conv2 operand addressing: k-step s, k-group q -> (tap, plane) pair 4 s + q: tap = s >> 1 (compile time),
  // plane = q + 4 (s & 1): one per-lane offset (q * plane pitch), everything else an immediate
  const int qoff = q * kXS;

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(p.tiles_total, t_begin + p.tiles_per_wg);
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = t / tiles_per_img;
    const int rem = t - img * tiles_per_img;
    const int ty = rem / p.tiles_x;
    oy0 = ty * kT; ox0 = (rem - ty * p.tiles_x) * kT;
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.bytes), rs_out = make_buf(p.out, p.bytes);
  // LDS-DMA of the x halo tile of tile t into buffer b: 2 pieces of 64 slots per plane, 64 pieces; wave w takes planes 4w .. 4w + 3
  auto issue_x = [&](int t, int b) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    char* xl = xl0 + b * kXBuf;
#pragma unroll
    for (int piece = 0; piece < 2; ++piece) {
      const int slot = piece * 64 + lane;
      const int my = slot / kHW, mx = slot - my * kHW;
      const int iy = oy0 - 1 + my, ix = ox0 - 1 + mx;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const uint32_t voff = ok ? (uint32_t)(img * 32 * HW + iy * p.W + ix) * 16u : BUF_OOB;   // padding: read as zeros
      if (slot < kHPix) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int pl = 4 * wave + k;
          dma16_buf(rs_in, voff, (uint32_t)(pl * HW) * 16u, xl + pl * kXS + piece * 1024);
        }
      }
    }
  };

  if (t_begin < t_end) issue_x(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int buf = 0;
  for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    if (t + 1 < t_end) issue_x(t + 1, buf ^ 1);            // streams in under this tile's three phases
    const char* xl = xl0 + buf * kXBuf;

    // ---- A: conv1 (1x1, 256 -> 64) on the halo pixels -> t1 ----
    {
      f32x4 acc[2][2];                                     // accumulators start at the bias of their rows
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float4 bs = *reinterpret_cast<const float4*>(bl + (2 * ch + mb) * 16 + q * 4);
        acc[mb][0] = f32x4{bs.x, bs.y, bs.z, bs.w}; acc[mb][1] = acc[mb][0];
      }
      int pidx[2];
      bool live[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int col = wq + 4 * c;                        // columns wq and wq + 4 (7 columns: column 7 does not exist)
        pidx[c] = col * 16 + r;
        live[c] = pidx[c] < kHPix;
        if (!live[c]) pidx[c] = kHPix - 1;
      }
      // fragments one k-step ahead of the MFMAs that use them (the compiler waits for an LDS read at its first use)
      auto load_a = [&](int s, frag_t* a, frag_t* b) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const frag_t*>(w1l + ((((s * 4 + 2 * ch + mb) * 4 + q) * 16 + r) * 16));
#pragma unroll
        for (int c = 0; c < 2; ++c) b[c] = *reinterpret_cast<const frag_t*>(xl + (4 * s + q) * kXS + pidx[c] * 16);
      };
      frag_t fa[2][2], fb[2][2];
      load_a(0, fa[0], fb[0]);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) load_a(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[mb][c] = mfma16<T>(fa[s & 1][mb], fb[s & 1][c], acc[mb][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int my = pidx[c] / kHW, mx = pidx[c] - my * kHW;
        const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
        const bool inimg = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          uint2 o;
          o.x = relu2_16(pack2<T>(acc[mb][c][0], acc[mb][c][1]), 0u);
          o.y = relu2_16(pack2<T>(acc[mb][c][2], acc[mb][c][3]), 0u);
          if (!inimg) o = make_uint2(0u, 0u);              // conv2's zero padding, not a conv1 output
          if (live[c]) *reinterpret_cast<uint2*>(t1l + (2 * (2 * ch + mb) + psel) * kXS + pidx[c] * 16 + 8 * (q >> 1)) = o;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // t1 complete

    // ---- B: conv2 (3x3, 64 -> 64) on the 64 output pixels -> t2 ----
    {
      f32x4 acc[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float4 bs = *reinterpret_cast<const float4*>(bl + 64 + (2 * ch + mb) * 16 + q * 4);
        acc[mb] = f32x4{bs.x, bs.y, bs.z, bs.w};
      }
      const int pix = wq * 16 + r;                         // output pixel of this lane: rows 2 wq, 2 wq + 1
      const int py = pix >> 3, px = pix & 7;
      const char* bcol = t1l + (py * kHW + px) * 16;
      const char* bq0 = bcol + qoff;
      auto k2imm = [](int s) { const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky; return 4 * (s & 1) * kXS + (ky * kHW + kx) * 16; };
      frag_t bq[3];                                       // fragments two k-steps ahead
      bq[0] = *reinterpret_cast<const frag_t*>(bq0 + k2imm(0));
      bq[1] = *reinterpret_cast<const frag_t*>(bq0 + k2imm(1));
#pragma unroll
      for (int s = 0; s < 18; ++s) {
        if (s + 2 < 18) bq[(s + 2) % 3] = *reinterpret_cast<const frag_t*>(bq0 + k2imm(s + 2));
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma16<T>(w2f[s][0], bq[s % 3], acc[0]);
        acc[1] = mfma16<T>(w2f[s][1], bq[s % 3], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        uint2 o;
        o.x = relu2_16(pack2<T>(acc[mb][0], acc[mb][1]), 0u);
        o.y = relu2_16(pack2<T>(acc[mb][2], acc[mb][3]), 0u);
        *reinterpret_cast<uint2*>(t2l + (2 * (2 * ch + mb) + psel) * kT2S + pix * 16 + 8 * (q >> 1)) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // t2 complete

    // ---- C: conv3 (1x1, 64 -> 256) + residual -> y ----
    // column pairs (0,1), (2,3), one after the other (keeps the live accumulators at 2 x 2): after the lane exchange the
    // lower half-wave owns the pixel of the even column, the upper half-wave that of the odd one, each lane the 8 channels
    // of plane 2 * block + psel
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      f32x4 acc[2][2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float4 bs = *reinterpret_cast<const float4*>(bl + 128 + (2 * wave + mb) * 16 + q * 4);
        acc[mb][0] = f32x4{bs.x, bs.y, bs.z, bs.w}; acc[mb][1] = acc[mb][0];
      }
      frag_t bc[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) bc[s][c] = *reinterpret_cast<const frag_t*>(t2l + (4 * s + q) * kT2S + ((2 * cp + c) * 16 + r) * 16);
      const int pix = (2 * cp + half) * 16 + r, py = pix >> 3, px = pix & 7;
      u32x4 rv[2];                                         // residual: centre of the x tile
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) rv[mb] = *reinterpret_cast<const u32x4*>(xl + (2 * (2 * wave + mb) + psel) * kXS + ((py + 1) * kHW + px + 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[mb][c] = mfma16<T>(w3f[s][mb], bc[s][c], acc[mb][c]);
      const int oy = oy0 + py, ox = ox0 + px;
      const bool store_ok = oy < p.H && ox < p.W;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const f32x4 a0 = acc[mb][0], a1 = acc[mb][1];
        uint32_t a[4], b[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { a[jj] = __float_as_uint(a0[jj]); b[jj] = __float_as_uint(a1[jj]); }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
          a[jj] = sw[0]; b[jj] = sw[1];
        }
        const int plane = 2 * (2 * wave + mb) + psel;
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { v[jj] = __uint_as_float(a[jj]); v[4 + jj] = __uint_as_float(b[jj]); }
        v[0] += from_bits<T>(rv[mb][0] & 0xffff); v[1] += from_bits<T>(rv[mb][0] >> 16);
        v[2] += from_bits<T>(rv[mb][1] & 0xffff); v[3] += from_bits<T>(rv[mb][1] >> 16);
        v[4] += from_bits<T>(rv[mb][2] & 0xffff); v[5] += from_bits<T>(rv[mb][2] >> 16);
        v[6] += from_bits<T>(rv[mb][3] & 0xffff); v[7] += from_bits<T>(rv[mb][3] >> 16);
        u32x4 ov;
        ov[0] = relu2_16(pack2<T>(v[0], v[1]), 0u); ov[1] = relu2_16(pack2<T>(v[2], v[3]), 0u);
        ov[2] = relu2_16(pack2<T>(v[4], v[5]), 0u); ov[3] = relu2_16(pack2<T>(v[6], v[7]), 0u);
        const uint32_t voff = store_ok ? (uint32_t)((img * 32 + plane) * HW + oy * p.W + ox) * 16u : BUF_OOB;
        store16_buf(rs_out, voff, 0u, ov);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next x tile landed (this wave's planes); residual reads done
    __builtin_amdgcn_s_barrier();                                   // x buffer, t1 and t2 are free again
  }
}

// ---- host ----
bool bottleneck_fusable(int cin, int cmid, int cout) { return cin == 256 && cmid == 64 && cout == 256; }

// w1: [64][256] (1x1), w2: [64][64][3][3], w3: [256][64] (1x1), BN-folded f32 OIHW.
void bottleneck_pack(const float* w1, const float* w2, const float* w3, const float* b1, const float* b2, const float* b3, int dtype,
                     std::vector<uint16_t>* pw1, std::vector<uint16_t>* pw2, std::vector<uint16_t>* pw3, std::vector<float>* pb) {
  pw1->assign((size_t)8 * 4 * 4 * 16 * 8, 0);
  pw2->assign((size_t)18 * 4 * 4 * 16 * 8, 0);
  pw3->assign((size_t)2 * 16 * 4 * 16 * 8, 0);
  pb->assign(64 + 64 + 256, 0.f);
  for (int s = 0; s < 8; ++s)
    for (int m = 0; m < 4; ++m)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * m + bneck_row_channel(r), plane = 4 * s + q;
          uint16_t* d = pw1->data() + ((((size_t)s * 4 + m) * 4 + q) * 16 + r) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(w1[(size_t)co * 256 + plane * 8 + j], dtype);
        }
  for (int s = 0; s < 18; ++s)
    for (int m = 0; m < 4; ++m)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * m + bneck_row_channel(r);
          const int t = 4 * s + q, tap = t >> 3, plane = t & 7, ky = tap / 3, kx = tap % 3;
          uint16_t* d = pw2->data() + ((((size_t)s * 4 + m) * 4 + q) * 16 + r) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(w2[((size_t)(co * 64 + plane * 8 + j) * 3 + ky) * 3 + kx], dtype);
        }
  for (int s = 0; s < 2; ++s)
    for (int m = 0; m < 16; ++m)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * m + bneck_row_channel(r), plane = 4 * s + q;
          uint16_t* d = pw3->data() + ((((size_t)s * 16 + m) * 4 + q) * 16 + r) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(w3[(size_t)co * 64 + plane * 8 + j], dtype);
        }
  for (int pos = 0; pos < 64; ++pos) {
    const int co = (pos & ~15) + bneck_row_channel(pos & 15);
    (*pb)[pos] = b1[co]; (*pb)[64 + pos] = b2[co];
  }
  for (int pos = 0; pos < 256; ++pos) (*pb)[128 + pos] = b3[(pos & ~15) + bneck_row_channel(pos & 15)];
}

int32_t bottleneck_launch(const void* in, const void* w1, const void* w2, const void* w3, const float* bias, int N, int H, int W,
                          int dtype, void* out, hipStream_t stream) {
  // the kernel addresses its tensors through 32-bit buffer descriptors: batches whose 256-channel tensor reaches 4 GiB
  // (~900 frames at 96 x 96) run as several launches over frame ranges
  const size_t per_frame = (size_t)32 * H * W * 16;
  SCP_REQUIRE(per_frame < 0xfffffff0ull, "bottleneck: one %dx%d frame does not fit a 32-bit buffer descriptor", H, W);
  const int max_n = (int)(0xfffffff0ull / per_frame);
  static LdsOptIn big_b, big_f;
  for (int n0 = 0; n0 < N; n0 += max_n) {
    const int n = N - n0 < max_n ? N - n0 : max_n;
    BneckLaunch L{};
    L.in = static_cast<const char*>(in) + (size_t)n0 * per_frame;
    L.out = static_cast<char*>(out) + (size_t)n0 * per_frame;
    L.w1 = w1; L.w2 = w2; L.w3 = w3; L.b1 = bias; L.b2 = bias + 64; L.b3 = bias + 128;
    L.bytes = (uint32_t)((size_t)n * per_frame);
    L.N = n; L.H = H; L.W = W;
    L.tiles_x = (W + kT - 1) / kT; L.tiles_y = (H + kT - 1) / kT;
    L.tiles_total = n * L.tiles_x * L.tiles_y;
    int grid = conv_device_cus();
    if (grid > L.tiles_total) grid = L.tiles_total;
    L.tiles_per_wg = (L.tiles_total + grid - 1) / grid;
    L.grid = (L.tiles_total + L.tiles_per_wg - 1) / L.tiles_per_wg;
    if (dtype == SCPOSE_DT_BF16) {
      { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(bottleneck_kernel<0>), kLds, &big_b); if (rc != SCPOSE_OK) return rc; }
      hipLaunchKernelGGL(bottleneck_kernel<0>, dim3(L.grid), dim3(512), kLds, stream, L);
    } else {
      { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(bottleneck_kernel<1>), kLds, &big_f); if (rc != SCPOSE_OK) return rc; }
      hipLaunchKernelGGL(bottleneck_kernel<1>, dim3(L.grid), dim3(512), kLds, stream, L);
    }
    SCP_CHECK_HIP(hipGetLastError());
  }
  return SCPOSE_OK;
}

}  // namespace scpose
