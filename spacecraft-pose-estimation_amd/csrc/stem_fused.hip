// Fused stem: conv1 (3 -> 64, 3x3, stride 2) + bn1 + ReLU + conv2 (64 -> 64, 3x3, stride 2) + bn2 + ReLU in ONE kernel,
// both convolutions on v_mfma_f32_16x16x32_{bf16,f16}.
//
// Replaces pose_hrnet.py:282-288 / :426-431 of the reference (conv1, bn1, relu, conv2, bn2, relu) and, for the uint8
// input format, ToTensor() + Normalize(mean, std) of landmark_regression/tools/test.py:106-114.
//
// Why: run as two layers, the stem writes and re-reads a 64 x H/2 x W/2 tensor (1.2 GB at 384x384 / batch 256) that
// exists only between them: 0.75 ms for conv1 (f32 VALU, write-bound) + 1.48 ms for the stride-2 conv2 = 6.7 % of the
// forward for 0.7 % of its FLOPs.  Fused, HBM sees the input image (3 B/pixel) and the 64 x H/4 x W/4 result only.
//
// One 512-thread workgroup per CU, persistent over tiles of 8 x 16 output pixels (H/4 x W/4 map).  Per tile:
//   stage   the (35 x 67 x 3) input patch as normalised 16-bit values in LDS (uint8: through a 3 x 256 table of
//           ((u / 255 - mean) / std rounded to the 16-bit type; float32: rounded); pixels outside the image are 0;
//   conv1   on the 17 x 33 conv1 pixels the tile's conv2 taps touch (+21 % halo recompute): K = 27 padded to ONE
//           32-deep MFMA k-step.  The k-slots are laid out so that a lane's 8 operands are four aligned dwords of the
//           staged patch (k-group q < 3: patch row q, elements 0..7 of its 9; group 3: the 9th element of each row);
//           bias + ReLU + 16-bit rounding -- the rounding the unfused path applies when it stores conv1's output --
//           then 8-byte half-slot writes into the intermediate tile [8 planes][561 pixels][16 B] in LDS;
//   conv2   18 k-steps of (tap, plane) pairs from that tile (stride-2 reads: 32-byte lane pitch, conflict-free with
//           an odd plane pitch).  Each wave owns 32 of the 64 output channels and two of the tile's eight rows and
//           keeps its 2 x 18 weight fragments in registers (144 VGPRs): LDS holds no weights at all.
//   store   bias + ReLU, v_permlane32_swap pairs the half-waves' channel halves, 16-byte stores.
// Two workgroup barriers per tile; the next tile's input is fetched into registers under conv1 and committed to LDS
// under conv2.
#include "common.h"
#include "conv_device.h"

namespace scpose {

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

constexpr int kTH = 8, kTW = 16;                 // output tile
constexpr int kMH = 2 * kTH + 1, kMW = 2 * kTW + 1, kMPix = kMH * kMW;   // conv1 pixels per tile: 17 x 33 = 561
constexpr int kIH = 2 * kMH + 1, kIW = 2 * kMW + 1;                      // input pixels per tile: 35 x 67
constexpr int kIRow = 202;                       // 16-bit elements per staged input row (67 * 3 = 201, padded: rows stay dword-aligned)
constexpr int kMS = kMPix * 16;                  // bytes of one intermediate plane (561 slots: odd, see header)
constexpr int kLutBytes = 3 * 256 * 2;
constexpr int kInBytes = ((kIH * kIRow * 2 + 63) / 64) * 64;
constexpr int kMidBytes = 8 * kMS;
constexpr int kColsA = (kMPix + 15) / 16;        // 36 columns of 16 conv1 pixels
constexpr int kDwRow = 52;                       // aligned dwords that cover one 201-byte row of a uint8 patch
constexpr int kStageU8 = (kIH * kDwRow + 511) / 512;            // dword loads per thread (4)
constexpr int kStageF32 = (kIH * kIW * 3 + 511) / 512;          // float loads per thread (14)

struct StemFusedLaunch {
  const void* in;
  const void* w1;        // [4 blocks][4 k-groups][16 rows][8]
  const void* w2;        // [18 k-steps][4 blocks][4 k-groups][16 rows][8]
  const float* b1;       // [64] in MFMA row order
  const float* b2;
  const float* mean_std; // [6] (uint8 input)
  void* out;
  int32_t N, H, W, Ho, Wo;
  int32_t tiles_x, tiles_y, tiles_total, grid;
  uint32_t* sched;       // dynamic tile queue (conv_device.h: tile_claim)
  FastDiv fd_tiles_img, fd_tiles_x;   // the tile decode's divisions (three decodes per tile were ~300 scalar instructions of division)
};

// MFMA accumulator row (4 * (lane >> 4) + reg) -> channel within the 16-channel block (same map as conv_igemm.hip):
// lane groups q = 0, 2 (lanes l, l + 32) share plane 0 (channels 0-3 | 4-7), q = 1, 3 plane 1.
inline int stem_row_channel(int row) {
  const int q = row >> 2, reg = row & 3;
  return (q & 1) * 8 + (q >> 1) * 4 + reg;
}

template <typename T> __device__ __forceinline__ uint16_t bits16(float f) {
  T t = (T)f;
  return __builtin_bit_cast(uint16_t, t);
}

}  // namespace

template <int DT, int FMT>
__global__ __launch_bounds__(512, 2) void stem_fused_kernel(const StemFusedLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  uint16_t* lut = reinterpret_cast<uint16_t*>(smem);
  char* in16 = smem + kLutBytes;
  char* mid = in16 + kInBytes;
  int* tq = reinterpret_cast<int*>(mid + kMidBytes);   // tile queue words

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int H2 = p.H >> 1, W2 = p.W >> 1;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  if constexpr (FMT == SCPOSE_IN_U8_NHWC) {
    // ToTensor: u/255 ; Normalize: (t - mean)/std (torchvision's op order), rounded once to the 16-bit operand type
    for (int e = tid; e < 768; e += 512) {
      const int c = e >> 8;
      lut[e] = bits16<T>(((float)(e & 255) / 255.0f - p.mean_std[c]) / p.mean_std[3 + c]);
    }
  }
  for (int e = tid; e < kInBytes / 4; e += 512) reinterpret_cast<uint32_t*>(in16)[e] = 0u;   // padding elements stay finite

  // ---- weights: conv1 (4 blocks) and this wave's half of conv2 (2 blocks x 18 k-steps) live in registers ----
  const int ch = wave & 1;           // conv2: output-channel half (blocks 2ch, 2ch + 1)
  const int cpair = wave >> 1;       // conv2: tile rows 2 cpair, 2 cpair + 1
  frag_t w1f[4], w2f[18][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
    w1f[m] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w1) + ((m * 4 + q) * 16 + r) * 16);
#pragma unroll
  for (int s = 0; s < 18; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      w2f[s][mb] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w2) + (((s * 4 + 2 * ch + mb) * 4 + q) * 16 + r) * 16);
  float4 bs1[4];
  // conv2's biases wait in LDS (behind the tile queue words) for the tile's epilogue: eight registers that the k-loops need more --
  // with them in registers the kernel spilled two values whose reload (and its vmcnt(0)) sat at the head of conv1, right behind the
  // next tile's input loads: the prefetch was waited for before conv1 instead of flying under it
  float* b2l = reinterpret_cast<float*>(tq + 4);
  for (int e = tid; e < 64; e += 512) b2l[e] = p.b2[e];
#pragma unroll
  for (int m = 0; m < 4; ++m) bs1[m] = *reinterpret_cast<const float4*>(p.b1 + m * 16 + q * 4);

  // conv1 operand addressing: dword i of this lane's fragment sits at B0(pixel) + dl[i]
  int dl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dl[i] = q < 3 ? q * (kIRow * 2) + 4 * i : (i < 3 ? i : 2) * (kIRow * 2) + 16;
  // conv2 operand addressing: k-step s, k-group q -> (tap, plane) pair t = 4 s + q: plane = t & 7 = q + 4 (s & 1), tap = t >> 3 =
  // s >> 1 -- so a fragment address is one of TWO per-lane plane offsets plus a compile-time tap offset (an immediate of the
  // ds_read); eighteen per-lane offsets in registers spilled, and the reload's vmcnt(0) sat in front of conv1, behind the next
  // tile's input loads
  const int k2p0 = q * kMS, k2p1 = (4 + q) * kMS;
  auto k2t = [](int s) constexpr -> int { const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky; return (ky * kMW + kx) * 16; };

  // tiles come from the launch's dynamic queue (conv_device.h): this tile and the next are always known (the next patch is
  // fetched under this tile's conv1); the one after is claimed at the top of the tile and published through LDS in front of
  // the tile's last barrier
  if (tid == 0) { tq[0] = tile_claim(p.sched, p.tiles_total); tq[1] = tq[0] < 0 ? -1 : tile_claim(p.sched, p.tiles_total); }
  auto fdiv = [](int n, const FastDiv& f) -> int { return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n * f.add) >> f.shift); };
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = fdiv(rem, p.fd_tiles_x);
    oy0 = ty * kTH; ox0 = (rem - ty * p.tiles_x) * kTW;
  };

  // ---- input staging: fetch() reads the patch of tile t into registers, commit() writes it to LDS as 16-bit ----
  constexpr int NST = FMT == SCPOSE_IN_U8_NHWC ? kStageU8 : kStageF32;
  uint32_t sreg[NST];
  auto fetch = [&](int t) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    const int iy0 = 4 * oy0 - 3, ix0 = 4 * ox0 - 3;
    if constexpr (FMT == SCPOSE_IN_U8_NHWC) {
      const int bs = 3 * ix0;                              // byte offset of the patch inside an image row (may be -9)
      const int d0 = (bs - (bs < 0 ? 3 : 0)) / 4;          // floor(bs / 4)
      const int rowdw = (p.W * 3) >> 2;                    // W % 4 == 0: rows are whole dwords
      const uint32_t* base = static_cast<const uint32_t*>(p.in) + (size_t)img * p.H * rowdw;
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int e = tid + 512 * k;
        const int ty = e / kDwRow, dw = d0 + (e - ty * kDwRow), iy = iy0 + ty;
        const bool ok = e < kIH * kDwRow && iy >= 0 && iy < p.H && dw >= 0 && dw < rowdw;
        sreg[k] = ok ? base[(size_t)iy * rowdw + dw] : 0xffffffffu;
      }
    } else {
      const float* base = static_cast<const float*>(p.in) + (size_t)img * 3 * p.H * p.W;
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int e = tid + 512 * k;                       // (c, ty, tx), tx fastest
        const int c = e / (kIH * kIW), rem = e - c * (kIH * kIW), ty = rem / kIW, tx = rem - ty * kIW;
        const int iy = iy0 + ty, ix = ix0 + tx;
        const bool ok = e < 3 * kIH * kIW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        sreg[k] = ok ? __float_as_uint(base[((size_t)c * p.H + iy) * p.W + ix]) : 0u;
      }
    }
  };
  auto commit = [&](int t) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    const int iy0 = 4 * oy0 - 3, ix0 = 4 * ox0 - 3;
    uint16_t* dst = reinterpret_cast<uint16_t*>(in16);
    if constexpr (FMT == SCPOSE_IN_U8_NHWC) {
      const int bs = 3 * ix0;
      const int d0 = (bs - (bs < 0 ? 3 : 0)) / 4;
      const int rowdw = (p.W * 3) >> 2;
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int e = tid + 512 * k;
        if (e >= kIH * kDwRow) continue;
        const int ty = e / kDwRow, dw = d0 + (e - ty * kDwRow), iy = iy0 + ty;
        const bool ok = iy >= 0 && iy < p.H && dw >= 0 && dw < rowdw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = 4 * dw + i - bs;                    // element index inside the staged row
          if (b < 0 || b >= kIW * 3) continue;
          const int c = b % 3;
          const uint32_t byte = (sreg[k] >> (8 * i)) & 255u;
          dst[ty * kIRow + b] = ok ? lut[c * 256 + byte] : (uint16_t)0;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int e = tid + 512 * k;
        if (e >= 3 * kIH * kIW) continue;
        const int c = e / (kIH * kIW), rem = e - c * (kIH * kIW), ty = rem / kIW, tx = rem - ty * kIW;
        dst[ty * kIRow + tx * 3 + c] = bits16<T>(__uint_as_float(sreg[k]));   // out-of-image pixels were fetched as 0
      }
    }
  };

  __syncthreads();   // table, zeroed patch, first two tile ids
  int t = tq[0], t_next = tq[1];
  if (t >= 0) { fetch(t); commit(t); }
  __syncthreads();

  for (; t >= 0; t = t_next, t_next = tq[2]) {             // tq[2]: published before this tile's last barrier, rewritten after the next tile's first
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    const bool more = t_next >= 0;
    // the tile after the next: the counter's RAW value is kept and compared with the tile count only where it is published (in front
    // of the tile's last barrier).  Turned into a tile id on the spot -- tile_claim() -- the compiler waited for the returning atomic
    // three instructions after issuing it (vmcnt(0)): wave 0 stood still for the atomic's round trip at the top of every tile.
    uint32_t claim_raw = 0xffffffffu;
    if (tid == 0 && more) claim_raw = __hip_atomic_fetch_add(p.sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (more) fetch(t_next);                               // global loads fly under conv1

    // ---- conv1 -> intermediate tile ----
#pragma unroll 1
    for (int col = wave; col < kColsA; col += 8) {
      const int pidx = col * 16 + r;
      const bool live = pidx < kMPix;
      const int pc = live ? pidx : kMPix - 1;
      const int my = pc / kMW, mx = pc - my * kMW;
      // staged row 2 my starts at byte 2 my * (2 kIRow); pixel column 2 mx at element 6 mx
      const char* b0 = in16 + my * (4 * kIRow) + mx * 12;
      uint32_t bw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bw[i] = *reinterpret_cast<const uint32_t*>(b0 + dl[i]);
      const frag_t bf = __builtin_bit_cast(frag_t, u32x4_t{bw[0], bw[1], bw[2], bw[3]});
      const int gy = 2 * oy0 - 1 + my, gx = 2 * ox0 - 1 + mx;
      const bool inimg = gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = mfma16<T>(w1f[m], bf, acc);
        const float4 bs = bs1[m];
        uint2 o;
        o.x = relu2_16(pack2<T>(acc[0] + bs.x, acc[1] + bs.y), 0u);
        o.y = relu2_16(pack2<T>(acc[2] + bs.z, acc[3] + bs.w), 0u);
        if (!inimg) o = make_uint2(0u, 0u);                // conv2's zero padding, not a conv1 output
        if (live) *reinterpret_cast<uint2*>(mid + (2 * m + psel) * kMS + pidx * 16 + 8 * (q >> 1)) = o;
      }
    }
    __syncthreads();                                       // intermediate tile complete; the staged patch is dead
    if (more) commit(t_next);                              // ... so the next patch goes in under conv2

    // ---- conv2 -> output ----
    f32x4 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) { acc[mb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const char* bcol0 = mid + ((2 * (2 * cpair)) * kMW + 2 * r) * 16;
    const char* bcol1 = bcol0 + 2 * kMW * 16;
    const char* const bc[2][2] = {{bcol0 + k2p0, bcol0 + k2p1}, {bcol1 + k2p0, bcol1 + k2p1}};   // [column][k-step parity]
    frag_t bq[3][2];                                       // fragments two k-steps ahead (see conv_s2r.hip)
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0) {
      bq[s0][0] = *reinterpret_cast<const frag_t*>(bc[0][s0 & 1] + k2t(s0));
      bq[s0][1] = *reinterpret_cast<const frag_t*>(bc[1][s0 & 1] + k2t(s0));
    }
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (s + 2 < 18) {
        bq[(s + 2) % 3][0] = *reinterpret_cast<const frag_t*>(bc[0][s & 1] + k2t(s + 2));
        bq[(s + 2) % 3][1] = *reinterpret_cast<const frag_t*>(bc[1][s & 1] + k2t(s + 2));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        acc[mb][0] = mfma16<T>(w2f[s][mb], bq[s % 3][0], acc[mb][0]);
        acc[mb][1] = mfma16<T>(w2f[s][mb], bq[s % 3][1], acc[mb][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // lower half-wave: column 0's pixel, upper half-wave: column 1's; each lane ends up with the 8 channels of plane
    // 2 * block + psel of its pixel
    const int oy = oy0 + 2 * cpair + half, ox = ox0 + r;
    const bool store_ok = oy < p.Ho && ox < p.Wo;
    const size_t plane_sz = (size_t)p.Ho * p.Wo;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float4 bs = *reinterpret_cast<const float4*>(b2l + (2 * ch + mb) * 16 + q * 4);
      uint32_t a[4], b[4];
      a[0] = __float_as_uint(acc[mb][0][0] + bs.x); a[1] = __float_as_uint(acc[mb][0][1] + bs.y);
      a[2] = __float_as_uint(acc[mb][0][2] + bs.z); a[3] = __float_as_uint(acc[mb][0][3] + bs.w);
      b[0] = __float_as_uint(acc[mb][1][0] + bs.x); b[1] = __float_as_uint(acc[mb][1][1] + bs.y);
      b[2] = __float_as_uint(acc[mb][1][2] + bs.z); b[3] = __float_as_uint(acc[mb][1][3] + bs.w);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
        a[jj] = sw[0]; b[jj] = sw[1];
      }
      u32x4_t ov;
      ov[0] = relu2_16(pack2<T>(__uint_as_float(a[0]), __uint_as_float(a[1])), 0u);
      ov[1] = relu2_16(pack2<T>(__uint_as_float(a[2]), __uint_as_float(a[3])), 0u);
      ov[2] = relu2_16(pack2<T>(__uint_as_float(b[0]), __uint_as_float(b[1])), 0u);
      ov[3] = relu2_16(pack2<T>(__uint_as_float(b[2]), __uint_as_float(b[3])), 0u);
      if (store_ok) {
        const int plane = 2 * (2 * ch + mb) + psel;
        *reinterpret_cast<u32x4_t*>(static_cast<char*>(p.out) + (((size_t)img * 8 + plane) * plane_sz + (size_t)oy * p.Wo + ox) * 16) = ov;
      }
    }
    if (tid == 0) tq[2] = claim_raw < (uint32_t)p.tiles_total ? (int)claim_raw : -1;
    __syncthreads();                                       // intermediate tile free again; next patch committed; tq[2] published
  }
  if (tid == 0) tile_retire(p.sched);
}

size_t stem_fused_lds_bytes() { return (size_t)kLutBytes + kInBytes + kMidBytes + 16 + 64 * 4; }

// Host: pack the two BN-folded convolutions (OIHW f32) for the kernel.  w1: [64][3][3][3], w2: [64][64][3][3].
void stem_fused_pack(const float* w1, const float* b1, const float* w2, const float* b2, int dtype, std::vector<uint16_t>* pw1,
                     std::vector<uint16_t>* pw2, std::vector<float>* pb1, std::vector<float>* pb2) {
  pw1->assign((size_t)4 * 4 * 16 * 8, 0);
  pw2->assign((size_t)18 * 4 * 4 * 16 * 8, 0);
  pb1->assign(64, 0.f); pb2->assign(64, 0.f);
  for (int m = 0; m < 4; ++m)
    for (int q = 0; q < 4; ++q)
      for (int r = 0; r < 16; ++r) {
        const int co = 16 * m + stem_row_channel(r);
        uint16_t* d = pw1->data() + ((size_t)(m * 4 + q) * 16 + r) * 8;
        for (int j = 0; j < 8; ++j) {
          int ky = -1, e = 0;
          if (q < 3) { ky = q; e = j; }
          else if (j == 0 || j == 2 || j == 4) { ky = j / 2; e = 8; }
          if (ky < 0) continue;
          const int kx = e / 3, c = e % 3;
          d[j] = host_f32_to_16(w1[((size_t)(co * 3 + c) * 3 + ky) * 3 + kx], dtype);
        }
      }
  for (int s = 0; s < 18; ++s)
    for (int m = 0; m < 4; ++m)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * m + stem_row_channel(r);
          const int t = 4 * s + q, tap = t >> 3, plane = t & 7, ky = tap / 3, kx = tap % 3;
          uint16_t* d = pw2->data() + ((((size_t)s * 4 + m) * 4 + q) * 16 + r) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(w2[((size_t)(co * 64 + plane * 8 + j) * 3 + ky) * 3 + kx], dtype);
        }
  for (int pos = 0; pos < 64; ++pos) {
    const int co = (pos & ~15) + stem_row_channel(pos & 15);
    (*pb1)[pos] = b1[co]; (*pb2)[pos] = b2[co];
  }
}

int32_t stem_fused_launch(const void* in, int in_fmt, const void* w1, const void* w2, const float* b1, const float* b2,
                          const float* mean_std, int N, int H, int W, int dtype, void* out, uint32_t* sched, hipStream_t stream) {
  SCP_REQUIRE(sched, "stem: null tile-queue words");
  SCP_REQUIRE(H % 32 == 0 && W % 32 == 0, "stem: H=%d W=%d must be multiples of 32", H, W);
  SCP_REQUIRE(in_fmt == SCPOSE_IN_F32_NCHW || in_fmt == SCPOSE_IN_U8_NHWC, "stem: input format %d", in_fmt);
  SCP_REQUIRE(in_fmt == SCPOSE_IN_F32_NCHW || mean_std, "stem: u8 input needs mean/std");
  SCP_REQUIRE(((size_t)in & 3) == 0, "stem: input pointer must be 4-byte aligned");
  StemFusedLaunch L{};
  L.in = in; L.w1 = w1; L.w2 = w2; L.b1 = b1; L.b2 = b2; L.mean_std = mean_std; L.out = out;
  L.N = N; L.H = H; L.W = W; L.Ho = H / 4; L.Wo = W / 4;
  L.tiles_x = (L.Wo + kTW - 1) / kTW; L.tiles_y = (L.Ho + kTH - 1) / kTH;
  L.tiles_total = N * L.tiles_x * L.tiles_y;
  L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y); L.fd_tiles_x = make_fastdiv(L.tiles_x);
  L.grid = conv_device_cus() < L.tiles_total ? conv_device_cus() : L.tiles_total;
  L.sched = sched;
  const size_t lds = stem_fused_lds_bytes();
#define STEM_LAUNCH(DTV, FMTV)                                                                                     \
  do {                                                                                                             \
    static LdsOptIn big;                                                                                           \
    const int32_t rc_ = lds_opt_in(reinterpret_cast<const void*>(stem_fused_kernel<DTV, FMTV>), (int)lds, &big);   \
    if (rc_ != SCPOSE_OK) return rc_;                                                                              \
    hipLaunchKernelGGL((stem_fused_kernel<DTV, FMTV>), dim3(L.grid), dim3(512), lds, stream, L);                   \
  } while (0)
  const bool bf = dtype == SCPOSE_DT_BF16;
  if (in_fmt == SCPOSE_IN_U8_NHWC) { if (bf) STEM_LAUNCH(0, SCPOSE_IN_U8_NHWC); else STEM_LAUNCH(1, SCPOSE_IN_U8_NHWC); }
  else { if (bf) STEM_LAUNCH(0, SCPOSE_IN_F32_NCHW); else STEM_LAUNCH(1, SCPOSE_IN_F32_NCHW); }
#undef STEM_LAUNCH
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
