// One pass over branch 0 of a HighResolutionModule's fuse stage (landmark_regression/lib/models/pose_hrnet.py:254-265):
//   * fuse row 0:  y0 = ReLU(x0 + up2(t1) + up4(t2) + up8(t3)), t_j = the 1x1-conv + BN outputs of the coarser branches
//     (:199-210; nn.Upsample 'nearest' read as (y >> s, x >> s), never materialised), summed in fp32 in the reference's j
//     order, one 16-bit rounding -- bit-identical to fuse_sum_kernel (elementwise.hip);
//   * the FIRST hop of every down path that starts at branch 0 (:211-239): 3x3 stride-2 conv + BN (+ ReLU when it is not the
//     row's last hop) for rows 1, 2, 3 -- C -> 2C (row 1, its only hop), C -> C, C -> C -- as ONE convolution C -> nb * C with
//     per-group output tensors, on v_mfma_f32_16x16x32 with the weights in registers exactly like conv_s2r_kernel (same
//     k-steps, same order: bit-identical to the separate launches).
// Why: branch 0 is the largest tensor of the module (226 MB at batch 256 for W48 / 384^2) and the unfused schedule read it
// once per consumer -- four times in stage 4 (fuse_sum + three stride-2 chains), three times in stage 3: 0.45 / 0.68 GB of
// HBM reads per module that bought nothing, in launches that are all HBM-bound.  Here its tile is staged in LDS once (the
// stride-2 convolutions' 17 x 33 input tile, by LDS-DMA, double-buffered) and everything that reads branch 0 is computed
// from that copy: the tile's own 16 x 32 input pixels are also the pixels of y0 it owns.
// The low-resolution terms of the tile (8 x 16, 4 x 8, 2 x 4 pixels per plane: 16 KB) ride in by LDS-DMA with the next input
// tile, so the fuse epilogue reads nothing but LDS.
//
// One NW * 64-thread workgroup per CU, persistent over tiles of 8 x 16 stride-2 output pixels; waves split G ways over the
// output-channel groups (16 * NBLK channels each: one group = C channels) and NW / G ways over the tile's rows.  G = nb
// (the module's branch count): 2, 3 (six waves) or 4.
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // dma16_buf, store16_buf, make_buf, BUF_OOB, u32x4

namespace scpose {

namespace {

constexpr int kTH = 8, kTW = 16;                    // stride-2 output tile
constexpr int kMH = 2 * (kTH - 1) + 3, kMW = 2 * (kTW - 1) + 3, kMPix = kMH * kMW;   // 17 x 33 staged input pixels
constexpr int kMS = (kMPix | 1) * 16;               // bytes of one staged plane (561 slots: odd pitch, see conv_s2r.hip)
constexpr int kCH = 2 * kTH, kCW = 2 * kTW;         // the tile's own input pixels: rows 1..16, columns 1..32 of the staged tile
constexpr int fd_term_slots(int planes, int s) { return ((planes * ((kCH >> s) * (kCW >> s)) + 63) / 64) * 64; }   // whole 64-slot DMA pieces
constexpr int fd_term_bytes(int planes) { return (fd_term_slots(planes, 1) + fd_term_slots(planes, 2) + fd_term_slots(planes, 3)) * 16; }

struct FdLaunch {
  const void* in;        // x0: [N][PLANES][H][W][8]
  const void* w;         // [k-step][G * NBLK cout blocks][4 k-groups][16 rows][8]  (conv_s2r_pack of the concatenated convolutions)
  const float* bias;     // MFMA row order, G * NBLK * 16
  void* out[4];          // per output-channel group: tensor, its size, its plane count, the group's first plane in it, ReLU
  uint32_t out_bytes[4];
  int32_t out_planes[4], plane0[4], relu[4];
  const void* term[3];   // low-resolution terms of fuse row 0, in the reference's j order
  uint32_t term_bytes[3];
  int32_t shift[3], nlow;
  void* y;               // fuse row 0's output (shape of x0)
  uint32_t in_bytes;
  int32_t N, H, W, Ho, Wo;
  int32_t tiles_x, tiles_y, tiles_total, tiles_per_wg, grid;
  FastDiv fd_tiles_img, fd_tiles_x;   // tile decode by exact multiply-shift
};

}  // namespace

template <int DT, int PLANES, int NBLK, int G, int NW>
__global__ __launch_bounds__(NW * 64, 2) void fuse_down_kernel(const FdLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int NPT = 9 * PLANES;              // (tap, plane) pairs
  constexpr int KS = (NPT + 3) / 4;            // 32-deep k-steps
  constexpr int RSETS = NW / G, NCOL = kTH / RSETS;   // row sets, tile rows per wave
  static_assert(NW % G == 0 && kTH % RSETS == 0 && NCOL % 2 == 0, "fuse_down: wave split");
  constexpr int XB = PLANES * kMS;             // bytes of one input-tile buffer
  constexpr int TB = fd_term_bytes(PLANES);    // bytes of one term-tile buffer
  constexpr int NSLOT = (kMPix + 63) / 64;     // 64-pixel DMA pieces per plane
  constexpr int NT = NW * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int g = wave % G, rset = wave / G;     // output-channel group, row set
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // ---- this wave's weights: NBLK blocks x KS k-steps, resident for the workgroup's whole life ----
  frag_t wf[KS][NBLK];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb)
      wf[s][mb] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w) + ((((size_t)s * (NBLK * G) + g * NBLK + mb) * 4 + q) * 16 + r) * 16);
  // the biases wait in LDS (behind the tile buffers) for the epilogues: 12 registers the k-loops need more
  float* bias_l = reinterpret_cast<float*>(smem + 2 * XB + 2 * TB);
  for (int e = tid; e < G * NBLK * 16; e += NT) bias_l[e] = p.bias[e];
  int koff[KS];   // k-step s, k-group q -> pair t = 4 s + q = (tap, plane); padding pairs read slot 0 (finite, weight 0)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int t = 4 * s + q, tap = t / PLANES, plane = t - tap * PLANES, ky = tap / 3, kx = tap - 3 * ky;
    koff[s] = t < NPT ? plane * kMS + (ky * kMW + kx) * 16 : 0;
  }

  // term k's tile in a term buffer: [PLANES][16 >> s][32 >> s] slots, padded to whole 64-slot pieces
  int tbase[3], tpieces[3];
  {
    int off = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int s = k < p.nlow ? p.shift[k] : 3;
      tpieces[k] = k < p.nlow ? (PLANES * (512 >> (2 * s)) + 63) >> 6 : 0;
      tbase[k] = off;
      off += tpieces[k] * 1024;
    }
  }

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(p.tiles_total, t_begin + p.tiles_per_wg);
  auto fdiv = [](int n, const FastDiv& f) -> int { return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n * f.add) >> f.shift); };
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = fdiv(rem, p.fd_tiles_x);
    oy0 = ty * kTH; ox0 = (rem - ty * p.tiles_x) * kTW;
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.in_bytes), rs_y = make_buf(p.y, p.in_bytes);
  const buf_rsrc_t rs_out = make_buf(p.out[g], p.out_bytes[g]);
  // LDS-DMA of tile t into buffer b: the input tile (piece = 64 pixel slots of one plane; wave w takes pieces w, w + NW, ...)
  // and the tile's share of every low-resolution term (pieces dealt round-robin over the waves, across the terms)
  auto issue_tile = [&](int t, int b) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    char* xl = smem + b * XB;
#pragma unroll 1
    for (int piece = wave; piece < NSLOT; piece += NW) {
      const int slot = piece * 64 + lane;
      const int my = slot / kMW, mx = slot - my * kMW;
      const int iy = 2 * oy0 - 1 + my, ix = 2 * ox0 - 1 + mx;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const uint32_t voff = ok ? (uint32_t)(img * PLANES * HW + iy * p.W + ix) * 16u : BUF_OOB;   // padding: read as zeros
      if (slot < kMPix) {   // lanes past the plane's last slot stay inactive: their LDS write would land in the next plane
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) dma16_buf(rs_in, voff, (uint32_t)(pl * HW) * 16u, xl + pl * kMS + piece * 1024);
      }
    }
    char* tl = smem + 2 * XB + b * TB;
    int dealt = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < p.nlow) {
        const int s = p.shift[k];
        const int Hk = p.H >> s, Wk = p.W >> s, lgc = 5 - s, per_plane = 512 >> (2 * s), total = PLANES * per_plane;
        const buf_rsrc_t rs_t = make_buf(p.term[k], p.term_bytes[k]);
        int first = (wave - dealt) % NW;
        if (first < 0) first += NW;
#pragma unroll 1
        for (int piece = first; piece < tpieces[k]; piece += NW) {
          const int slot = piece * 64 + lane;
          const int pl = slot >> (9 - 2 * s), rem = slot & (per_plane - 1), ly = rem >> lgc, lx = rem & ((1 << lgc) - 1);
          const int gy = ((2 * oy0) >> s) + ly, gx = ((2 * ox0) >> s) + lx;
          const bool ok = gy < Hk && gx < Wk;
          const uint32_t voff = ok ? (uint32_t)(((img * PLANES + pl) * Hk + gy) * Wk + gx) * 16u : BUF_OOB;
          if (slot < total) dma16_buf(rs_t, voff, 0u, tl + tbase[k] + piece * 1024);
        }
        dealt += tpieces[k];
      }
    }
  };

  if (t_begin < t_end) issue_tile(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const size_t plane_sz = (size_t)p.Ho * p.Wo;
  const uint32_t relu_floor = p.relu[g] ? 0u : 0x80008000u;
  const int oplanes = p.out_planes[g], oplane0 = p.plane0[g];
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    if (t + 1 < t_end) issue_tile(t + 1, buf ^ 1);          // streams in under this tile's work
    const char* xl = smem + buf * XB;

    // ---- the stride-2 convolutions: this wave's group, tile rows rset * NCOL + c, two at a time ----
#pragma unroll 1
    for (int c0 = 0; c0 < NCOL; c0 += 2) {
      f32x4 acc[NBLK][2];
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb) { acc[mb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mb][1] = acc[mb][0]; }
      const int py0 = rset * NCOL + c0;
      const char* bcol = xl + ((2 * py0) * kMW + 2 * r) * 16;
      frag_t bf[3][2];   // fragments two k-steps ahead of the MFMAs that use them; the scheduling barriers pin that order
#pragma unroll
      for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[s0][c] = *reinterpret_cast<const frag_t*>(bcol + c * (2 * kMW * 16) + koff[s0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 2 < KS) {
#pragma unroll
          for (int c = 0; c < 2; ++c) bf[(s + 2) % 3][c] = *reinterpret_cast<const frag_t*>(bcol + c * (2 * kMW * 16) + koff[s + 2]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < NBLK; ++mb)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[mb][c] = mfma16<T>(wf[s][mb], bf[s % 3][c], acc[mb][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: v_permlane32_swap gives the lower half-wave the 8 channels (plane 2 * block + psel) of row py0's pixel and the
      // upper half-wave those of row py0 + 1's
      const int oy = oy0 + py0 + half, ox = ox0 + r;
      const bool store_ok = oy < p.Ho && ox < p.Wo;
      // (asm reads: the compiler puts `s_waitcnt vmcnt(0)` in front of an ordinary LDS load it cannot tell from the LDS-DMA
      // destination, which would drain the next tile's DMA -- and every store still in flight -- in front of each epilogue)
      f32x4 bsv[NBLK];
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb) lds_read16<0>(bsv[mb], (uint32_t)(size_t)(bias_l + (g * NBLK + mb) * 16 + q * 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb) lds_landed(bsv[mb]);
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb) {
        const float4 b4 = make_float4(bsv[mb][0], bsv[mb][1], bsv[mb][2], bsv[mb][3]);
        uint32_t a[4], b[4];
        a[0] = __float_as_uint(acc[mb][0][0] + b4.x); a[1] = __float_as_uint(acc[mb][0][1] + b4.y);
        a[2] = __float_as_uint(acc[mb][0][2] + b4.z); a[3] = __float_as_uint(acc[mb][0][3] + b4.w);
        b[0] = __float_as_uint(acc[mb][1][0] + b4.x); b[1] = __float_as_uint(acc[mb][1][1] + b4.y);
        b[2] = __float_as_uint(acc[mb][1][2] + b4.z); b[3] = __float_as_uint(acc[mb][1][3] + b4.w);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
          a[jj] = sw[0]; b[jj] = sw[1];
        }
        u32x4 ov;
        ov[0] = relu2_16(pack2<T>(__uint_as_float(a[0]), __uint_as_float(a[1])), relu_floor);
        ov[1] = relu2_16(pack2<T>(__uint_as_float(a[2]), __uint_as_float(a[3])), relu_floor);
        ov[2] = relu2_16(pack2<T>(__uint_as_float(b[0]), __uint_as_float(b[1])), relu_floor);
        ov[3] = relu2_16(pack2<T>(__uint_as_float(b[2]), __uint_as_float(b[3])), relu_floor);
        const int plane = oplane0 + 2 * mb + psel;
        const uint32_t voff = store_ok ? (uint32_t)(((size_t)img * oplanes + plane) * plane_sz + (size_t)oy * p.Wo + ox) * 16u : BUF_OOB;
        store16_buf(rs_out, voff, 0u, ov);
      }
    }

    // ---- fuse row 0 on the tile's own 16 x 32 input pixels: every operand is in LDS ----
    // A thread takes one (plane, column) of TWO vertically adjacent pixels: they share every low-resolution term, which is read and
    // unpacked once; lanes run along x, so both stores are whole rows of 16-byte vectors.  The phase is VALU-issue-bound (two
    // waves per SIMD): packed fp32 adds, one v_cvt_pk per output dword.  Sum order per pixel: ((x0 + t1) + t2) + t3, as
    // fuse_sum_kernel forms it (its leading 0 + x0 is exact: x0 is a ReLU output, never -0).
    {
      const char* tl = smem + 2 * XB + buf * TB;
      constexpr int NV = PLANES * (kCH / 2) * kCW, NIT = (NV + NT - 1) / NT;
      auto unpack = [](uint32_t w) -> f32x2 { return f32x2{from_bits<T>((uint16_t)(w & 0xffff)), from_bits<T>((uint16_t)(w >> 16))}; };
#pragma unroll 1
      for (int i = 0; i < NIT; ++i) {
        const int v0 = i * NT + tid;
        const bool mine = NV % NT == 0 || v0 < NV;         // (threads past the last item redo item 0 and drop the stores)
        const int v = mine ? v0 : 0;
        const int pl = v >> 8, y = (v >> 4) & 14, x = v & 31;   // v = (plane, row pair, column)
        const char* xp = xl + pl * kMS + ((y + 1) * kMW + x + 1) * 16;
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xp), xb = *reinterpret_cast<const u32x4*>(xp + kMW * 16);
        f32x2 sa[4], sb[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { sa[d] = unpack(xa[d]); sb[d] = unpack(xb[d]); }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k < p.nlow) {
            const int sh = p.shift[k];
            const u32x4 tv = *reinterpret_cast<const u32x4*>(tl + tbase[k] + ((((pl << (4 - sh)) + (y >> sh)) << (5 - sh)) + (x >> sh)) * 16);
#pragma unroll
            for (int d = 0; d < 4; ++d) { const f32x2 t2 = unpack(tv[d]); sa[d] += t2; sb[d] += t2; }
          }
        }
        u32x4 oa, ob;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          oa[d] = pack2<T>(fmaxf(sa[d][0], 0.f), fmaxf(sa[d][1], 0.f));
          ob[d] = pack2<T>(fmaxf(sb[d][0], 0.f), fmaxf(sb[d][1], 0.f));
        }
        const int gy = 2 * oy0 + y, gx = 2 * ox0 + x;
        const uint32_t va = (mine && gy < p.H && gx < p.W) ? (uint32_t)((img * PLANES + pl) * HW + gy * p.W + gx) * 16u : BUF_OOB;
        const uint32_t vb = (va != BUF_OOB && gy + 1 < p.H) ? va + (uint32_t)p.W * 16u : BUF_OOB;
        store16_buf(rs_y, va, 0u, oa);
        store16_buf(rs_y, vb, 0u, ob);
      }
    }
    // next tile landed (this wave's pieces).  vmcnt(0) also waits for this tile's stores; a counted wait that leaves them in
    // flight (the DMA was issued first) measured the same here and 1-2 % SLOWER in the fused BasicBlock and the producer/consumer
    // kernels (round 4), and round 2 saw sporadic wrong results with it in the stride-2 kernels
    // (one asm statement with a memory clobber: the builtin barrier is IntrNoMem to the compiler, which may then hoist the next
    // tile's ordinary LDS loads above it, into a buffer other waves' LDS-DMA is still filling -- conv_block2_kernel.h)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                          // ... and everybody is done with this buffer (LDS reads are consumed)
  }
}

// ---- host ----
// nb = branches of the module (2..4), c0 / c1 = channels of branches 0 / 1
bool fuse_down_supported(int nb, int c0, int c1) {
  return (c0 == 32 || c0 == 48) && c1 == 2 * c0 && nb >= 2 && nb <= 4;
}

// w / bias: the nb - 1 folded convolutions [c1 | c0 | c0][c0][3][3] concatenated along the output channels (row 1's first)
int32_t fuse_down_upload(const float* w, const float* bias, int nb, int c0, int dtype, FuseDownPacked* fd) {
  const int cout = nb * c0;
  fd->c0 = c0; fd->nb = nb; fd->dtype = dtype;
  const size_t wb = conv_s2r_pack(nullptr, cout, c0, dtype, nullptr);
  std::vector<uint16_t> pw(wb / 2);
  conv_s2r_pack(w, cout, c0, dtype, pw.data());
  std::vector<float> pb(cout);
  conv_s2r_pack_bias(bias, cout, pb.data());
  SCP_CHECK_HIP(hipMalloc(&fd->d_w, wb));
  SCP_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&fd->d_b), pb.size() * 4));
  SCP_CHECK_HIP(hipMemcpy(fd->d_w, pw.data(), wb, hipMemcpyHostToDevice));
  SCP_CHECK_HIP(hipMemcpy(fd->d_b, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
  return SCPOSE_OK;
}

void fuse_down_free(FuseDownPacked* fd) {
  if (fd->d_w) (void)hipFree(fd->d_w);
  if (fd->d_b) (void)hipFree(fd->d_b);
  fd->d_w = nullptr; fd->d_b = nullptr;
}

template <int DT, int PLANES, int NBLK, int G, int NW>
static int32_t fd_launch_one(const FdLaunch& L, hipStream_t st) {
  auto kern = fuse_down_kernel<DT, PLANES, NBLK, G, NW>;
  constexpr int lds = 2 * PLANES * kMS + 2 * fd_term_bytes(PLANES) + G * NBLK * 16 * 4;
  static LdsOptIn big;
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), lds, &big); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(NW * 64), lds, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

template <int DT>
static int32_t fd_dispatch(int c0, int nb, const FdLaunch& L, hipStream_t st) {
  if (c0 == 48) {
    if (nb == 2) return fd_launch_one<DT, 6, 3, 2, 8>(L, st);
    if (nb == 3) return fd_launch_one<DT, 6, 3, 3, 6>(L, st);
    return fd_launch_one<DT, 6, 3, 4, 8>(L, st);
  }
  if (nb == 2) return fd_launch_one<DT, 4, 2, 2, 8>(L, st);
  if (nb == 3) return fd_launch_one<DT, 4, 2, 3, 6>(L, st);
  return fd_launch_one<DT, 4, 2, 4, 8>(L, st);
}

// x0: [N][c0 / 8][H][W][8]; terms[k]: branch k + 1's 1x1-conv output at (H >> (k + 1)) x (W >> (k + 1)), c0 channels;
// y: fuse row 0's output; outs[0]: row 1's down-path result (2 c0 channels), outs[1], outs[2]: the first hops of rows 2 and 3
// (c0 channels, ReLU), all at (H / 2) x (W / 2).
int32_t fuse_down_launch(const FuseDownPacked& fd, const void* x0, int N, int H, int W, const void* const* terms, void* y,
                         void* const* outs, hipStream_t stream) {
  const int nb = fd.nb, c0 = fd.c0, planes = c0 / 8;
  SCP_REQUIRE(fuse_down_supported(nb, c0, 2 * c0) && fd.d_w, "fuse_down: %d branches of %d channels not eligible", nb, c0);
  SCP_REQUIRE(H % 8 == 0 && W % 8 == 0, "fuse_down: branch 0 is %dx%d (multiples of 8: the coarsest term is 1/8 of it)", H, W);
  const int Ho = H / 2, Wo = W / 2;
  const size_t in_frame = (size_t)planes * H * W * 16;
  SCP_REQUIRE(in_frame < 0xfffffff0ull, "fuse_down: one %dx%d frame does not fit a 32-bit buffer descriptor", H, W);
  // tensors are addressed through 32-bit buffer descriptors: batches whose branch-0 tensor reaches 4 GiB run as several launches
  int max_n = (int)(0xfffffff0ull / in_frame);
  { static const char* e = dev_env("SCPOSE_FD_MAXN"); if (e && atoi(e) > 0 && atoi(e) < max_n) max_n = atoi(e); }   // tests: force the split on a small batch
  for (int n0 = 0; n0 < N; n0 += max_n) {
    const int n = N - n0 < max_n ? N - n0 : max_n;
    FdLaunch L{};
    L.in = static_cast<const char*>(x0) + (size_t)n0 * in_frame;
    L.y = static_cast<char*>(y) + (size_t)n0 * in_frame;
    L.in_bytes = (uint32_t)((size_t)n * in_frame);
    L.w = fd.d_w; L.bias = fd.d_b;
    for (int gidx = 0; gidx < nb; ++gidx) {
      const int o = gidx < 2 ? 0 : gidx - 1;            // groups 0, 1: the two halves of row 1's 2 c0 channels
      const int op = (o == 0 ? 2 : 1) * planes;
      const size_t out_frame = (size_t)op * Ho * Wo * 16;
      L.out[gidx] = static_cast<char*>(outs[o]) + (size_t)n0 * out_frame;
      L.out_bytes[gidx] = (uint32_t)((size_t)n * out_frame);
      L.out_planes[gidx] = op;
      L.plane0[gidx] = gidx == 1 ? planes : 0;
      L.relu[gidx] = o == 0 ? 0 : 1;                    // row 1's hop is its last (no ReLU, pose_hrnet.py:218-226); the others go on (:227-236)
    }
    L.nlow = nb - 1;
    for (int k = 0; k < nb - 1; ++k) {
      const size_t t_frame = (size_t)planes * (H >> (k + 1)) * (W >> (k + 1)) * 16;
      L.term[k] = static_cast<const char*>(terms[k]) + (size_t)n0 * t_frame;
      L.term_bytes[k] = (uint32_t)((size_t)n * t_frame);
      L.shift[k] = k + 1;
    }
    L.N = n; L.H = H; L.W = W; L.Ho = Ho; L.Wo = Wo;
    L.tiles_x = (Wo + kTW - 1) / kTW; L.tiles_y = (Ho + kTH - 1) / kTH;
    L.tiles_total = n * L.tiles_x * L.tiles_y;
    L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y); L.fd_tiles_x = make_fastdiv(L.tiles_x);
    int grid = conv_device_cus();
    if (grid > L.tiles_total) grid = L.tiles_total;
    L.tiles_per_wg = (L.tiles_total + grid - 1) / grid;
    L.grid = (L.tiles_total + L.tiles_per_wg - 1) / L.tiles_per_wg;
    const int32_t rc = fd.dtype == SCPOSE_DT_BF16 ? fd_dispatch<0>(c0, nb, L, stream) : fd_dispatch<1>(c0, nb, L, stream);
    if (rc != SCPOSE_OK) return rc;
  }
  return SCPOSE_OK;
}

}  // namespace scpose
