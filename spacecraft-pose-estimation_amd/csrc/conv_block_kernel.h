// Fused BasicBlock for the high-resolution branch:  out = ReLU(conv2(ReLU(conv1(x))) + x),
// both 3x3 / stride 1 / C -> C with folded BatchNorm, C = 16*MREP <= 64
// (landmark_regression/lib/models/pose_hrnet.py:41-57; branch 0 of every HighResolutionModule :139-185).
//
// Why: at C = 48 the two convolutions are HBM-bound (172.8 flop/B) and, run separately, move five
// activation tensors per block (conv1 in+out, conv2 in+residual+out).  Fused, the block reads x once
// (with a 2-pixel halo) and writes its output once: 2.3x fewer bytes and memory instructions, for 27 %
// more MFMAs (conv1 is evaluated on the 18x18 halo of each 16x16 output tile).
//
// One 512-thread workgroup per CU (two waves per SIMD, which a 16x16x32 MFMA stream needs to fill the
// matrix pipe), persistent over 16x16 output tiles.  LDS (157.7 KB at C = 48): both layers' packed weights
// resident (2 x 42 KB), the 20x20 input tile (6 planes, 38.4 KB), the 18x18 intermediate tile (32 KB).
// Per tile:
//   A. conv1 on the 18x18 intermediate pixels (21 MFMA columns over 8 waves), bias + ReLU + 16-bit rounding
//      -- the same rounding the unfused path applies when it stores conv1's output -- written to the
//      intermediate tile in LDS as 8-byte half-slots (no lane exchange needed); pixels outside the image
//      are written as zeros (they are conv2's zero padding, not conv1 outputs);
//      before that, every lane copies the residual slots it will need (the centre of the input tile) into
//      registers, so that after the barrier that ends this phase the input tile is dead;
//   B. the LDS-DMA of the NEXT tile's input is issued into it and streams in under phase C;
//   C. conv2 on the 16x16 output pixels from the intermediate tile, bias + residual + ReLU, 16-byte stores.
// Two workgroup barriers per tile.  Both k-loops are fully unrolled, hand-scheduled asm (fragment reads
// one step ahead, tap offsets in registers, weight offsets immediates), as in conv_pipe_kernel.h.
#pragma once
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // dma16, u32x4, pipe_fdiv

namespace scpose {

struct BlockLaunch {
  const void* in;        // blocked N x C x H x W
  const void* w1;        // packed weights of conv1 / conv2 (pack_conv_weights, mt = C, one chunk)
  const void* w2;
  const float* b1;       // folded biases in packed row order
  const float* b2;
  void* out;             // blocked N x C x H x W
  const void* zero16;
  uint32_t bytes;        // size of in (= out) when < 4 GiB: buffer-addressed global traffic (conv_pipe_kernel.h), else 0
  int32_t N, H, W;
  int32_t tiles_x, tiles_y, tiles_total, tiles_per_wg, grid;
  FastDiv fd_tiles_img, fd_tiles_x;
  unsigned long long* dbg_buf;   // development: per-wave phase cycle sums (SCPOSE_DBG & 8), else null
};

template <int S, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (S < E) {
    f(std::integral_constant<int, S>{});
    static_for<S + 1, E>(f);
  }
}

constexpr int kBlockTile = 16;                                   // output tile edge
constexpr int block_ksteps(int mrep) { return (mrep * 9 + 1) / 2; }
constexpr int block_xs() { return 20 * 20 * 16; }               // bytes of one input-tile plane
constexpr int block_ms() { return (18 * 18 * 16 + 255) & ~255; } // bytes of one intermediate-tile plane
constexpr size_t block_lds_bytes(int mrep) {
  return 1024 + 2 * (size_t)block_ksteps(mrep) * 4 * (16 * mrep) * 16 + 2 * mrep * (size_t)(block_xs() + block_ms());
}

template <int DT, int MREP>
__global__ __launch_bounds__(512, 2) void conv_block_kernel(const BlockLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int MT = 16 * MREP, PLANES = 2 * MREP, KSTEPS = block_ksteps(MREP), NPT = MREP * 9;
  constexpr int XS = block_xs(), MS = block_ms();
  constexpr int WBYTES = KSTEPS * 4 * MT * 16;

  int* koffA = reinterpret_cast<int*>(smem);            // [16 steps][4 k-groups] offsets into the input tile
  int* koffB = koffA + 64;                              // ... into the intermediate tile
  float* bias1 = reinterpret_cast<float*>(smem + 512);
  float* bias2 = bias1 + 64;
  char* w1l = smem + 1024;
  char* w2l = w1l + WBYTES;
  char* xl = w2l + WBYTES;
  char* ml = xl + PLANES * XS;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases / M0
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  if (tid < 128) {   // tap offsets of MFMA k-step st, k-group qq (plane 2pp + (qq&1), tap pt = 2 st + (qq>>1))
    const int tbl = tid >> 6, e = tid & 63;
    const int st = e >> 2, qq = e & 3;
    const int pt = 2 * st + (qq >> 1);
    const int pp = pt / 9, tap = pt - pp * 9;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int v = tbl == 0 ? (2 * pp + (qq & 1)) * XS + (ky * 20 + kx) * 16 : (2 * pp + (qq & 1)) * MS + (ky * 18 + kx) * 16;
    (tbl == 0 ? koffA : koffB)[e] = pt < NPT ? v : 0;
  }
  if (tid < MT) { bias1[tid] = p.b1[tid]; bias2[tid] = p.b2[tid]; }

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(p.tiles_total, t_begin + p.tiles_per_wg);

  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = pipe_fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = pipe_fdiv(rem, p.fd_tiles_x);
    oy0 = ty * kBlockTile; ox0 = (rem - ty * p.tiles_x) * kBlockTile;
  };
  // input tile (20 x 20 halo of the 16 x 16 outputs): one pixel per thread, all planes
  const int hy = tid / 20, hx = tid - hy * 20;
  const bool buf = p.bytes != 0;
  const buf_rsrc_t rs_in = make_buf(p.in, p.bytes), rs_out = make_buf(p.out, p.bytes);
  auto issue_x = [&](int t) {
    if (tid < 400) {
      int img, oy0, ox0;
      decode(t, img, oy0, ox0);
      const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      if (buf) {   // 32-bit lane offset, plane displacement in an SGPR, padding pixels out of range (read as zeros)
        const uint32_t voff = ok ? (uint32_t)(img * PLANES * HW + iy * p.W + ix) * 16u : BUF_OOB;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) dma16_buf(rs_in, voff, (uint32_t)(pl * HW) * 16u, xl + pl * XS + wave * 1024);
        return;
      }
      const char* src0 = static_cast<const char*>(p.in) + ((size_t)img * PLANES * HW + (size_t)(ok ? iy * p.W + ix : 0)) * 16;
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) {
        const char* src = ok ? src0 + (size_t)pl * HW * 16 : static_cast<const char*>(p.zero16);
        dma16(src, xl + pl * XS + wave * 1024);
      }
    }
  };

  // prologue: both weight sets, the first input tile
  for (int o = 0; o < WBYTES; o += 8192) {
    const int mine = o + tid * 16;
    if (mine < WBYTES) {
      dma16(static_cast<const char*>(p.w1) + mine, w1l + o + wave * 1024);
      dma16(static_cast<const char*>(p.w2) + mine, w2l + o + wave * 1024);
    }
  }
  if (t_begin < t_end) issue_x(t_begin);
  __syncthreads();

  int kA[KSTEPS], kB[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) { kA[s] = koffA[s * 4 + q]; kB[s] = koffB[s * 4 + q]; }

  // phase A columns of this wave: intermediate pixels (n*8 + wave)*16 + r, n = 0..2 (21 columns of 16 over 324 pixels)
  int offA[3], myA[3], mxA[3];
  bool okA[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int pidx = (n * 8 + wave) * 16 + r;
    okA[n] = pidx < 18 * 18;
    myA[n] = okA[n] ? pidx / 18 : 0;
    mxA[n] = okA[n] ? pidx - myA[n] * 18 : 0;
    offA[n] = (myA[n] * 20 + mxA[n]) * 16;
  }
  // A tile whose left neighbour was this workgroup's previous tile re-uses that tile's two rightmost intermediate columns
  // (the same image pixels: they are copied inside LDS between the tiles) and evaluates conv1 only on its 18 x 16 new
  // pixels: one MFMA column per intermediate row (18 columns instead of 21; five of the six tiles of a 96-pixel row).
  // The chip is power-limited in this kernel: 12 % fewer conv1 MFMAs are time, not just idle issue slots.
  int offN[3], myN[3], slotN[3];
  bool okN[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int c = n * 8 + wave;            // intermediate row
    okN[n] = c < 18;
    myN[n] = okN[n] ? c : 0;
    offN[n] = (myN[n] * 20 + 2 + r) * 16;
    slotN[n] = (myN[n] * 18 + 2 + r) * 16;
  }
  // phase C columns: output pixels (n*8 + wave)*16 + r, n = 0..1; after the lane exchange the lower half-wave
  // owns column 0's pixel, the upper half-wave column 1's
  int offB[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int pidx = (n * 8 + wave) * 16 + r;
    offB[n] = ((pidx >> 4) * 18 + (pidx & 15)) * 16;
  }
  const int opix = (half * 8 + wave) * 16 + r;       // the output pixel whose 16-byte slots this lane stores
  const int opy = opix >> 4, opx = opix & 15;

  // fully unrolled k-loop: NCOL columns, fragments one step ahead
  auto kloop = [&](auto ncol_c, const char* wl, const char* bl, const int* kv, const int* off, f32x4 (*acc)[3]) {
    constexpr int NCOL = decltype(ncol_c)::value;
    const uint32_t wa = (uint32_t)(size_t)wl + (q * MT + r) * 16;
    const uint32_t ba = (uint32_t)(size_t)bl;
    frag_t a0[MREP], b0[NCOL], a1[MREP], b1[NCOL];
    auto issue = [&](auto sc, frag_t* a, frag_t* b) {
      constexpr int S = decltype(sc)::value;
      lds_read16<S * (4 * MT * 16)>(a[0], wa);
      if constexpr (MREP > 1) lds_read16<S * (4 * MT * 16) + 256>(a[1], wa);
      if constexpr (MREP > 2) lds_read16<S * (4 * MT * 16) + 512>(a[2], wa);
      if constexpr (MREP > 3) lds_read16<S * (4 * MT * 16) + 768>(a[3], wa);
#pragma unroll
      for (int n = 0; n < NCOL; ++n) lds_read16<0>(b[n], ba + kv[S] + off[n]);
    };
    auto landed = [&](frag_t* a, frag_t* b) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < MREP; ++m) lds_landed(a[m]);
#pragma unroll
      for (int n = 0; n < NCOL; ++n) lds_landed(b[n]);
    };
    auto cols = [&](const frag_t* a, const frag_t* b, int n0, int n1) {
#pragma unroll
      for (int n = 0; n < NCOL; ++n)
        if (n >= n0 && n < n1)
#pragma unroll
          for (int m = 0; m < MREP; ++m) mfma16_acc<T>(acc[m][n], a[m], b[n]);
    };
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < NCOL; ++n) mfma_input_fence<false>(acc[m][n]);
    issue(std::integral_constant<int, 0>{}, a0, b0);
    landed(a0, b0);
    static_for<0, KSTEPS>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      frag_t* ca = (S & 1) ? a1 : a0; frag_t* cb = (S & 1) ? b1 : b0;
      frag_t* na = (S & 1) ? a0 : a1; frag_t* nb = (S & 1) ? b0 : b1;
#ifdef SCPOSE_BLOCK_NO_OVERLAP
      cols(ca, cb, 0, NCOL);
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      if constexpr (S + 1 < KSTEPS) issue(std::integral_constant<int, S + 1>{}, na, nb);
#else
      cols(ca, cb, 0, 1);
      if constexpr (S + 1 < KSTEPS) issue(std::integral_constant<int, S + 1>{}, na, nb);
      cols(ca, cb, 1, NCOL);
#endif
      if constexpr (S + 1 < KSTEPS) landed(na, nb);
    });
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last MFMA's result visible to the VALU
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < NCOL; ++n) mfma_result_fence<false>(acc[m][n]);
  };

  float4 bs1[MREP], bs2[MREP];   // this lane's bias values (rows 4q..4q+3 of every 16-row tile), tile-invariant
#pragma unroll
  for (int m = 0; m < MREP; ++m) {
    bs1[m] = *reinterpret_cast<const float4*>(bias1 + m * 16 + q * 4);
    bs2[m] = *reinterpret_cast<const float4*>(bias2 + m * 16 + q * 4);
  }

  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { return SCP_DBG_BUF(p) ? __builtin_amdgcn_s_memtime() : 0ull; };
  for (int t = t_begin; t < t_end; ++t) {
    const unsigned long long t0 = now();
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    f32x4 acc[MREP][3];
    const bool narrow = t > t_begin && ox0 > 0;          // columns 0, 1 of the intermediate tile are already there (see offN)

    // ---- A: conv1 -> intermediate tile ----
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{bs1[m].x, bs1[m].y, bs1[m].z, bs1[m].w};   // accumulators start at the bias of their rows
    // 21 columns over 8 waves: waves 0-4 have a third column, waves 5-7 do not (their third accumulators are never written out);
    // they run the two-column loop instead of spending 42 MFMAs per tile on a column nobody reads
    if (wave < (narrow ? 2 : 5)) kloop(std::integral_constant<int, 3>{}, w1l, xl, kA, narrow ? offN : offA, acc);
    else kloop(std::integral_constant<int, 2>{}, w1l, xl, kA, narrow ? offN : offA, acc);
    const unsigned long long t1 = now();
    // residual slots of this lane's output pixel (the centre of the input tile), kept in registers until the end
    u32x4 resv[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
      resv[m] = *reinterpret_cast<const u32x4*>(xl + (2 * m + psel) * XS + ((opy + 2) * 20 + opx + 2) * 16);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int gy = oy0 - 1 + (narrow ? myN[n] : myA[n]), gx = ox0 - 1 + (narrow ? 2 + r : mxA[n]);
      const bool inimg = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const bool okn = narrow ? okN[n] : okA[n];
      const int slot = narrow ? slotN[n] : ((n * 8 + wave) * 16 + r) * 16;
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        uint2 o;
        o.x = relu2_16(pack2<T>(acc[m][n][0], acc[m][n][1]), 0u);
        o.y = relu2_16(pack2<T>(acc[m][n][2], acc[m][n][3]), 0u);
        if (!inimg) o = make_uint2(0u, 0u);              // conv2's zero padding
        if (okn)
          *reinterpret_cast<uint2*>(ml + (2 * m + psel) * MS + slot + 8 * (q >> 1)) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t2 = now();
    __builtin_amdgcn_s_barrier();                        // intermediate tile complete; nobody reads the input tile any more
    const unsigned long long t3 = now();
    if (t + 1 < t_end) issue_x(t + 1);                   // ... so the next tile's input streams in under conv2

    // ---- C: conv2 -> output ----
#pragma unroll
    for (int m = 0; m < MREP; ++m) { acc[m][0] = f32x4{bs2[m].x, bs2[m].y, bs2[m].z, bs2[m].w}; acc[m][1] = acc[m][0]; }
    const unsigned long long t4 = now();
    kloop(std::integral_constant<int, 2>{}, w2l, ml, kB, offB, acc);
    // MFMA result -> v_permlane32_swap needs more wait states than hipcc pads (conv_device.h: mfma_swap_pad)
    if constexpr (MREP == 3) mfma_swap_pad(acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1]);
    else mfma_swap_pad(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
    // The next input tile (requested after the first barrier, a whole conv2 ago) must have landed before the barrier that ends the
    // tile.  Wait for it HERE, in front of this tile's output stores: a wave's vector-memory operations are counted together and stores
    // may retire before older loads, so the only safe wait is vmcnt(0) -- and behind the stores that would be a wait for their
    // acknowledgement (hundreds of cycles per tile) instead of for a DMA that has long arrived.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t5 = now();
    const int oy = oy0 + opy, ox = ox0 + opx;
    const bool store_ok = oy < p.H && ox < p.W;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
      uint32_t a[4], b[4];
      a[0] = __float_as_uint(acc[m][0][0]); a[1] = __float_as_uint(acc[m][0][1]);
      a[2] = __float_as_uint(acc[m][0][2]); a[3] = __float_as_uint(acc[m][0][3]);
      b[0] = __float_as_uint(acc[m][1][0]); b[1] = __float_as_uint(acc[m][1][1]);
      b[2] = __float_as_uint(acc[m][1][2]); b[3] = __float_as_uint(acc[m][1][3]);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
        a[jj] = sw[0]; b[jj] = sw[1];
      }
      // lower half-wave: a = own (column 0, channels 0-3), b = partner's (column 0, channels 4-7);
      // upper half-wave: a = partner's (column 1, channels 0-3), b = own (column 1, channels 4-7)
      float v[8];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) { v[jj] = __uint_as_float(a[jj]); v[4 + jj] = __uint_as_float(b[jj]); }
      const u32x4 rv = resv[m];
      v[0] += from_bits<T>(rv[0] & 0xffff); v[1] += from_bits<T>(rv[0] >> 16);
      v[2] += from_bits<T>(rv[1] & 0xffff); v[3] += from_bits<T>(rv[1] >> 16);
      v[4] += from_bits<T>(rv[2] & 0xffff); v[5] += from_bits<T>(rv[2] >> 16);
      v[6] += from_bits<T>(rv[3] & 0xffff); v[7] += from_bits<T>(rv[3] >> 16);
      u32x4 ov;
      ov[0] = relu2_16(pack2<T>(v[0], v[1]), 0u); ov[1] = relu2_16(pack2<T>(v[2], v[3]), 0u);
      ov[2] = relu2_16(pack2<T>(v[4], v[5]), 0u); ov[3] = relu2_16(pack2<T>(v[6], v[7]), 0u);
      if (buf)
        store16_buf(rs_out, store_ok ? (uint32_t)((img * PLANES + psel) * HW + oy * p.W + ox) * 16u : BUF_OOB, (uint32_t)(2 * m * HW) * 16u, ov);
      else if (store_ok)
        *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + (((size_t)img * PLANES + 2 * m + psel) * HW + (size_t)oy * p.W + ox) * 16) = ov;
    }
    const unsigned long long t6 = now();
    // (the next input tile has landed: waited for in front of the stores, above.  Deferring the conv2 epilogue of waves 4-7 into the
    // next tile -- a stagger between the two waves of a SIMD -- measured no gain.)
    // the two rightmost intermediate columns become the next tile's two leftmost ones when that tile is this one's right
    // neighbour: read here (the intermediate tile is read-only in this phase), written behind the barrier (nobody reads
    // the intermediate tile then, and conv1 of the next tile writes columns 2..17 only)
    bool keep_next = false;
    u32x4 keep = {0u, 0u, 0u, 0u};
    if (t + 1 < t_end) {
      int img1, oy1, ox1;
      decode(t + 1, img1, oy1, ox1);
      keep_next = ox1 > 0;
    }
    const int kpl = tid / 36, krem = tid - kpl * 36;                // plane, (row, column 16 / 17)
    const bool keeper = keep_next && tid < PLANES * 36;
    if (keeper) keep = *reinterpret_cast<const u32x4*>(ml + kpl * MS + ((krem >> 1) * 18 + 16 + (krem & 1)) * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // ... and the intermediate tile is free again
    if (keeper) *reinterpret_cast<u32x4*>(ml + kpl * MS + ((krem >> 1) * 18 + (krem & 1)) * 16) = keep;
    if (SCP_DBG_BUF(p)) {   // [conv1 loop][conv1 epilogue][barrier][DMA issue][conv2 loop][conv2 epilogue + end wait/barrier]
      const unsigned long long t7 = now();
      tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3; tph[4] += t5 - t4; tph[5] += t7 - t5;
      (void)t6;
    }
  }
  if (SCP_DBG_BUF(p) && lane == 0)
    for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave) * 6 + k] = tph[k];
}

template <int DT, int MREP>
int32_t block_launch_one(const BlockLaunch& L, hipStream_t st) {
  auto kern = conv_block_kernel<DT, MREP>;
  static LdsOptIn big_lds;   // per device (common.h)
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(512), block_lds_bytes(MREP), st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
