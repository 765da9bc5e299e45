// Fused BasicBlock, second form: the two convolutions run on DIFFERENT waves at the same time, each with its whole
// weight set in registers.
//
//   out = ReLU(conv2(ReLU(conv1(x))) + x),  3x3 / stride 1 / C -> C, C = 16 * MREP in {32, 48}
//   (landmark_regression/lib/models/pose_hrnet.py:41-57; branch 0 of every HighResolutionModule :139-185)
//
// Why (round 3, profiles/round3_final_*): in conv_block_kernel.h all eight waves run conv1, then all eight run conv2.
// The matrix pipe is busy 49 % of the time -- the epilogues (ReLU / pack / LDS writes; lane exchange / residual / stores)
// and two barriers per tile are serial with the MFMA loops -- and every wave re-reads both layers' weights from LDS for
// every tile (672 of the 1 150 KB of LDS reads per tile are A fragments; LDS is active 49 % of the time too).
// Here a wave belongs to ONE layer for the kernel's whole life:
//   waves 0-3 (one per SIMD)  conv1 of tile i      x tile (LDS)  -> intermediate tile i (LDS, ReLU, 16-bit)
//   waves 4-7 (one per SIMD)  conv2 of tile i - 1   intermediate tile i - 1 (LDS) + residual (the tile's own input) -> output (global)
// * its layer's packed weights never leave its registers (C = 48: 14 k-steps x 3 row blocks x 4 VGPRs = 168), so the MFMA
//   A operands cost no LDS read at all and LDS holds nothing but the double-buffered input tile (2 x 38.4 KB) and the
//   double-buffered intermediate tile (2 x 32 KB);
// * while one layer's wave is in its epilogue the other layer's wave on the same SIMD is issuing MFMAs; one workgroup
//   barrier per tile.
// * residual (round 4): the block's residual IS the centre of its input tile, which is in LDS -- but conv2 of tile i - 1 runs
//   while that buffer is being refilled with tile i + 1.  So at the END of step i - 1, when the buffer is still whole (conv1 has
//   been reading it all step; the LDS-DMA of that step went to the other buffer) and the conv2 waves' accumulator and fragment
//   registers are free, every conv2 lane fetches the residual vectors of ITS output pixels of tile i - 1: the first pass's
//   (MREP vectors) stay in registers across the barrier, the second pass's go through a lane-private LDS slot (rl0), because
//   2 x MREP vectors next to the first pass's accumulators and fragments are 12 registers more than the 168 of weights leave.
//   Round 3 re-read them from global memory (through L2): 226 MB of the launch's 422 MB of fabric fetches at batch 256.
// Same MFMA (v_mfma_f32_16x16x32), same K order, same accumulator initialisation (bias) and the same rounding points as
// conv_block_kernel.h: results are bit-identical to it.
#pragma once
#include "conv_block_kernel.h"

namespace scpose {

// + the residual hand-over area: MREP 16-byte vectors per conv2 lane (see "residual" in the kernel)
constexpr size_t block2_lds_bytes(int mrep) { return 1024 + 2 * 2 * mrep * (size_t)(block_xs() + block_ms()) + (size_t)mrep * 256 * 16; }

template <int DT, int MREP>
__global__ __launch_bounds__(512, 2) void conv_block2_kernel(const BlockLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int MT = 16 * MREP, PLANES = 2 * MREP, KSTEPS = block_ksteps(MREP), NPT = MREP * 9;
  constexpr int XS = block_xs(), MS = block_ms();
  char* const xl0 = smem + 1024;                       // [2][PLANES][20 x 20][16 B]
  char* const ml0 = xl0 + 2 * PLANES * XS;             // [2][PLANES][18 x 18 (+ pad)][16 B]
  char* const rl0 = ml0 + 2 * PLANES * MS;             // [MREP][256 conv2 lanes][16 B]: residual vectors of the second pass (below)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1, hi = q >> 1;
  const int role = wave >> 2, rw = wave & 3, rtid = tid & 255;
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // ---- this wave's layer: weights and bias, for the kernel's whole life ----
  const char* wsrc = static_cast<const char*>(role ? p.w2 : p.w1);
  frag_t wf[KSTEPS][MREP];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int m = 0; m < MREP; ++m)
      wf[s][m] = *reinterpret_cast<const frag_t*>(wsrc + (size_t)s * (4 * MT * 16) + (q * MT + m * 16 + r) * 16);
  // biases (MFMA row order) stay in LDS: [conv1: 64][conv2: 64] floats, re-read when a pass initialises its accumulators
  float* const bias_l = reinterpret_cast<float*>(smem) + role * 64;
  if (tid < MT) { reinterpret_cast<float*>(smem)[tid] = p.b1[tid]; reinterpret_cast<float*>(smem)[64 + tid] = p.b2[tid]; }

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(p.tiles_total, t_begin + p.tiles_per_wg);
  const int ntiles = t_end - t_begin;
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = pipe_fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = pipe_fdiv(rem, p.fd_tiles_x);
    oy0 = ty * kBlockTile; ox0 = (rem - ty * p.tiles_x) * kBlockTile;
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.bytes), rs_out = make_buf(p.out, p.bytes);

  // input tile (20 x 20 halo of the 16 x 16 outputs) by LDS-DMA: issued by the four conv2 waves (they have the shorter step:
  // 672 against 882 MFMAs per tile), pixel slots rtid and rtid + 256.  (An LDS-DMA instruction blocks its wave while the path
  // is busy; splitting the planes between the two layers' waves made BOTH block as long: +4 %.)
  // slot group i (pixel slots rtid + 256 i) of tile t: the lane's offset in x, or BUF_OOB (padding: read as zeros)
  auto x_voff = [&](int t, int i) -> uint32_t {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    int rt = rtid;
    asm volatile("" : "+v"(rt));   // per-lane geometry is recomputed per tile, not hoisted out of the tile loop: the registers belong to the weights
    const int slot = rt + 256 * i;
    const int hy = slot / 20, hx = slot - hy * 20;
    const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
    const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    return ok ? (uint32_t)(img * PLANES * HW + iy * p.W + ix) * 16u : BUF_OOB;
  };
  auto x_dma = [&](uint32_t voff, int i, int pl, int b) {   // one LDS-DMA instruction: plane pl of slot group i into buffer b
    if (rtid + 256 * i < 400)    // lanes past the plane's last slot stay inactive: their LDS write would land in the next plane
      dma16_buf(rs_in, voff, (uint32_t)(pl * HW) * 16u, xl0 + b * (PLANES * XS) + pl * XS + (rw * 64 + 256 * i) * 16);
  };
  auto issue_x = [&](int t, int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t voff = x_voff(t, i);
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) x_dma(voff, i, pl, b);
    }
  };

  // B-operand addressing.  k-step s, k-group q covers (plane pair, tap) pair pt = 2 s + (q >> 1) and plane 2 (pt / 9) + (q & 1):
  //   offset = (q & 1) * PS  +  F(2 s + hi),   F(pt) = 2 (pt / 9) PS + ((pt % 9) / 3 * ROWW + (pt % 9) % 3) * 16
  // F(2 s) goes into the ds_read offset field, F(2 s + 1) - F(2 s) is added for the lanes with hi = 1 (one v_cndmask per k-step);
  // a padding pair (pt >= NPT: zero weights) reads the lane's base slot.
  auto koff = [](int pt, int PS, int ROWW) constexpr -> int {
    return pt < NPT ? 2 * (pt / 9) * PS + ((pt % 9) / 3 * ROWW + (pt % 9) % 3) * 16 : 0;
  };

  // k-loop: NCOL columns, the layer's A fragments from registers, B fragments two k-steps ahead (a k-step is 3-6 MFMAs = 48-96
  // cycles on this wave's pipe: less than an LDS read takes when the other seven waves read too)
  auto kloop = [&](auto ncol_c, auto role_c, const char* tile, const int* pixoff, f32x4 (*acc)[3], auto&& side, int psel_k) {
    constexpr int NCOL = decltype(ncol_c)::value;
    constexpr int ROLE = decltype(role_c)::value;
    constexpr int PS = ROLE ? MS : XS, ROWW = ROLE ? 18 : 20;
    const char* base[NCOL];
#pragma unroll
    for (int n = 0; n < NCOL; ++n) base[n] = tile + psel_k * PS + pixoff[n];
    int hi_ = hi;
    asm volatile("" : "+v"(hi_));   // the per-k-step selects below stay in the loop (one VALU each) instead of being hoisted
                                    // out of the tile loop into registers the weights need
    frag_t bf[3][NCOL];
    auto fetch = [&](auto sc) {
      constexpr int S = decltype(sc)::value;
      if constexpr (S < KSTEPS) {
        constexpr int F0 = koff(2 * S, PS, ROWW), F1 = koff(2 * S + 1, PS, ROWW);
        const int d = hi_ ? F1 - F0 : 0;
#pragma unroll
        for (int n = 0; n < NCOL; ++n) bf[S % 3][n] = *reinterpret_cast<const frag_t*>(base[n] + d + F0);
      }
    };
    fetch(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 1>{});
    static_for<0, KSTEPS>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      fetch(std::integral_constant<int, S + 2>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NCOL; ++n) acc[m][n] = mfma16<T>(wf[S][m], bf[S % 3][n], acc[m][n]);
      side(sc);                       // (conv2 waves: one LDS-DMA instruction of the next input tile behind this k-step's MFMAs)
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // (s_setprio 3 on the conv2 waves shortens their k-loops 4.5k -> 3.4k cycles per tile and lengthens their epilogues and the
  // conv1 waves' loops by as much: no gain, not kept)
  if (role == 1 && ntiles > 0) issue_x(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};   // development (SCPOSE_DBG & 8): [DMA issue / residual request][k-loops][epilogues][-][-][end wait + barrier]
  auto now = [&]() -> unsigned long long {
    if (!SCP_DBG_BUF(p)) return 0ull;
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
  };
  // step i: conv1 of tile t_begin + i (i < ntiles) beside conv2 of tile t_begin + i - 1 (i >= 1); one barrier per step.
  // (Two loops, one per role, with the same number of barriers: values one role carries from step to step -- the conv2 waves'
  // residual registers -- are then not live in the other role's code, which has no register to spare either.)
  if (role == 0) {
    for (int i = 0; i <= ntiles; ++i) {
      if (i < ntiles) {
        const int t = t_begin + i;
        int img, oy0, ox0;
        decode(t, img, oy0, ox0);
        const char* xl = xl0 + (i & 1) * (PLANES * XS);
        char* ml = ml0 + (i & 1) * (PLANES * MS);
        // NARROW: the tile is the right neighbour of this workgroup's previous tile, whose conv2 waves copy the two shared
        // intermediate columns (above): conv1 runs on the 18 x 16 new pixels only, one MFMA column per intermediate row
        // (18 columns: 5, 5, 4, 4 per wave instead of 21: 6, 5, 5, 5)
        const bool narrow = i > 0 && ox0 > 0;
        // full tile: 21 columns of 16 intermediate pixels (18 x 18 = 324); wave rw owns columns rw + 4 j, two at a time (three
        // accumulator columns and their fragments do not fit beside 168 registers of weights)
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
          int rr = r;
          asm volatile("" : "+v"(rr));                          // (see issue_x)
          auto geom = [&](int n, int& pidx, int& my, int& mx) {   // pidx: slot in the 18 x 18 intermediate tile, >= 324: none
            const int c = rw + 4 * (2 * pass + n);
            pidx = narrow ? (c < 18 ? c * 18 + 2 + rr : 18 * 18) : c * 16 + rr;
            const bool ok = pidx < 18 * 18;
            my = ok ? pidx / 18 : 0;
            mx = ok ? pidx - my * 18 : 0;
          };
          // columns this wave has in this pass: two, except in the last pass (full tile: two for wave 0 (columns 16, 20), one for
          // the others; narrow tile: one for waves 0, 1 (rows 16, 17), none for waves 2, 3) -- no MFMAs on columns nobody reads
          const int cmax = narrow ? 18 : 21;
          const int ncol = (rw + 8 * pass < cmax) + (rw + 8 * pass + 4 < cmax);
          if (ncol == 0) continue;
          int pixoff[2];
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            int pidx, my, mx;
            geom(n, pidx, my, mx);
            pixoff[n] = (my * 20 + mx) * 16;
          }
          f32x4 acc[MREP][3];
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_l + m * 16 + q * 4);
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{b4.x, b4.y, b4.z, b4.w};   // accumulators start at the bias of their rows
          }
          const unsigned long long s1 = now();
          auto none = [](auto) {};
          if (ncol == 2) kloop(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, xl, pixoff, acc, none, psel);
          else kloop(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, xl, pixoff, acc, none, psel);
          const unsigned long long s2 = now();
          tph[1] += s2 - s1;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            int pidx, my, mx;
            asm volatile("" : "+v"(rr));
            geom(n, pidx, my, mx);                              // recomputed: nothing but the accumulators lives across the k-loop
            const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
            const bool inimg = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const bool okn = pidx < 18 * 18 && n < ncol;
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
              uint2 o;
              o.x = relu2_16(pack2<T>(acc[m][n][0], acc[m][n][1]), 0u);
              o.y = relu2_16(pack2<T>(acc[m][n][2], acc[m][n][3]), 0u);
              if (!inimg) o = make_uint2(0u, 0u);              // conv2's zero padding
              if (okn) *reinterpret_cast<uint2*>(ml + (2 * m + psel) * MS + pidx * 16 + 8 * hi) = o;
            }
          }
          tph[2] += now() - s2;
        }
      }
      const unsigned long long s9 = now();
      // intermediate tile i complete and input tile i + 1 landed | intermediate tile i - 1 and input tile i free.
      // ONE statement with a memory clobber: s_barrier is IntrNoMem to the compiler, which may move LDS loads of the next step
      // above it -- in the product build of the first two-loop form of this kernel (no instrumentation asm behind the barrier) it
      // did: the conv1 waves read the next input tile before the other waves' LDS-DMA had been waited for (non-deterministic
      // results that the development build did not show; tools_dev/check_block_determinism.py)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      tph[5] += now() - s9;
    }
  } else {
    u32x4 res0[MREP];   // residual vectors of the coming step's first pass
#pragma unroll
    for (int m = 0; m < MREP; ++m) res0[m] = u32x4{0u, 0u, 0u, 0u};
    for (int i = 0; i <= ntiles; ++i) {
      const unsigned long long s0 = now();
      // the next input tile (its buffer held tile i - 1: conv1 finished reading it a barrier ago) is requested from inside the k-loops
      // below, one LDS-DMA instruction per k-step: an LDS-DMA instruction blocks its wave while the path is busy (~140 cycles each
      // with four waves issuing), which the MFMAs already issued for the k-step cover -- up front, the twelve of them cost
      // the conv2 waves 1 700 cycles per tile.  Step 0 has no k-loop: there the tile is requested here.
      if (i == 0 && 1 < ntiles) issue_x(t_begin + 1, 1);
      // the two rightmost intermediate columns of tile i - 1 are the two leftmost ones of tile i when that is its right neighbour
      // (the same image pixels): copied from this step's read-only intermediate buffer into the one conv1 is filling, whose
      // conv1 then skips them (NARROW below)
      if (i >= 1 && i < ntiles) {
        int img1, oy1, ox1;
        decode(t_begin + i, img1, oy1, ox1);
        int rt = rtid;
        asm volatile("" : "+v"(rt));                             // (see issue_x)
        if (ox1 > 0 && rt < PLANES * 36) {
          const int kpl = rt / 36, krem = rt - kpl * 36;         // plane, (row, column 16 / 17)
          const u32x4 keep = *reinterpret_cast<const u32x4*>(ml0 + ((i - 1) & 1) * (PLANES * MS) + kpl * MS + ((krem >> 1) * 18 + 16 + (krem & 1)) * 16);
          *reinterpret_cast<u32x4*>(ml0 + (i & 1) * (PLANES * MS) + kpl * MS + ((krem >> 1) * 18 + (krem & 1)) * 16) = keep;
        }
      }
      tph[0] += now() - s0;
      if (i >= 1) {
        const int t = t_begin + i - 1;
        int img, oy0, ox0;
        decode(t, img, oy0, ox0);
        const char* ml = ml0 + ((i - 1) & 1) * (PLANES * MS);
        // 16 columns = the tile's 16 rows: wave rw owns rows rw + 4 j, two at a time (the lower half-wave ends up with the
        // first row's pixel, the upper half-wave with the second row's)
        // (the two passes are two instances of one body: in a rolled loop the first pass's residual registers would be live through
        // the second pass too)
        auto conv2_pass = [&](auto pass_c) {
          constexpr int pass = decltype(pass_c)::value;
          const int row0 = rw + 8 * pass, row1 = row0 + 4;
          int rt2 = tid;
          asm volatile("" : "+v"(rt2));                           // (see issue_x: per-pass lane geometry is formed here, from an opaque
                                                                  // copy of the thread id, or the unrolled passes' constants get hoisted)
          const int rr = rt2 & 15, half_ = (rt2 >> 5) & 1, psel_ = (rt2 >> 4) & 1;
          const int oy = oy0 + (half_ ? row1 : row0), ox = ox0 + rr;
          const bool store_ok = oy < p.H && ox < p.W;
          const uint32_t gvoff = store_ok ? (uint32_t)((img * PLANES + psel_) * HW + oy * p.W + ox) * 16u : BUF_OOB;
          u32x4 resv[MREP];                                       // residual: this lane's output pixel of x, planes 2 m + psel
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            if constexpr (pass == 0) resv[m] = res0[m];             // fetched at the end of the previous step, kept in registers
            else resv[m] = *reinterpret_cast<const u32x4*>(rl0 + m * 4096 + (rt2 & 255) * 16);
          }
          int pixoff[2] = {(row0 * 18 + rr) * 16, (row1 * 18 + rr) * 16};
          f32x4 acc[MREP][3];
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_l + m * 16 + q * 4);
            acc[m][0] = f32x4{b4.x, b4.y, b4.z, b4.w}; acc[m][1] = acc[m][0];
          }
          const unsigned long long s1 = now();
          // (slot group = pass; all twelve in the first pass would need one more register than there is.  One per k-step in the
          // FIRST six k-steps of the pass: spread over every other k-step the block takes 241 instead of 206 us -- both layers'
          // k-loops slow down while DMA data is arriving, and the tile lands later.  Issuing them and the residual loads as
          // inline asm with hand-counted waits, so that the compiler's conservative waits behind LDS-DMA disappear: no change.)
          const bool dma_on = i + 1 < ntiles;
          const uint32_t xv = dma_on ? x_voff(t_begin + i + 1, pass) : BUF_OOB;
          auto side = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if constexpr (S < PLANES) { if (dma_on) x_dma(xv, pass, S, (i + 1) & 1); }
          };
          kloop(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, ml, pixoff, acc, side, psel_);
          // v_permlane32_swap below reads MFMA results: explicit wait states tied to the accumulators (conv_device.h: mfma_swap_pad).
          // Until round 4 the residual loads' wait happened to sit in between; with the first pass's residual already in registers
          // the swap follows the last MFMA directly.
          if constexpr (MREP == 3) mfma_swap_pad(acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1]);
          else mfma_swap_pad(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
          const unsigned long long s2 = now();
          tph[1] += s2 - s1;
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            uint32_t a[4], b[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { a[jj] = __float_as_uint(acc[m][0][jj]); b[jj] = __float_as_uint(acc[m][1][jj]); }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
              a[jj] = sw[0]; b[jj] = sw[1];
            }
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
            store16_buf(rs_out, gvoff, (uint32_t)(2 * m * HW) * 16u, ov);
          }
          tph[2] += now() - s2;
        };
        conv2_pass(std::integral_constant<int, 0>{});
        conv2_pass(std::integral_constant<int, 1>{});
      }
      // residual of tile i for the next step (see the header comment): its input buffer is whole until the barrier below
      if (i < ntiles) {
        int tt = tid;
        asm volatile("" : "+v"(tt));                               // lane geometry recomputed here (see issue_x): nothing of it lives across the passes
        const int l_ = tt & 63, rr = l_ & 15, ps_ = (l_ >> 4) & 1, hf_ = l_ >> 5, rw_ = (tt >> 6) & 3;
        const char* xc = xl0 + (i & 1) * (PLANES * XS) + ps_ * XS + ((2 + rw_ + 4 * hf_) * 20 + 2 + rr) * 16;   // centre pixel (rw | rw + 4, rr), plane psel
        char* rl = rl0 + (tt & 255) * 16;
#pragma unroll
        for (int m = 0; m < MREP; ++m) res0[m] = *reinterpret_cast<const u32x4*>(xc + 2 * m * XS);                // first pass: rows rw | rw + 4
#pragma unroll
        for (int m = 0; m < MREP; ++m)                                                                           // second pass: 8 rows further down
          *reinterpret_cast<u32x4*>(rl + m * 4096) = *reinterpret_cast<const u32x4*>(xc + 2 * m * XS + 8 * 20 * 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next input tile has landed (this wave's pieces)
      const unsigned long long s9 = now();
      // intermediate tile i complete and input tile i + 1 landed | intermediate tile i - 1 and input tile i free.
      // ONE statement with a memory clobber: s_barrier is IntrNoMem to the compiler, which may move LDS loads of the next step
      // above it -- in the product build of the first two-loop form of this kernel (no instrumentation asm behind the barrier) it
      // did: the conv1 waves read the next input tile before the other waves' LDS-DMA had been waited for (non-deterministic
      // results that the development build did not show; tools_dev/check_block_determinism.py)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      tph[5] += now() - s9;
    }
  }
  if (SCP_DBG_BUF(p) && lane == 0)
    for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave) * 6 + k] = tph[k];
}

template <int DT, int MREP>
int32_t block2_launch_one(const BlockLaunch& L, hipStream_t stream) {
  const size_t lds = block2_lds_bytes(MREP);
  auto kern = conv_block2_kernel<DT, MREP>;
  static bool attr_set[16] = {};
  int dev = 0;
  SCP_CHECK_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    SCP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(512), lds, stream, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
