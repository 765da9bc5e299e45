// Producer/consumer form of the 32x32x16-MFMA 3x3 convolution (see conv_m32_kernel.h for the GEMM
// view, fragment layouts and the reference lines it replaces).
//
// Measured on MI355X (tools_dev/micro): a wave that issues memory instructions -- LDS-DMA,
// residual loads, output stores -- stalls in the issue of each one while the memory pipeline is
// busy, and cannot issue MFMAs meanwhile; with one wave per SIMD that stall is matrix-pipe idle time
// (the single-role kernel spends ~35 % of its cycles there).  Two resident waves per SIMD cap the
// wave at 256 registers, so instead of two symmetric workgroups the 512-thread workgroup is split
// by ROLE:
//   * waves 0-3, one per SIMD: consumers.  They execute only LDS reads, MFMAs and the VALU epilogue;
//     a 32x32x16 MFMA stream from one wave keeps its SIMD's matrix pipe full.
//   * waves 4-7, one per SIMD: producers.  They execute every global-memory instruction: LDS-DMA of the
//     next K-chunk (weights + halos), LDS-DMA of the tile's residual rows into an LDS "retire buffer",
//     and the stores of the previous tile's results out of that buffer.
// Retire buffer (RO): [Cout-block planes][tile pixels] x 16 B.  For tile i the producers store tile
// i-1's results from RO during the first chunks, DMA tile i's residual into RO during the middle
// chunks, and after the last chunk's MFMAs the consumers replace each residual slot by
// ReLU(acc + bias + residual) in place.  Every stage ends in one workgroup barrier, which orders all
// of this.  Needs >= 3 K-chunks per tile (host falls back to the single-role kernel otherwise).
#pragma once
#include "conv_m32_kernel.h"

namespace scpose {

// ---- 16x16x32 consumers (C16 > 0): compile-time schedule ----------------------------------------------------------------
// k-step S of a pair of plane pairs: class 0 reads tap m16_tap0(S), class 1 the tap one pixel (or, m16_rowpair, one row) further
constexpr int m16_tap0(int S) { return S == 0 ? 0 : S == 1 ? 3 : S == 2 ? 6 : S == 3 ? 2 : S == 4 ? 8 : S == 5 ? 0 : S == 6 ? 3 : S == 7 ? 6 : 5; }
constexpr bool m16_rowpair(int S) { return S == 3 || S == 8; }
// B fragments live in a ring of NB + 1 registers: column N of k-step S sits in slot (N - S) mod RING, so the request for column N
// of k-step S + 1 goes to the slot column N - 1 released one column earlier (never to a register an MFMA in flight still reads)
constexpr int m16_ring(int N, int S, int RING) { return ((N - S) % RING + 4 * RING) % RING; }
// behind column n a prefetching k-step requests m16_na(NB, n) of the next k-step's six A fragments, then its B fragment of column n
constexpr int m16_na(int NB, int n) { return NB == 2 ? 3 : NB == 3 ? (n < 2 ? 3 : 0) : (n < 2 ? 2 : n < 4 ? 1 : 0); }   // NB 6: 2 2 1 1 0 0, 5: 2 2 1 1 0, 4: 2 2 1 1, 3: 3 3 0, 2: 3 3
constexpr int m16_afirst(int NB, int n) { int a = 0; for (int k = 0; k < n; ++k) a += m16_na(NB, k); return a; }
constexpr int m16_issued(int NB, int n) { return m16_afirst(NB, n) + n; }   // requests of a prefetching k-step in front of its column n
constexpr int m16_ja(int NB) { return NB <= 3 ? 1 : 3; }                    // the column that requests the last A fragment
// LDS returns in order: lgkmcnt value to wait for in front of column n (-1: what it needs is older than something already waited for).
// first: the k-step's fragments were requested in one burst, A0..5 B0..NB-1; more: the k-step itself prefetches.
constexpr int m16_wait(int NB, int n, bool first, bool more) {
  if (first) return (NB - 1 - n) + (more ? m16_issued(NB, n) : 0);
  if (n == 0) return NB - m16_ja(NB);
  if (n < m16_ja(NB)) return -1;
  return (NB - 1 - n) + (more ? m16_issued(NB, n) : 0);
}
static_assert(m16_wait(6, 0, true, true) == 5 && m16_wait(6, 1, true, true) == 7 && m16_wait(6, 3, true, true) == 10 && m16_wait(6, 5, true, true) == 11, "");
static_assert(m16_wait(6, 0, false, true) == 3 && m16_wait(6, 2, false, true) == -1 && m16_wait(6, 3, false, true) == 10 && m16_wait(6, 4, false, true) == 11, "");
static_assert(m16_wait(6, 0, false, false) == 3 && m16_wait(6, 3, false, false) == 2 && m16_wait(6, 5, false, false) == 0, "");
static_assert(m16_wait(3, 0, false, true) == 2 && m16_wait(3, 1, false, true) == 5 && m16_wait(3, 2, false, true) == 8 && m16_wait(3, 2, false, false) == 0 && m16_wait(3, 0, true, true) == 2, "");
static_assert(m16_wait(2, 0, false, true) == 1 && m16_wait(2, 1, false, true) == 4 && m16_wait(5, 4, false, true) == 10 && m16_wait(4, 3, false, true) == 8, "");

// 48-row Cout blocks (three 16-row blocks, NB = 8 columns per consumer wave: 3 x 8 accumulators).  A prefetching k-step requests the
// next one's NB + 3 fragments in the order A0 B0 A1 B1 A2 B2 B3 .. (one A and one B behind columns 0-2, one B behind the others);
// a burst (first k-step of a half) in the order A0 A1 A2 B0 B1 ...  lgkmcnt value in front of column n (-1: nothing new to wait for).
constexpr int m48_issued(int n) { return n <= 3 ? 2 * n : n + 3; }   // requests of a prefetching k-step in front of its column n
constexpr int m48_wait(int NB, int n, bool first, bool more) {
  const int total = NB + 3, inew = more ? m48_issued(n) : 0;
  if (first) return total - (4 + n) + inew;
  if (n == 0) return total - 5;
  if (n == 1) return -1;
  return total - (n + 4) + inew;
}
static_assert(m48_wait(8, 0, true, true) == 7 && m48_wait(8, 7, true, true) == 10 && m48_wait(8, 2, false, true) == 9 && m48_wait(8, 5, false, true) == 10 &&
              m48_wait(8, 7, false, false) == 0 && m48_wait(8, 2, false, false) == 5 && m48_wait(8, 0, false, true) == 6, "");

// WREG > 0: the layer has ONE Cout block of WREG K-chunks and the producers keep all of its packed weights in
// their (otherwise idle) registers -- 7 x 16 B per thread and chunk -- and refill the LDS chunk buffers with
// ds_write_b128 instead of LDS-DMA: 7 of the ~16 memory instructions a producer wave issues per stage disappear.
//
// C16 = 1 (round 4): the CONSUMERS run v_mfma_f32_16x16x32 instead of 32x32x16 -- same tile, same LDS images, same producers.
// Why: these layers run at the clock the chip grants under load, and on random data it grants the 16x16x32 stream more
// (tools_dev/micro/shape_bench.hip: this consumer loop beside 7 LDS-DMA per producer wave and stage, 3 015 vs 3 043 cycles per
// stage but 1.90 vs 1.71 GHz: +12 % FLOP/s; MI355X micro-architecture guide, "DVFS give-back" item 7).
//   * K = 32 of one MFMA is FOUR groups of 8 channels, and every 16-lane group q of a wave reads its own fragment address, so
//     the four groups may be any (plane, tap) pairs.  The packed weight image is already [group = 2 tap + plane][row][16 B], so
//     a k-step takes two taps of both planes: q & 1 = plane, q >> 1 = which tap of the pair.  Nothing about the producers, the
//     LDS layout or the host-side packing changes.
//   * 18 groups per stage are 4.5 k-steps: stages are processed in PAIRS (the host routes layers with an even number of
//     2-plane chunks here).  Over a pair the lanes with q >> 1 = 0 walk taps 0 3 6 2 8 | 0 3 6 5, the others 1 4 7 5 | 2 1 4 7 8:
//     k-step 4 straddles the stage barrier.  Its lower half-wave's fragments (tap 8 of the even stage) are requested before
//     that barrier and kept in registers across it; behind it the upper half-wave's (tap 2 of the odd stage) are read into
//     the same registers with EXEC = lanes 32-63.
//   * a wave owns MT/16 x 2 NR accumulators of 16 x 16 (6 x 6 x 4 = 144 registers, as before); the A fragments of a k-step are
//     double-buffered (all six stay live through its six columns), the B fragments sit in a ring of 2 NR + 1 registers; the
//     twelve reads of the next k-step are issued BETWEEN this k-step's MFMAs, at most one per two MFMAs (an in-order wave that
//     issues seven reads back to back leaves its matrix pipe idle: 3 360 vs 3 015 cycles per stage in shape_bench), and
//     waited for with counted lgkmcnt (LDS returns in order).
//   * fp32 summation order differs from the 32x32x16 form (taps pair up differently inside an MFMA); it is fixed for a layer
//     shape, so the invariance properties (batch position, batch size, eager = captured) hold as before.
// M16 > 0 (round 4): the Cout block is M16 16-row blocks instead of MR 32-row ones -- M16 = 3 with C16 = 8: 48-row blocks on 3 x 8
// accumulators per consumer wave (512-pixel tile groups), for the one stride-1 layer with 48 output channels and a deep K
// (transition1: 256 -> 48), which the 64-row form ran with a quarter of its MFMAs on padding rows.
// ADDR: 0 = every tensor of the launch is buffer-addressed (< 4 GiB), 1 = 64-bit pointers, 2 = decided at run time (p.in_bytes).  The
// 16x16x32-consumer variants are built for 0 and 1 separately: with both paths in one kernel the producers' stage loop is twice the
// code and spills 70 SGPRs (round 5).
// CW2 = 1 (round 6): TWO consumer waves per SIMD -- a 768-thread workgroup, waves 0-7 consumers (each C16 / 2 columns of the same
// C16 x 64-pixel tile group: 6 x 3 accumulators, 72 registers), waves 8-11 producers; three waves per SIMD leave 168 registers per
// wave, so the producers cannot hold the layer's weights (WREG = 0 only).  Why: with one consumer wave per SIMD the matrix pipe
// idles whenever that wave waits for fragments, runs its epilogue or sits at the stage barrier (busy 0.43-0.53);
// with two, one wave's waits sit under its partner's MFMAs.  Costs 1.5x the LDS fragment reads (9 per 18 MFMAs instead of 12 per 36).
template <int DT, int KS, int STRIDE, int MR, int NR, int WREG = 0, int C16 = 0, int M16 = 0, int ADDR = 2, int CW2 = 0>
__global__ __launch_bounds__(CW2 ? 768 : 512, CW2 ? 3 : 2) void conv_m32p_kernel(const ConvLaunch p) {
  static_assert(!CW2 || (C16 > 0 && C16 % 2 == 0 && WREG == 0 && M16 == 0), "two consumer waves per SIMD: 16x16x32 consumers, even column count, weights by LDS-DMA");
  constexpr int NCW = CW2 ? 8 : 4;             // consumer waves
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int MT = M16 > 0 ? 16 * M16 : 32 * MR;
  constexpr int KK = KS * KS;
  constexpr int MAXP = 4;
  constexpr int PXCAP = C16 ? 64 * C16 : 4 * NR * 32;   // pixel slots of a tile group (16x16x32 consumers: four waves x C16 columns of 16)
  constexpr int ROPL = MT / 8;                 // planes of a Cout block
  constexpr int NQ = (PXCAP + 255) / 256;      // retire-buffer pixels per producer thread and plane

  // LDS: [bias][W: 2 chunk buffers, or all chunks when the layer's weights stay resident][X x2][RO]
  const bool w_resident = p.nbuf_w != 2;          // nbuf_w = nchunks: one Cout block whose whole K fits
  float* bias_l = reinterpret_cast<float*>(smem);
  char* wl0 = smem + p.lds_bias;
  char* xl0 = wl0 + p.nbuf_w * p.lds_w;
  char* ro = xl0 + 2 * p.lds_x;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: role branch, LDS-DMA bases and M0 values stay scalar
  const bool producer = wave_all >= NCW;
  const int cw = wave_all;                     // consumer wave index, 0 .. NCW - 1 (16x16x32 consumers)
  const int wave = wave_all & 3, ptid = tid & 255;
  const int half = lane >> 5, r = lane & 31;
  const int HW = p.H * p.W;
  const int HP = p.halo_h * p.halo_w;
  const int npix = p.th * p.tw;
  const int nseg = p.nt;
  const int P = nseg * npix;
  const int ksteps_full = (p.cp >> 1) * KK;
  const int cout_planes = (p.cout + 7) >> 3;
  const size_t HoWo = (size_t)p.Ho * p.Wo;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const size_t chunk_wbytes = (size_t)ksteps_full * (2 * MT * 16);

  for (int i = tid; i < p.n_mblk * MT; i += (CW2 ? 768 : 512)) bias_l[i] = p.bias[i];

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int it_begin = wg * p.items_per_wg;
  const int it_end = min(p.items_total, it_begin + p.items_per_wg);

  auto pixel_of = [&](int pidx, int& ps, int& py, int& px) {   // flattened pixel -> (segment, y, x); ps < 0 = none
    if (pidx < P) {
      ps = fdiv(pidx, p.fd_npix);
      const int rem = pidx - ps * npix;
      py = fdiv(rem, p.fd_tw); px = rem - py * p.tw;
    } else {
      ps = -1; py = px = 0;
    }
  };
  auto decode_tile = [&](int it, int seg, int& img, int& oy0, int& ox0) {
    const int t = fdiv(it, p.fd_nmblk) * nseg + seg;
    if (seg < 0 || t >= p.tiles_total) { img = -1; oy0 = ox0 = 0; return; }
    img = fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = fdiv(rem, p.fd_tiles_x);
    oy0 = ty * p.th; ox0 = (rem - ty * p.tiles_x) * p.tw;
  };
  // number of stages at the head of a tile in which the previous tile's results are stored
  // (total_blocks doubles as a host-side override of the store-stage count for this kernel)
  const int n_st = p.total_blocks > 0 ? p.total_blocks : (p.nchunks >= 6 ? 2 : 1);

  if (producer) {
    // =====================================================================================
    // producers: all global-memory traffic
    // =====================================================================================
    int hs[MAXP], hy[MAXP], hx[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int hp = i * 256 + ptid;
      if (hp < nseg * HP) {
        hs[i] = fdiv(hp, p.fd_hp);
        const int rem = hp - hs[i] * HP;
        hy[i] = fdiv(rem, p.fd_halo_w); hx[i] = rem - hy[i] * p.halo_w;
      } else {
        hs[i] = -1; hy[i] = hx[i] = 0;
      }
    }
    // Retire-buffer traffic is split over the producer waves by pixel slot: slots [0, 256) belong to thread ptid.  The NX 64-slot
    // chunks past 256 (PXCAP = 320: one, 384: two) used to belong to waves 0 .. NX - 1 for EVERY plane, which gave those waves twice the
    // stores and residual loads of the others -- and their SIMDs' consumer waves a busier partner, at whose pace all eight waves then
    // met the stage barrier (round 5).  They now go round the waves by plane: NX = 2: waves 0, 1 on even planes, 2, 3 on odd ones;
    // NX = 1: wave pl & 3.  A thread still owns at most two slots (ptid, and 256 + 64 (wave & 1) + lane).
    constexpr int NX = (PXCAP > 256 && PXCAP < 512) ? (PXCAP - 256) / 64 : 0;
    static_assert(NX <= 2, "the retire-buffer slot rotation is written for one or two extra 64-slot chunks (PXCAP 320 / 384)");
    auto slot_of = [&](int k) -> int { return (k == 0 || NX == 0) ? k * 256 + ptid : 256 + (NX == 2 ? (wave & 1) : 0) * 64 + lane; };
    auto wslot_of = [&](int k) -> int { return (k == 0 || NX == 0) ? k * 256 + wave * 64 : 256 + (NX == 2 ? (wave & 1) : 0) * 64; };   // the wave's first slot
    auto moves = [&](int k, int pl) -> bool {   // wave-uniform: does this wave move pixel group k of plane pl?
      if (k == 0 || NX == 0) return k * 256 + wave * 64 < PXCAP;
      return NX == 2 ? ((pl & 1) == (wave >> 1)) : ((pl & 3) == wave);
    };
    int qs[NQ], qy[NQ], qx[NQ];   // retire-buffer pixels of this thread
#pragma unroll
    for (int k = 0; k < NQ; ++k) pixel_of(slot_of(k) < PXCAP ? slot_of(k) : P, qs[k], qy[k], qx[k]);

    // Tensors below 4 GiB are addressed through buffer descriptors (conv_pipe_kernel.h: dma16_buf): per-item 32-bit
    // lane offsets, chunk / plane displacement in an SGPR, padding lanes out of range.  Larger ones keep 64-bit pointers.
    const bool buf = ADDR == 2 ? p.in_bytes != 0 : ADDR == 0;
    const buf_rsrc_t rs_w = make_buf(p.wpk, (uint32_t)(p.n_mblk * p.nchunks * (int)chunk_wbytes));
    const buf_rsrc_t rs_in = make_buf(p.in, p.in_bytes), rs_res = make_buf(p.res ? p.res : p.out, p.out_bytes),
                     rs_out = make_buf(p.out, p.out_bytes);
    size_t xoff[MAXP];
    uint32_t xvo[MAXP];
    auto locate_halo = [&](int it) {
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        int img, oy0, ox0;
        decode_tile(it, hs[i], img, oy0, ox0);
        const int iy = oy0 * STRIDE - (KS / 2) + hy[i], ix = ox0 * STRIDE - (KS / 2) + hx[i];
        const bool ok = img >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        if (buf) xvo[i] = ok ? (uint32_t)(img * p.cin_planes * HW + (iy * p.W + ix)) * 16u : BUF_OOB;
        else xoff[i] = ok ? ((size_t)img * p.cin_planes * HW + (size_t)(iy * p.W + ix)) * 16 : ~(size_t)0;
      }
    };
    auto issue_x = [&](int cl, int xb) {
      const int c = cl;   // chunks in natural order: results do not depend on the workgroup (batch position)
      const char* inb = static_cast<const char*>(p.in) + (size_t)c * p.cp * HW * 16;
      char* xl = xl0 + xb * p.lds_x;
      if (buf) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
          if (hs[i] >= 0)
            for (int pl = 0; pl < p.cp; ++pl)
              dma16_buf(rs_in, xvo[i], (uint32_t)((c * p.cp + pl) * HW) * 16u, xl + pl * p.plane_stride + (i * 256 + wave * 64) * 16);
        return;
      }
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (hs[i] >= 0) {
          const bool ok = xoff[i] != ~(size_t)0;
          for (int pl = 0; pl < p.cp; ++pl) {
            const char* src = ok ? inb + xoff[i] + (size_t)pl * HW * 16 : static_cast<const char*>(p.zero16);
            dma16(src, xl + pl * p.plane_stride + (i * 256 + wave * 64) * 16);
          }
        }
      }
    };
    // the first tile's first halo chunk is requested before anything else: it comes from HBM and lands under the weight loads below
    if (it_begin < it_end) { locate_halo(it_begin); issue_x(0, 0); }
    constexpr int WSLOTS = (KK * 2 * MT * 16 + 4095) / 4096;   // 16-byte slots per thread of a 2-plane chunk
    u32x4 wreg[WREG > 0 ? WREG : 1][WSLOTS];
    if constexpr (WREG > 0) {
      const int nbytes = ksteps_full * (2 * MT * 16);
#pragma unroll
      for (int c = 0; c < WREG; ++c)
#pragma unroll
        for (int j = 0; j < WSLOTS; ++j) {
          const int mine = j * 4096 + ptid * 16;
          wreg[c][j] = mine < nbytes ? *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.wpk) + (size_t)c * chunk_wbytes + mine)
                                     : u32x4{0u, 0u, 0u, 0u};
        }
    }
    auto issue_w = [&](int it, int cl, int wb) {
      const int c = cl;   // chunks in natural order: results do not depend on the workgroup (batch position)
      const int nbytes = ksteps_full * (2 * MT * 16);
      if constexpr (WREG > 0) {   // registers -> LDS (same image as the DMA would write)
        char* wl = wl0 + wb * p.lds_w + ptid * 16;
        auto put = [&](auto cc) {
          constexpr int C = decltype(cc)::value;
          if constexpr (C < WREG) {
#pragma unroll
            for (int j = 0; j < WSLOTS; ++j)
              if (j * 4096 + ptid * 16 < nbytes) *reinterpret_cast<u32x4*>(wl + j * 4096) = wreg[C][j];
          }
        };
        switch (c) {
          case 0: put(std::integral_constant<int, 0>{}); break;
          case 1: put(std::integral_constant<int, 1>{}); break;
          case 2: put(std::integral_constant<int, 2>{}); break;
          case 3: put(std::integral_constant<int, 3>{}); break;
          case 4: put(std::integral_constant<int, 4>{}); break;
          default: put(std::integral_constant<int, 5>{}); break;
        }
        return;
      }
      const uint32_t wchunk = (uint32_t)(((it - fdiv(it, p.fd_nmblk) * p.n_mblk) * p.nchunks + c) * (int)chunk_wbytes);
      char* wl = wl0 + wb * p.lds_w;
      for (int o = 0; o < nbytes; o += 4096) {   // packed weights are far below 4 GiB: always buffer-addressed
        const int mine = o + ptid * 16;
        if (mine < nbytes) dma16_buf(rs_w, (uint32_t)ptid * 16u, wchunk + (uint32_t)o, wl + o + wave * 1024);
      }
    };
    // byte offset of (image, plane 0, pixel) in out / res for this thread's retire-buffer pixels
    auto locate_out = [&](int it, size_t* qb) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        int img, oy0, ox0;
        decode_tile(it, qs[k], img, oy0, ox0);
        const int oy = oy0 + qy[k], ox = ox0 + qx[k];
        const bool ok = img >= 0 && oy < p.Ho && ox < p.Wo;
        // buffer addressing: the 32-bit offset lives in the same slot (BUF_OOB for pixels outside the map)
        if (buf) qb[k] = ok ? (size_t)((uint32_t)(img * cout_planes * HoWo + oy * p.Wo + ox) * 16u) : (size_t)BUF_OOB;
        else qb[k] = ok ? ((size_t)img * cout_planes * HoWo + (size_t)oy * p.Wo + ox) * 16 : ~(size_t)0;
      }
    };
    // planes [pl0, pl1) of the retire buffer: residual rows of tile `it` -> RO (LDS-DMA)
    auto load_residual = [&](int it, const size_t* qb, int pl0, int pl1) {
      const int plane0 = (it - fdiv(it, p.fd_nmblk) * p.n_mblk) * ROPL;
      if (buf) {
        for (int pl = pl0; pl < pl1; ++pl)
#pragma unroll
          for (int k = 0; k < NQ; ++k)
            if (moves(k, pl))
              dma16_buf(rs_res, plane0 + pl < cout_planes ? (uint32_t)qb[k] : BUF_OOB, (uint32_t)((plane0 + pl) * HoWo) * 16u,
                        ro + ((pl * PXCAP) + wslot_of(k)) * 16);
        return;
      }
      for (int pl = pl0; pl < pl1; ++pl)
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (moves(k, pl)) {
            const bool ok = qb[k] != ~(size_t)0 && plane0 + pl < cout_planes;
            const char* src = ok ? static_cast<const char*>(p.res) + qb[k] + (size_t)(plane0 + pl) * HoWo * 16
                                 : static_cast<const char*>(p.zero16);
            dma16(src, ro + ((pl * PXCAP) + wslot_of(k)) * 16);
          }
    };
    // planes [pl0, pl1) of the retire buffer: results of tile `it` -> global
    auto store_results = [&](int it, const size_t* qb, int pl0, int pl1) {
      const int plane0 = (it - fdiv(it, p.fd_nmblk) * p.n_mblk) * ROPL;
      if (buf) {
        for (int pl = pl0; pl < pl1 && plane0 + pl < cout_planes; ++pl)
#pragma unroll
          for (int k = 0; k < NQ; ++k)
            if (moves(k, pl))     // pixels outside the map carry BUF_OOB: the store is dropped by the hardware
              store16_buf(rs_out, (uint32_t)qb[k], (uint32_t)((plane0 + pl) * HoWo) * 16u,
                          *reinterpret_cast<const u32x4*>(ro + ((pl * PXCAP) + slot_of(k)) * 16));
        return;
      }
      for (int pl = pl0; pl < pl1; ++pl)
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (moves(k, pl) && qb[k] != ~(size_t)0 && plane0 + pl < cout_planes) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(ro + ((pl * PXCAP) + slot_of(k)) * 16);
            *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + qb[k] + (size_t)(plane0 + pl) * HoWo * 16) = v;
          }
    };

    // memory instructions ahead of the consumers' MFMA stream -- for the 32x32x16 consumers (a 32-cycle MFMA leaves the issue port free
    // three quarters of the time; stride 2: 130 against 146 us without it).  Beside the 16x16x32 consumers, which issue twice as often,
    // the producers' address arithmetic at priority 3 delays MFMA issue: without it 96 -> 96 97.2 against 99.3 us, 192 -> 192 80.9 / 81.7,
    // 384 -> 384 88.8 / 88.5 inside the forward (round 5, development build, profiles/round5_priority_ab.txt)
    if (!SCP_DBG(p, 128) && !(C16 != 0 && M16 == 0)) __builtin_amdgcn_s_setprio(3);
    size_t qb_cur[NQ], qb_prev[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) qb_cur[k] = qb_prev[k] = buf ? (size_t)BUF_OOB : ~(size_t)0;
    int wc = 0, xb = 0;
    if (it_begin < it_end) {
      if (w_resident) {   // every chunk, in storage order, once per launch
        const int nbytes = p.nchunks * ksteps_full * (2 * MT * 16);
        for (int o = 0; o < nbytes; o += 4096) {
          const int mine = o + ptid * 16;
          if (mine < nbytes) dma16(static_cast<const char*>(p.wpk) + mine, wl0 + o + wave * 1024);
        }
      } else {
        issue_w(it_begin, 0, 0);
      }
    }
    __syncthreads();   // bias, stage 0 (compiler: vmcnt(0) + lgkmcnt(0) + barrier)

    const int n_ld = p.nchunks - 1 - n_st;   // stages that load residual rows
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return SCP_DBG(p, 8) ? __builtin_amdgcn_s_memtime() : 0ull; };
    for (int it = it_begin; it < it_end; ++it) {
      locate_out(it, qb_cur);
      for (int c = 0; c < p.nchunks; ++c, ++wc) {
        const unsigned long long t0 = now();
        const bool last = c == p.nchunks - 1;
        const int nit = last ? it + 1 : it, nc = last ? 0 : c + 1;
        // Issue order = expected latency, longest first: the retire-buffer traffic and the halos come from
        // HBM, the weights from L2; everything is waited for once, at the end of the stage.
        if (c < n_st) {
          if (it > it_begin && !SCP_DBG(p, 2)) store_results(it - 1, qb_prev, ROPL * c / n_st, ROPL * (c + 1) / n_st);
        } else if (!last && p.res && !SCP_DBG(p, 2)) {
          load_residual(it, qb_cur, ROPL * (c - n_st) / n_ld, ROPL * (c - n_st + 1) / n_ld);
        }
        const unsigned long long t1 = now();
        unsigned long long t2 = t1;
        if (nit < it_end) {   // DMA for the next stage
          if (nc == 0) locate_halo(nit);
          if (!SCP_DBG(p, 4)) issue_x(nc, xb ^ 1);
          t2 = now();
          if (!w_resident) issue_w(nit, nc, (wc + 1) & 1);
        }
        const unsigned long long t3 = now();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t4 = now();
        __builtin_amdgcn_s_barrier();
        if SCP_DBG(p, 8) {   // producers: [retire-buffer traffic][locate + X issue][W issue][wait][barrier]
          const unsigned long long t5 = now();
          tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3; tph[4] += t5 - t4;
          if (c < n_st) tph[5] += t4 - t3;   // ... of which in the stages that store the previous tile
        }
        xb ^= 1;
      }
#pragma unroll
      for (int k = 0; k < NQ; ++k) qb_prev[k] = qb_cur[k];
    }
    if (it_begin < it_end && !SCP_DBG(p, 2)) store_results(it_end - 1, qb_prev, 0, ROPL);   // drain the last tile
    if (SCP_DBG(p, 8) && SCP_DBG_BUF(p) && lane == 0)
      for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + 4 + wave) * 6 + k] = tph[k];   // slots 4-7: the producer waves
  } else if constexpr (C16 != 0) {
    // =====================================================================================
    // consumers, 16x16x32 form (see the header comment): LDS reads, MFMAs, epilogue into the retire buffer
    // =====================================================================================
    static_assert(KS == 3 && (MT == 96 || MT == 48), "16x16x32 consumers: 3x3 layers, 96- or 48-row Cout blocks");
    constexpr int MB = MT / 16;          // 16-row blocks of the Cout block
    constexpr int NB = CW2 ? C16 / 2 : C16;   // 16-pixel columns of a consumer wave (PXCAP = 16 NB NCW pixel slots per tile group)
    constexpr int RING = NB + 1;
    constexpr int TAPB = 2 * MT * 16;    // bytes of one tap (both planes) in the packed weight image
    static_assert(MB == 6 ? (NB == 2 || NB == 3 || NB == 4 || NB == 5 || NB == 6) : (MB == 3 && NB == 8), "16x16x32 consumers: column counts with a prefetch schedule");
    typedef f32x4 acc_t;
    // Lane (q = lane >> 4, l15 = lane & 15): plane q & 1 of a plane pair; class q >> 1 picks the tap of a k-step's pair.
    // Pairs are chosen so that the two classes' fragment addresses differ by a constant: one pixel (taps kx, kx + 1 of a row) or
    // one row (taps (ky, 2), (ky + 1, 2)), so a B address is  pixoffq[n] + class shift + (uniform: buffer + tap)  -- one v_add3
    // with the uniform part in an SGPR -- and an A address is a per-half base + an immediate.  Over a PAIR of plane pairs
    // (32 channels: nine k-steps; m16_tap0 / m16_rowpair):
    //   k-step    0      1      2      3      4 (straddle)        5      6      7      8
    //   taps    0 | 1  3 | 4  6 | 7  2 | 5  8 even | 2 odd       0 | 1  3 | 4  6 | 7  5 | 8
    // With 2-plane stages (p.cp == 2) the even and the odd plane pair are two STAGES and k-step 4 straddles their barrier; with
    // deeper stages (small batches) both halves read the same stage's buffers.  Either way a pixel's
    // fp32 sums are formed by the same MFMAs in the same order, whatever the tile, the column count or the stage depth.
    int pixoffq[NB];   // byte offset of this lane's pixel (column n) in its plane of a staged plane pair
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      int ps, py, px;
      pixel_of((cw * NB + n) * 16 + (lane & 15), ps, py, px);
      pixoffq[n] = (ps >= 0 ? (ps * HP + (py * STRIDE) * p.halo_w + px * STRIDE) * 16 : 0) + ((lane >> 4) & 1) * p.plane_stride;
    }
    const uint32_t hw16 = (uint32_t)p.halo_w * 16u;
    const uint32_t sh_px = (uint32_t)(lane >> 5) * 16u, sh_row = (uint32_t)(lane >> 5) * hw16;   // class shifts of the B address
    acc_t acc[MB][NB];
    frag_t a0[MB], a1[MB], bR[RING];
    const bool wave_idle = cw * NB * 16 >= P;   // wave-uniform
    if (SCP_DBG(p, 512)) __builtin_amdgcn_s_setprio(3);   // (development: consumers at the producers' priority -- arbitration by age)
    const int npp = p.cp >> 1;                    // plane pairs per stage: 1 (the halves of a pair are two stages), 2 or 4
    int wc = 0, xb = 0;
    __syncthreads();   // matches the producers' prologue barrier

    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return SCP_DBG(p, 8) ? __builtin_amdgcn_s_memtime() : 0ull; };
    auto add3 = [](uint32_t a, uint32_t b, uint32_t c_uniform) -> uint32_t {   // opaque to the compiler: dozens of hoisted addresses otherwise
      uint32_t d;
      asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
      return d;
    };
    // request fragment M of k-step S (bases of the half that k-step reads) / the B fragment of column N of k-step S
    auto rd_a = [&](auto s_, auto m_, frag_t* dst, uint32_t wa1, uint32_t wa3) {
      constexpr int S = decltype(s_)::value, M = decltype(m_)::value;
      lds_read16<m16_tap0(S) * TAPB + M * 256>(dst[M], m16_rowpair(S) ? wa3 : wa1);
    };
    auto rd_b = [&](auto s_, auto n_, uint32_t xl) {
      constexpr int S = decltype(s_)::value, N = decltype(n_)::value;
      lds_read16<0>(bR[m16_ring(N, S, RING)], add3((uint32_t)pixoffq[N], m16_rowpair(S) ? sh_row : sh_px,
                                                    xl + (uint32_t)(m16_tap0(S) / 3) * hw16 + (uint32_t)(m16_tap0(S) % 3) * 16u));
    };
    // One column of a k-step: wait (counted) for its B fragment, six MFMAs, and between them this column's share of the next
    // k-step's requests (m16_na of its A fragments, its B fragment of this column); wa1n / wa3n / xln: bases of k-step S + 1.
    auto kcol = [&](auto s_, auto n_, auto more_, auto first_, frag_t* CA, frag_t* NA, uint32_t wa1n, uint32_t wa3n, uint32_t xln) {
      constexpr int S = decltype(s_)::value, N = decltype(n_)::value;
      constexpr bool MORE = decltype(more_)::value, FIRST = decltype(first_)::value;
      if constexpr (MB == 3) {   // three MFMAs per column: one A request (columns 0-2) and one B request of the next k-step between them
        constexpr int W3 = m48_wait(NB, N, FIRST, MORE);
        if constexpr (W3 >= 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(W3 < 0 ? 0 : W3) : "memory");
        if constexpr (N == 0) {
#pragma unroll
          for (int m = 0; m < MB; ++m) lds_landed(CA[m]);
        }
        frag_t& b3 = bR[m16_ring(N, S, RING)];
        lds_landed(b3);
        using SN3 = std::integral_constant<int, S + 1>;
        mfma16_acc<T>(acc[0][N], CA[0], b3);
        if constexpr (MORE && N < 3) rd_a(SN3{}, std::integral_constant<int, N < 3 ? N : 0>{}, NA, wa1n, wa3n);
        mfma16_acc<T>(acc[1][N], CA[1], b3);
        if constexpr (MORE) rd_b(SN3{}, n_, xln);
        mfma16_acc<T>(acc[2][N], CA[2], b3);
      } else {
      constexpr int W = m16_wait(NB, N, FIRST, MORE);
      if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(W < 0 ? 0 : W) : "memory");
      if constexpr (N == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) lds_landed(CA[m]);
      }
      frag_t& b = bR[m16_ring(N, S, RING)];
      lds_landed(b);
      constexpr int NA_ = MORE ? m16_na(NB, N) : 0, AF = m16_afirst(NB, N);
      using SN = std::integral_constant<int, S + 1>;
      mfma16_acc<T>(acc[0][N], CA[0], b);
      if constexpr (NA_ == 3) rd_a(SN{}, std::integral_constant<int, AF < 6 ? AF : 0>{}, NA, wa1n, wa3n);
      mfma16_acc<T>(acc[1][N], CA[1], b);
      if constexpr (NA_ == 1 || NA_ == 2) rd_a(SN{}, std::integral_constant<int, AF < 6 ? AF : 0>{}, NA, wa1n, wa3n);
      mfma16_acc<T>(acc[2][N], CA[2], b);
      if constexpr (NA_ == 3) rd_a(SN{}, std::integral_constant<int, AF + 1 < 6 ? AF + 1 : 0>{}, NA, wa1n, wa3n);
      mfma16_acc<T>(acc[3][N], CA[3], b);
      if constexpr (NA_ >= 2) rd_a(SN{}, std::integral_constant<int, AF + NA_ - 1 < 6 ? AF + NA_ - 1 : 0>{}, NA, wa1n, wa3n);
      mfma16_acc<T>(acc[4][N], CA[4], b);
      if constexpr (MORE) rd_b(SN{}, n_, xln);
      mfma16_acc<T>(acc[5][N], CA[5], b);
      }
    };
    auto kstep = [&](auto s_, auto more_, auto first_, frag_t* CA, frag_t* NA, uint32_t wa1n, uint32_t wa3n, uint32_t xln) {
      kcol(s_, std::integral_constant<int, 0>{}, more_, first_, CA, NA, wa1n, wa3n, xln);
      kcol(s_, std::integral_constant<int, 1>{}, more_, first_, CA, NA, wa1n, wa3n, xln);
      if constexpr (NB > 2) kcol(s_, std::integral_constant<int, 2>{}, more_, first_, CA, NA, wa1n, wa3n, xln);
      if constexpr (NB > 3) kcol(s_, std::integral_constant<int, 3>{}, more_, first_, CA, NA, wa1n, wa3n, xln);
      if constexpr (NB > 4) kcol(s_, std::integral_constant<int, 4>{}, more_, first_, CA, NA, wa1n, wa3n, xln);
      if constexpr (NB > 5) kcol(s_, std::integral_constant<int, 5>{}, more_, first_, CA, NA, wa1n, wa3n, xln);
      if constexpr (NB > 6) { kcol(s_, std::integral_constant<int, 6>{}, more_, first_, CA, NA, wa1n, wa3n, xln); kcol(s_, std::integral_constant<int, 7>{}, more_, first_, CA, NA, wa1n, wa3n, xln); }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>; using I8 = std::integral_constant<int, 8>;
    using Yes = std::true_type; using No = std::false_type;
    // per-half A bases, formed behind an opaque copy of the lane id so that they are not kept live across halves
#define M16_BASES(WL)                                                                                    \
      uint32_t lane_s = (uint32_t)lane;                                                                  \
      asm volatile("" : "+v"(lane_s));                                                                   \
      const uint32_t wa0 = (WL) + ((((lane_s >> 4) & 1) * MT + (lane_s & 15)) * 16);                     \
      const uint32_t wa1 = wa0 + (lane_s >> 5) * TAPB, wa3 = wa0 + (lane_s >> 5) * (3 * TAPB);
    if (wave_idle) {
      // a consumer wave whose pixel slots all lie past the work item's pixels (two 12 x 12 images fill 288 of 384 slots) has
      // nothing to compute: it only keeps the barriers.  (Its own loop: a conditional around the k-steps would keep every
      // fragment register live across the epilogue.)
      for (int it = it_begin; it < it_end; ++it)
        for (int c = 0; c < p.nchunks; ++c) __builtin_amdgcn_s_barrier();
    } else
    for (int it = it_begin; it < it_end; ++it) {
      const int mb = it - fdiv(it, p.fd_nmblk) * p.n_mblk;
      for (int c = 0; c < p.nchunks; c += (npp == 1 ? 2 : 1)) {
        const bool last = c + (npp == 1 ? 2 : 1) >= p.nchunks;
        unsigned long long t0 = now(), t_mid = 0;
        for (int u = 0; 2 * u < npp || u == 0; ++u) {   // pairs of plane pairs of this stage (one, split over two stages, when npp == 1)
          // ---------------- even plane pair: taps 0|1 3|4 6|7 2|5, and the lower half of the straddling k-step (tap 8) ----------------
          {
            const uint32_t xl = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(xl0 + xb * p.lds_x) + (uint32_t)(4 * u) * (uint32_t)p.plane_stride);
            M16_BASES((uint32_t)(size_t)(wl0 + (w_resident ? c : (wc & 1)) * p.lds_w) + (uint32_t)(2 * u) * (9 * TAPB))
            (void)wa0;
            {   // the first fragments fly while the accumulators are initialised
              rd_a(I0{}, I0{}, a0, wa1, wa3); rd_a(I0{}, I1{}, a0, wa1, wa3); rd_a(I0{}, I2{}, a0, wa1, wa3);
              if constexpr (MB > 3) { rd_a(I0{}, I3{}, a0, wa1, wa3); rd_a(I0{}, I4{}, a0, wa1, wa3); rd_a(I0{}, I5{}, a0, wa1, wa3); }
              rd_b(I0{}, I0{}, xl); rd_b(I0{}, I1{}, xl);
              if constexpr (NB > 2) rd_b(I0{}, I2{}, xl);
              if constexpr (NB > 3) rd_b(I0{}, I3{}, xl);
              if constexpr (NB > 4) rd_b(I0{}, I4{}, xl);
              if constexpr (NB > 5) rd_b(I0{}, I5{}, xl);
              if constexpr (NB > 6) { rd_b(I0{}, I6{}, xl); rd_b(I0{}, I7{}, xl); }
            }
            if (c == 0 && u == 0) {   // accumulators start at the bias of their rows: row(j) = 16 m + 4 q + j
              const float* bq = bias_l + mb * MT + 4 * (lane_s >> 4);
#pragma unroll
              for (int m = 0; m < MB; ++m) {
                const float4 b4 = *reinterpret_cast<const float4*>(bq + m * 16);
#pragma unroll
                for (int n = 0; n < NB; ++n) { acc[m][n][0] = b4.x; acc[m][n][1] = b4.y; acc[m][n][2] = b4.z; acc[m][n][3] = b4.w; }
              }
              // VALU write -> MFMA SrcC needs wait states the hazard recogniser cannot see around inline-asm MFMAs (conv_device.h)
#pragma unroll
              for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) mfma_input_fence<false>(acc[m][n]);
            }
            if (!SCP_DBG(p, 1)) {   // (development build: bit 1 skips the MFMA loops -- wrong results, timing ablation; folds away in the product)
              kstep(I0{}, Yes{}, Yes{}, a0, a1, wa1, wa3, xl); kstep(I1{}, Yes{}, No{}, a1, a0, wa1, wa3, xl);
              kstep(I2{}, Yes{}, No{}, a0, a1, wa1, wa3, xl);
              kstep(I3{}, Yes{}, No{}, a1, a0, wa1, wa3, xl);   // requests k-step 4 from THIS half's buffers: tap 8 for class 0 (class 1's lanes are refilled below)
#pragma unroll
              for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) mfma_result_fence<false>(acc[m][n]);
            }
          }
          if (npp == 1) {   // the halves are two stages: the straddling fragments land before the producers may refill these buffers
            const unsigned long long t1 = now();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t2 = now();
            asm volatile("s_barrier" ::: "memory");   // (an asm with a memory clobber, not the builtin: nothing the compiler schedules may cross it)
            if SCP_DBG(p, 8) {   // slots 3-5: the barrier wait again, by class of the stage that ends -- stores the previous tile / loads residual rows / last
              const unsigned long long t3 = now(); tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2;
              tph[3 + (c < n_st ? 0 : c == p.nchunks - 1 ? 2 : 1)] += t3 - t2; t0 = t3;
            }
            xb ^= 1; ++wc;
          }
          {
#pragma unroll
            for (int m = 0; m < MB; ++m) lds_landed(a0[m]);
#pragma unroll
            for (int n = 0; n < NB; ++n) lds_landed(bR[m16_ring(n, 4, RING)]);
          }
          // ---------------- odd plane pair: upper half of the straddling k-step (tap 2), then taps 0|1 3|4 6|7 5|8 ----------------
          {
            const uint32_t xl = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(xl0 + xb * p.lds_x) +
                                                              (npp == 1 ? 0u : (uint32_t)(4 * u + 2) * (uint32_t)p.plane_stride));
            M16_BASES((uint32_t)(size_t)(wl0 + (w_resident ? c + (npp == 1 ? 1 : 0) : (wc & 1)) * p.lds_w) + (npp == 1 ? 0u : (uint32_t)(2 * u + 1) * (9 * TAPB)))
            {
              // lanes 32-63 (class 1): their group of k-step 4 is tap 2 of this plane pair; lanes 0-31 keep what they hold.  ONE asm
              // statement, so that nothing the compiler schedules runs under the narrowed EXEC.
              uint32_t xa[NB];
#pragma unroll
              for (int n = 0; n < NB; ++n) xa[n] = add3((uint32_t)pixoffq[n], 0u, xl + 32u);
              unsigned long long keep;
              if constexpr (NB == 6) {
                asm volatile(
                  "s_mov_b64 %[keep], exec\n\t"
                  "s_mov_b32 exec_lo, 0\n\t"
                  "ds_read_b128 %[a0], %[wm] offset:%[o0]\n\t"
                  "ds_read_b128 %[a1], %[wm] offset:%[o1]\n\t"
                  "ds_read_b128 %[a2], %[wm] offset:%[o2]\n\t"
                  "ds_read_b128 %[a3], %[wm] offset:%[o3]\n\t"
                  "ds_read_b128 %[a4], %[wm] offset:%[o4]\n\t"
                  "ds_read_b128 %[a5], %[wm] offset:%[o5]\n\t"
                  "ds_read_b128 %[b0], %[x0]\n\t"
                  "ds_read_b128 %[b1], %[x1]\n\t"
                  "ds_read_b128 %[b2], %[x2]\n\t"
                  "ds_read_b128 %[b3], %[x3]\n\t"
                  "ds_read_b128 %[b4], %[x4]\n\t"
                  "ds_read_b128 %[b5], %[x5]\n\t"
                  "s_mov_b64 exec, %[keep]"
                  : [keep] "=&s"(keep), [a0] "+v"(a0[0]), [a1] "+v"(a0[1]), [a2] "+v"(a0[2]), [a3] "+v"(a0[3]), [a4] "+v"(a0[4]), [a5] "+v"(a0[5]),
                    [b0] "+v"(bR[m16_ring(0, 4, RING)]), [b1] "+v"(bR[m16_ring(1, 4, RING)]), [b2] "+v"(bR[m16_ring(2, 4, RING)]), [b3] "+v"(bR[m16_ring(3, 4, RING)]), [b4] "+v"(bR[m16_ring(4, 4, RING)]), [b5] "+v"(bR[m16_ring(5, 4, RING)])
                  : [wm] "v"(wa0), [x0] "v"(xa[0]), [x1] "v"(xa[1]), [x2] "v"(xa[2]), [x3] "v"(xa[3]), [x4] "v"(xa[4]), [x5] "v"(xa[5]),
                    [o0] "n"(2 * TAPB + 0), [o1] "n"(2 * TAPB + 256), [o2] "n"(2 * TAPB + 512), [o3] "n"(2 * TAPB + 768), [o4] "n"(2 * TAPB + 1024), [o5] "n"(2 * TAPB + 1280)
                  : "memory");
              } else if constexpr (NB == 8) {   // 48-row blocks: three A fragments (2 TAPB + m * 256 with TAPB = 2 * 48 * 16), eight B
                asm volatile(
                  "s_mov_b64 %[keep], exec\n\t"
                  "s_mov_b32 exec_lo, 0\n\t"
                  "ds_read_b128 %[a0], %[wm] offset:3072\n\t"
                  "ds_read_b128 %[a1], %[wm] offset:3328\n\t"
                  "ds_read_b128 %[a2], %[wm] offset:3584\n\t"
                  "ds_read_b128 %[b0], %[x0]\n\t"
                  "ds_read_b128 %[b1], %[x1]\n\t"
                  "ds_read_b128 %[b2], %[x2]\n\t"
                  "ds_read_b128 %[b3], %[x3]\n\t"
                  "ds_read_b128 %[b4], %[x4]\n\t"
                  "ds_read_b128 %[b5], %[x5]\n\t"
                  "ds_read_b128 %[b6], %[x6]\n\t"
                  "ds_read_b128 %[b7], %[x7]\n\t"
                  "s_mov_b64 exec, %[keep]"
                  : [keep] "=&s"(keep), [a0] "+v"(a0[0]), [a1] "+v"(a0[1]), [a2] "+v"(a0[2]),
                    [b0] "+v"(bR[m16_ring(0, 4, RING)]), [b1] "+v"(bR[m16_ring(1, 4, RING)]), [b2] "+v"(bR[m16_ring(2, 4, RING)]), [b3] "+v"(bR[m16_ring(3, 4, RING)]),
                    [b4] "+v"(bR[m16_ring(4, 4, RING)]), [b5] "+v"(bR[m16_ring(5, 4, RING)]), [b6] "+v"(bR[m16_ring(NB == 8 ? 6 : 0, 4, RING)]), [b7] "+v"(bR[m16_ring(NB == 8 ? 7 : 0, 4, RING)])
                  : [wm] "v"(wa0), [x0] "v"(xa[0]), [x1] "v"(xa[1]), [x2] "v"(xa[2]), [x3] "v"(xa[3]), [x4] "v"(xa[4]), [x5] "v"(xa[5]),
                    [x6] "v"(xa[NB == 8 ? 6 : 0]), [x7] "v"(xa[NB == 8 ? 7 : 0])
                  : "memory");
                static_assert(NB != 8 || 2 * TAPB == 3072, "");
              } else if constexpr (NB == 5) {
                asm volatile(
                  "s_mov_b64 %[keep], exec\n\t"
                  "s_mov_b32 exec_lo, 0\n\t"
                  "ds_read_b128 %[a0], %[wm] offset:%[o0]\n\t"
                  "ds_read_b128 %[a1], %[wm] offset:%[o1]\n\t"
                  "ds_read_b128 %[a2], %[wm] offset:%[o2]\n\t"
                  "ds_read_b128 %[a3], %[wm] offset:%[o3]\n\t"
                  "ds_read_b128 %[a4], %[wm] offset:%[o4]\n\t"
                  "ds_read_b128 %[a5], %[wm] offset:%[o5]\n\t"
                  "ds_read_b128 %[b0], %[x0]\n\t"
                  "ds_read_b128 %[b1], %[x1]\n\t"
                  "ds_read_b128 %[b2], %[x2]\n\t"
                  "ds_read_b128 %[b3], %[x3]\n\t"
                  "ds_read_b128 %[b4], %[x4]\n\t"
                  "s_mov_b64 exec, %[keep]"
                  : [keep] "=&s"(keep), [a0] "+v"(a0[0]), [a1] "+v"(a0[1]), [a2] "+v"(a0[2]), [a3] "+v"(a0[3]), [a4] "+v"(a0[4]), [a5] "+v"(a0[5]),
                    [b0] "+v"(bR[m16_ring(0, 4, RING)]), [b1] "+v"(bR[m16_ring(1, 4, RING)]), [b2] "+v"(bR[m16_ring(2, 4, RING)]), [b3] "+v"(bR[m16_ring(3, 4, RING)]), [b4] "+v"(bR[m16_ring(4, 4, RING)])
                  : [wm] "v"(wa0), [x0] "v"(xa[0]), [x1] "v"(xa[1]), [x2] "v"(xa[2]), [x3] "v"(xa[3]), [x4] "v"(xa[4]),
                    [o0] "n"(2 * TAPB + 0), [o1] "n"(2 * TAPB + 256), [o2] "n"(2 * TAPB + 512), [o3] "n"(2 * TAPB + 768), [o4] "n"(2 * TAPB + 1024), [o5] "n"(2 * TAPB + 1280)
                  : "memory");
              } else if constexpr (NB == 4) {
                asm volatile(
                  "s_mov_b64 %[keep], exec\n\t"
                  "s_mov_b32 exec_lo, 0\n\t"
                  "ds_read_b128 %[a0], %[wm] offset:%[o0]\n\t"
                  "ds_read_b128 %[a1], %[wm] offset:%[o1]\n\t"
                  "ds_read_b128 %[a2], %[wm] offset:%[o2]\n\t"
                  "ds_read_b128 %[a3], %[wm] offset:%[o3]\n\t"
                  "ds_read_b128 %[a4], %[wm] offset:%[o4]\n\t"
                  "ds_read_b128 %[a5], %[wm] offset:%[o5]\n\t"
                  "ds_read_b128 %[b0], %[x0]\n\t"
                  "ds_read_b128 %[b1], %[x1]\n\t"
                  "ds_read_b128 %[b2], %[x2]\n\t"
                  "ds_read_b128 %[b3], %[x3]\n\t"
                  "s_mov_b64 exec, %[keep]"
                  : [keep] "=&s"(keep), [a0] "+v"(a0[0]), [a1] "+v"(a0[1]), [a2] "+v"(a0[2]), [a3] "+v"(a0[3]), [a4] "+v"(a0[4]), [a5] "+v"(a0[5]),
                    [b0] "+v"(bR[m16_ring(0, 4, RING)]), [b1] "+v"(bR[m16_ring(1, 4, RING)]), [b2] "+v"(bR[m16_ring(2, 4, RING)]), [b3] "+v"(bR[m16_ring(3, 4, RING)])
                  : [wm] "v"(wa0), [x0] "v"(xa[0]), [x1] "v"(xa[1]), [x2] "v"(xa[2]), [x3] "v"(xa[3]),
                    [o0] "n"(2 * TAPB + 0), [o1] "n"(2 * TAPB + 256), [o2] "n"(2 * TAPB + 512), [o3] "n"(2 * TAPB + 768), [o4] "n"(2 * TAPB + 1024), [o5] "n"(2 * TAPB + 1280)
                  : "memory");
              } else if constexpr (NB == 3) {
                asm volatile(
                  "s_mov_b64 %[keep], exec\n\t"
                  "s_mov_b32 exec_lo, 0\n\t"
                  "ds_read_b128 %[a0], %[wm] offset:%[o0]\n\t"
                  "ds_read_b128 %[a1], %[wm] offset:%[o1]\n\t"
                  "ds_read_b128 %[a2], %[wm] offset:%[o2]\n\t"
                  "ds_read_b128 %[a3], %[wm] offset:%[o3]\n\t"
                  "ds_read_b128 %[a4], %[wm] offset:%[o4]\n\t"
                  "ds_read_b128 %[a5], %[wm] offset:%[o5]\n\t"
                  "ds_read_b128 %[b0], %[x0]\n\t"
                  "ds_read_b128 %[b1], %[x1]\n\t"
                  "ds_read_b128 %[b2], %[x2]\n\t"
                  "s_mov_b64 exec, %[keep]"
                  : [keep] "=&s"(keep), [a0] "+v"(a0[0]), [a1] "+v"(a0[1]), [a2] "+v"(a0[2]), [a3] "+v"(a0[3]), [a4] "+v"(a0[4]), [a5] "+v"(a0[5]),
                    [b0] "+v"(bR[m16_ring(0, 4, RING)]), [b1] "+v"(bR[m16_ring(1, 4, RING)]), [b2] "+v"(bR[m16_ring(NB == 3 ? 2 : 0, 4, RING)])
                  : [wm] "v"(wa0), [x0] "v"(xa[0]), [x1] "v"(xa[1]), [x2] "v"(xa[NB == 3 ? 2 : 0]),
                    [o0] "n"(2 * TAPB + 0), [o1] "n"(2 * TAPB + 256), [o2] "n"(2 * TAPB + 512), [o3] "n"(2 * TAPB + 768), [o4] "n"(2 * TAPB + 1024), [o5] "n"(2 * TAPB + 1280)
                  : "memory");
              } else if constexpr (NB == 2) {
                asm volatile(
                  "s_mov_b64 %[keep], exec\n\t"
                  "s_mov_b32 exec_lo, 0\n\t"
                  "ds_read_b128 %[a0], %[wm] offset:%[o0]\n\t"
                  "ds_read_b128 %[a1], %[wm] offset:%[o1]\n\t"
                  "ds_read_b128 %[a2], %[wm] offset:%[o2]\n\t"
                  "ds_read_b128 %[a3], %[wm] offset:%[o3]\n\t"
                  "ds_read_b128 %[a4], %[wm] offset:%[o4]\n\t"
                  "ds_read_b128 %[a5], %[wm] offset:%[o5]\n\t"
                  "ds_read_b128 %[b0], %[x0]\n\t"
                  "ds_read_b128 %[b1], %[x1]\n\t"
                  "s_mov_b64 exec, %[keep]"
                  : [keep] "=&s"(keep), [a0] "+v"(a0[0]), [a1] "+v"(a0[1]), [a2] "+v"(a0[2]), [a3] "+v"(a0[3]), [a4] "+v"(a0[4]), [a5] "+v"(a0[5]),
                    [b0] "+v"(bR[m16_ring(0, 4, RING)]), [b1] "+v"(bR[m16_ring(1, 4, RING)])
                  : [wm] "v"(wa0), [x0] "v"(xa[0]), [x1] "v"(xa[1]),
                    [o0] "n"(2 * TAPB + 0), [o1] "n"(2 * TAPB + 256), [o2] "n"(2 * TAPB + 512), [o3] "n"(2 * TAPB + 768), [o4] "n"(2 * TAPB + 1024), [o5] "n"(2 * TAPB + 1280)
                  : "memory");
              }
              if (!SCP_DBG(p, 1)) {
                kstep(I4{}, Yes{}, Yes{}, a0, a1, wa1, wa3, xl); kstep(I5{}, Yes{}, No{}, a1, a0, wa1, wa3, xl);
                kstep(I6{}, Yes{}, No{}, a0, a1, wa1, wa3, xl); kstep(I7{}, Yes{}, No{}, a1, a0, wa1, wa3, xl);
                kstep(I8{}, No{}, No{}, a0, a1, wa1, wa3, xl);
              }
#pragma unroll
              for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) mfma_result_fence<false>(acc[m][n]);
            }
          }
        }
        t_mid = now();
        if (last) {
          // last MFMA's result visible to the VALU -- only the tile's last stage is followed by VALU reads of the accumulators
          asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) mfma_result_fence<false>(acc[m][n]);
          // retire: RO <- ReLU(acc + residual), in place.  A lane holds rows 16 m + 4 q .. + 3 of its pixel: channels 4 (q & 1) .. + 3
          // of output plane 2 m + (q >> 1), i.e. one 8-byte half-slot per accumulator -- no lane exchange, no separate bias add.
          const uint32_t relu_floor = p.relu ? 0u : 0x80008000u;
          // (the slot address is formed HERE, behind an opaque copy of the lane id: hoisted out of the tile loop it would be
          // live across the k-steps, which have no register to spare)
          uint32_t lane_e = (uint32_t)lane;
          asm volatile("" : "+v"(lane_e));
          char* const slot0 = ro + ((((lane_e >> 5) & 1) * PXCAP) + cw * NB * 16 + (lane_e & 15)) * 16 + 8 * ((lane_e >> 4) & 1);
          auto slot = [&](int m, int n) -> char* { return slot0 + ((2 * m) * PXCAP + n * 16) * 16; };
          if (!p.res) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
              for (int n = 0; n < NB; ++n) {
                uint2 o;
                o.x = relu2_16(pack2<T>(acc[m][n][0], acc[m][n][1]), relu_floor);
                o.y = relu2_16(pack2<T>(acc[m][n][2], acc[m][n][3]), relu_floor);
                *reinterpret_cast<uint2*>(slot(m, n)) = o;     // padding pixels write their own slots too: the producers never store those
              }
          } else {
            // residual half-slots are read one row block ahead of the one being finalised (the in-place writes would otherwise
            // order every read behind the previous write)
            if constexpr (NB > 6) {   // 3 x 8 form: four columns at a time (no row-ahead prefetch)
#pragma unroll
              for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n0 = 0; n0 < NB; n0 += 4) {
                  uint2 r4[4];
#pragma unroll
                  for (int k = 0; k < 4; ++k) r4[k] = *reinterpret_cast<const uint2*>(slot(m, n0 + k));
#pragma unroll
                  for (int k = 0; k < 4; ++k) {
                    const int n = n0 + k;
                    const uint2 x = r4[k];
                    const float v0 = acc[m][n][0] + from_bits<T>(x.x & 0xffff), v1 = acc[m][n][1] + from_bits<T>(x.x >> 16);
                    const float v2 = acc[m][n][2] + from_bits<T>(x.y & 0xffff), v3 = acc[m][n][3] + from_bits<T>(x.y >> 16);
                    uint2 o;
                    o.x = relu2_16(pack2<T>(v0, v1), relu_floor); o.y = relu2_16(pack2<T>(v2, v3), relu_floor);
                    *reinterpret_cast<uint2*>(slot(m, n)) = o;
                  }
                }
            } else {
            uint2 rr[2][NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) rr[0][n] = *reinterpret_cast<const uint2*>(slot(0, n));
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              if (m + 1 < MB) {
#pragma unroll
                for (int n = 0; n < NB; ++n) rr[(m + 1) & 1][n] = *reinterpret_cast<const uint2*>(slot(m + 1, n));
              }
#pragma unroll
              for (int n = 0; n < NB; ++n) {
                const uint2 x = rr[m & 1][n];
                const float v0 = acc[m][n][0] + from_bits<T>(x.x & 0xffff), v1 = acc[m][n][1] + from_bits<T>(x.x >> 16);
                const float v2 = acc[m][n][2] + from_bits<T>(x.y & 0xffff), v3 = acc[m][n][3] + from_bits<T>(x.y >> 16);
                uint2 o;
                o.x = relu2_16(pack2<T>(v0, v1), relu_floor); o.y = relu2_16(pack2<T>(v2, v3), relu_floor);
                *reinterpret_cast<uint2*>(slot(m, n)) = o;
              }
            }
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = now();
        asm volatile("s_barrier" ::: "memory");
        if SCP_DBG(p, 8) {
          const unsigned long long t3 = now(); tph[0] += t_mid - t0; tph[1] += t2 - t_mid; tph[2] += t3 - t2;
          const int ce = c + (npp == 1 ? 1 : 0);
          tph[3 + (ce < n_st ? 0 : ce == p.nchunks - 1 ? 2 : 1)] += t3 - t2;
        }
        xb ^= 1; ++wc;
      }
    }
#undef M16_BASES
    if (SCP_DBG(p, 8) && SCP_DBG_BUF(p) && lane == 0 && wave_all < 4)   // slots 0-3 (two consumer waves per SIMD: the first four)
      for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave_all) * 6 + k] = tph[k];
  } else {
    // =====================================================================================
    // consumers: LDS reads, MFMAs, epilogue into the retire buffer
    // =====================================================================================
    int pixoff[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      int ps, py, px;
      pixel_of((wave * NR + n) * 32 + r, ps, py, px);
      pixoff[n] = ps >= 0 ? (ps * HP + (py * STRIDE) * p.halo_w + px * STRIDE) * 16 : 0;
    }
    f32x16 acc[MR][NR];
    const bool wave_idle = wave * NR * 32 >= P;   // wave-uniform
    int wc = 0, xb = 0;
    __syncthreads();   // matches the producers' prologue barrier

    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return SCP_DBG(p, 8) ? __builtin_amdgcn_s_memtime() : 0ull; };
    for (int it = it_begin; it < it_end; ++it) {
      const int mb = it - fdiv(it, p.fd_nmblk) * p.n_mblk;
      for (int c = 0; c < p.nchunks; ++c, ++wc) {
        const unsigned long long t0 = now();
        const bool last = c == p.nchunks - 1;
        {  // MFMA loop: plane pairs x KK taps, taps unrolled so that every LDS offset is an immediate
          // a consumer wave whose pixel slots all lie past the work item's pixels (two 12 x 12 images fill 288 of 384 slots: all of
          // wave 3's) has nothing to compute: it only keeps the barriers
          const int npp = (SCP_DBG(p, 1) || wave_idle) ? 0 : (SCP_DBG(p, 256) ? 2 : 1) * (p.cp >> 1);   // dbg 256: every stage's MFMA loop twice (timing experiment, wrong results)
          const uint32_t xl = (uint32_t)(size_t)(xl0 + xb * p.lds_x) + half * p.plane_stride;
          const int cidx = c;
          uint32_t wa = (uint32_t)(size_t)(wl0 + (w_resident ? cidx : (wc & 1)) * p.lds_w) + (half * MT + r) * 16;
          const int hw16 = p.halo_w * 16;
          frag_t a0[MR], b0[NR], a1[MR], b1[NR];
          uint32_t brow[NR];
          auto set_row = [&](int pp, int ky) {
#pragma unroll
            for (int n = 0; n < NR; ++n) brow[n] = xl + pp * 2 * p.plane_stride + ky * hw16 + pixoff[n];
          };
          auto issue = [&](auto tapn, frag_t* a, frag_t* b) {
            constexpr int TAPN = decltype(tapn)::value;
            constexpr int AOFF = TAPN * (2 * MT * 16);
            lds_read16<AOFF>(a[0], wa);
            if constexpr (MR > 1) lds_read16<AOFF + 512>(a[1], wa);
            if constexpr (MR > 2) lds_read16<AOFF + 1024>(a[2], wa);
            if constexpr (MR > 3) lds_read16<AOFF + 1536>(a[3], wa);
#pragma unroll
            for (int n = 0; n < NR; ++n) lds_read16<((TAPN % KK) % KS) * 16>(b[n], brow[n]);
          };
          auto landed = [&](frag_t* a, frag_t* b) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < MR; ++m) lds_landed(a[m]);
#pragma unroll
            for (int n = 0; n < NR; ++n) lds_landed(b[n]);
          };
          auto mfmas = [&](const frag_t* a, const frag_t* b, int n0, int n1) {
#pragma unroll
            for (int n = 0; n < NR; ++n)
              if (n >= n0 && n < n1)
#pragma unroll
                for (int m = 0; m < MR; ++m) mfma32_acc<T, false>(acc[m][n], a[m], b[n]);
          };
          if (npp > 0) {
            set_row(0, 0);
            issue(std::integral_constant<int, 0>{}, a0, b0);     // the first fragments fly while the accumulators are initialised
          }
          if (c == 0) {   // accumulators start at the bias of their rows: row(j) = 8*(j/4) + 4*half + j%4
  #pragma unroll
            for (int m = 0; m < MR; ++m) {
              const float* bp = bias_l + mb * MT + m * 32 + 4 * half;
              float bv[16];
  #pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * q);
                bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
              }
  #pragma unroll
              for (int n = 0; n < NR; ++n)
  #pragma unroll
                for (int j = 0; j < 16; ++j) acc[m][n][j] = bv[j];
            }
          }
          if (npp > 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
              for (int n = 0; n < NR; ++n) mfma_input_fence<false>(acc[m][n]);
            landed(a0, b0);
          }
          for (int pp = 0; pp < npp; ++pp) {
            const bool more = pp + 1 < npp;
            auto tap = [&](auto tapc) {
              constexpr int TAP = decltype(tapc)::value;
              frag_t* ca = (TAP & 1) ? a1 : a0; frag_t* cb = (TAP & 1) ? b1 : b0;
              frag_t* na = (TAP & 1) ? a0 : a1; frag_t* nb = (TAP & 1) ? b0 : b1;
              mfmas(ca, cb, 0, 1);
              if constexpr (TAP + 1 < KK) {
                if constexpr ((TAP + 1) % KS == 0) set_row(pp, (TAP + 1) / KS);
                issue(std::integral_constant<int, TAP + 1>{}, na, nb);
              } else if (more) {
                set_row(pp + 1, 0);
                issue(std::integral_constant<int, KK>{}, na, nb);
              }
              mfmas(ca, cb, 1, NR);
              if (TAP + 1 < KK || more) landed(na, nb);
            };
            tap(std::integral_constant<int, 0>{});
            if constexpr (KK > 1) {
              tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
              tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{});
              tap(std::integral_constant<int, 5>{}); tap(std::integral_constant<int, 6>{});
              tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
            }
            if (more) {
              wa += KK * (2 * MT * 16);
              if constexpr (KK & 1) {
#pragma unroll
                for (int m = 0; m < MR; ++m) a0[m] = a1[m];
#pragma unroll
                for (int n = 0; n < NR; ++n) b0[n] = b1[n];
              }
            }
          }
          // last MFMA's result visible to the VALU -- only the tile's last stage is followed by VALU reads of the accumulators
          // (the epilogue); between stages the next MFMA takes its accumulator as SrcC, which the hardware interlocks
          if (last) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n) mfma_result_fence<false>(acc[m][n]);
        }
        const unsigned long long t1 = now();
        if (last && !wave_idle) {  // retire: RO <- ReLU(acc + residual), in place.  Before any lane exchange a lane holds
                     // channels [4*half, 4*half+4) of the two planes 2g and 2g+1 of its pixel: it reads and
                     // writes those two 8-byte half-slots directly, so no permlane and no bias add are needed.
          // residual half-slots are read one (m, g) batch ahead of the batch being finalised: the
          // in-place writes would otherwise order every read behind the previous write
          const uint32_t relu_floor = p.relu ? 0u : 0x80008000u;
          uint2 ra[2][NR], rb[2][NR];
          auto slot_a = [&](int b, int n) -> char* {   // b = 2*m + g
            return ro + ((((b >> 1) * 4 + 2 * (b & 1)) * PXCAP) + (wave * NR + n) * 32 + r) * 16 + 8 * half;
          };
          auto fetch = [&](int b, int buf) {
#pragma unroll
            for (int n = 0; n < NR; ++n) {
              ra[buf][n] = *reinterpret_cast<const uint2*>(slot_a(b, n));
              rb[buf][n] = *reinterpret_cast<const uint2*>(slot_a(b, n) + PXCAP * 16);
            }
          };
          if (!p.res) {   // no residual (conv1 of a BasicBlock, transitions): round, ReLU, write -- a third of the VALU work of the path below
#pragma unroll
            for (int b = 0; b < 2 * MR; ++b) {
              const int m = b >> 1, g = b & 1;
#pragma unroll
              for (int n = 0; n < NR; ++n) {
                uint2 oa, ob;
                oa.x = relu2_16(pack2<T>(acc[m][n][8 * g + 0], acc[m][n][8 * g + 1]), relu_floor); oa.y = relu2_16(pack2<T>(acc[m][n][8 * g + 2], acc[m][n][8 * g + 3]), relu_floor);
                ob.x = relu2_16(pack2<T>(acc[m][n][8 * g + 4], acc[m][n][8 * g + 5]), relu_floor); ob.y = relu2_16(pack2<T>(acc[m][n][8 * g + 6], acc[m][n][8 * g + 7]), relu_floor);
                *reinterpret_cast<uint2*>(slot_a(b, n)) = oa;          // padding pixels (no tile pixel behind this lane) write their own
                *reinterpret_cast<uint2*>(slot_a(b, n) + PXCAP * 16) = ob;   // slots too: the producers never store those (BUF_OOB / ~0 offsets)
              }
            }
          } else {
          fetch(0, 0);
#pragma unroll
          for (int b = 0; b < 2 * MR; ++b) {
            const int m = b >> 1, g = b & 1, buf = b & 1;
            if (b + 1 < 2 * MR) fetch(b + 1, buf ^ 1);
#pragma unroll
            for (int n = 0; n < NR; ++n) {
              const uint2 xa = ra[buf][n], xb2 = rb[buf][n];
              float v[8];
              v[0] = acc[m][n][8 * g + 0] + from_bits<T>(xa.x & 0xffff); v[1] = acc[m][n][8 * g + 1] + from_bits<T>(xa.x >> 16);
              v[2] = acc[m][n][8 * g + 2] + from_bits<T>(xa.y & 0xffff); v[3] = acc[m][n][8 * g + 3] + from_bits<T>(xa.y >> 16);
              v[4] = acc[m][n][8 * g + 4] + from_bits<T>(xb2.x & 0xffff); v[5] = acc[m][n][8 * g + 5] + from_bits<T>(xb2.x >> 16);
              v[6] = acc[m][n][8 * g + 6] + from_bits<T>(xb2.y & 0xffff); v[7] = acc[m][n][8 * g + 7] + from_bits<T>(xb2.y >> 16);
              uint2 oa, ob;
              oa.x = relu2_16(pack2<T>(v[0], v[1]), relu_floor); oa.y = relu2_16(pack2<T>(v[2], v[3]), relu_floor);
              ob.x = relu2_16(pack2<T>(v[4], v[5]), relu_floor); ob.y = relu2_16(pack2<T>(v[6], v[7]), relu_floor);
              *reinterpret_cast<uint2*>(slot_a(b, n)) = oa;            // (padding pixels: see above)
              *reinterpret_cast<uint2*>(slot_a(b, n) + PXCAP * 16) = ob;
            }
          }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = now();
        __builtin_amdgcn_s_barrier();
        if SCP_DBG(p, 8) {   // consumers: [zero + MFMA loop][epilogue][barrier]
          const unsigned long long t3 = now();
          tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2;
        }
        xb ^= 1;
      }
    }
    if (SCP_DBG(p, 8) && SCP_DBG_BUF(p) && lane == 0)
      for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave_all) * 6 + k] = tph[k];
  }
}

template <int DT, int STRIDE, int MR, int NR, int WREG = 0, int C16 = 0, int M16 = 0, int ADDR = 2, int CW2 = 0>
int32_t m32p_launch_addr(const ConvLaunch& L, size_t lds, hipStream_t st) {
  auto kern = conv_m32p_kernel<DT, 3, STRIDE, MR, NR, WREG, C16, M16, ADDR, CW2>;
  static LdsOptIn big_lds;   // per device (common.h)
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(CW2 ? 768 : 512), lds, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

template <int DT, int STRIDE, int MR, int NR, int WREG = 0, int C16 = 0, int M16 = 0, int CW2 = 0>
int32_t m32p_launch_one(const ConvLaunch& L, size_t lds, hipStream_t st) {
  if constexpr (C16 != 0) {   // one addressing mode per kernel (see ADDR)
    return L.in_bytes != 0 ? m32p_launch_addr<DT, STRIDE, MR, NR, WREG, C16, M16, 0, CW2>(L, lds, st)
                           : m32p_launch_addr<DT, STRIDE, MR, NR, WREG, C16, M16, 1, CW2>(L, lds, st);
  } else {
    return m32p_launch_addr<DT, STRIDE, MR, NR, WREG, C16, M16, 2>(L, lds, st);
  }
}

template <int DT>
int32_t m32p_dispatch(int stride, int mr, int nr, int c16, int cw2, const ConvLaunch& L, size_t lds, hipStream_t st) {
  if (c16 && cw2) {   // two consumer waves per SIMD (768 threads)
    if (stride == 1 && mr == 3 && c16 == 6) return m32p_launch_one<DT, 1, 3, 3, 0, 6, 0, 1>(L, lds, st);
    if (stride == 1 && mr == 3 && c16 == 4) return m32p_launch_one<DT, 1, 3, 2, 0, 4, 0, 1>(L, lds, st);
    set_error("conv m32p: two-consumer-wave variant stride=%d mr=%d columns=%d not built", stride, mr, c16);
    return SCPOSE_E_INVALID;
  }
  if (c16) {   // 16x16x32 consumers, c16 = 16-pixel columns per consumer wave (conv_launch_m32 routes only stride 1 and 96-row blocks here)
    if (stride == 1 && mr == 3) {
      if (c16 == 6) return L.groups == 6 ? m32p_launch_one<DT, 1, 3, 3, 6, 6>(L, lds, st) : m32p_launch_one<DT, 1, 3, 3, 0, 6>(L, lds, st);
      if (c16 == 5) return m32p_launch_one<DT, 1, 3, 3, 0, 5>(L, lds, st);
      if (c16 == 4) return m32p_launch_one<DT, 1, 3, 2, 0, 4>(L, lds, st);
      if (c16 == 2) return m32p_launch_one<DT, 1, 3, 1, 0, 2>(L, lds, st);
    }
    if (stride == 1 && mr == kMrep48 && c16 == 8) return m32p_launch_one<DT, 1, 2, 4, 0, 8, 3>(L, lds, st);   // 48-row blocks: 3 x 8 accumulators
    set_error("conv m32p: 16x16x32 consumer variant stride=%d mr=%d columns=%d not built", stride, mr, c16);
    return SCPOSE_E_INVALID;
  }
  if (stride == 2) {
    if (mr == 3 && nr == 1) return m32p_launch_one<DT, 2, 3, 1>(L, lds, st);
    if (mr == 3 && nr == 2) return m32p_launch_one<DT, 2, 3, 2>(L, lds, st);
    if (mr == 2 && nr == 1) return m32p_launch_one<DT, 2, 2, 1>(L, lds, st);
    if (mr == 2 && nr == 2) return m32p_launch_one<DT, 2, 2, 2>(L, lds, st);
    set_error("conv m32p: stride-2 variant mr=%d nr=%d not built", mr, nr);
    return SCPOSE_E_INVALID;
  }
  if (mr == 3 && nr == 1) return m32p_launch_one<DT, 1, 3, 1>(L, lds, st);
  if (mr == 3 && nr == 2) return m32p_launch_one<DT, 1, 3, 2>(L, lds, st);
  if (mr == 3 && nr == 3 && L.groups == 6) return m32p_launch_one<DT, 1, 3, 3, 6>(L, lds, st);   // weights in producer registers
  if (mr == 3 && nr == 3) return m32p_launch_one<DT, 1, 3, 3>(L, lds, st);
  if (mr == 2 && nr == 1) return m32p_launch_one<DT, 1, 2, 1>(L, lds, st);
  if (mr == 2 && nr == 2) return m32p_launch_one<DT, 1, 2, 2>(L, lds, st);
  if (mr == 2 && nr == 3) return m32p_launch_one<DT, 1, 2, 3>(L, lds, st);
  set_error("conv m32p: variant mr=%d nr=%d not built", mr, nr);
  return SCPOSE_E_INVALID;
}

}  // namespace scpose
