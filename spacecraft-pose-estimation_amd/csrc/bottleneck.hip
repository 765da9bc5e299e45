// Fused Bottleneck of layer1:  y = ReLU( bn3(conv3( ReLU(bn2(conv2( ReLU(bn1(conv1(x))) ))) )) + x ),
// conv1 1x1 256 -> 64, conv2 3x3 64 -> 64, conv3 1x1 64 -> 256, all on v_mfma_f32_16x16x32_{bf16,f16}
// (landmark_regression/lib/models/pose_hrnet.py:60-98, the blocks of layer1, :374-391; the first one, whose residual is
// projected by `downsample`, runs as the PROJ instantiation -- see the kernel template).
//
// Why: run as three layers a Bottleneck moves 4.8 GB at batch 256 / 96x96 -- the 256-channel tensor is read by conv1,
// read again as the residual and written by conv3, and two 64-channel tensors are written and re-read in between --
// for 1.0 ms at 5.0-5.4 TB/s (12 % of the forward for 4 % of its FLOPs).  Fused, HBM sees x once (plus the halo
// overlap of neighbouring tiles, mostly served by L2) and y once.
//
// One 512-thread workgroup per CU, persistent over tiles of 16 x 16 output pixels.  x is never staged in LDS: conv1 is
// a 1x1 convolution, so a lane's B fragment is one 16-byte vector of the blocked layout and comes straight from global
// memory into registers (buffer loads, out-of-image halo pixels read as zeros); the residual of phase C is re-read the
// same way.  LDS (108.8 KB):
//   t1 tile  [8 planes][18 x 18 halo pixels][16 B]        conv1 output on the halo (zero outside the image)
//   t2 tile  [8 planes][16 x 16][16 B]                    conv2 output
//   W1       [8 k-steps][4 blocks][4 k-groups][16 rows][16 B] = 32 KB (conv1's weights)
// Per tile, three phases and two workgroup barriers:
//   A  conv1 on the 324 halo pixels (21 columns of 16): wave w owns columns w, w + 8, w + 16 and all four 16-channel
//      blocks (one x vector feeds four MFMAs; no x vector is loaded by two waves).  Its 8 x 3 x vectors were requested a
//      whole tile ahead (96 registers per wave = 192 KB in flight per CU); bias + ReLU + 16-bit rounding -> t1
//   B  conv2 (3x3) on the 256 output pixels: a wave owns 32 output channels x four tile rows; its 2 x 18 weight
//      fragments (144 registers) are re-read from L2 per tile at the end of phase A -- resident they would leave no
//      room for the x and residual vectors; bias + ReLU + rounding -> t2
//   C  every residual vector of the tile is requested up front (64 registers), then conv3 runs on the 256 pixels (a wave
//      owns 32 of the 256 output channels, two tile rows at a time): bias, lane exchange, + residual, ReLU, 16-byte
//      stores; row pair k first requests the next tile's x vectors of k-step k.  A wave's vector-memory operations
//      complete in order, so a load issued behind stores cannot be used before those stores are acknowledged: with the
//      residual loads only one to three row pairs ahead of their stores, phase C took 12 us per tile.  (The x requests
//      do sit behind earlier pairs' stores, but are not needed before the next tile's phase A.)
// Measured at batch 256, 96 x 96 (MI355X): 0.78 ms per launch (three layers: 1.00 ms).  Ablations: without the residual
// reads 0.71 ms, without the stores 0.61 ms, without both 0.52 ms.  The first version of this kernel (8 x 8 tiles, x tile
// double-buffered in LDS by LDS-DMA, three barriers per 64 pixels, 56 % halo recompute in conv1) was bound by LDS reads
// at 0.89 ms; a version of this one with 3 k-steps of look-ahead and duplicated x loads was latency-bound at 0.97 ms.
// Register-allocation notes: anything defined on only one path of the tile loop (an `if (next tile exists)` around the
// requests) is live around the whole loop and spilled 91 registers; the requests therefore always run, with out-of-range
// offsets on the last tile.
// Rounding points are exactly those of the unfused layers, so the oracle's storage model is unchanged.
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // dma16_buf, store16_buf, make_buf, BUF_OOB, u32x4

namespace scpose {

namespace {

constexpr int kT = 16;                         // output tile edge
constexpr int kHW = kT + 2, kHPix = kHW * kHW; // 18 x 18 halo pixels
constexpr int kCols1 = (kHPix + 15) / 16;      // 21 MFMA columns of conv1
constexpr int kXS = (kHPix | 1) * 16;          // bytes of one t1 plane (325 slots: odd pitch)
constexpr int kT2S = (kT * kT | 1) * 16;       // bytes of one t2 plane (257 slots)
constexpr int kW1Bytes = 8 * 4 * 4 * 16 * 16;  // 32 KB
constexpr int kW2LSteps = 13;                  // identity Bottleneck: k-steps of conv2's weights resident in LDS (52 KB: what 160 KB leave); the other 5 stay in registers
constexpr int kLds = 8 * kXS + 8 * kT2S + kW1Bytes + (64 + 64 + 256) * 4 + 16 + kW2LSteps * 4 * 1024;   // + the tile queue words + W2
// first Bottleneck (PROJ): x has 64 channels, W1 is 8 KB, and the tile's 256 centre pixels of x (8 planes) are kept in LDS,
// double-buffered, as the operand of the projection half of conv3
constexpr int kW1BytesProj = 2 * 4 * 4 * 16 * 16;
constexpr int kLdsProj = 8 * kXS + 8 * kT2S + 2 * 8 * kT2S + kW1BytesProj + (64 + 64 + 256) * 4 + 16;

struct BneckLaunch {
  const void* in;
  const void* w1;   // [Cin / 32][4][4][16][8]
  const void* w2;   // [18][4][4][16][8]   (tap, plane) pairs t = 4 s + q: tap = t >> 3, plane = t & 7
  const void* w3;   // [2 or 4][16][4][16][8]   (4: [W3 | Wds] of the first Bottleneck)
  const float* b1;  // MFMA row order
  const float* b2;
  const float* b3;
  void* out;
  uint32_t in_bytes, out_bytes;
  int32_t N, H, W;
  int32_t tiles_x, tiles_y, tiles_total, grid;
  uint32_t* sched;               // dynamic tile queue (conv_device.h: tile_claim_xcd)
  int32_t xcd_local;
  unsigned long long* dbg_buf;   // development (SCPOSE_BNECK_DBG=1): cycles per phase and wave
};

inline int bneck_row_channel(int row) {
  const int q = row >> 2, reg = row & 3;
  return (q & 1) * 8 + (q >> 1) * 4 + reg;
}

}  // namespace

// PROJ = false: identity residual, Cin = 256 (layer1 blocks 1-3).  PROJ = true: the first Bottleneck (pose_hrnet.py:378-384,
// :78-98 with `downsample`): Cin = 64, and conv3 runs over the concatenated operand [t2 ; x] with weights [W3 | Wds] and bias
// b3 + bds -- the 1x1 projection of the residual is two more k-steps of conv3 and is summed in the fp32 accumulators.
template <int S, int E, typename F>
__device__ __forceinline__ void static_for_b(F&& f) {
  if constexpr (S < E) { f(std::integral_constant<int, S>{}); static_for_b<S + 1, E>(f); }
}

// k-steps of the NEXT tile's x vectors that are requested in phase B instead of phase C (identity Bottleneck).  Measured, layer1 of
// W48 384^2 at batch 256, same box: 0: 3.58 ms, 2: 3.44 ms (-47 us per Bottleneck); 3 and 4 spill (the ring's registers hold the
// residual vectors during phase B: 64 + 12 per early k-step + conv2's 40 weight registers + its fragments).
// The kernel sits on the CU's vector-memory instruction rate: per wave and tile 24 x loads + 16 residual loads + 16 stores +
// 10 weight reloads = 66 one-KiB instructions x 8 waves x ~70 cycles = 37 k cycles, the tile takes 36 k (SCPOSE_BNECK_DBG=1).
// Moving instructions between phases only fills idle slots: keeping the last 2 (identity) / 4 (first Bottleneck) row pairs'
// output vectors in registers and storing them in the next tile's phase B measured -0.6 % and was dropped.
#ifndef SCPOSE_BNECK_EARLY
#define SCPOSE_BNECK_EARLY 2
#endif
#ifndef SCPOSE_BNECK_EARLY_PROJ   // the first Bottleneck has two k-steps of x in all (64 input channels).  Same box: 0: 426 us, 1: 370 us; 2 spills
#define SCPOSE_BNECK_EARLY_PROJ 1   // (six registers, whatever part of conv2's weights moves to the 10 KB of LDS that are left)
#endif
template <int DT, bool PROJ>
__global__ __launch_bounds__(512, 2) void bottleneck_kernel(const BneckLaunch p) {
  constexpr int EARLY = PROJ ? SCPOSE_BNECK_EARLY_PROJ : SCPOSE_BNECK_EARLY;   // k-steps of the next tile's x vectors requested in phase B
  constexpr int CINP = PROJ ? 8 : 32;          // input planes
  constexpr int KA = CINP / 4;                 // conv1 k-steps
  constexpr int KC = PROJ ? 4 : 2;             // conv3 k-steps
  constexpr int W1B = PROJ ? kW1BytesProj : kW1Bytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  char* t1l = smem;
  char* t2l = t1l + 8 * kXS;
  char* xcl = t2l + 8 * kT2S;                  // PROJ: 2 x [8 planes][256 centre pixels] of x
  char* w1l = xcl + (PROJ ? 2 * 8 * kT2S : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15, half = lane >> 5, psel = q & 1;
  const int ch = wave & 1, wq = wave >> 1;   // phases A / B: output-channel half (blocks 2ch, 2ch + 1) and column group
  const int HW = p.H * p.W;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // ---- weights ----
  for (int o = tid * 16; o < W1B; o += 512 * 16)
    *reinterpret_cast<u32x4*>(w1l + o) = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.w1) + o);
  frag_t w3f[KC][2];
#pragma unroll
  for (int s = 0; s < KC; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      w3f[s][mb] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(p.w3) + ((((size_t)s * 16 + 2 * wave + mb) * 4 + q) * 16 + r) * 16);
  // biases stay in LDS (behind W1) and are re-read per phase
  float* bl = reinterpret_cast<float*>(w1l + W1B);
  for (int e = tid; e < 64 + 64 + 256; e += 512) bl[e] = e < 64 ? p.b1[e] : e < 128 ? p.b2[e - 64] : p.b3[e - 128];
  // conv2 operand addressing: k-step s, k-group q -> (tap, plane) pair 4 s + q: tap = s >> 1 (compile time),
  // plane = q + 4 (s & 1): one per-lane offset (q * plane pitch), everything else an immediate
  const int qoff = q * kXS;
  const uint32_t w2vo = (uint32_t)(((2 * ch) * 4 + q) * 16 + r) * 16u;   // this lane's slot in a conv2 weight fragment; + (s * 4 + mb) * 1024

  // tiles come from the launch's dynamic queue: this tile and the next are always known (the next one's x vectors are
  // requested during this tile's phase C); the one after is claimed at the top of the tile and published through LDS
  // in front of the tile's second barrier
  int* tq = reinterpret_cast<int*>(bl + 64 + 64 + 256);
  // conv2's weights: the first W2L k-steps (all four 16-channel blocks, 4 KB each) live in LDS, the rest in registers.  A wave
  // used to re-read its 36 fragments (36 KB) from L2 for every tile -- 288 KB per tile and CU, 40 % of all the bytes the CU's
  // vector-memory path moved, and phase C (residual loads, x requests, stores) is bound by exactly that path.  (The first
  // Bottleneck's LDS is taken by the tile's copy of x: it keeps the register form.)
  constexpr int W2L = PROJ ? 0 : kW2LSteps;
  char* w2ll = reinterpret_cast<char*>(tq) + 16;
  for (int o = tid * 16; o < W2L * 4 * 1024; o += 512 * 16)
    *reinterpret_cast<u32x4*>(w2ll + o) = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.w2) + o);
  const int xcd = xcc_id();
  auto claim = [&]() { return p.xcd_local ? tile_claim_xcd(p.sched, xcd, p.N, tiles_per_img) : tile_claim(p.sched + 9, p.tiles_total); };
  if (tid == 0) { tq[0] = claim(); tq[1] = tq[0] < 0 ? -1 : claim(); }
  __syncthreads();
  int t = tq[0], t_next = tq[1];
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = t / tiles_per_img;
    const int rem = t - img * tiles_per_img;
    const int ty = rem / p.tiles_x;
    oy0 = ty * kT; ox0 = (rem - ty * p.tiles_x) * kT;
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.in_bytes), rs_out = make_buf(p.out, p.out_bytes), rs_w2 = make_buf(p.w2, 18u * 4 * 4 * 16 * 16);
  const uint32_t kstep_bytes = (uint32_t)(4 * HW) * 16u;     // four planes = one k-step of conv1

  // conv1 columns of this wave: wave + 8 i, i = 0..2 (21 columns: the third exists for waves 0-4 only).
  // vo[i] = byte offset of (image, plane q, halo pixel of this lane) in x, BUF_OOB outside the image / the halo
  const bool has_col2 = wave + 16 < kCols1;
  uint32_t vo[3];
  int cx[3] = {-1, -1, -1};   // PROJ: byte offset of this lane's pixel in a centre plane, -1 for halo pixels
  auto locate = [&](int t) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int slot = (wave + 8 * i) * 16 + r;
      const int my = slot / kHW, mx = slot - my * kHW;
      const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
      const bool ok = slot < kHPix && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      vo[i] = ok ? (uint32_t)((img * CINP + q) * HW + gy * p.W + gx) * 16u : BUF_OOB;
      if constexpr (PROJ) cx[i] = (slot < kHPix && my >= 1 && my <= kT && mx >= 1 && mx <= kT) ? ((my - 1) * kT + mx - 1) * 16 : -1;
    }
  };
  // x vectors of conv1: all 8 k-steps of the wave's three columns (96 registers, 192 KB in flight per CU), requested
  // one tile ahead -- at the start of the previous tile's phase C, BEFORE that phase's stores: a wave's vector-memory
  // operations complete in order, so a load issued behind stores is not usable until those stores are acknowledged
  // (measured: with loads and stores interleaved per row pair, phase C took 12 us per tile)
  u32x4 xr[KA][3];
  auto request = [&](auto sc) {
    constexpr int S = decltype(sc)::value;
    if constexpr (S < KA)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        xr[S][i] = load16_buf(rs_in, vo[i], (uint32_t)S * kstep_bytes);   // a column that does not exist reads zeros (BUF_OOB)
  };
  auto request_head = [&]() {
    request(std::integral_constant<int, 0>{}); request(std::integral_constant<int, 1>{});
    request(std::integral_constant<int, 2>{}); request(std::integral_constant<int, 3>{});
    request(std::integral_constant<int, 4>{}); request(std::integral_constant<int, 5>{});
    request(std::integral_constant<int, 6>{}); request(std::integral_constant<int, 7>{});
  };

  if (t >= 0) {
    locate(t);
    request_head();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // W1 and the biases are in LDS (and every wave has read tq[0], tq[1])

  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { return SCP_DBG_BUF(p) ? __builtin_amdgcn_s_memtime() : 0ull; };
  int xbuf = 0;
  for (; t >= 0; t = t_next, t_next = tq[2], xbuf ^= 1) {   // tq[2]: published before this tile's second barrier, not rewritten before the next tile's
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    const unsigned long long ts0 = now();
    int t_after = -1;
    if (tid == 0 && t_next >= 0) t_after = claim();   // returns under phases A and B

    // ---- A: conv1 (1x1, 256 -> 64) on the halo pixels -> t1: all four 16-channel blocks of the wave's columns ----
    frag_t w2f[18 - W2L][2];                               // conv2's weights of this wave that are not in LDS: re-read per tile (below)
    // phase C's residual vectors (the tile's centre pixels of x, 64 registers): requested at the end of phase A, so that
    // they arrive under conv2 -- requested at the start of phase C their latency (2-3 us under load, with nothing else
    // for the wave to do: conv3 is 64 MFMAs) was exposed on every tile.  The ring of x vectors is dead during phase B,
    // which is where the registers come from.
    const uint32_t pl_bytes = (uint32_t)HW * 16u;
    auto out_off = [&](int cp) -> uint32_t {
      const int oy = oy0 + 2 * cp + half, ox = ox0 + r;
      return (oy < p.H && ox < p.W) ? (uint32_t)((img * 32 + psel) * HW + oy * p.W + ox) * 16u : BUF_OOB;
    };
    u32x4 rv[PROJ ? 1 : 8][2];
    {
      f32x4 acc[4][3];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {                     // accumulators start at the bias of their rows
        const float4 bs = *reinterpret_cast<const float4*>(bl + mb * 16 + q * 4);
        acc[mb][0] = f32x4{bs.x, bs.y, bs.z, bs.w}; acc[mb][1] = acc[mb][0]; acc[mb][2] = acc[mb][0];
      }
      auto load_a = [&](int s, frag_t* a) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = *reinterpret_cast<const frag_t*>(w1l + ((((s * 4 + mb) * 4 + q) * 16 + r) * 16));
      };
      frag_t fa[2][4];
      load_a(0, fa[0]);
      auto step = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S + 1 < KA) load_a(S + 1, fa[(S + 1) & 1]);
        if constexpr (PROJ) {   // the centre pixels' x vectors are the projection operand of conv3: into this tile's LDS copy
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (cx[i] >= 0) *reinterpret_cast<u32x4*>(xcl + (xbuf * 8 + 4 * S + q) * kT2S + cx[i]) = xr[S][i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          acc[mb][0] = mfma16<T>(fa[S & 1][mb], __builtin_bit_cast(frag_t, xr[S][0]), acc[mb][0]);
          acc[mb][1] = mfma16<T>(fa[S & 1][mb], __builtin_bit_cast(frag_t, xr[S][1]), acc[mb][1]);
          if (has_col2) acc[mb][2] = mfma16<T>(fa[S & 1][mb], __builtin_bit_cast(frag_t, xr[S][2]), acc[mb][2]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      if constexpr (KA > 2) {
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
      }
      // t1 <- ReLU(acc), zero outside the image
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int slot = (wave + 8 * i) * 16 + r;
        const bool inimg = vo[i] != BUF_OOB;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          uint2 o;
          o.x = relu2_16(pack2<T>(acc[mb][i][0], acc[mb][i][1]), 0u);
          o.y = relu2_16(pack2<T>(acc[mb][i][2], acc[mb][i][3]), 0u);
          if (!inimg) o = make_uint2(0u, 0u);              // conv2's zero padding, not a conv1 output
          if (slot < kHPix) *reinterpret_cast<uint2*>(t1l + (2 * mb + psel) * kXS + slot * 16 + 8 * (q >> 1)) = o;
        }
      }
      // conv2's 36 weight fragments of this wave (36 KB per wave and tile, L2 hits): resident they would take 144 of the
      // 256 registers, which is what conv1's x vectors and conv3's residual vectors need to keep enough bytes in flight.
      // Buffer loads on purpose: through a laundered generic pointer hipcc emitted flat_load, and the results were
      // sporadically wrong (non-deterministic at the 1e-1 level; found by the determinism check of tools_dev/dump_tap.py).
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = W2L; s2 < 18; ++s2)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          w2f[s2 - W2L][mb] = __builtin_bit_cast(frag_t, load16_buf(rs_w2, w2vo, (uint32_t)((s2 * 4 + mb) * 1024)));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!PROJ) {
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) {
          const uint32_t o = out_off(cp);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) rv[cp][mb] = load16_buf(rs_in, o, (uint32_t)(2 * (2 * wave + mb)) * pl_bytes);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long ts1 = now();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // t1 complete (and every wave is done with t2 of the previous tile)
    const unsigned long long ts2 = now();

    // ---- B: conv2 (3x3, 64 -> 64) on the 256 output pixels -> t2: tile rows wq, wq + 4, wq + 8, wq + 12 ----
    // The next tile's x vectors of the first EARLY k-steps are requested HERE, not in phase C: the vector-memory path of the CU is
    // idle during phases A and B and carries all of a tile's loads and stores in phase C, which is 55 % of the tile time.  (Their
    // registers are free: the ring is consumed in phase A.)
    if constexpr (EARLY > 0) {
      __builtin_amdgcn_sched_barrier(0);
      if (t_next >= 0) locate(t_next);
      else { vo[0] = vo[1] = vo[2] = BUF_OOB; cx[0] = cx[1] = cx[2] = -1; }
      static_for_b<0, EARLY>([&](auto sc) { request(sc); });
      __builtin_amdgcn_sched_barrier(0);
    }
    // two row groups at a time where the weights come from LDS (one A fragment then feeds two MFMAs)
    constexpr int CC = PROJ ? 1 : 2;
#pragma unroll 1
    for (int c0 = 0; c0 < 4; c0 += CC) {
      f32x4 acc[2][CC];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float4 bs = *reinterpret_cast<const float4*>(bl + 64 + (2 * ch + mb) * 16 + q * 4);
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) acc[mb][cc] = f32x4{bs.x, bs.y, bs.z, bs.w};
      }
      const int px = r;                                    // output pixels of this lane: rows wq + 4 (c0 + cc)
      const char* bq0 = t1l + ((wq + 4 * c0) * kHW + px) * 16 + qoff;
      const char* aq0 = w2ll + (2 * ch) * 1024 + lane * 16;
      auto k2imm = [](int s) { const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky; return 4 * (s & 1) * kXS + (ky * kHW + kx) * 16; };
      frag_t bq[3][CC], aq[3][2];                          // fragments two k-steps ahead
      auto fetch = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < 18) {
#pragma unroll
          for (int cc = 0; cc < CC; ++cc) bq[S % 3][cc] = *reinterpret_cast<const frag_t*>(bq0 + cc * (4 * kHW * 16) + k2imm(S));
          if constexpr (S < W2L) {
            aq[S % 3][0] = *reinterpret_cast<const frag_t*>(aq0 + S * 4096);
            aq[S % 3][1] = *reinterpret_cast<const frag_t*>(aq0 + S * 4096 + 1024);
          }
        }
      };
      fetch(std::integral_constant<int, 0>{});
      fetch(std::integral_constant<int, 1>{});
      auto kstep = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        fetch(std::integral_constant<int, S + 2>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          frag_t a;
          if constexpr (S < W2L) a = aq[S % 3][mb]; else a = w2f[S - W2L][mb];
#pragma unroll
          for (int cc = 0; cc < CC; ++cc) acc[mb][cc] = mfma16<T>(a, bq[S % 3][cc], acc[mb][cc]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
      kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
      kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{}); kstep(std::integral_constant<int, 8>{});
      kstep(std::integral_constant<int, 9>{}); kstep(std::integral_constant<int, 10>{}); kstep(std::integral_constant<int, 11>{});
      kstep(std::integral_constant<int, 12>{}); kstep(std::integral_constant<int, 13>{}); kstep(std::integral_constant<int, 14>{});
      kstep(std::integral_constant<int, 15>{}); kstep(std::integral_constant<int, 16>{}); kstep(std::integral_constant<int, 17>{});
#pragma unroll
      for (int cc = 0; cc < CC; ++cc) {
        const int pix = (wq + 4 * (c0 + cc)) * kT + px;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          uint2 o;
          o.x = relu2_16(pack2<T>(acc[mb][cc][0], acc[mb][cc][1]), 0u);
          o.y = relu2_16(pack2<T>(acc[mb][cc][2], acc[mb][cc][3]), 0u);
          *reinterpret_cast<uint2*>(t2l + (2 * (2 * ch + mb) + psel) * kT2S + pix * 16 + 8 * (q >> 1)) = o;
        }
      }
    }
    const unsigned long long ts3 = now();
    if (tid == 0) tq[2] = t_after;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // t2 complete (and every wave is done with t1); tq[2] published
    const unsigned long long ts4 = now();

    // ---- C: conv3 (1x1, 64 -> 256) + residual -> y ----
    // row pairs (0,1), (2,3), ... one after the other (keeps the live accumulators at 2 x 2): after the lane exchange the
    // lower half-wave owns the pixel of the even row, the upper half-wave that of the odd one, each lane the 8 channels
    // of plane 2 * block + psel
    {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (EARLY == 0) {
        if (t_next >= 0) locate(t_next);
        else { vo[0] = vo[1] = vo[2] = BUF_OOB; cx[0] = cx[1] = cx[2] = -1; }   // last tile: the requests still run (they read nothing), so that the
                                                             // ring is redefined on every path and is not live across phase B
      }
      auto pair = [&](auto cpc) {
        constexpr int cp = decltype(cpc)::value;
        request(std::integral_constant<int, cp + EARLY>{});   // next tile's x vectors of k-step cp + EARLY (none past the last): in front of this pair's stores
        f32x4 acc[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const float4 bs = *reinterpret_cast<const float4*>(bl + 128 + (2 * wave + mb) * 16 + q * 4);
          acc[mb][0] = f32x4{bs.x, bs.y, bs.z, bs.w}; acc[mb][1] = acc[mb][0];
        }
        frag_t bc[KC][2];
#pragma unroll
        for (int s = 0; s < KC; ++s)
#pragma unroll
          for (int c = 0; c < 2; ++c)   // k-steps 0, 1: t2; PROJ k-steps 2, 3: the tile's copy of x
            bc[s][c] = *reinterpret_cast<const frag_t*>((s < 2 ? t2l + (4 * s + q) * kT2S : xcl + (xbuf * 8 + 4 * (s - 2) + q) * kT2S) + ((2 * cp + c) * 16 + r) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KC; ++s)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[mb][c] = mfma16<T>(w3f[s][mb], bc[s][c], acc[mb][c]);
        // MFMA result -> v_permlane32_swap: explicit wait states tied to the accumulators (conv_device.h: mfma_swap_pad) -- nothing
        // guarantees that the scheduler keeps other work between the last MFMA and the swaps
        mfma_swap_pad(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
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
          float v[8];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { v[jj] = __uint_as_float(a[jj]); v[4 + jj] = __uint_as_float(b[jj]); }
          if constexpr (!PROJ) {
          const u32x4 rr = rv[cp][mb];
          v[0] += from_bits<T>(rr[0] & 0xffff); v[1] += from_bits<T>(rr[0] >> 16);
          v[2] += from_bits<T>(rr[1] & 0xffff); v[3] += from_bits<T>(rr[1] >> 16);
          v[4] += from_bits<T>(rr[2] & 0xffff); v[5] += from_bits<T>(rr[2] >> 16);
          v[6] += from_bits<T>(rr[3] & 0xffff); v[7] += from_bits<T>(rr[3] >> 16);
          }
          u32x4 ov;
          ov[0] = relu2_16(pack2<T>(v[0], v[1]), 0u); ov[1] = relu2_16(pack2<T>(v[2], v[3]), 0u);
          ov[2] = relu2_16(pack2<T>(v[4], v[5]), 0u); ov[3] = relu2_16(pack2<T>(v[6], v[7]), 0u);
          store16_buf(rs_out, out_off(cp), (uint32_t)(2 * (2 * wave + mb)) * pl_bytes, ov);
        }
      };
      pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 1>{}); pair(std::integral_constant<int, 2>{});
      pair(std::integral_constant<int, 3>{}); pair(std::integral_constant<int, 4>{}); pair(std::integral_constant<int, 5>{});
      pair(std::integral_constant<int, 6>{}); pair(std::integral_constant<int, 7>{});
    }
    if (SCP_DBG_BUF(p)) {   // [A][barrier][B][barrier][C]
      const unsigned long long ts5 = now();
      tph[0] += ts1 - ts0; tph[1] += ts2 - ts1; tph[2] += ts3 - ts2; tph[3] += ts4 - ts3; tph[4] += ts5 - ts4;
    }
  }
  if (tid == 0) { if (p.xcd_local) tile_retire_xcd(p.sched); else tile_retire(p.sched + 9); }
  if (SCP_DBG_BUF(p) && lane == 0)
    for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave) * 6 + k] = tph[k];
}

// ---- host ----
bool bottleneck_fusable(int cin, int cmid, int cout) { return (cin == 256 || cin == 64) && cmid == 64 && cout == 256; }

// w1: [64][cin] (1x1), w2: [64][64][3][3], w3: [256][64] (1x1), BN-folded f32 OIHW.  cin = 256: identity residual (wds, bds
// null).  cin = 64: the first Bottleneck -- wds [256][64] / bds are its `downsample` 1x1 conv + BN, appended to conv3 as
// k-steps 2 and 3 (input planes 8..15 of the concatenated operand [t2 ; x]) with the biases summed.
void bottleneck_pack(const float* w1, const float* w2, const float* w3, const float* wds, const float* b1, const float* b2, const float* b3,
                     const float* bds, int cin, int dtype,
                     std::vector<uint16_t>* pw1, std::vector<uint16_t>* pw2, std::vector<uint16_t>* pw3, std::vector<float>* pb) {
  const int ka = cin / 32, kc = wds ? 4 : 2;
  pw1->assign((size_t)ka * 4 * 4 * 16 * 8, 0);
  pw2->assign((size_t)18 * 4 * 4 * 16 * 8, 0);
  pw3->assign((size_t)kc * 16 * 4 * 16 * 8, 0);
  pb->assign(64 + 64 + 256, 0.f);
  for (int s = 0; s < ka; ++s)
    for (int m = 0; m < 4; ++m)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * m + bneck_row_channel(r), plane = 4 * s + q;
          uint16_t* d = pw1->data() + ((((size_t)s * 4 + m) * 4 + q) * 16 + r) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(w1[(size_t)co * cin + plane * 8 + j], dtype);
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
  for (int s = 0; s < kc; ++s)
    for (int m = 0; m < 16; ++m)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) {
          const int co = 16 * m + bneck_row_channel(r), plane = 4 * s + q;
          uint16_t* d = pw3->data() + ((((size_t)s * 16 + m) * 4 + q) * 16 + r) * 8;
          const float* src = plane < 8 ? w3 + (size_t)co * 64 + plane * 8 : wds + (size_t)co * 64 + (plane - 8) * 8;
          for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(src[j], dtype);
        }
  for (int pos = 0; pos < 64; ++pos) {
    const int co = (pos & ~15) + bneck_row_channel(pos & 15);
    (*pb)[pos] = b1[co]; (*pb)[64 + pos] = b2[co];
  }
  for (int pos = 0; pos < 256; ++pos) {
    const int co = (pos & ~15) + bneck_row_channel(pos & 15);
    (*pb)[128 + pos] = b3[co] + (bds ? bds[co] : 0.f);
  }
}

template <int DT, bool PROJ>
static int32_t bneck_launch_one(const BneckLaunch& L, hipStream_t stream) {
  auto kern = bottleneck_kernel<DT, PROJ>;
  constexpr int lds = PROJ ? kLdsProj : kLds;
  static LdsOptIn big;   // per device (common.h)
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), lds, &big); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(512), lds, stream, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

// cin = 256: identity residual; cin = 64: first Bottleneck (projection folded into conv3, see bottleneck_pack)
int32_t bottleneck_launch(const void* in, const void* w1, const void* w2, const void* w3, const float* bias, int N, int H, int W,
                          int cin, int dtype, void* out, uint32_t* sched, hipStream_t stream) {
  SCP_REQUIRE(sched, "bottleneck: null tile-queue words");
  // the kernel addresses its tensors through 32-bit buffer descriptors: batches whose 256-channel tensor reaches 4 GiB
  // (~900 frames at 96 x 96) run as several launches over frame ranges
  const bool proj = cin == 64;
  SCP_REQUIRE(cin == 64 || cin == 256, "bottleneck: Cin = %d (64 or 256)", cin);
  const size_t in_frame = (size_t)(cin / 8) * H * W * 16, out_frame = (size_t)32 * H * W * 16;
  SCP_REQUIRE(out_frame < 0xfffffff0ull, "bottleneck: one %dx%d frame does not fit a 32-bit buffer descriptor", H, W);
  const int max_n = (int)(0xfffffff0ull / out_frame);
  for (int n0 = 0; n0 < N; n0 += max_n) {
    const int n = N - n0 < max_n ? N - n0 : max_n;
    BneckLaunch L{};
    L.in = static_cast<const char*>(in) + (size_t)n0 * in_frame;
    L.out = static_cast<char*>(out) + (size_t)n0 * out_frame;
    L.w1 = w1; L.w2 = w2; L.w3 = w3; L.b1 = bias; L.b2 = bias + 64; L.b3 = bias + 128;
    L.in_bytes = (uint32_t)((size_t)n * in_frame); L.out_bytes = (uint32_t)((size_t)n * out_frame);
    L.N = n; L.H = H; L.W = W;
    L.tiles_x = (W + kT - 1) / kT; L.tiles_y = (H + kT - 1) / kT;
    L.tiles_total = n * L.tiles_x * L.tiles_y;
    L.grid = conv_device_cus() < L.tiles_total ? conv_device_cus() : L.tiles_total;
    L.sched = sched;
    { static const char* e = dev_env("SCPOSE_BNECK_GLOBALQ"); L.xcd_local = !(e && atoi(e)); }
    { static const char* e = dev_env("SCPOSE_BNECK_DBG"); L.dbg_buf = (kDevBuild && e && atoi(e)) ? conv_dbg_buffer(stream) : nullptr; if (L.dbg_buf) conv_dbg_set_grid(L.grid); }
    int32_t rc;
    if (dtype == SCPOSE_DT_BF16) rc = proj ? bneck_launch_one<0, true>(L, stream) : bneck_launch_one<0, false>(L, stream);
    else rc = proj ? bneck_launch_one<1, true>(L, stream) : bneck_launch_one<1, false>(L, stream);
    if (rc != SCPOSE_OK) return rc;
  }
  return SCPOSE_OK;
}

}  // namespace scpose
