// Branch chain: the BasicBlocks of one low-resolution branch of a HighResolutionModule in ONE launch, a frame per workgroup.
//
//   x_{k+1} = ReLU(bn2(conv2( ReLU(bn1(conv1(x_k))) )) + x_k),  k = 0 .. NBLK - 1,  3x3 / stride 1 / C -> C
//   (landmark_regression/lib/models/pose_hrnet.py:28-57 BasicBlock, :142-154 _make_one_branch, :247-253 the branch loop of
//   HighResolutionModule.forward; four blocks per branch in every shipped configuration)
//
// Why (round 6): at small batches the deep branches are chains of 10-20 us launches -- prologue (bias, first weight
// and halo chunk from L2), one or two work items, drain -- that keep the whole chip for a fraction of its throughput: HRNet-W32
// 256 x 256 at batch 64 spends 31 % of its forward in the 128-channel / 16 x 16 and 256-channel / 8 x 8 branches at 9-12 % of the
// MFMA peak.  A frame's activations of such a branch fit in LDS (128 x 16 x 16: 65.5 KB, 256 x 8 x 8: 32.8 KB per buffer), so one
// persistent workgroup can walk all eight convolutions of a frame with the activations ping-ponging between two LDS buffers:
//   * X   [C/8 planes][H][W + 2][16 B]   block input; conv2's result goes over its own residual in place (the epilogue reads and
//                                         writes the same 8 bytes per lane); the left / right zero columns are never written
//   * MID [same]                          conv1's result
//   * a zero slot that every out-of-image row reads
// * weights are MFMA A operands streamed from L2 straight into registers, three k-steps ahead: a wave owns two 16-row blocks of
//   Cout, so every 1 KiB fragment it loads feeds NCW MFMAs; nothing but activations lives in LDS and there is NO barrier inside a
//   convolution (one per convolution, between the epilogue's writes and the next layer's reads);
// * a k-step is one tap of four input planes (32 channels); B fragments are read one k-step ahead;
// * frames are claimed from a device-wide queue (conv_device.h: tile_claim), so CUs that other lanes of the captured forward
//   hold delay nothing.
// Rounding points are those of the per-layer path (16-bit after every convolution's ReLU, residual added in fp32 before it), so
// the oracle's storage model is unchanged; the fp32 summation order over K (tap-major, plane quads inside) is fixed per layer
// shape, so a frame's result does not depend on the batch or on the workgroup that computes it.
#include <type_traits>

#include "common.h"
#include "conv_device.h"

namespace scpose {

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4c;

struct ChainLaunch {
  const void* in;       // blocked [N][C/8][H][W][8]
  void* out;            // same shape (may alias in)
  const void* wpk;      // [nconv][9 * C/32 k-steps][C/16 row blocks][4 k-groups][16 rows][8]
  const float* bias;    // [nconv][C]
  int32_t N, nconv;
  uint32_t* sched;      // frame queue (2 zero-initialised words)
};

template <int S, int E, typename F>
__device__ __forceinline__ void static_for_c(F&& f) {
  if constexpr (S < E) { f(std::integral_constant<int, S>{}); static_for_c<S + 1, E>(f); }
}

template <int C, int H, int W>
struct ChainGeom {
  static constexpr int PLANES = C / 8, PQ = C / 32, KS = 9 * PQ, MBK = C / 16;
  static constexpr int MGROUPS = MBK / 2, CGROUPS = 8 / MGROUPS, NC = H * W / 16, NCW = NC / CGROUPS;
  static constexpr int PITCH = W + 2, PS = H * PITCH * 16, BUF = PLANES * PS;
  static constexpr int LDS = 256 + 2 * BUF;   // [zero slot (256 B)][X][MID]
  static_assert(MBK % 2 == 0 && 8 % MGROUPS == 0 && NC % CGROUPS == 0 && (H * W) % 16 == 0 && 16 % W == 0, "chain geometry");
  static_assert(LDS <= 160 * 1024, "two activation buffers must fit 160 KB of LDS");
};

template <int DT, int C, int H, int W>
__global__ __launch_bounds__(512, 2) void conv_chain_kernel(const ChainLaunch p) {
  typedef ChainGeom<C, H, W> G;
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const zero = smem;
  char* const xbuf = smem + 256;
  char* const mbuf = xbuf + G::BUF;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, l15 = lane & 15;
  const int mg = wave % G::MGROUPS, cg = wave / G::MGROUPS;

  // ---- LDS: everything zero once (padding columns and the zero slot stay zero for the kernel's life) ----
  for (int o = tid * 16; o < G::LDS; o += 512 * 16) *reinterpret_cast<u32x4c*>(smem + o) = u32x4c{0u, 0u, 0u, 0u};

  // ---- per-lane geometry of this wave's NCW columns (16 consecutive pixels each, row-major) ----
  int boff[G::NCW];     // B fragment: byte offset of (pixel, plane q) inside an activation buffer
  int ooff[G::NCW];     // epilogue: byte offset of the lane's 8-byte half-slot of (pixel, plane q >> 1 of row block 0)
  int yrow[G::NCW];
#pragma unroll
  for (int n = 0; n < G::NCW; ++n) {
    const int pix = (cg * G::NCW + n) * 16 + l15;
    const int y = pix / W, x = pix - y * W;
    yrow[n] = y;
    boff[n] = q * G::PS + (y * G::PITCH + x + 1) * 16;
    ooff[n] = (q >> 1) * G::PS + (y * G::PITCH + x + 1) * 16 + (q & 1) * 8;
  }

  // (the queue hand-over word lives in the 256-byte header behind the 16 zero bytes: no static LDS, so that the dynamic segment may be opted in whole)
  volatile int& next_frame = *reinterpret_cast<volatile int*>(smem + 128);
  if (tid == 0) next_frame = tile_claim(p.sched, p.N);
  __syncthreads();
  int frame = next_frame;
  while (frame >= 0) {
    // ---- block input -> X (interior pixels only) ----
    {
      const char* src = static_cast<const char*>(p.in) + (size_t)frame * G::PLANES * (H * W) * 16;
      for (int v = tid; v < G::PLANES * H * W; v += 512) {
        const int pl = v / (H * W), pix = v - pl * (H * W);
        const int y = pix / W, x = pix - y * W;
        *reinterpret_cast<u32x4c*>(xbuf + pl * G::PS + (y * G::PITCH + x + 1) * 16) = *reinterpret_cast<const u32x4c*>(src + (size_t)v * 16);
      }
    }
    __syncthreads();
    if (tid == 0) next_frame = tile_claim(p.sched, p.N);   // published by the barriers below, read after the last one

    for (int cv = 0; cv < p.nconv; ++cv) {
      const bool second = cv & 1;                 // conv2 of a block: reads MID, adds X, writes X; conv1: reads X, writes MID
      const char* const src = second ? mbuf : xbuf;
      char* const dst = second ? xbuf : mbuf;
      const char* const wl = static_cast<const char*>(p.wpk) + (size_t)cv * G::KS * G::MBK * 1024 + ((size_t)(2 * mg) * 4 + q) * 256 + l15 * 16;
      const float* const bq = p.bias + cv * C + (2 * mg) * 16 + 4 * q;

      f32x4 acc[2][G::NCW];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float4 b4 = *reinterpret_cast<const float4*>(bq + m * 16);
#pragma unroll
        for (int n = 0; n < G::NCW; ++n) acc[m][n] = f32x4{b4.x, b4.y, b4.z, b4.w};   // accumulators start at the bias of their rows
      }
      frag_t af[4][2];            // A fragments, a ring of four k-steps (three ahead)
      frag_t bf[2][G::NCW];       // B fragments, one k-step ahead
      auto fetch_a = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < G::KS) {
#pragma unroll
          for (int m = 0; m < 2; ++m) af[S & 3][m] = *reinterpret_cast<const frag_t*>(wl + (size_t)S * (G::MBK * 1024) + m * 1024);
        }
      };
      auto fetch_b = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < G::KS) {
          constexpr int TAP = S / G::PQ, PQI = S % G::PQ, DY = TAP / 3 - 1, DX = TAP % 3 - 1;
          constexpr int OFF = PQI * 4 * G::PS + (DY * G::PITCH + DX) * 16;
#pragma unroll
          for (int n = 0; n < G::NCW; ++n) {
            const bool ok = DY == 0 || (DY < 0 ? yrow[n] > 0 : yrow[n] < H - 1);
            const char* a = ok ? src + (boff[n] + OFF) : zero;
            bf[S & 1][n] = *reinterpret_cast<const frag_t*>(a);
          }
        }
      };
      fetch_a(std::integral_constant<int, 0>{});
      fetch_a(std::integral_constant<int, 1>{});
      fetch_a(std::integral_constant<int, 2>{});
      fetch_b(std::integral_constant<int, 0>{});
      static_for_c<0, G::KS>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        fetch_a(std::integral_constant<int, S + 3>{});
        fetch_b(std::integral_constant<int, S + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < G::NCW; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m][n] = mfma16<T>(af[S & 3][m], bf[S & 1][n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
      });

      // ---- epilogue: (+ residual) ReLU, 16-bit, into the destination buffer.  A lane holds rows 4 q .. + 3 of its pixel: channels
      // 4 (q & 1) .. + 3 of plane 2 mb + (q >> 1) -- one 8-byte half-slot per accumulator, no lane exchange ----
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < G::NCW; ++n) {
          const int o = (2 * (2 * mg + m)) * G::PS + ooff[n];
          float v0 = acc[m][n][0], v1 = acc[m][n][1], v2 = acc[m][n][2], v3 = acc[m][n][3];
          if (second) {
            const uint2 r = *reinterpret_cast<const uint2*>(xbuf + o);
            v0 += from_bits<T>(r.x & 0xffff); v1 += from_bits<T>(r.x >> 16);
            v2 += from_bits<T>(r.y & 0xffff); v3 += from_bits<T>(r.y >> 16);
          }
          uint2 w;
          w.x = relu2_16(pack2<T>(v0, v1), 0u); w.y = relu2_16(pack2<T>(v2, v3), 0u);
          *reinterpret_cast<uint2*>(dst + o) = w;
        }
      __syncthreads();
    }

    // ---- X -> block output ----
    {
      char* dstg = static_cast<char*>(p.out) + (size_t)frame * G::PLANES * (H * W) * 16;
      for (int v = tid; v < G::PLANES * H * W; v += 512) {
        const int pl = v / (H * W), pix = v - pl * (H * W);
        const int y = pix / W, x = pix - y * W;
        *reinterpret_cast<u32x4c*>(dstg + (size_t)v * 16) = *reinterpret_cast<const u32x4c*>(xbuf + pl * G::PS + (y * G::PITCH + x + 1) * 16);
      }
    }
    frame = next_frame;      // (written before the convolutions' barriers)
    __syncthreads();         // X is free again; next_frame may be overwritten
  }
  if (tid == 0) tile_retire(p.sched);
}

template <int DT, int C, int H, int W>
int32_t chain_launch_one(const ChainLaunch& L, int grid, hipStream_t st) {
  auto kern = conv_chain_kernel<DT, C, H, W>;
  static LdsOptIn big_lds;
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  typedef ChainGeom<C, H, W> G;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace

// Shapes with a chain kernel: (channels, map) whose two activation buffers fit LDS and whose row blocks fill eight waves.
// OFF unless SCPOSE_CHAIN=1 (development switch).  Measured, one box, W32 256 x 256, captured forward (profiles/round6_chain_ab.txt): batch 256
// +8.6 % (32 650 against 30 050 poses/s: every CU has a frame, and a chain is 160 us against 8 x 29 us), batch 64 -1 ... -4 %, batch 16 -6 %:
// below one frame per CU the chain (94-104 us alone, 160-180 us beside the other lanes) is the longest lane of its module, and the
// module is paced by the launch chains of branches 0 and 1 anyway (kernel timeline in the same file).  A frame's result must not depend
// on its batch (the CLI's engine-batch coalescing relies on it), so the choice cannot follow the batch size, and
// BASELINE's W32 configurations are quoted at batch 64: the per-layer path stays the default.
static bool chain_enabled() {
  static const char* e = dev_env("SCPOSE_CHAIN");
  return e && atoi(e) == 1;
}
bool conv_chain_supported(int C, int H, int W) {
  return chain_enabled() && ((C == 128 && H == 16 && W == 16) || (C == 256 && H == 8 && W == 8));
}
bool conv_chain_channels(int C) { return chain_enabled() && (C == 128 || C == 256); }

// w: nconv folded 3x3 weights (OIHW fp32, C x C x 3 x 3 each, back to back); k-step s = tap * (C / 32) + plane quad
size_t conv_chain_pack(const float* w, int nconv, int C, int dtype, uint16_t* dst) {
  const int PQ = C / 32, KS = 9 * PQ, MBK = C / 16;
  const size_t per = (size_t)KS * MBK * 512;   // 16-bit words per convolution
  if (!dst) return per * nconv * 2;
  for (int cv = 0; cv < nconv; ++cv) {
    const float* wc = w + (size_t)cv * C * C * 9;
    for (int s = 0; s < KS; ++s) {
      const int tap = s / PQ, pq = s % PQ;
      for (int mb = 0; mb < MBK; ++mb)
        for (int qq = 0; qq < 4; ++qq)
          for (int r = 0; r < 16; ++r) {
            uint16_t* d = dst + (size_t)cv * per + ((((size_t)s * MBK + mb) * 4 + qq) * 16 + r) * 8;
            const int co = mb * 16 + r, ci0 = (pq * 4 + qq) * 8;
            for (int j = 0; j < 8; ++j) d[j] = host_f32_to_16(wc[((size_t)co * C + ci0 + j) * 9 + tap], dtype);
          }
    }
  }
  return per * nconv * 2;
}

int32_t conv_chain_launch(const void* in, void* out, const void* wpk, const float* bias, int nconv, int N, int C, int H, int W,
                          int dtype, uint32_t* sched, hipStream_t stream) {
  SCP_REQUIRE(conv_chain_supported(C, H, W), "conv_chain: %d channels at %dx%d has no chain kernel", C, H, W);
  SCP_REQUIRE(nconv > 0 && nconv % 2 == 0 && N > 0, "conv_chain: %d convolutions, batch %d", nconv, N);
  ChainLaunch L{in, out, wpk, bias, N, nconv, sched};
  const int cus = conv_device_cus();
  const int grid = N < cus ? N : cus;
  const bool bf = dtype == SCPOSE_DT_BF16;
  if (C == 128) return bf ? chain_launch_one<0, 128, 16, 16>(L, grid, stream) : chain_launch_one<1, 128, 16, 16>(L, grid, stream);
  return bf ? chain_launch_one<0, 256, 8, 8>(L, grid, stream) : chain_launch_one<1, 256, 8, 8>(L, grid, stream);
}

}  // namespace scpose
