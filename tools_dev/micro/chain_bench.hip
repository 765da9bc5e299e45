// Microbenchmark 13 (round 6): what keeps the matrix pipes ~45 % busy even in a kernel with NO producer / consumer coupling and NO barrier inside a convolution?
// The branch-chain kernel of csrc/conv_chain.hip (copied below with ablation switches; 128 channels at 16 x 16, eight convolutions per frame, a frame per
// workgroup, activations in LDS, weights L2 -> registers) on N frames of N(0,1) data:  N = 256 (a frame on every CU) and N = 16 (16 CUs busy).
//   ABL 0 full | 1 weights loaded for the first k-steps of a convolution only (no L2 stream) | 2 B fragments likewise (no LDS reads in the loop) | 3 both | 4 no MFMAs
// Prints us per launch, the in-kernel shader clock (s_memtime / s_memrealtime x 100 MHz, median over workgroups) and MFMA busy = MFMA cycles / kernel cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spacecraft-pose-estimation_amd/csrc -o /tmp/chain_bench tools_dev/micro/chain_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include <vector>
#include <algorithm>
#include "conv_device.h"

using namespace scpose;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4c;
struct ChainLaunch { const void* in; void* out; const void* wpk; const float* bias; int32_t N, nconv; uint32_t* sched; };
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

// ABL bits: 1 = A fragments (weights, L2 -> registers) loaded for the first k-steps of a convolution only, 2 = B fragments (LDS) likewise, 4 = no MFMAs
template <int DT, int C, int H, int W, int ABL>
__global__ __launch_bounds__(512, 2) void conv_chain_kernel(const ChainLaunch p, unsigned long long* clk) {
  typedef ChainGeom<C, H, W> G;
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const zero = smem;
  char* const xbuf = smem + 256;
  char* const mbuf = xbuf + G::BUF;

  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
        if constexpr (S < G::KS && (!(ABL & 1) || S < 4)) {
#pragma unroll
          for (int m = 0; m < 2; ++m) af[S & 3][m] = *reinterpret_cast<const frag_t*>(wl + (size_t)S * (G::MBK * 1024) + m * 1024);
        }
      };
      auto fetch_b = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < G::KS && (!(ABL & 2) || S < 2)) {
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
          for (int m = 0; m < 2; ++m) { if constexpr (!(ABL & 4)) acc[m][n] = mfma16<T>(af[S & 3][m], bf[S & 1][n], acc[m][n]); else acc[m][n][0] += (float)af[S & 3][m][0] + (float)bf[S & 1][n][0]; }
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
  if (tid == 0) { tile_retire(p.sched); clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}


static unsigned short bf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float gauss() { const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)); }

template <int ABL>
void run(const char* name, int N, const ChainLaunch& L0, unsigned long long* clk, bool print = true) {
  typedef ChainGeom<128, 16, 16> G;
  auto kern = conv_chain_kernel<0, 128, 16, 16, ABL>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  ChainLaunch L = L0; L.N = N;
  const int grid = N < 256 ? N : 256, iters = 30;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS, 0, L, clk);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS, 0, L, clk);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> ghz, cyc;
  for (int i = 0; i < grid; ++i) { ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); cyc.push_back((double)h[2 * i]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double us = ms * 1e3 / iters, frames_per_wg = (double)N / grid;
  const double mfma_cycles = frames_per_wg * 8.0 * 36 * 2 * 16 * 16;   // per SIMD: 8 convolutions x 36 k-steps x 2 waves x 16 MFMAs x 16 cycles
  if (print) printf("%-58s N=%3d  %7.1f us/launch  clock %.2f GHz  kernel %7.0f cycles (median WG)  MFMA busy %.2f\n", name, N, us, ghz[grid / 2], cyc[grid / 2], (ABL & 4) ? 0.0 : mfma_cycles / cyc[grid / 2]);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const int NMAX = 256, C = 128, HW = 256, nconv = 8;
  const size_t act = (size_t)NMAX * C * HW, wel = (size_t)nconv * 36 * 8 * 512;
  std::vector<unsigned short> hx(act), hw(wel);
  srand(7);
  for (auto& v : hx) v = bf16(fabsf(gauss()));                       // post-ReLU-like input
  for (auto& v : hw) v = bf16(gauss() / 48.0f);                      // He-like scale: activations stay O(1) through the chain
  std::vector<float> hb(nconv * C, 0.01f);
  void *dx, *dy, *dw; float* db; uint32_t* sched; unsigned long long* clk;
  hipMalloc(&dx, act * 2); hipMalloc(&dy, act * 2); hipMalloc(&dw, wel * 2); hipMalloc(&db, hb.size() * 4); hipMalloc(&sched, 64); hipMalloc(&clk, 2 * 256 * 8);
  hipMemcpy(dx, hx.data(), act * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wel * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); hipMemset(sched, 0, 64);
  ChainLaunch L{dx, dy, dw, db, NMAX, nconv, sched};
  for (int w = 0; w < 3; ++w) run<0>("warm-up", 256, L, clk, false);
  for (int round = 0; round < 2; ++round) {
    printf("-- round %d\n", round);
    for (int N : {256, 16}) {
      run<0>("full", N, L, clk);
      run<1>("weights loaded for the first k-steps only (no L2 stream)", N, L, clk);
      run<2>("B fragments for the first k-steps only (no LDS reads)", N, L, clk);
      run<3>("neither (MFMAs + epilogues + frame load / store)", N, L, clk);
      run<4>("no MFMAs (loads, LDS reads, epilogues)", N, L, clk);
    }
  }
  return 0;
}
