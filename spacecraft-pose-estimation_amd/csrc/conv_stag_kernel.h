// Staggered two-group implicit-GEMM 3x3 convolution for the weight-streaming (wide, low
// resolution) branches: same math, data layout, K ordering and epilogue as conv_pipe_kernel.h.
//
// Why a second schedule.  Measured on MI355X (tools_dev/micro/dma_bench.hip and the in-kernel phase
// stamps): the LDS-DMA path of a CU accepts ~16 B/clk, and a wave that issues global_load_lds into a
// full queue simply stalls -- with one wave per SIMD the matrix pipe idles for bytes/16 cycles per
// stage, about as long as the MFMAs themselves (37 % DMA issue vs 35 % MFMA in the lock-step
// version).  Feeding the pieces from inside the MFMA loop is worse (the stall just moves).  The
// only thing that hides it is ANOTHER wave on the same SIMD that has MFMAs to issue meanwhile.
//
// Schedule.  One 512-thread workgroup per CU = two wave groups g = 0,1 (two waves per SIMD, <= 256
// registers each).  Both walk the same sequence of stages s = (work item, K-chunk) on two
// DIFFERENT pixel tiles, sharing ONE staged copy of each weight chunk (halves the weight bytes per
// MFMA), but half a stage apart.  Phases are separated by workgroup barriers:
//       phase 2s+1+g : group g runs the MFMAs of stage s           (matrix pipe)
//       phase 2s+2+g : group g stores stage s's results if it retired a tile, then issues the
//                      LDS-DMA of stage s+2: its own input tile and ITS HALF of weight chunk s+2
// so in every phase one group is on the matrix pipe while the other sits in the DMA queue.
// Hand-off rules (DMA data is visible to other waves only after the issuing wave's vmcnt wait AND a
// barrier):
//   * a group drains its DMA (s_waitcnt vmcnt(0)) at the END of its MFMA phase, i.e. one full phase
//     after issuing it -- nothing is waited for while it is still landing;
//   * input tiles: 2 buffers per group (stage s+2 overwrites the buffer of stage s, whose MFMAs the
//     same group finished one phase earlier);
//   * weights: 3 buffers.  Chunk s+2 is written during phases 2s+2 .. 2s+4 into buffer (s+2) % 3,
//     last read (chunk s-1, by group 1) in phase 2s; both halves are visible from phase 2s+5, the
//     first MFMA phase that needs them.
#pragma once
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"

namespace scpose {

// s_waitcnt takes an immediate: a wave-uniform runtime count goes through a switch.
#define SCP_WAITVM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    SCP_WAITVM_CASE(0) SCP_WAITVM_CASE(1) SCP_WAITVM_CASE(2) SCP_WAITVM_CASE(3) SCP_WAITVM_CASE(4)
    SCP_WAITVM_CASE(5) SCP_WAITVM_CASE(6) SCP_WAITVM_CASE(7) SCP_WAITVM_CASE(8) SCP_WAITVM_CASE(9)
    SCP_WAITVM_CASE(10) SCP_WAITVM_CASE(11) SCP_WAITVM_CASE(12) SCP_WAITVM_CASE(13) SCP_WAITVM_CASE(14)
    SCP_WAITVM_CASE(15) SCP_WAITVM_CASE(16) SCP_WAITVM_CASE(17) SCP_WAITVM_CASE(18) SCP_WAITVM_CASE(19)
    SCP_WAITVM_CASE(20) SCP_WAITVM_CASE(21) SCP_WAITVM_CASE(22) SCP_WAITVM_CASE(23) SCP_WAITVM_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // conservative: drain everything
  }
}
#undef SCP_WAITVM_CASE

template <int DT, int MREP, int NREP>
__global__ __launch_bounds__(512, 2) void conv_stag_kernel(const ConvLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int KS = 3, KK = 9, MAXP = 2;
  constexpr int MT = 16 * MREP;
  constexpr int NPAIR = (NREP + 1) / 2;

  // LDS: [k-offset tables 512 B][bias, packed row order][W x 3][X x 4 = (group, parity)]
  int* koff = reinterpret_cast<int*>(smem);
  float* bias_l = reinterpret_cast<float*>(smem + 512);
  char* wl0 = smem + 512 + p.lds_bias;
  char* xl0 = wl0 + 3 * p.lds_w;

  const int lane = threadIdx.x & 63;
  const int grp = threadIdx.x >> 8;                          // wave group = tile of the pair
  const int wave = (threadIdx.x >> 6) & 3, tid = threadIdx.x & 255;
  const int q = lane >> 4, r = lane & 15;
  const int HW = p.H * p.W;
  const int HP = p.halo_h * p.halo_w;
  const int npix = p.th * p.tw;
  const int planes_last = p.cin_planes - (p.nchunks - 1) * p.cp;

  if (threadIdx.x < 128) {  // K-offset tables: LDS byte offset of k-group qq at k-step st (0 for padding)
    const int tbl = threadIdx.x >> 6, e = threadIdx.x & 63;
    const int planes = tbl ? planes_last : p.cp;
    const int npt = (planes >> 1) * KK;
    const int st = e >> 2, qq = e & 3;
    const int pt = 2 * st + (qq >> 1);
    const int pp = pt / KK, tap = pt - pp * KK;
    const int ky = tap / KS, kx = tap - ky * KS;
    koff[threadIdx.x] = pt < npt ? (2 * pp + (qq & 1)) * p.plane_stride + (ky * p.halo_w + kx) * 16 : 0;
  }
  for (int i = threadIdx.x; i < p.n_mblk * MT; i += 512) bias_l[i] = p.bias[i];

  int hy[MAXP], hx[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int hp = i * 256 + tid;
    hy[i] = hp < HP ? hp / p.halo_w : -1;
    hx[i] = hp < HP ? hp - hy[i] * p.halo_w : 0;
  }
  auto pix_yx = [&](int n, int& y, int& x) {
    const int pidx = (wave * NREP + n) * 16 + r;
    if (pidx < npix) { y = pidx / p.tw; x = pidx - y * p.tw; } else { y = -1; x = 0; }
  };
  int pixoff[NREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n) {
    int y, x;
    pix_yx(n, y, x);
    pixoff[n] = y >= 0 ? (y * p.halo_w + x) * 16 : 0;
  }
  const int half = lane >> 5, psel = q & 1;
  int epy[NPAIR], epx[NPAIR];
#pragma unroll
  for (int np = 0; np < NPAIR; ++np) {
    const bool paired = 2 * np + 1 < NREP;
    int y, x;
    pix_yx(half && paired ? 2 * np + 1 : 2 * np, y, x);
    epy[np] = (half && !paired) ? -1 : y;
    epx[np] = x;
  }
  const int cout_planes = (p.cout + 7) >> 3;
  const size_t HoWo = (size_t)p.Ho * p.Wo;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int it_begin = wg * p.items_per_wg;
  const int it_end = min(p.items_total, it_begin + p.items_per_wg);
  const int S = (it_end - it_begin) * p.nchunks;            // stages of this workgroup
  const size_t chunk_wbytes = (size_t)p.ksteps_full * (4 * MT * 16);

  auto stage_of = [&](int s, int& it, int& c) {
    const int k = s / p.nchunks;
    it = it_begin + k; c = s - k * p.nchunks;
  };
  auto decode_tile = [&](int it, int& img, int& oy0, int& ox0) {   // this group's tile of the pair
    const int t = (it / p.n_mblk) * 2 + grp;
    if (t >= p.tiles_total) { img = -1; oy0 = ox0 = 0; return; }
    img = t / tiles_per_img;
    const int rem = t - img * tiles_per_img;
    const int ty = rem / p.tiles_x;
    oy0 = ty * p.th; ox0 = (rem - ty * p.tiles_x) * p.tw;
  };

  // LDS-DMA of stage s issued by THIS group: its input tile chunk and its half of the weight chunk
  auto issue_stage = [&](int s) -> int {      // returns the number of DMA instructions THIS WAVE issued
    int cnt = 0;
    if (s >= S) return cnt;
    int it, c;
    stage_of(s, it, c);
    const int planes = c == p.nchunks - 1 ? planes_last : p.cp;
    {  // weight half
      const int ksteps = (((planes >> 1) * KK) + 1) >> 1;
      const int nbytes = ksteps * (4 * MT * 16);
      const int slice = ((nbytes + 1) / 2 + 4095) & ~4095;
      const int lo = grp * slice, hi = min(nbytes, lo + slice);
      const char* ws = static_cast<const char*>(p.wpk) + ((size_t)(it % p.n_mblk) * p.nchunks + c) * chunk_wbytes;
      char* wl = wl0 + (s % 3) * p.lds_w;
      for (int o = lo; o < hi; o += 4096) {
        const int mine = o + tid * 16;
        if (o + wave * 1024 < hi) ++cnt;            // lane 0 of this wave is inside the slice
        if (mine < hi) dma16(ws + mine, wl + o + wave * 1024);
      }
    }
    int img, oy0, ox0;
    decode_tile(it, img, oy0, ox0);
    if (img < 0 || SCP_DBG(p, 4)) return cnt;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const char* inb = static_cast<const char*>(p.in) + ((size_t)img * p.cin_planes + (size_t)c * p.cp) * HW * 16;
    char* xl = xl0 + (grp * 2 + (s & 1)) * p.lds_x;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i * 256 + wave * 64 < HP) cnt += planes;  // lane 0 of this wave is inside the halo tile
      if (hy[i] >= 0) {
        const int iy = iy0 + hy[i], ix = ix0 + hx[i];
        const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const size_t g = ok ? (size_t)(iy * p.W + ix) * 16 : 0;
        for (int pl = 0; pl < planes; ++pl) {
          const char* src = ok ? inb + (size_t)pl * HW * 16 + g : static_cast<const char*>(p.zero16);
          dma16(src, xl + pl * p.plane_stride + (i * 256 + wave * 64) * 16);
        }
      }
    }
    return cnt;
  };

  f32x4 acc[MREP][NREP];
  auto slot_off = [&](int m, int np, int mb, int img, int oy0, int ox0) -> uint32_t {
    const int co_plane = mb * MT + m * 16 + psel * 8;
    const int oy = oy0 + epy[np], ox = ox0 + epx[np];
    const bool ok = img >= 0 && epy[np] >= 0 && oy < p.Ho && ox < p.Wo && co_plane < p.cout && !SCP_DBG(p, 2);
    return ok ? (uint32_t)((((size_t)(co_plane >> 3)) * HoWo + (size_t)oy * p.Wo + ox) * 16) : 0xffffffffu;
  };

  // development instrumentation (dbg & 8)
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { return SCP_DBG(p, 8) ? __builtin_amdgcn_s_memtime() : 0ull; };

  // phase 0: stages 0 and 1 (each group its tile chunks and weight halves)
  issue_stage(0);
  issue_stage(1);
  __syncthreads();   // tables, bias, stages 0/1 (vmcnt(0) + barrier)

  for (int phi = 1; phi <= 2 * S + 1; ++phi) {
    const int u = phi - 1 - grp;
    const unsigned long long t0 = now();
    if (u >= 0 && !(u & 1) && (u >> 1) < S) {
      // ------------------------------ MFMA phase of stage s ------------------------------
      const int s = u >> 1;
      int it, c;
      stage_of(s, it, c);
      const bool last = c == p.nchunks - 1;
      int img, oy0, ox0;
      decode_tile(it, img, oy0, ox0);
      if (c == 0) {
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int n = 0; n < NREP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const unsigned long long t1 = now();
      if (img >= 0) {
        const int planes = last ? planes_last : p.cp;
        const int npt = (planes >> 1) * KK;
        const int ksteps = SCP_DBG(p, 1) ? 0 : (npt + 1) >> 1;
        const char* xl = xl0 + (grp * 2 + (s & 1)) * p.lds_x;
        const char* wq = wl0 + (s % 3) * p.lds_w + (q * MT + r) * 16;
        const int* kt = koff + (last ? 64 : 0) + q;
        // fragments one k-step ahead, k-offsets two: the partner wave on this SIMD sits in the DMA queue
        // during this phase, so nothing else hides the LDS latency
        const int klast = ksteps - 1;
        frag_t a0[MREP], b0[NREP], a1[MREP], b1[NREP];
        auto load_frags = [&](int st, int ko, frag_t* a, frag_t* b) {
#pragma unroll
          for (int m = 0; m < MREP; ++m) a[m] = *reinterpret_cast<const frag_t*>(wq + st * (4 * MT * 16) + m * 256);
#pragma unroll
          for (int n = 0; n < NREP; ++n) b[n] = *reinterpret_cast<const frag_t*>(xl + ko + pixoff[n]);
        };
        auto mfmas = [&](const frag_t* a, const frag_t* b) {
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = mfma16<T>(a[m], b[n], acc[m][n]);
        };
        if (ksteps > 0) {
          int ko_a = kt[0], ko_b = kt[4];
          load_frags(0, ko_a, a0, b0);
          int st = 0;
          for (; st + 1 < ksteps; st += 2) {
            ko_a = kt[min(st + 2, klast) * 4];
            load_frags(st + 1, ko_b, a1, b1);
            mfmas(a0, b0);
            ko_b = kt[min(st + 3, klast) * 4];
            load_frags(min(st + 2, klast), ko_a, a0, b0);
            mfmas(a1, b1);
          }
          if (ksteps & 1) mfmas(a0, b0);
        }
      }
      const unsigned long long t2 = now();
      // the DMA and stores this group issued one phase ago have had this whole phase to complete
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t3 = now();
      if SCP_DBG(p, 8) { tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; }
    } else if (u >= 1 && (u & 1)) {
      // ------- issue the DMA of stage s+2; while it queues, retire stage s's tile if s was its last chunk -------
      const int s = (u - 1) >> 1;
      if (s < S) {
        int it, c;
        stage_of(s, it, c);
        int img, oy0, ox0;
        decode_tile(it, img, oy0, ox0);
        const bool retire = c == p.nchunks - 1 && img >= 0;
        const int mb = it % p.n_mblk;
        const size_t img_off = (size_t)(img < 0 ? 0 : img) * cout_planes * HoWo * 16;
        issue_stage(s + 2);   // this wave now sits in the DMA queue while the other group runs its MFMAs
        if (retire) {
          // Row by row: residual slots (plain loads), bias from LDS, finalize, 16-byte stores.  hipcc
          // drains vmcnt for them, i.e. this phase also waits for the burst above to land -- only on
          // the one stage in nchunks that retires a tile, and it keeps the kernel free of long-lived
          // epilogue registers (spills cost a vmcnt(0) each, in the middle of the DMA stream).
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            u32x4 rv[NPAIR];
            uint32_t off[NPAIR];
#pragma unroll
            for (int np = 0; np < NPAIR; ++np) {
              off[np] = slot_off(m, np, mb, img, oy0, ox0);
              rv[np] = u32x4{0u, 0u, 0u, 0u};
              if (p.res && off[np] != 0xffffffffu)
                rv[np] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.res) + img_off + off[np]);
            }
            const float4 bs = *reinterpret_cast<const float4*>(bias_l + mb * MT + m * 16 + q * 4);
#pragma unroll
            for (int np = 0; np < NPAIR; ++np) {
              const int n0 = 2 * np, n1 = (2 * np + 1 < NREP) ? 2 * np + 1 : 2 * np;
              uint32_t a[4], b[4];
              a[0] = __float_as_uint(acc[m][n0][0] + bs.x); a[1] = __float_as_uint(acc[m][n0][1] + bs.y);
              a[2] = __float_as_uint(acc[m][n0][2] + bs.z); a[3] = __float_as_uint(acc[m][n0][3] + bs.w);
              b[0] = __float_as_uint(acc[m][n1][0] + bs.x); b[1] = __float_as_uint(acc[m][n1][1] + bs.y);
              b[2] = __float_as_uint(acc[m][n1][2] + bs.z); b[3] = __float_as_uint(acc[m][n1][3] + bs.w);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
                a[jj] = sw[0]; b[jj] = sw[1];
              }
              float v[8];
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) { v[jj] = __uint_as_float(a[jj]); v[4 + jj] = __uint_as_float(b[jj]); }
              v[0] += from_bits<T>(rv[np][0] & 0xffff); v[1] += from_bits<T>(rv[np][0] >> 16);
              v[2] += from_bits<T>(rv[np][1] & 0xffff); v[3] += from_bits<T>(rv[np][1] >> 16);
              v[4] += from_bits<T>(rv[np][2] & 0xffff); v[5] += from_bits<T>(rv[np][2] >> 16);
              v[6] += from_bits<T>(rv[np][3] & 0xffff); v[7] += from_bits<T>(rv[np][3] >> 16);
              if (p.relu) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = fmaxf(v[jj], 0.f);
              }
              u32x4 ov;
              ov[0] = (uint32_t)to_bits<T>(v[0]) | ((uint32_t)to_bits<T>(v[1]) << 16);
              ov[1] = (uint32_t)to_bits<T>(v[2]) | ((uint32_t)to_bits<T>(v[3]) << 16);
              ov[2] = (uint32_t)to_bits<T>(v[4]) | ((uint32_t)to_bits<T>(v[5]) << 16);
              ov[3] = (uint32_t)to_bits<T>(v[6]) | ((uint32_t)to_bits<T>(v[7]) << 16);
              if (off[np] != 0xffffffffu) *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + img_off + off[np]) = ov;
            }
          }
        }
      }
      if SCP_DBG(p, 8) { tph[5] += now() - t0; }
    }
    const unsigned long long tb = now();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if SCP_DBG(p, 8) tph[4] += now() - tb;
  }
  if (SCP_DBG(p, 8) && SCP_DBG_BUF(p) && lane == 0)
    for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 6 + k] = tph[k];
}

template <int DT, int MREP, int NREP>
int32_t stag_launch_one(const ConvLaunch& L, size_t lds, hipStream_t st) {
  auto kern = conv_stag_kernel<DT, MREP, NREP>;
  static LdsOptIn big_lds;   // per device (common.h)
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(512), lds, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

template <int DT>
int32_t stag_dispatch(int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
#define SCP_STAG(M, N) if (mrep == M && nrep == N) return stag_launch_one<DT, M, N>(L, lds, st);
  SCP_STAG(4, 1) SCP_STAG(4, 2) SCP_STAG(4, 3) SCP_STAG(4, 4)
  SCP_STAG(6, 1) SCP_STAG(6, 2) SCP_STAG(6, 3) SCP_STAG(6, 4)
#undef SCP_STAG
  set_error("conv: staggered variant mrep %d nrep %d not built", mrep, nrep);
  return SCPOSE_E_INVALID;
}

}  // namespace scpose
