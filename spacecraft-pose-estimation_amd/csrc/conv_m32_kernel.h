// Persistent implicit-GEMM 3x3 convolution on v_mfma_f32_32x32x16_{bf16,f16} (gfx950).
//
// Same role and data layout as conv_pipe_kernel.h (reference: conv3x3 + eval BatchNorm2d + ReLU /
// residual add of landmark_regression/lib/models/pose_hrnet.py:22-25, :41-57), for the layers whose
// Cout is a multiple of 32.  Why a second kernel -- measured on MI355X (tools_dev/micro):
//   * ONE wave per SIMD issues a 16x16x32 MFMA every ~26 cycles (61 % of the matrix pipe) but a
//     32x32x16 MFMA every 32 cycles (100 %): the 16x16 shape needs two waves per SIMD, i.e. half
//     the registers per wave, to fill the pipe; the 32x32 shape fills it with one wave that owns
//     all 512 registers of its SIMD lanes;
//   * a burst of LDS-DMA (global_load_lds) issued in front of a 32x32x16 MFMA loop overlaps with
//     it (48 KiB per 108 MFMAs: +18 % over the MFMAs alone), which was not the case for the
//     16x16x32 loop (+190 %).
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] X[k][pixel];  A = weights, B = activations.
//   * k-step = (input plane pair pp, tap): k-group kg = lane>>5 reads the 16-byte (pixel, 8-channel)
//     vector of plane 2*pp + kg at that tap.  The two planes live in different ds_read_b128 lane
//     groups, so no bank constraint couples them.
//   * accumulator tile 32 rows x 32 pixels: lane (half = lane>>5, col = lane&31) holds rows
//     8*(i/4) + 4*half + (i%4), i = 0..15, of pixel col.  One v_permlane32_swap per register pair
//     gives the lower half-wave the 8 channels of output plane 2g, the upper half-wave those of
//     plane 2g+1 (g = 0,1): 16-byte stores, 512 contiguous bytes per half-wave, natural channel
//     order (no row permutation at pack time).
//   * workgroup = 4 waves = WM (along Cout) x WN (along pixels); a wave owns MR x NR accumulator
//     tiles (MT = 32*MR*WM rows of one Cout block, 32*NR pixels).
//   * work item = (Cout block, group of `nseg` spatial tiles th x tw).  The segments are flattened
//     into one pixel index space of nseg*th*tw pixels that is cut into 32-pixel MFMA columns, so a
//     12x12 map (144 pixels) packs two images into 9 columns instead of wasting half a column per
//     image.  All segments' halos of a K-chunk are staged together with that chunk's weights.
//   * pipeline: stage = one K-chunk (cp planes).  DMA of chunk c+1 (weights + halos) is issued as
//     one burst, then the MFMA loop of chunk c runs, then vmcnt(0) + one barrier.  Residual slots
//     are prefetched (inline-asm loads) ahead of the last chunk's loop; results are finalised into
//     16-byte slots before the barrier and stored after it.
#pragma once
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // dma16, u32x4, address-space typedefs

namespace scpose {

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename T>
__device__ __forceinline__ f32x16 mfma32(typename FragOf<T>::type a, typename FragOf<T>::type b, f32x16 c) {
  if constexpr (__is_same(T, __bf16))
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// The k-loop is written with single-instruction asm statements: hipcc otherwise (a) keeps the
// accumulators in VGPRs around the loop and copies all of them to AGPRs and back once per K-chunk,
// and (b) waits lgkmcnt(0) right after issuing each fragment read.  "+a" pins the accumulators to
// AGPRs; the reads of k-step s+1 are issued before the MFMAs of step s and waited for after them.
// AGPR = true pins the accumulator to AGPRs (one wave per SIMD, 512 registers); with two waves per
// SIMD hipcc splits the 256 registers 128/128 as soon as a kernel touches AGPRs, so those variants
// keep everything in VGPRs.
template <typename T, bool AGPR>
__device__ __forceinline__ void mfma32_acc(f32x16& c, const typename FragOf<T>::type& a, const typename FragOf<T>::type& b) {
  if constexpr (__is_same(T, __bf16)) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  } else {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n * f.add) >> f.shift);
}

// halo pixels staged per thread and plane: 4 (<= 1024 per chunk) with one workgroup per CU, 2 with two
constexpr int m32_maxp(int occ) { return occ == 1 ? 4 : 2; }

// OCC = resident workgroups per CU the variant is built for (2: <= 256 registers per wave).
template <int DT, int KS, int STRIDE, int MR, int NR, int WM, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_m32_kernel(const ConvLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int WN = 4 / WM;
  constexpr int MT = 32 * MR * WM;
  constexpr int KK = KS * KS;
  constexpr int MAXP = m32_maxp(OCC);

  // LDS: [k-offset table 1 KiB][bias, natural channel order][W buffers x nbuf_w][X buffers x 2]
  int* koff = reinterpret_cast<int*>(smem);
  float* bias_l = reinterpret_cast<float*>(smem + 1024);
  char* wl0 = smem + 1024 + p.lds_bias;
  char* xl0 = wl0 + p.nbuf_w * p.lds_w;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases / M0
  const int half = lane >> 5, r = lane & 31;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int HW = p.H * p.W;
  const int HP = p.halo_h * p.halo_w;
  const int npix = p.th * p.tw;
  const int nseg = p.nt;
  const int P = nseg * npix;
  const int ksteps_full = (p.cp >> 1) * KK;
  const int planes_last = p.cin_planes - (p.nchunks - 1) * p.cp;
  const bool w_resident = p.nbuf_w == 1;

  if (tid < 2 * ksteps_full) {   // LDS byte offset of k-group kg at k-step st
    const int st = tid >> 1, kg = tid & 1;
    const int pp = st / KK, tap = st - pp * KK;
    const int ky = tap / KS, kx = tap - ky * KS;
    koff[tid] = (2 * pp + kg) * p.plane_stride + (ky * p.halo_w + kx) * 16;
  }
  for (int i = tid; i < p.n_mblk * MT; i += 256) bias_l[i] = p.bias[i];

  // this thread's halo pixels (segment, y, x) and this lane's output pixels per MFMA column
  int hs[MAXP], hy[MAXP], hx[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int hp = i * 256 + tid;
    if (hp < nseg * HP) {
      hs[i] = fdiv(hp, p.fd_hp);
      const int rem = hp - hs[i] * HP;
      hy[i] = fdiv(rem, p.fd_halo_w); hx[i] = rem - hy[i] * p.halo_w;
    } else {
      hs[i] = -1; hy[i] = hx[i] = 0;
    }
  }
  int pixoff[NR];
  auto pixel_of = [&](int n, int& ps, int& py, int& px) {   // column n, this lane -> (segment, y, x); ps < 0 = none
    const int pidx = (wn * NR + n) * 32 + r;
    if (pidx < P) {
      ps = fdiv(pidx, p.fd_npix);
      const int rem = pidx - ps * npix;
      py = fdiv(rem, p.fd_tw); px = rem - py * p.tw;
    } else {
      ps = -1; py = px = 0;
    }
  };
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    int ps, py, px;
    pixel_of(n, ps, py, px);
    pixoff[n] = ps >= 0 ? (ps * HP + (py * STRIDE) * p.halo_w + px * STRIDE) * 16 : 0;
  }
  const int cout_planes = (p.cout + 7) >> 3;
  const size_t HoWo = (size_t)p.Ho * p.Wo;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int it_begin = wg * p.items_per_wg;
  const int it_end = min(p.items_total, it_begin + p.items_per_wg);
  const size_t chunk_wbytes = (size_t)ksteps_full * (2 * MT * 16);

  auto decode_tile = [&](int it, int seg, int& img, int& oy0, int& ox0) {
    const int t = fdiv(it, p.fd_nmblk) * nseg + seg;
    if (seg < 0 || t >= p.tiles_total) { img = -1; oy0 = ox0 = 0; return; }
    img = fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = fdiv(rem, p.fd_tiles_x);
    oy0 = ty * p.th; ox0 = (rem - ty * p.tiles_x) * p.tw;
  };

  // global byte offset (plane 0 of its image) of each halo pixel of the item being staged; ~0 = zero page
  size_t xoff[MAXP];
  auto locate_halo = [&](int it) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      int img, oy0, ox0;
      decode_tile(it, hs[i], img, oy0, ox0);
      const int iy = oy0 * STRIDE - (KS / 2) + hy[i], ix = ox0 * STRIDE - (KS / 2) + hx[i];
      const bool ok = img >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      // buffer addressing (p.in_bytes != 0): the slot holds a 32-bit offset, BUF_OOB for padding pixels
      if (p.in_bytes) xoff[i] = ok ? (size_t)((uint32_t)(img * p.cin_planes * HW + iy * p.W + ix) * 16u) : (size_t)BUF_OOB;
      else xoff[i] = ok ? ((size_t)(SCP_DBG(p, 64) ? 0 : img) * p.cin_planes * HW + (size_t)(iy * p.W + ix)) * 16 : ~(size_t)0;
    }
  };
  const buf_rsrc_t rs_in = make_buf(p.in, p.in_bytes);
  // K-chunks are summed in their natural order in every workgroup.  (A per-workgroup rotated order, meant to
  // spread the weight-chunk reads of lock-stepped CUs over L2, measured no faster and made a frame's result depend
  // on which workgroup computed it, i.e. on its position in the batch.)
  auto issue_x = [&](int cl, int xb) {
    if SCP_DBG(p, 4) return;
    const int c = cl;
    const int planes = c == p.nchunks - 1 ? planes_last : p.cp;
    const char* inb = static_cast<const char*>(p.in) + (size_t)c * p.cp * HW * 16;
    char* xl = xl0 + xb * p.lds_x;
    if (p.in_bytes) {   // conv_pipe_kernel.h: dma16_buf (plane displacement in an SGPR, hardware zero fill for padding)
#pragma unroll
      for (int i = 0; i < MAXP; ++i)
        if (hs[i] >= 0)
          for (int pl = 0; pl < planes; ++pl)
            dma16_buf(rs_in, (uint32_t)xoff[i], (uint32_t)((c * p.cp + pl) * HW) * 16u, xl + pl * p.plane_stride + (i * 256 + wave * 64) * 16);
      return;
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (hs[i] >= 0) {
        const bool ok = xoff[i] != ~(size_t)0;
        for (int pl = 0; pl < planes; ++pl) {
          const char* src = ok ? inb + xoff[i] + (size_t)pl * HW * 16 : static_cast<const char*>(p.zero16);
          dma16(src, xl + pl * p.plane_stride + (i * 256 + wave * 64) * 16);
        }
      }
    }
  };
  auto issue_w = [&](int it, int cl, int wb) {
    const int c = cl;
    const int planes = c == p.nchunks - 1 ? planes_last : p.cp;
    const int nbytes = (planes >> 1) * KK * (2 * MT * 16);
    const buf_rsrc_t rs_w = make_buf(p.wpk, (uint32_t)(p.n_mblk * p.nchunks * (int)chunk_wbytes));   // packed weights: far below 4 GiB
    const uint32_t wchunk = (uint32_t)(((it - fdiv(it, p.fd_nmblk) * p.n_mblk) * p.nchunks + c) * (int)chunk_wbytes);
    char* wl = wl0 + wb * p.lds_w;
    for (int o = 0; o < nbytes; o += 4096) {
      const int mine = o + tid * 16;
      if (mine < nbytes) dma16_buf(rs_w, (uint32_t)tid * 16u, wchunk + (uint32_t)o, wl + o + wave * 1024);
    }
  };

  f32x16 acc[MR][NR];
  constexpr int NSLOT = OCC == 1 ? MR : 1;
  u32x4 slot[NSLOT][NR][2];  // 16-byte slots of the tile being retired: residual in, result out
  size_t pbase[NR];          // byte offset of (image, plane 0, pixel) in out / res; ~0 = masked

  int wc = 0;                // running chunk counter: W buffer = wc & 1
  if (it_begin < it_end) {
    locate_halo(it_begin);
    issue_w(it_begin, 0, 0);
    issue_x(0, 0);
  }
  __syncthreads();           // table, bias, stage 0

  int xb = 0;
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { return SCP_DBG(p, 8) ? __builtin_amdgcn_s_memtime() : 0ull; };
  for (int it = it_begin; it < it_end; ++it) {
    const int mb = it - fdiv(it, p.fd_nmblk) * p.n_mblk;
    const int plane0 = (mb * MT + wm * (MR * 32)) >> 3;   // first output plane of this wave's rows
    for (int c = 0; c < p.nchunks; ++c, ++wc) {
      const unsigned long long t0 = now();
      const bool last = c == p.nchunks - 1;
      const int nit = last ? it + 1 : it, nc = last ? 0 : c + 1;
      const bool have_next = nit < it_end;
      if (c == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][n][j] = 0.f;
      }
      // (1) residual slots: inline-asm loads (not counted by the compiler), consumed after the wait
      if (last) {
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          int ps, py, px, img, oy0, ox0;
          pixel_of(n, ps, py, px);
          decode_tile(it, ps, img, oy0, ox0);
          const int oy = oy0 + py, ox = ox0 + px;
          const bool ok = img >= 0 && oy < p.Ho && ox < p.Wo && !SCP_DBG(p, 2);
          pbase[n] = ok ? ((size_t)img * cout_planes * HoWo + (size_t)oy * p.Wo + ox) * 16 : ~(size_t)0;
        }
        if constexpr (OCC == 1)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              slot[m][n][g] = u32x4{0u, 0u, 0u, 0u};
              if (p.res) {
                const int pl = plane0 + m * 4 + 2 * g + half;
                const bool ok = pbase[n] != ~(size_t)0 && pl < cout_planes;
                const char* rp = ok ? static_cast<const char*>(p.res) + pbase[n] + (size_t)pl * HoWo * 16
                                    : static_cast<const char*>(p.zero16);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(slot[m][n][g]) : "v"(rp) : "memory");
              }
            }
      }
      // (2) DMA burst for the next stage
      const unsigned long long ta = now();
      unsigned long long tb = ta, tc = ta;
      if (have_next) {
        if (nc == 0) locate_halo(nit);
        tb = now();
        issue_x(nc, xb ^ 1);
        tc = now();
        if (!w_resident) issue_w(nit, nc, (wc + 1) & 1);
      }
      const unsigned long long t1 = now();
      {  // (3) MFMA loop: plane pairs x KK taps, taps unrolled so that every LDS offset is an immediate
        const int planes = last ? planes_last : p.cp;
        const int npp = SCP_DBG(p, 1) ? 0 : planes >> 1;
        const uint32_t xl = (uint32_t)(size_t)(xl0 + xb * p.lds_x) + half * p.plane_stride;
        uint32_t wa = (uint32_t)(size_t)(wl0 + (w_resident ? 0 : (wc & 1)) * p.lds_w) + (half * MT + wm * (MR * 32) + r) * 16;
        const int hw16 = p.halo_w * 16;
        frag_t a0[MR], b0[NR], a1[MR], b1[NR];
        uint32_t brow[NR];   // LDS address of (plane pair, tap row ky) for each MFMA column of this lane
        auto set_row = [&](int pp, int ky) {
#pragma unroll
          for (int n = 0; n < NR; ++n) brow[n] = xl + pp * 2 * p.plane_stride + ky * hw16 + pixoff[n];
        };
        // fragment reads of tap TAPN (0..KK: KK = tap 0 of the next plane pair) relative to wa / brow
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
        // MFMAs of column range [n0, n1): the next tap's reads are issued after the first column so that
        // their issue slots sit in the shadow of running MFMAs and the tap boundary holds nothing but a
        // wait that is already satisfied
        auto mfmas = [&](const frag_t* a, const frag_t* b, int n0, int n1) {
#pragma unroll
          for (int n = 0; n < NR; ++n)
            if (n >= n0 && n < n1)
#pragma unroll
              for (int m = 0; m < MR; ++m) mfma32_acc<T, OCC == 1>(acc[m][n], a[m], b[n]);
        };
        if (npp > 0) {
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n) mfma_input_fence<OCC == 1>(acc[m][n]);
          set_row(0, 0);
          issue(std::integral_constant<int, 0>{}, a0, b0);
          landed(a0, b0);
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
              if constexpr (KK & 1) {   // odd tap count: the prefetched fragments sit in the other buffer
#pragma unroll
                for (int m = 0; m < MR; ++m) a0[m] = a1[m];
#pragma unroll
                for (int n = 0; n < NR; ++n) b0[n] = b1[n];
              }
            }
          }
          asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA's result visible to the VALU
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n) mfma_result_fence<OCC == 1>(acc[m][n]);
        }
      }
      const unsigned long long t2 = now();
      // (4) this wave's share of the next stage has landed; residual slots too
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t3 = now();

      // one 16-byte output slot: bias + residual + ReLU + rounding of accumulator (m, n), row group g
      auto finalize = [&](int m, int n, int g, const float4 bs0, const float4 bs1, const u32x4 rv) -> u32x4 {
        uint32_t a[4], b[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          a[jj] = __float_as_uint(acc[m][n][8 * g + jj]);
          b[jj] = __float_as_uint(acc[m][n][8 * g + 4 + jj]);
          const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
          a[jj] = sw[0]; b[jj] = sw[1];
        }
        // lower half-wave: the 8 channels of plane 2g (a own, b from the partner);
        // upper half-wave: plane 2g+1 (a from the partner, b own)
        float v[8];
        v[0] = __uint_as_float(a[0]) + bs0.x; v[1] = __uint_as_float(a[1]) + bs0.y;
        v[2] = __uint_as_float(a[2]) + bs0.z; v[3] = __uint_as_float(a[3]) + bs0.w;
        v[4] = __uint_as_float(b[0]) + bs1.x; v[5] = __uint_as_float(b[1]) + bs1.y;
        v[6] = __uint_as_float(b[2]) + bs1.z; v[7] = __uint_as_float(b[3]) + bs1.w;
        v[0] += from_bits<T>(rv[0] & 0xffff); v[1] += from_bits<T>(rv[0] >> 16);
        v[2] += from_bits<T>(rv[1] & 0xffff); v[3] += from_bits<T>(rv[1] >> 16);
        v[4] += from_bits<T>(rv[2] & 0xffff); v[5] += from_bits<T>(rv[2] >> 16);
        v[6] += from_bits<T>(rv[3] & 0xffff); v[7] += from_bits<T>(rv[3] >> 16);
        const uint32_t relu_floor = p.relu ? 0u : 0x80008000u;   // packed ReLU: signed 16-bit max (conv_device.h)
        u32x4 ov;
        ov[0] = relu2_16(pack2<T>(v[0], v[1]), relu_floor); ov[1] = relu2_16(pack2<T>(v[2], v[3]), relu_floor);
        ov[2] = relu2_16(pack2<T>(v[4], v[5]), relu_floor); ov[3] = relu2_16(pack2<T>(v[6], v[7]), relu_floor);
        return ov;
      };
      if (last) {  // (5) retire the tile
        if constexpr (OCC == 1) {   // finalize into the prefetched slots (registers only); stores after the barrier
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
              for (int g = 0; g < 2; ++g) asm volatile("" : "+v"(slot[m][n][g]));
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const float* bp = bias_l + mb * MT + wm * (MR * 32) + m * 32 + (2 * g + half) * 8;
              const float4 bs0 = *reinterpret_cast<const float4*>(bp);
              const float4 bs1 = *reinterpret_cast<const float4*>(bp + 4);
#pragma unroll
              for (int n = 0; n < NR; ++n) slot[m][n][g] = finalize(m, n, g, bs0, bs1, slot[m][n][g]);
            }
        } else {   // two workgroups per CU: the other one owns the matrix pipe while this one waits for its residual rows
#pragma unroll
          for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int pl = plane0 + m * 4 + 2 * g + half;
                const bool ok = p.res && pbase[n] != ~(size_t)0 && pl < cout_planes;
                slot[0][n][g] = ok ? *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.res) + pbase[n] + (size_t)pl * HoWo * 16)
                                   : u32x4{0u, 0u, 0u, 0u};
              }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const float* bp = bias_l + mb * MT + wm * (MR * 32) + m * 32 + (2 * g + half) * 8;
              const float4 bs0 = *reinterpret_cast<const float4*>(bp);
              const float4 bs1 = *reinterpret_cast<const float4*>(bp + 4);
              const int pl = plane0 + m * 4 + 2 * g + half;
#pragma unroll
              for (int n = 0; n < NR; ++n) {
                const u32x4 ov = finalize(m, n, g, bs0, bs1, slot[0][n][g]);
                if (pbase[n] != ~(size_t)0 && pl < cout_planes)
                  *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + pbase[n] + (size_t)pl * HoWo * 16) = ov;
              }
            }
          }
        }
      }
      // (6) one barrier per stage
      const unsigned long long t4 = now();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned long long t5 = now();

      if constexpr (OCC == 1)
      if (last) {  // (7) stores complete under the following stages
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int pl = plane0 + m * 4 + 2 * g + half;
              if (pbase[n] != ~(size_t)0 && pl < cout_planes)
                *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + pbase[n] + (size_t)pl * HoWo * 16) = slot[m][n][g];
            }
      }
      if SCP_DBG(p, 8) {
        const unsigned long long t6 = now();
        if SCP_DBG(p, 256) {   // finer split of the front of the stage: [zero+residual][locate][X issue][W issue][MFMA][rest]
          tph[0] += ta - t0; tph[1] += tb - ta; tph[2] += tc - tb; tph[3] += t1 - tc; tph[4] += t2 - t1; tph[5] += t6 - t2;
        } else {
          tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3; tph[4] += t5 - t4; tph[5] += t6 - t5;
        }
      }
      xb ^= 1;
    }
  }
  if (SCP_DBG(p, 8) && SCP_DBG_BUF(p) && lane == 0)
    for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave) * 6 + k] = tph[k];
}

// ---- launch dispatch (instantiated per dtype in conv_m32_bf16.hip / conv_m32_f16.hip) ----
template <int DT, int MR, int NR, int WM, int OCC>
int32_t m32_launch_one(const ConvLaunch& L, size_t lds, hipStream_t st) {
  auto kern = conv_m32_kernel<DT, 3, 1, MR, NR, WM, OCC>;
  static LdsOptIn big_lds;   // per device (common.h)
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(256), lds, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

// (MR, WM, NR) variants built; conv_m32_supported() on the host mirrors this list
template <int DT>
int32_t m32_dispatch(int mr, int wm, int nr, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
#define SCP_M32_CASE(MR_, WM_, NR_, OCC_) if (mr == MR_ && wm == WM_ && nr == NR_ && occ == OCC_) return m32_launch_one<DT, MR_, NR_, WM_, OCC_>(L, lds, st)
  SCP_M32_CASE(3, 1, 3, 1);
  SCP_M32_CASE(3, 1, 2, 2);
  SCP_M32_CASE(2, 1, 2, 1);
  SCP_M32_CASE(2, 1, 4, 1);
  SCP_M32_CASE(2, 1, 3, 2);
#undef SCP_M32_CASE
  set_error("conv m32: variant mr=%d wm=%d nr=%d occ=%d not built", mr, wm, nr, occ);
  return SCPOSE_E_INVALID;
}

}  // namespace scpose
