// Software-pipelined, persistent implicit-GEMM convolution (same math and data layout as
// conv_igemm.hip; see that file for the GEMM orientation and K ordering).
//
// What changes is HOW the operands reach LDS:
//   * one workgroup per CU, resident for the whole launch, walks a contiguous range of
//     (tile, Cout-block) work items; the XCD remap keeps a range inside one XCD's L2;
//   * every K-chunk (cp input planes + their packed weights) is brought in with LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, 1 KiB per wave-instruction) into one of TWO
//     LDS buffers, so the loads of stage s+1 are in flight while the MFMAs of stage s run;
//     one barrier per stage;
//   * a layer whose whole K fits one chunk and whose Cout fits one block (the C<=48 high
//     resolution branch: 35 % of the forward) keeps its packed weights resident in LDS for
//     the whole launch and only streams activation tiles;
//   * zero padding comes from a zero page: an out-of-image halo pixel's DMA source address is
//     redirected to 16 zero bytes, lanes past the halo tile are masked off (EXEC).
#include <stdlib.h>

#include "common.h"
#include "conv_device.h"

namespace scpose {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: usable as an inline-asm register operand

__device__ __forceinline__ void dma16(const void* g, void* l_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l_wave_base, 16, 0, 0);
}

// s_waitcnt takes an immediate: a wave-uniform runtime count goes through a switch.
#define SCP_WAITVM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    SCP_WAITVM_CASE(0) SCP_WAITVM_CASE(1) SCP_WAITVM_CASE(2) SCP_WAITVM_CASE(3) SCP_WAITVM_CASE(4)
    SCP_WAITVM_CASE(5) SCP_WAITVM_CASE(6) SCP_WAITVM_CASE(7) SCP_WAITVM_CASE(8) SCP_WAITVM_CASE(9)
    SCP_WAITVM_CASE(10) SCP_WAITVM_CASE(11) SCP_WAITVM_CASE(12) SCP_WAITVM_CASE(13) SCP_WAITVM_CASE(14)
    SCP_WAITVM_CASE(15) SCP_WAITVM_CASE(16) SCP_WAITVM_CASE(17) SCP_WAITVM_CASE(18) SCP_WAITVM_CASE(19)
    SCP_WAITVM_CASE(20) SCP_WAITVM_CASE(21) SCP_WAITVM_CASE(22) SCP_WAITVM_CASE(23) SCP_WAITVM_CASE(24)
    SCP_WAITVM_CASE(25) SCP_WAITVM_CASE(26) SCP_WAITVM_CASE(27) SCP_WAITVM_CASE(28) SCP_WAITVM_CASE(29)
    SCP_WAITVM_CASE(30) SCP_WAITVM_CASE(31) SCP_WAITVM_CASE(32) SCP_WAITVM_CASE(33) SCP_WAITVM_CASE(34)
    SCP_WAITVM_CASE(35) SCP_WAITVM_CASE(36) SCP_WAITVM_CASE(37) SCP_WAITVM_CASE(38) SCP_WAITVM_CASE(39)
    SCP_WAITVM_CASE(40)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // conservative: drain everything
  }
}
#undef SCP_WAITVM_CASE

template <int DT, int KS, int STRIDE, int MREP, int NREP>
__global__ __launch_bounds__(256) void conv_pipe_kernel(const ConvLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int MT = 16 * MREP;
  constexpr int MAXP = (STRIDE == 1) ? 2 : 3;
  constexpr int KK = KS * KS;
  constexpr int NPAIR = (NREP + 1) / 2;

  // LDS: [k-offset tables 512 B][bias, packed row order][W buffers x nbuf_w][X buffers x nbuf_x]
  int* koff = reinterpret_cast<int*>(smem);
  float* bias_l = reinterpret_cast<float*>(smem + 512);
  char* wl0 = smem + 512 + p.lds_bias;
  char* xl0 = wl0 + p.nbuf_w * p.lds_w;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int HW = p.H * p.W;
  const int HP = p.halo_h * p.halo_w;
  const int npix = p.th * p.tw;
  const int planes_last = p.cin_planes - (p.nchunks - 1) * p.cp;
  const bool w_resident = p.nbuf_w == 1;     // whole K in one chunk and one Cout block: weights stay in LDS
  const int depth = p.nbuf_x - 1;            // stages in flight ahead of the one being computed

  if (tid < 128) {  // K-offset tables: LDS byte offset of k-group qq at k-step st (0 for padding)
    const int tbl = tid >> 6, e = tid & 63;
    const int planes = tbl ? planes_last : p.cp;
    const int npt = (planes >> 1) * KK;
    const int st = e >> 2, qq = e & 3;
    const int pt = 2 * st + (qq >> 1);
    const int pp = pt / KK, tap = pt - pp * KK;
    const int ky = tap / KS, kx = tap - ky * KS;
    koff[tid] = pt < npt ? (2 * pp + (qq & 1)) * p.plane_stride + (ky * p.halo_w + kx) * 16 : 0;
  }
  for (int i = tid; i < p.n_mblk * MT; i += 256) bias_l[i] = p.bias[i];

  // tile-independent geometry of this thread's halo pixels and this lane's output pixels
  int hy[MAXP], hx[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int hp = i * 256 + tid;
    hy[i] = hp < HP ? hp / p.halo_w : -1;
    hx[i] = hp < HP ? hp - hy[i] * p.halo_w : 0;
  }
  int pixoff[NREP], py[NREP], px[NREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n) {
    const int pidx = (wave * NREP + n) * 16 + r;
    if (pidx < npix) {
      py[n] = pidx / p.tw; px[n] = pidx - py[n] * p.tw;
      pixoff[n] = ((py[n] * STRIDE) * p.halo_w + px[n] * STRIDE) * 16;
    } else {
      py[n] = -1; px[n] = 0; pixoff[n] = 0;
    }
  }
  // Epilogue geometry.  Accumulator rows are channel-permuted at pack time (conv_row_channel) so
  // that lanes l and l+32 hold the low/high 4 channels of the SAME 8-channel plane.  Two pixel
  // tiles (n0, n1) retire together: one v_permlane32_swap per dword gives the lower half-wave all
  // 8 channels of its n0 pixel and the upper half-wave all 8 of its n1 pixel, i.e. one aligned
  // 16-byte slot of the blocked tensor per lane (8-byte half-slot stores measured ~4x slower).
  const int half = lane >> 5, psel = q & 1;
  int epy[NPAIR], epx[NPAIR];
#pragma unroll
  for (int np = 0; np < NPAIR; ++np) {
    const int n0 = 2 * np, n1 = (2 * np + 1 < NREP) ? 2 * np + 1 : 2 * np;
    const bool paired = 2 * np + 1 < NREP;
    epy[np] = half ? (paired ? py[n1] : -1) : py[n0];
    epx[np] = half ? px[n1] : px[n0];
  }
  const int cout_planes = (p.cout + 7) >> 3;
  const size_t HoWo = (size_t)p.Ho * p.Wo;

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int it_begin = wg * p.items_per_wg;
  const int it_end = min(p.items_total, it_begin + p.items_per_wg);
  const size_t chunk_wbytes = (size_t)p.ksteps_full * (4 * MT * 16);

  auto decode_item = [&](int it, int& mb, int& img, int& oy0, int& ox0) {
    mb = it % p.n_mblk;
    int t = it / p.n_mblk;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    img = t / p.tiles_y;
    oy0 = ty * p.th; ox0 = tx * p.tw;
  };

  // Issue the LDS-DMA of stage (it, c) into W buffer wb / X buffer xb.  Returns the number of
  // DMA instructions THIS WAVE issued (wave-uniform), which the counted vmcnt waits need.
  auto issue = [&](int it, int c, int wb, int xb, bool with_weights) -> int {
    int cnt = 0;
    int mb, img, oy0, ox0;
    decode_item(it, mb, img, oy0, ox0);
    const int planes = c == p.nchunks - 1 ? planes_last : p.cp;
    if (with_weights) {
      const int ksteps = (((planes >> 1) * KK) + 1) >> 1;
      const int nbytes = ksteps * (4 * MT * 16);
      const char* ws = static_cast<const char*>(p.wpk) + ((size_t)mb * p.nchunks + c) * chunk_wbytes;
      char* wl = wl0 + wb * p.lds_w;
      for (int o = 0; o < nbytes; o += 4096) {
        const int mine = o + tid * 16;
        if (o + wave * 1024 < nbytes) ++cnt;          // any lane of this wave active
        if (mine < nbytes) dma16(ws + mine, wl + o + wave * 1024);
      }
    }
    if (!(p.dbg & 4)) {
      const int iy0 = oy0 * STRIDE - (KS / 2), ix0 = ox0 * STRIDE - (KS / 2);
      const char* inb = static_cast<const char*>(p.in) + ((size_t)img * p.cin_planes + (size_t)c * p.cp) * HW * 16;
      char* xl = xl0 + xb * p.lds_x;
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (i * 256 + wave * 64 < HP) cnt += planes;   // lane 0 of the wave is inside the halo tile
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
    }
    return cnt;
  };

  // bias of the (single) Cout block stays in registers for the whole launch; with several Cout
  // blocks it is re-read from LDS per item (those layers drain the DMA at that point anyway)
  float4 bsv[MREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m) bsv[m] = *reinterpret_cast<const float4*>(p.bias + m * 16 + q * 4);

  f32x4 acc[MREP][NREP];
  u32x4 slot[MREP][NPAIR];    // the lane's 16-byte slots of the item being retired: residual in, result out
  uint32_t ooff[MREP][NPAIR]; // byte offset of the slot inside the image (out and res share it), ~0 = masked

  int ld_it = it_begin, ld_c = 0, ld_s = 0;   // next stage to load
  bool w_loaded = false;
  auto issue_next = [&]() -> int {
    int cnt = 0;
    if (ld_it < it_end) {
      cnt = issue(ld_it, ld_c, w_resident ? 0 : (ld_s & 1), ld_s % p.nbuf_x, !(w_resident && w_loaded));
      w_loaded = true;
      ++ld_s;
      if (++ld_c == p.nchunks) { ld_c = 0; ++ld_it; }
    }
    return cnt;
  };
  for (int k = 0; k < depth; ++k) issue_next();
  __syncthreads();   // k-offset tables, bias, and the first `depth` stages are in LDS

  int s_idx = 0;
  for (int it = it_begin; it < it_end; ++it) {
    int mb, img, oy0, ox0;
    decode_item(it, mb, img, oy0, ox0);
    const size_t img_off = (size_t)img * cout_planes * HoWo * 16;
    for (int c = 0; c < p.nchunks; ++c, ++s_idx) {
      if (c == 0) {
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int n = 0; n < NREP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const bool last = c == p.nchunks - 1;
      const bool retire = last && !p.out_nchw_f32;

      // (1) residual slots of this item: issued FIRST so that they are older than the DMA below
      //     (inline asm: hipcc must not count or wait for them -- it would drain the DMA too)
      if (retire) {
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int co_plane = mb * MT + m * 16 + psel * 8;
#pragma unroll
          for (int np = 0; np < NPAIR; ++np) {
            const int oy = oy0 + epy[np], ox = ox0 + epx[np];
            const bool ok = epy[np] >= 0 && oy < p.Ho && ox < p.Wo && co_plane < p.cout && !(p.dbg & 2);
            ooff[m][np] = ok ? (uint32_t)((((size_t)(co_plane >> 3)) * HoWo + (size_t)oy * p.Wo + ox) * 16) : 0xffffffffu;
            slot[m][np] = u32x4{0u, 0u, 0u, 0u};
          }
        }
        if (p.res) {
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int np = 0; np < NPAIR; ++np) {
              const char* rp = ooff[m][np] != 0xffffffffu ? static_cast<const char*>(p.res) + img_off + ooff[m][np]
                                                          : static_cast<const char*>(p.zero16);
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(slot[m][np]) : "v"(rp) : "memory");
            }
        }
      }

      // (2) DMA of the stage `depth` ahead into the buffer released at the previous barrier
      const int newest = __builtin_amdgcn_readfirstlane(issue_next());   // wave-uniform -> scalar branch

      {  // (3) MFMA loop over the chunk's k-steps: fragments one step ahead, k-offsets two
        const int planes = last ? planes_last : p.cp;
        const int npt = (planes >> 1) * KK;
        const int ksteps = (p.dbg & 1) ? 0 : (npt + 1) >> 1;
        const int klast = ksteps - 1;
        const char* xl = xl0 + (s_idx % p.nbuf_x) * p.lds_x;
        const char* wq = wl0 + (w_resident ? 0 : (s_idx & 1)) * p.lds_w + (q * MT + r) * 16;
        const int* kt = koff + (last ? 64 : 0) + q;
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
          int ko_a = kt[0], ko_b = kt[4];              // steps 0 and 1 (the table is zero padded)
          load_frags(0, ko_a, a0, b0);
          int st = 0;
          for (; st + 1 < ksteps; st += 2) {            // branch-free body: indices clamp instead
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

      // (4) Everything older than the newest stage's DMA must be complete: the residual loads,
      //     the previous item's stores, and the DMA of the stage computed next.  With a single
      //     stage of lookahead (double buffer) that newest stage IS the next one: drain.
      wait_vm(depth >= 2 ? newest : 0);

      if (retire) {  // (5) finalize the item into 16-byte slots (registers only)
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int np = 0; np < NPAIR; ++np) asm volatile("" : "+v"(slot[m][np]));   // loads above have landed
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          if (p.n_mblk > 1) bsv[m] = *reinterpret_cast<const float4*>(bias_l + mb * MT + m * 16 + q * 4);
          const float4 bs = bsv[m];
#pragma unroll
          for (int np = 0; np < NPAIR; ++np) {
            const int n0 = 2 * np, n1 = (2 * np + 1 < NREP) ? 2 * np + 1 : 2 * np;
            uint32_t a[4], b[4];
            a[0] = __float_as_uint(acc[m][n0][0] + bs.x); a[1] = __float_as_uint(acc[m][n0][1] + bs.y);
            a[2] = __float_as_uint(acc[m][n0][2] + bs.z); a[3] = __float_as_uint(acc[m][n0][3] + bs.w);
            b[0] = __float_as_uint(acc[m][n1][0] + bs.x); b[1] = __float_as_uint(acc[m][n1][1] + bs.y);
            b[2] = __float_as_uint(acc[m][n1][2] + bs.z); b[3] = __float_as_uint(acc[m][n1][3] + bs.w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const auto sw = __builtin_amdgcn_permlane32_swap(a[j], b[j], false, false);
              a[j] = sw[0]; b[j] = sw[1];
            }
            // lower half-wave: a = own (n0, ch 0-3), b = partner's (n0, ch 4-7)
            // upper half-wave: a = partner's (n1, ch 0-3), b = own (n1, ch 4-7)
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = __uint_as_float(a[j]); v[4 + j] = __uint_as_float(b[j]); }
            const u32x4 rv = slot[m][np];
            v[0] += from_bits<T>(rv[0] & 0xffff); v[1] += from_bits<T>(rv[0] >> 16);
            v[2] += from_bits<T>(rv[1] & 0xffff); v[3] += from_bits<T>(rv[1] >> 16);
            v[4] += from_bits<T>(rv[2] & 0xffff); v[5] += from_bits<T>(rv[2] >> 16);
            v[6] += from_bits<T>(rv[3] & 0xffff); v[7] += from_bits<T>(rv[3] >> 16);
            if (p.relu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            u32x4 ov;
            ov[0] = (uint32_t)to_bits<T>(v[0]) | ((uint32_t)to_bits<T>(v[1]) << 16);
            ov[1] = (uint32_t)to_bits<T>(v[2]) | ((uint32_t)to_bits<T>(v[3]) << 16);
            ov[2] = (uint32_t)to_bits<T>(v[4]) | ((uint32_t)to_bits<T>(v[5]) << 16);
            ov[3] = (uint32_t)to_bits<T>(v[6]) | ((uint32_t)to_bits<T>(v[7]) << 16);
            slot[m][np] = ov;
          }
        }
      }

      // (6) One barrier per stage: every wave has drained its share of the next stage's DMA and
      //     finished reading this stage's buffers, which the next issue may overwrite.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();

      if (last) {  // (7) stores fly under the following stages; nothing waits for them explicitly
        if (retire) {
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int np = 0; np < NPAIR; ++np)
              if (ooff[m][np] != 0xffffffffu)
                *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + img_off + ooff[m][np]) = slot[m][np];
        } else if (!(p.dbg & 2)) {   // final layer: few channels, float32 NCHW, 4-byte stores
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const int co = mb * MT + m * 16 + psel * 8 + half * 4;
            const float4 bs = *reinterpret_cast<const float4*>(bias_l + mb * MT + m * 16 + q * 4);
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
              if (py[n] < 0 || co >= p.cout) continue;
              const int oy = oy0 + py[n], ox = ox0 + px[n];
              if (oy >= p.Ho || ox >= p.Wo) continue;
              const float v[4] = {acc[m][n][0] + bs.x, acc[m][n][1] + bs.y, acc[m][n][2] + bs.z, acc[m][n][3] + bs.w};
              float* o = static_cast<float*>(p.out) + ((size_t)img * p.cout + co) * HoWo + (size_t)oy * p.Wo + ox;
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (co + j < p.cout) o[j * HoWo] = p.relu ? fmaxf(v[j], 0.f) : v[j];
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
static void* g_zero_page[16] = {nullptr};

const void* conv_zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_zero_page[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    g_zero_page[dev] = p;
  }
  return g_zero_page[dev];
}

static int device_cus() {
  static int cus[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

template <int DT, int KS, int STRIDE, int MREP>
static int32_t pipe_nrep(int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
  dim3 grid(L.grid), block(256);
#define SCP_LAUNCH(NR)                                                                          \
  case NR: {                                                                                    \
    auto kern = conv_pipe_kernel<DT, KS, STRIDE, MREP, NR>;                                      \
    static bool big_lds_enabled = false;                                                        \
    if (!big_lds_enabled) {                                                                     \
      SCP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                        160 * 1024));                                           \
      big_lds_enabled = true;                                                                   \
    }                                                                                           \
    hipLaunchKernelGGL(kern, grid, block, lds, st, L);                                          \
    break;                                                                                      \
  }
  switch (nrep) {
    SCP_LAUNCH(1) SCP_LAUNCH(2) SCP_LAUNCH(3) SCP_LAUNCH(4)
    default: set_error("conv: nrep %d unsupported", nrep); return SCPOSE_E_INVALID;
  }
#undef SCP_LAUNCH
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

template <int DT, int KS, int STRIDE>
static int32_t pipe_mrep(int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
  switch (mrep) {
    case 1: return pipe_nrep<DT, KS, STRIDE, 1>(nrep, L, lds, st);
    case 2: return pipe_nrep<DT, KS, STRIDE, 2>(nrep, L, lds, st);
    case 3: return pipe_nrep<DT, KS, STRIDE, 3>(nrep, L, lds, st);
    case 4: return pipe_nrep<DT, KS, STRIDE, 4>(nrep, L, lds, st);
    case 6: return pipe_nrep<DT, KS, STRIDE, 6>(nrep, L, lds, st);
  }
  set_error("conv: mrep %d unsupported", mrep);
  return SCPOSE_E_INVALID;
}

template <int DT>
static int32_t pipe_ks(int ks, int stride, int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
  if (ks == 3 && stride == 1) return pipe_mrep<DT, 3, 1>(mrep, nrep, L, lds, st);
  if (ks == 3 && stride == 2) return pipe_mrep<DT, 3, 2>(mrep, nrep, L, lds, st);
  if (ks == 1 && stride == 1) return pipe_mrep<DT, 1, 1>(mrep, nrep, L, lds, st);
  set_error("conv: k=%d stride=%d unsupported", ks, stride);
  return SCPOSE_E_INVALID;
}

int32_t conv_launch_pipe(const PackedConv& pc, ConvLaunch& L, int nrep, hipStream_t stream, bool* fits) {
  L.lds_w = pc.ksteps_full * 4 * pc.mt * 16;
  L.lds_x = pc.cp * L.plane_stride;
  L.lds_bias = ((pc.n_mblk * pc.mt * 4) + 511) & ~511;
  const bool resident = pc.nchunks == 1 && pc.n_mblk == 1;
  L.nbuf_w = resident ? 1 : 2;
  L.nbuf_x = 2;
  const size_t fixed = 512 + (size_t)L.lds_bias + (size_t)L.nbuf_w * L.lds_w;
  if (resident && fixed + 3 * (size_t)L.lds_x <= 160 * 1024) L.nbuf_x = 3;   // two stages of DMA in flight
  const size_t lds = fixed + (size_t)L.nbuf_x * L.lds_x;
  *fits = lds <= 160 * 1024;
  if (!*fits) return SCPOSE_E_INVALID;
  L.zero16 = conv_zero_page();
  SCP_REQUIRE(L.zero16, "conv: cannot allocate the zero page");
  L.items_total = L.total_blocks;
  { static const char* e = getenv("SCPOSE_DBG"); L.dbg = e ? atoi(e) : 0; }
  const int cus = device_cus();
  const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  int grid = cus * (per_cu > 2 ? 2 : per_cu);
  if (grid > L.items_total) grid = L.items_total;
  L.items_per_wg = (L.items_total + grid - 1) / grid;
  L.grid = (L.items_total + L.items_per_wg - 1) / L.items_per_wg;
  if (pc.dtype == SCPOSE_DT_BF16) return pipe_ks<0>(pc.ks, pc.stride, pc.mrep, nrep, L, lds, stream);
  return pipe_ks<1>(pc.ks, pc.stride, pc.mrep, nrep, L, lds, stream);
}

}  // namespace scpose
