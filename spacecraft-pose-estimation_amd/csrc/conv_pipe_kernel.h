// Software-pipelined, persistent implicit-GEMM convolution kernel for gfx950 (MFMA 16x16x32).
//
// Replaces every nn.Conv2d + eval BatchNorm2d (+ReLU, +residual add) of the reference
// landmark_regression/lib/models/pose_hrnet.py (conv3x3 :22-25, BasicBlock :41-57,
// Bottleneck :78-98, transition :343-368, fuse down path :216-237, fuse 1x1 :199-205,
// final_layer :323-329) except the 3-channel stem conv (stem.hip).
//
// GEMM view:  D[cout][pixel] = sum_k Wt[cout][k] * X[k][pixel],  k = (input plane, tap, 8 ch).
//   * A operand = weights (M = Cout), B operand = activations (N = pixels): an accumulator lane
//     holds 4 consecutive rows (channels) of ONE pixel, and a B fragment is exactly one 16-byte
//     (pixel, 8-channel) vector of the blocked [N][C/8][H][W][8] input -- no transposes.
//   * K order inside a chunk is (plane pair, tap); MFMA k-group q (lane>>4) reads plane
//     2*pp + (q&1) at tap pt = 2*s + (q>>1).  The two k-groups that share a ds_read_b128 lane
//     group differ by one whole LDS plane, whose stride is padded to keep them on disjoint banks.
//   * Cout rows inside each 16-row MFMA tile are permuted at pack time so that lanes l and l+32
//     hold the two halves of one 8-channel plane: the epilogue pairs two pixel tiles, swaps
//     halves with v_permlane32_swap and stores aligned 16-byte slots (8-byte half-slot stores
//     measured ~4x slower).
//
// Data movement (measured on MI355X: one CU fills LDS by LDS-DMA at ~16 B/clk from L2 and the
// HBM share of a CU is ~10 B/clk, so the design minimises bytes staged per MFMA):
//   * one or two resident workgroups per CU walk a contiguous range of work items; the XCD remap
//     keeps a range inside one XCD's L2;
//   * operands reach LDS by global_load_lds_dwordx4 (no staging registers), double buffered: the
//     DMA of stage s+1 flies while the MFMAs of stage s run; one barrier per stage;
//   * a work item is a GROUP of NT pixel tiles x one Cout block.  For each K-chunk the packed
//     weights are staged ONCE and used for all NT tiles (NT accumulator sets in registers), which
//     divides the dominant weight traffic of the wide low-resolution branches by NT; the next
//     chunk's weights are streamed in NT slices, one per sub-stage, to keep every stage's DMA
//     about the same size;
//   * a layer whose whole K fits one chunk and whose Cout fits one block (the high-resolution
//     branch) keeps its weights resident in LDS for the whole launch;
//   * measured: the LDS-DMA path of a CU accepts ~16 B/clk and a wave that issues into a full queue
//     stalls, MFMAs included (feeding the pieces from inside the MFMA loop made the loop 2.8x
//     slower), so a stage's pieces go out in one burst before the MFMA loop; the overlap comes from
//     the second resident workgroup (weight-resident layers) or from the staggered two-group
//     schedule of conv_stag_kernel.h (weight-streaming layers);
//   * zero padding comes from a zero page (an out-of-image halo pixel's DMA source is redirected
//     to 16 zero bytes); lanes past the halo tile are masked off (EXEC);
//   * residual slots are prefetched (inline-asm loads, issued before the stage's DMA) and the
//     results are finalised in registers before the stage barrier; the stores are issued after it
//     and complete under the following stages.
#pragma once
#include "common.h"
#include "conv_device.h"

namespace scpose {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: usable as an inline-asm operand

// exact n / d for a launch-time constant d (common.h: FastDiv)
__device__ __forceinline__ int pipe_fdiv(int n, const FastDiv& f) {
  return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n * f.add) >> f.shift);
}

__device__ __forceinline__ void dma16(const void* g, void* l_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l_wave_base, 16, 0, 0);
}

// Buffer-addressed variants (raw buffer, stride 0, num_records = tensor bytes < 4 GiB): the address is
// base + 32-bit VGPR offset + SGPR offset, so a per-chunk / per-plane displacement costs no vector arithmetic, and a
// lane whose offset lies past num_records reads zeros (loads, LDS-DMA included) or is dropped (stores): the padding
// pixels of a halo need neither a zero page nor a 64-bit select.  BUF_OOB is the offset given to such lanes.
// (tools_dev/micro/bufdma_bench.hip: same LDS image, 15-23 % fewer producer cycles per instruction.)
constexpr uint32_t BUF_OOB = 0xfffffff0u;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_buf(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16_buf(buf_rsrc_t r, uint32_t voff, uint32_t soff, void* l_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)l_wave_base, 16, voff, soff, 0, 0);
}
// The s_nop behind the store: a 16-byte buffer store reads its four data VGPRs over a few cycles after issue, and hipcc (ROCm 7.2,
// gfx950) puts NO wait state between such a store and a VALU instruction that overwrites one of them.  Round 4 hit it in
// conv_block2_kernel: `buffer_store_dwordx4 v[82:85], ..., s42 offen` directly followed by `v_mov_b32 v82, v0` (the next pass's
// thread id) stored the thread id in the first dword of the vector for lanes 12-15 of every 16 -- only in a workgroup's first tile,
// only in the product build (tools_dev/where_block_differs.py; the guide prescribes the same pad for asm stores: cdna_hip_programming.md
// 5.7).  Two wait states here cost nothing next to the store's issue and protect every kernel that stores through this helper.
__device__ __forceinline__ void store16_buf(buf_rsrc_t r, uint32_t voff, uint32_t soff, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
  asm volatile("s_nop 1" ::"v"(v) : "memory");   // the data registers are an operand: they stay live (unwritten) up to and including the pad
}
__device__ __forceinline__ u32x4 load16_buf(buf_rsrc_t r, uint32_t voff, uint32_t soff) {   // out-of-range lanes read zeros
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ void store4_buf(buf_rsrc_t r, uint32_t voff, uint32_t v) {   // out-of-range lanes are dropped
  __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, 0u, 0);
}
#else   // host pass: the resource type does not exist there; these are never called
struct buf_rsrc_t {};
__device__ inline buf_rsrc_t make_buf(const void*, uint32_t) { return buf_rsrc_t{}; }
__device__ inline void dma16_buf(buf_rsrc_t, uint32_t, uint32_t, void*) {}
__device__ inline void store16_buf(buf_rsrc_t, uint32_t, uint32_t, u32x4) {}
__device__ inline u32x4 load16_buf(buf_rsrc_t, uint32_t, uint32_t) { return u32x4{0u, 0u, 0u, 0u}; }
__device__ inline void store4_buf(buf_rsrc_t, uint32_t, uint32_t) {}
#endif

// OCC = resident workgroups per CU the variant is built for: 2 caps the wave at 256 registers
// (two waves per SIMD, which also hide LDS latency, so the explicit fragment prefetch is dropped).
// G = wave groups per workgroup (1 or 2; 256 threads each).  With G = 2 the two groups work on
// two DIFFERENT pixel tiles in parallel from ONE staged copy of the weight chunk: the weight
// bytes per MFMA halve like NT = 2, but with two waves per SIMD one group's DMA-issue stalls
// (the LDS-DMA path accepts ~16 B/clk per CU and the issuing wave waits for it) overlap the
// other group's MFMAs.
template <int DT, int KS, int STRIDE, int MREP, int NREP, int NT, int OCC, int G>
__global__ __launch_bounds__(256 * G, (G == 2) ? 2 : OCC) void conv_pipe_kernel(const ConvLaunch p) {
  static_assert(G == 1 || NT == 1, "parallel tile groups and sequential tile groups are exclusive");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int MT = 16 * MREP;
  constexpr int MAXP = (STRIDE == 1) ? 2 : 3;
  constexpr int KK = KS * KS;
  constexpr int NPAIR = (NREP + 1) / 2;

  // LDS: [k-offset tables 512 B][bias, packed row order][W buffers x nbuf_w][X buffers x 2]
  int* koff = reinterpret_cast<int*>(smem);
  float* bias_l = reinterpret_cast<float*>(smem + 512);
  char* wl0 = smem + 512 + p.lds_bias;
  char* xl0 = wl0 + p.nbuf_w * p.lds_w;

  const int lane = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 .. 4G-1 (scalar): weight-chunk DMA is spread over all waves
  const int grp = wave_all >> 2;                     // this wave's tile group
  const int wave = wave_all & 3, tid = threadIdx.x & 255;   // position inside the group
  const int q = lane >> 4, r = lane & 15;
  const int HW = p.H * p.W;
  const int HP = p.halo_h * p.halo_w;
  const int npix = p.th * p.tw;
  const int planes_last = p.cin_planes - (p.nchunks - 1) * p.cp;
  const bool w_resident = p.nbuf_w == 1;     // whole K in one chunk and one Cout block: weights stay in LDS

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
  for (int i = threadIdx.x; i < p.n_mblk * MT; i += 256 * G) bias_l[i] = p.bias[i];

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
  const int half = lane >> 5, psel = q & 1;
  int epy[NPAIR], epx[NPAIR];   // the pixel whose 16-byte slot this lane stores for tile pair np
#pragma unroll
  for (int np = 0; np < NPAIR; ++np) {
    const int n0 = 2 * np, n1 = (2 * np + 1 < NREP) ? 2 * np + 1 : 2 * np;
    const bool paired = 2 * np + 1 < NREP;
    epy[np] = half ? (paired ? py[n1] : -1) : py[n0];
    epx[np] = half ? px[n1] : px[n0];
  }
  const int cout_planes = (p.cout + 7) >> 3;
  const size_t HoWo = (size_t)p.Ho * p.Wo;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  const int wg = xcd_remap(blockIdx.x, p.grid);
  const int it_begin = wg * p.items_per_wg;
  const int it_end = min(p.items_total, it_begin + p.items_per_wg);
  const size_t chunk_wbytes = (size_t)p.ksteps_full * (4 * MT * 16);

  // item -> (Cout block, first tile of the group); tile -> (image, tile origin); img < 0 = no tile
  auto item_mb = [&](int it) { return it - pipe_fdiv(it, p.fd_nmblk) * p.n_mblk; };
  auto decode_tile = [&](int it, int j, int& img, int& oy0, int& ox0) {
    const int t = pipe_fdiv(it, p.fd_nmblk) * (NT * G) + j + grp * NT;
    if (t >= p.tiles_total) { img = -1; oy0 = ox0 = 0; return; }
    img = pipe_fdiv(t, p.fd_tiles_img);
    const int rem = t - img * tiles_per_img;
    const int ty = pipe_fdiv(rem, p.fd_tiles_x);
    oy0 = ty * p.th; ox0 = (rem - ty * p.tiles_x) * p.tw;
  };

  // ---- LDS-DMA issue ----
  auto issue_x = [&](int it, int j, int c, int xb) {       // input planes of chunk c for tile j of item it
    int img, oy0, ox0;
    decode_tile(it, j, img, oy0, ox0);
    if (img < 0 || SCP_DBG(p, 4)) return;
    const int planes = c == p.nchunks - 1 ? planes_last : p.cp;
    const int iy0 = oy0 * STRIDE - (KS / 2), ix0 = ox0 * STRIDE - (KS / 2);
    // K-concatenated 1x1 layers read the chunks past split_planes from a second tensor (its own plane count)
    const bool second = p.in2 && c * p.cp >= p.split_planes;
    const int own_planes = p.in2 ? (second ? p.cin_planes - p.split_planes : p.split_planes) : p.cin_planes;
    const char* inb = static_cast<const char*>(second ? p.in2 : p.in) +
                      ((size_t)img * own_planes + (size_t)(c * p.cp - (second ? p.split_planes : 0))) * HW * 16;
    char* xl = xl0 + (grp * 2 + xb) * p.lds_x;
    if (p.in_bytes) {   // buffer addressing (tensors < 4 GiB): image / chunk / plane displacement in an SGPR, padding out of range
      const buf_rsrc_t rs = second ? make_buf(p.in2, p.out_bytes) : make_buf(p.in, p.in_bytes);   // out_bytes: size of in2 here
      const uint32_t sbase = (uint32_t)((img * own_planes + (c * p.cp - (second ? p.split_planes : 0))) * HW) * 16u;
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (hy[i] >= 0) {
          const int iy = iy0 + hy[i], ix = ix0 + hx[i];
          const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
          const uint32_t voff = ok ? (uint32_t)(iy * p.W + ix) * 16u : BUF_OOB;
          for (int pl = 0; pl < planes; ++pl)
            dma16_buf(rs, voff, sbase + (uint32_t)(pl * HW) * 16u, xl + pl * p.plane_stride + (i * 256 + wave * 64) * 16);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
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
  };
  auto issue_w = [&](int it, int c, int wb, int part, int nparts) {   // slice `part` of nparts of a weight chunk
    const int planes = c == p.nchunks - 1 ? planes_last : p.cp;
    const int ksteps = (((planes >> 1) * KK) + 1) >> 1;
    const int nbytes = ksteps * (4 * MT * 16);
    const int slice = (((nbytes + nparts - 1) / nparts) + 4096 * G - 1) / (4096 * G) * (4096 * G);   // whole rounds of the workgroup
    const int lo = part * slice, hi = min(nbytes, lo + slice);
    const buf_rsrc_t rs_w = make_buf(p.wpk, (uint32_t)(p.n_mblk * p.nchunks * (int)chunk_wbytes));   // packed weights: far below 4 GiB
    const uint32_t wchunk = (uint32_t)((item_mb(it) * p.nchunks + c) * (int)chunk_wbytes);
    char* wl = wl0 + wb * p.lds_w;
    for (int o = lo; o < hi; o += 4096 * G) {
      const int mine = o + (int)threadIdx.x * 16;
      if (mine < hi) dma16_buf(rs_w, threadIdx.x * 16u, wchunk + (uint32_t)o, wl + o + wave_all * 1024);
    }
  };

  // weight-resident single-chunk 3x3 layers (the HBM-bound high-resolution branch): the k-step offsets are
  // tile-invariant, so they live in registers and the k-loop below is fully unrolled, hand-scheduled asm
  constexpr bool ASM_LOOP = KS == 3 && STRIDE == 1 && NT == 1 && G == 1 && MREP * NREP <= 8 && MT <= 64;
  constexpr int ASM_STEPS = 14;
  int koffv[ASM_LOOP ? ASM_STEPS : 1];
  float4 bsv[MREP];   // bias of the (single) Cout block stays in registers; several blocks: re-read per item
#pragma unroll
  for (int m = 0; m < MREP; ++m) bsv[m] = *reinterpret_cast<const float4*>(p.bias + m * 16 + q * 4);

  f32x4 acc[NT][MREP][NREP];
  u32x4 slot[MREP][NPAIR];    // 16-byte slots of the tile being retired: residual in, result out
  uint32_t ooff[MREP][NPAIR]; // byte offset of the slot inside the image (out and res share it), ~0 = masked

  // prologue: first weight chunk (whole) and the first input tile
  int wc = 0;                 // running chunk counter of this workgroup: W buffer = wc & 1
  if (it_begin < it_end) {
    issue_w(it_begin, 0, 0, 0, 1);
    issue_x(it_begin, 0, 0, 0);
  }
  __syncthreads();            // tables, bias, stage 0 (vmcnt(0) + barrier)
  if constexpr (ASM_LOOP) {
#pragma unroll
    for (int s2 = 0; s2 < ASM_STEPS; ++s2) koffv[s2] = koff[s2 * 4 + q];   // single chunk: table 0 (zero padded to 16 steps)
  }

  int xb = 0;                 // X buffer of the stage being computed
  // development instrumentation (dbg & 8): cycles spent per phase, summed over the launch
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { return SCP_DBG(p, 8) ? __builtin_amdgcn_s_memtime() : 0ull; };
  for (int it = it_begin; it < it_end; ++it) {
    const int mb = item_mb(it);
    for (int c = 0; c < p.nchunks; ++c, ++wc) {
      const bool last = c == p.nchunks - 1;
      // the chunk after this one (same item, or chunk 0 of the next item)
      const int nit = last ? it + 1 : it, nc = last ? 0 : c + 1;
      const bool have_next_chunk = nit < it_end;

      auto stage = [&](auto jtag) {
        constexpr int J = decltype(jtag)::value;
        const unsigned long long t0 = now();
        int img, oy0, ox0;
        decode_tile(it, J, img, oy0, ox0);
        const size_t img_off = (size_t)(img < 0 ? 0 : img) * cout_planes * HoWo * 16;
        const bool retire = last && !p.out_nchw_f32;
        if (c == 0) {
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[J][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // (1) residual slots of this tile: inline-asm loads (hipcc must neither count nor wait for
        //     them), issued before the DMA and consumed after the stage's wait
        if (retire) {
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            const int co_plane = mb * MT + m * 16 + psel * 8;
#pragma unroll
            for (int np = 0; np < NPAIR; ++np) {
              const int oy = oy0 + epy[np], ox = ox0 + epx[np];
              const bool ok = img >= 0 && epy[np] >= 0 && oy < p.Ho && ox < p.Wo && co_plane < p.cout && !SCP_DBG(p, 2);
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
        // (2) DMA for the next stage: its input tile, plus this sub-stage's slice of the next weight chunk
        // single-chunk layers with several Cout blocks (1x1 64->256): consecutive items are the Cout
        // blocks of ONE pixel tile, whose staged input is reused instead of being fetched again
        const bool same_x = NT == 1 && G == 1 && p.nchunks == 1 && have_next_chunk &&
                            pipe_fdiv(nit, p.fd_nmblk) == pipe_fdiv(it, p.fd_nmblk);
        if (J + 1 < NT) issue_x(it, J + 1, c, xb ^ 1);
        else if (have_next_chunk && !same_x) issue_x(nit, 0, nc, xb ^ 1);
        if (have_next_chunk && !w_resident) issue_w(nit, nc, (wc + 1) & 1, J, NT);

        const unsigned long long t1 = now();
        if (img >= 0) {  // (3) MFMA loop over the chunk's k-steps: fragments one step ahead, k-offsets two
          const int planes = last ? planes_last : p.cp;
          const int npt = (planes >> 1) * KK;
          const int ksteps = SCP_DBG(p, 1) ? 0 : (npt + 1) >> 1;
          const int klast = ksteps - 1;
          const char* xl = xl0 + (grp * 2 + xb) * p.lds_x;
          const char* wq = wl0 + (w_resident ? 0 : (wc & 1)) * p.lds_w + (q * MT + r) * 16;
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
              for (int n = 0; n < NREP; ++n) acc[J][m][n] = mfma16<T>(a[m], b[n], acc[J][m][n]);
          };
          bool done_asm = false;
          if constexpr (ASM_LOOP) {
            if (w_resident && p.nchunks == 1 && ksteps <= ASM_STEPS && ksteps > 0) {
              done_asm = true;
              const uint32_t wa = (uint32_t)(size_t)wq;
              const uint32_t xa = (uint32_t)(size_t)xl;
              auto issue = [&](auto sc, frag_t* a, frag_t* b) {
                constexpr int S = decltype(sc)::value;
                lds_read16<S * (4 * MT * 16)>(a[0], wa);
                if constexpr (MREP > 1) lds_read16<S * (4 * MT * 16) + 256>(a[1], wa);
                if constexpr (MREP > 2) lds_read16<S * (4 * MT * 16) + 512>(a[2], wa);
                if constexpr (MREP > 3) lds_read16<S * (4 * MT * 16) + 768>(a[3], wa);
#pragma unroll
                for (int n = 0; n < NREP; ++n) lds_read16<0>(b[n], xa + koffv[S] + pixoff[n]);
              };
              auto landed = [&](frag_t* a, frag_t* b) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int m = 0; m < MREP; ++m) lds_landed(a[m]);
#pragma unroll
                for (int n = 0; n < NREP; ++n) lds_landed(b[n]);
              };
              auto mfma_cols = [&](const frag_t* a, const frag_t* b, int n0, int n1) {
#pragma unroll
                for (int n = 0; n < NREP; ++n)
                  if (n >= n0 && n < n1)
#pragma unroll
                    for (int m = 0; m < MREP; ++m) mfma16_acc<T>(acc[J][m][n], a[m], b[n]);
              };
#pragma unroll
              for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int n = 0; n < NREP; ++n) mfma_input_fence<false>(acc[J][m][n]);
              issue(std::integral_constant<int, 0>{}, a0, b0);
              landed(a0, b0);
              auto step = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                if (S < ksteps) {
                  frag_t* ca = (S & 1) ? a1 : a0; frag_t* cb = (S & 1) ? b1 : b0;
                  frag_t* na = (S & 1) ? a0 : a1; frag_t* nb = (S & 1) ? b0 : b1;
                  mfma_cols(ca, cb, 0, 1);
                  if constexpr (S + 1 < ASM_STEPS) {
                    if (S + 1 < ksteps) issue(std::integral_constant<int, S + 1>{}, na, nb);
                  }
                  mfma_cols(ca, cb, 1, NREP);
                  if constexpr (S + 1 < ASM_STEPS) {
                    if (S + 1 < ksteps) landed(na, nb);
                  }
                }
              };
              step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
              step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
              step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
              step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
              step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{});
              step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
              step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
              asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last MFMA's result visible to the VALU
#pragma unroll
              for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int n = 0; n < NREP; ++n) mfma_result_fence<false>(acc[J][m][n]);
            }
          }
          if (done_asm) {
          } else if constexpr (OCC >= 2 && MREP * NREP > 8) {   // big tiles at two waves per SIMD: no room for a second fragment set
            for (int st = 0; st < ksteps; ++st) {
              load_frags(st, kt[st * 4], a0, b0);
              mfmas(a0, b0);
            }
          } else if (ksteps > 0) {
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

        const unsigned long long t2 = now();
        // (4) next stage's operands have landed (this wave's share); residual loads too
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = now();

        if (retire) {  // (5) finalize the tile into 16-byte slots (registers only)
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int np = 0; np < NPAIR; ++np) asm volatile("" : "+v"(slot[m][np]));
#pragma unroll
          for (int m = 0; m < MREP; ++m) {
            if (p.n_mblk > 1) bsv[m] = *reinterpret_cast<const float4*>(bias_l + mb * MT + m * 16 + q * 4);
            const float4 bs = bsv[m];
#pragma unroll
            for (int np = 0; np < NPAIR; ++np) {
              const int n0 = 2 * np, n1 = (2 * np + 1 < NREP) ? 2 * np + 1 : 2 * np;
              uint32_t a[4], b[4];
              a[0] = __float_as_uint(acc[J][m][n0][0] + bs.x); a[1] = __float_as_uint(acc[J][m][n0][1] + bs.y);
              a[2] = __float_as_uint(acc[J][m][n0][2] + bs.z); a[3] = __float_as_uint(acc[J][m][n0][3] + bs.w);
              b[0] = __float_as_uint(acc[J][m][n1][0] + bs.x); b[1] = __float_as_uint(acc[J][m][n1][1] + bs.y);
              b[2] = __float_as_uint(acc[J][m][n1][2] + bs.z); b[3] = __float_as_uint(acc[J][m][n1][3] + bs.w);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const auto sw = __builtin_amdgcn_permlane32_swap(a[jj], b[jj], false, false);
                a[jj] = sw[0]; b[jj] = sw[1];
              }
              // lower half-wave: a = own (n0, ch 0-3), b = partner's (n0, ch 4-7)
              // upper half-wave: a = partner's (n1, ch 0-3), b = own (n1, ch 4-7)
              float v[8];
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) { v[jj] = __uint_as_float(a[jj]); v[4 + jj] = __uint_as_float(b[jj]); }
              const u32x4 rv = slot[m][np];
              v[0] += from_bits<T>(rv[0] & 0xffff); v[1] += from_bits<T>(rv[0] >> 16);
              v[2] += from_bits<T>(rv[1] & 0xffff); v[3] += from_bits<T>(rv[1] >> 16);
              v[4] += from_bits<T>(rv[2] & 0xffff); v[5] += from_bits<T>(rv[2] >> 16);
              v[6] += from_bits<T>(rv[3] & 0xffff); v[7] += from_bits<T>(rv[3] >> 16);
              const uint32_t relu_floor = p.relu ? 0u : 0x80008000u;   // packed ReLU: signed 16-bit max (conv_device.h)
              u32x4 ov;
              ov[0] = relu2_16(pack2<T>(v[0], v[1]), relu_floor); ov[1] = relu2_16(pack2<T>(v[2], v[3]), relu_floor);
              ov[2] = relu2_16(pack2<T>(v[4], v[5]), relu_floor); ov[3] = relu2_16(pack2<T>(v[6], v[7]), relu_floor);
              slot[m][np] = ov;
            }
          }
        }

        // (6) one barrier per stage: every wave has drained its share of the next stage's DMA and
        //     finished reading this stage's buffers, which the next issue may overwrite
        const unsigned long long t4 = now();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long t5 = now();

        if (last && img >= 0) {  // (7) stores complete under the following stages
          if (retire) {
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
              for (int np = 0; np < NPAIR; ++np)
                if (ooff[m][np] != 0xffffffffu)
                  *reinterpret_cast<u32x4*>(static_cast<char*>(p.out) + img_off + ooff[m][np]) = slot[m][np];
          } else if (!SCP_DBG(p, 2)) {   // final layer: few channels, float32 NCHW, 4-byte stores
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
              const int co = mb * MT + m * 16 + psel * 8 + half * 4;
              const float4 bs = *reinterpret_cast<const float4*>(bias_l + mb * MT + m * 16 + q * 4);
#pragma unroll
              for (int n = 0; n < NREP; ++n) {
                if (py[n] < 0 || co >= p.cout) continue;
                const int oy = oy0 + py[n], ox = ox0 + px[n];
                if (oy >= p.Ho || ox >= p.Wo) continue;
                const float v[4] = {acc[J][m][n][0] + bs.x, acc[J][m][n][1] + bs.y, acc[J][m][n][2] + bs.z, acc[J][m][n][3] + bs.w};
                float* o = static_cast<float*>(p.out) + ((size_t)img * p.cout + co) * HoWo + (size_t)oy * p.Wo + ox;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                  if (co + jj < p.cout) o[jj * HoWo] = p.relu ? fmaxf(v[jj], 0.f) : v[jj];
              }
            }
          }
        }
        if SCP_DBG(p, 8) {
          const unsigned long long t6 = now();
          tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3; tph[4] += t5 - t4; tph[5] += t6 - t5;
        }
        if (!same_x) xb ^= 1;
      };

      stage(std::integral_constant<int, 0>{});
      if constexpr (NT > 1) stage(std::integral_constant<int, 1>{});
      if constexpr (NT > 2) stage(std::integral_constant<int, 2>{});
    }
  }
  if (SCP_DBG(p, 8) && SCP_DBG_BUF(p) && lane == 0)
    for (int k = 0; k < 6; ++k) SCP_DBG_BUF(p)[((size_t)blockIdx.x * 8 + wave_all) * 6 + k] = tph[k];
}

// ---- launch dispatch (instantiated per dtype in conv_pipe_bf16.hip / conv_pipe_f16.hip) ----
template <int DT, int KS, int STRIDE, int MREP, int NREP, int NT, int OCC, int G = 1>
int32_t pipe_launch_one(const ConvLaunch& L, size_t lds, hipStream_t st) {
  auto kern = conv_pipe_kernel<DT, KS, STRIDE, MREP, NREP, NT, OCC, G>;
  static LdsOptIn big_lds;   // per device (common.h)
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(kern), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(kern, dim3(L.grid), dim3(256 * G), lds, st, L);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

template <int DT, int KS, int STRIDE, int MREP, int NREP>
int32_t pipe_nt(int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
  if constexpr (KS == 3 && STRIDE == 1 && MREP >= 4) {
    if (nt == 2 && occ == 2 && MREP * NREP * 8 <= 160) return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 2, 2>(L, lds, st);
    if (nt == 2) return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 2, 1>(L, lds, st);
    if (nt == 3 && MREP * NREP * 12 <= 224) return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 3, 1>(L, lds, st);
  }
  if constexpr (KS == 3 && STRIDE == 2 && MREP >= 4 && NREP <= 2) {   // stride 2: a staged weight chunk serves 3 (or 2) pixel tiles
    if (nt == 3) return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 3, 1>(L, lds, st);
    if (nt == 2) return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 2, 1>(L, lds, st);
  }
  if (nt != 1) { set_error("conv: tile group %d unsupported for this variant", nt); return SCPOSE_E_INVALID; }
  if (occ == 2 && MREP * NREP * 4 <= 128) return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 1, 2>(L, lds, st);
  return pipe_launch_one<DT, KS, STRIDE, MREP, NREP, 1, 1>(L, lds, st);
}

template <int DT, int KS, int STRIDE, int MREP>
int32_t pipe_nrep(int nrep, int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
  switch (nrep) {
    case 1: return pipe_nt<DT, KS, STRIDE, MREP, 1>(nt, occ, L, lds, st);
    case 2: return pipe_nt<DT, KS, STRIDE, MREP, 2>(nt, occ, L, lds, st);
    case 3: return pipe_nt<DT, KS, STRIDE, MREP, 3>(nt, occ, L, lds, st);
    case 4: return pipe_nt<DT, KS, STRIDE, MREP, 4>(nt, occ, L, lds, st);
  }
  set_error("conv: nrep %d unsupported", nrep);
  return SCPOSE_E_INVALID;
}

template <int DT, int KS, int STRIDE>
int32_t pipe_mrep(int mrep, int nrep, int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
  switch (mrep) {
    case 1: return pipe_nrep<DT, KS, STRIDE, 1>(nrep, nt, occ, L, lds, st);
    case 2: return pipe_nrep<DT, KS, STRIDE, 2>(nrep, nt, occ, L, lds, st);
    case 3: return pipe_nrep<DT, KS, STRIDE, 3>(nrep, nt, occ, L, lds, st);
    case 4: return pipe_nrep<DT, KS, STRIDE, 4>(nrep, nt, occ, L, lds, st);
    case 6: return pipe_nrep<DT, KS, STRIDE, 6>(nrep, nt, occ, L, lds, st);
  }
  set_error("conv: mrep %d unsupported", mrep);
  return SCPOSE_E_INVALID;
}

template <int DT>
int32_t pipe_dispatch(int ks, int stride, int mrep, int nrep, int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
  if (ks == 3 && stride == 1) return pipe_mrep<DT, 3, 1>(mrep, nrep, nt, occ, L, lds, st);
  if (ks == 3 && stride == 2) return pipe_mrep<DT, 3, 2>(mrep, nrep, nt, occ, L, lds, st);
  if (ks == 1 && stride == 1) return pipe_mrep<DT, 1, 1>(mrep, nrep, nt, occ, L, lds, st);
  set_error("conv: k=%d stride=%d unsupported", ks, stride);
  return SCPOSE_E_INVALID;
}

}  // namespace scpose
