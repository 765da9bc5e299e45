// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 (CDNA4) on v_mfma_f32_16x16x32_{bf16,f16}.
//
// Replaces every nn.Conv2d + eval BatchNorm2d (+ReLU, +residual add) of the reference
// landmark_regression/lib/models/pose_hrnet.py (conv3x3 :22-25, BasicBlock :41-57,
// Bottleneck :78-98, transition :343-368, fuse down path :216-237, fuse 1x1 :199-205,
// final_layer :323-329) except the 3-channel stem conv (stem.hip).
//
// GEMM view:  D[cout][pixel] = sum_k Wt[cout][k] * X[k][pixel],  k = (input plane, tap, 8 ch).
//   * A operand = weights (M = Cout), B operand = activations (N = pixels).  With this
//     orientation an accumulator lane holds 4 consecutive Cout of ONE pixel, i.e. 8
//     contiguous bytes of the blocked [N][C/8][H][W][8] output, and a B fragment is exactly
//     one 16-byte (pixel, 8-channel) vector of the blocked input: no transposes anywhere.
//   * Workgroup = 4 waves; every wave owns all MT = 16*MREP output channels of the block and
//     NREP 16-pixel tiles; the (th x tw) output tile's input halo is staged in LDS one
//     K-chunk (cp planes = 8*cp channels) at a time together with that chunk's weights.
//   * K order inside a chunk is (plane pair, tap); MFMA k-group q (lane>>4) reads plane
//     2*pp + (q&1) at tap pt = 2*s + (q>>1).  The two k-groups that share a ds_read_b128
//     lane group (q = 0,1 and q = 2,3) therefore differ by one whole LDS plane, whose stride
//     is padded so both land on disjoint banks.
#include <stdlib.h>

#include "common.h"
#include "conv_device.h"

namespace scpose {

template <int DT, int KS, int STRIDE, int MREP, int NREP>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvLaunch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  constexpr int MT = 16 * MREP;
  constexpr int MAXP = (STRIDE == 1) ? 2 : 3;   // halo pixels per thread (host guarantees)
  constexpr int KK = KS * KS;

  int* koff = reinterpret_cast<int*>(smem);                 // 64 entries
  char* wl = smem + 256;                                    // weights chunk
  char* xl = wl + p.ksteps_full * (4 * MT * 16);            // input chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;

  const int bid = xcd_remap(blockIdx.x, p.total_blocks);
  const int mb = bid % p.n_mblk;
  int t = bid / p.n_mblk;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int oy0 = ty * p.th, ox0 = tx * p.tw;
  const int iy0 = oy0 * STRIDE - (KS / 2), ix0 = ox0 * STRIDE - (KS / 2);
  const int HW = p.H * p.W;
  const int HP = p.halo_h * p.halo_w;

  // this thread's halo pixels: global pixel offset (or -1 = zero padding, -2 = none)
  int goff[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int hp = tid + i * 256;
    int g = -2;
    if (hp < HP) {
      const int hy = hp / p.halo_w, hx = hp - hy * p.halo_w;
      const int iy = iy0 + hy, ix = ix0 + hx;
      g = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? iy * p.W + ix : -1;
    }
    goff[i] = g;
  }

  // this lane's pixels (B-operand column) in each of the wave's NREP tiles
  int pixoff[NREP];
  const int npix = p.th * p.tw;
#pragma unroll
  for (int n = 0; n < NREP; ++n) {
    const int pidx = (wave * NREP + n) * 16 + r;
    int off = 0;
    if (pidx < npix) {
      const int y = pidx / p.tw, x = pidx - y * p.tw;
      off = ((y * STRIDE) * p.halo_w + x * STRIDE) * 16;
    }
    pixoff[n] = off;
  }

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* inb = static_cast<const char*>(p.in) + (size_t)img * p.cin_planes * HW * 16;
  const size_t chunk_wbytes = (size_t)p.ksteps_full * (4 * MT * 16);
  const char* wb = static_cast<const char*>(p.wpk) + (size_t)mb * p.nchunks * chunk_wbytes;

  for (int c = 0; c < p.nchunks; ++c) {
    const int plane0 = c * p.cp;
    const int planes = min(p.cp, p.cin_planes - plane0);
    const int npt = (planes >> 1) * KK;
    const int ksteps = (npt + 1) >> 1;

    __syncthreads();  // all waves finished reading the previous chunk
    if (tid < (ksteps + 1) * 4) {
      const int s = tid >> 2, qq = tid & 3;
      const int pt = 2 * s + (qq >> 1);
      int off = 0;
      if (pt < npt) {
        const int pp = pt / KK, tap = pt - pp * KK;
        const int ky = tap / KS, kx = tap - ky * KS;
        off = (2 * pp + (qq & 1)) * p.plane_stride + (ky * p.halo_w + kx) * 16;
      }
      koff[tid] = off;
    }
    {  // weights: one contiguous LDS image per (Cout block, chunk)
      const char* ws = wb + (size_t)c * chunk_wbytes;
      const int nbytes = ksteps * (4 * MT * 16);
      for (int i = tid * 16; i < nbytes; i += 256 * 16)
        *reinterpret_cast<uint4*>(wl + i) = *reinterpret_cast<const uint4*>(ws + i);
    }
    for (int pl = 0; pl < planes; ++pl) {
      const char* src = inb + (size_t)(plane0 + pl) * HW * 16;
      char* dst = xl + pl * p.plane_stride + tid * 16;
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        if (goff[i] != -2) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (goff[i] >= 0) v = *reinterpret_cast<const uint4*>(src + (size_t)goff[i] * 16);
          *reinterpret_cast<uint4*>(dst + i * 4096) = v;
        }
      }
    }
    __syncthreads();

    const char* wq = wl + (q * MT + r) * 16;
    for (int s = 0; s < ksteps; ++s) {
      const int ko = koff[s * 4 + q];
      frag_t a[MREP], b[NREP];
#pragma unroll
      for (int m = 0; m < MREP; ++m)
        a[m] = *reinterpret_cast<const frag_t*>(wq + s * (4 * MT * 16) + m * 256);
#pragma unroll
      for (int n = 0; n < NREP; ++n)
        b[n] = *reinterpret_cast<const frag_t*>(xl + ko + pixoff[n]);
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma16<T>(a[m], b[n], acc[m][n]);
    }
  }

  // ---- epilogue: + bias (folded BN) [+ residual] [ReLU] -> 16-bit blocked / f32 NCHW ----
  const int cout_planes = (p.cout + 7) >> 3;
  const size_t HoWo = (size_t)p.Ho * p.Wo;
#pragma unroll
  for (int n = 0; n < NREP; ++n) {
    const int pidx = (wave * NREP + n) * 16 + r;
    if (pidx >= npix) continue;
    const int y = pidx / p.tw, x = pidx - y * p.tw;
    const int oy = oy0 + y, ox = ox0 + x;
    if (oy >= p.Ho || ox >= p.Wo) continue;
    const size_t opix = (size_t)oy * p.Wo + ox;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
      const int co = mb * MT + m * 16 + q * 4;
      if (co >= p.cout) continue;
      const float4 bs = *reinterpret_cast<const float4*>(p.bias + co);
      float v0 = acc[m][n][0] + bs.x, v1 = acc[m][n][1] + bs.y;
      float v2 = acc[m][n][2] + bs.z, v3 = acc[m][n][3] + bs.w;
      const size_t boff =
          (((size_t)img * cout_planes + (co >> 3)) * HoWo + opix) * 16 + (co & 7) * 2;
      if (p.res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.res) + boff);
        v0 += from_bits<T>(rv.x & 0xffff); v1 += from_bits<T>(rv.x >> 16);
        v2 += from_bits<T>(rv.y & 0xffff); v3 += from_bits<T>(rv.y >> 16);
      }
      if (p.relu) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      }
      if (p.out_nchw_f32) {
        float* o = static_cast<float*>(p.out) + ((size_t)img * p.cout + co) * HoWo + opix;
        o[0] = v0;
        if (co + 1 < p.cout) o[HoWo] = v1;
        if (co + 2 < p.cout) o[2 * HoWo] = v2;
        if (co + 3 < p.cout) o[3 * HoWo] = v3;
      } else {
        uint2 ov;
        ov.x = (uint32_t)to_bits<T>(v0) | ((uint32_t)to_bits<T>(v1) << 16);
        ov.y = (uint32_t)to_bits<T>(v2) | ((uint32_t)to_bits<T>(v3) << 16);
        *reinterpret_cast<uint2*>(static_cast<char*>(p.out) + boff) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int largest_divisor_leq(int n, int lim) {
  for (int d = lim < n ? lim : n; d >= 1; --d)
    if (n % d == 0) return d;
  return 1;
}

void choose_mrep_cp(int cin, int cout, int ks, int stride, int* mrep, int* cp) {
  const int mtiles = (cout + 15) / 16;
  int mr = 1;
  const int cand[5] = {6, 4, 3, 2, 1};
  for (int i = 0; i < 5; ++i)
    if (mtiles % cand[i] == 0) { mr = cand[i]; break; }
  if (mtiles >= 5 && mr == 1) mr = 4;  // awkward Cout: pad the last block
  *mrep = mr;
  const int planes = cin / 8;
  int lim;
  if (ks == 1) lim = 8;
  else if (stride == 2) lim = (mr >= 4) ? 2 : 4;
  else lim = (planes == 6) ? 6 : 4;
  int c = 2;
  for (int d = lim; d >= 2; d -= 2)
    if (planes % d == 0) { c = d; break; }
  *cp = c;
}

void choose_tile(int ks, int stride, int Ho, int Wo, int* nrep, int* th, int* tw) {
  const int ptmax = stride == 1 ? 256 : 128;
  int w = (Wo % 16 == 0 || Wo > 32) ? 16 : Wo;
  int h = largest_divisor_leq(Ho, ptmax / w > 0 ? ptmax / w : 1);
  // staged halo must fit MAXP * 256 pixels
  const int maxhp = (stride == 1 ? 2 : 3) * 256;
  const int k2 = ks / 2;
  while (h > 1 && ((h - 1) * stride + 1 + 2 * k2) * ((w - 1) * stride + 1 + 2 * k2) > maxhp)
    h = largest_divisor_leq(Ho, h - 1);
  *th = h; *tw = w;
  *nrep = (h * w + 63) / 64;
}

uint16_t host_f32_to_16(float f, int dtype) {
  uint32_t u; memcpy(&u, &f, 4);
  if (dtype == SCPOSE_DT_BF16) {
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  _Float16 h = (_Float16)f;
  uint16_t b; memcpy(&b, &h, 2);
  return b;
}

float host_16_to_f32(uint16_t v, int dtype) {
  if (dtype == SCPOSE_DT_BF16) {
    uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f;
  }
  _Float16 h; memcpy(&h, &v, 2);
  return (float)h;
}

size_t pack_conv_weights(const float* w, int cout, int cin, int ks, int mt, int cp, int dtype,
                         uint16_t* dst, int* nchunks_out, int* ksteps_full_out) {
  const int planes = cin / 8, kk = ks * ks;
  const int nchunks = (planes + cp - 1) / cp;
  const int ksteps_full = ((cp / 2) * kk + 1) / 2;
  const int n_mblk = (cout + mt - 1) / mt;
  const size_t chunk_elems = (size_t)ksteps_full * 4 * mt * 8;
  const size_t total = (size_t)n_mblk * nchunks * chunk_elems;
  if (nchunks_out) *nchunks_out = nchunks;
  if (ksteps_full_out) *ksteps_full_out = ksteps_full;
  if (!dst) return total * 2;
  memset(dst, 0, total * 2);
  for (int mb = 0; mb < n_mblk; ++mb)
    for (int c = 0; c < nchunks; ++c) {
      const int plane0 = c * cp;
      const int pl = (planes - plane0) < cp ? (planes - plane0) : cp;
      const int npt = (pl / 2) * kk;
      uint16_t* base = dst + ((size_t)mb * nchunks + c) * chunk_elems;
      for (int s = 0; s < (npt + 1) / 2; ++s)
        for (int q = 0; q < 4; ++q) {
          const int pt = 2 * s + (q >> 1);
          if (pt >= npt) continue;
          const int pp = pt / kk, tap = pt % kk;
          const int plane = plane0 + 2 * pp + (q & 1);
          for (int r = 0; r < mt; ++r) {
            const int co = mb * mt + r;
            if (co >= cout) continue;
            uint16_t* d = base + ((size_t)(s * 4 + q) * mt + r) * 8;
            for (int j = 0; j < 8; ++j) {
              const int ci = plane * 8 + j;
              d[j] = host_f32_to_16(w[((size_t)co * cin + ci) * kk + tap], dtype);
            }
          }
        }
    }
  return total * 2;
}

int32_t conv_upload(const float* w, const float* bias, int cout, int cin, int ks, int stride,
                    int dtype, PackedConv* pc) {
  SCP_REQUIRE(ks == 1 || ks == 3, "conv: kernel size %d unsupported (1 or 3)", ks);
  SCP_REQUIRE(stride == 1 || (stride == 2 && ks == 3), "conv: stride %d with k=%d unsupported", stride, ks);
  SCP_REQUIRE(cin % 16 == 0 && cin > 0, "conv: Cin=%d must be a multiple of 16", cin);
  SCP_REQUIRE(cout > 0, "conv: Cout=%d", cout);
  SCP_REQUIRE(dtype == SCPOSE_DT_BF16 || dtype == SCPOSE_DT_F16, "conv: dtype %d", dtype);
  pc->cin = cin; pc->cout = cout; pc->ks = ks; pc->stride = stride; pc->dtype = dtype;
  choose_mrep_cp(cin, cout, ks, stride, &pc->mrep, &pc->cp);
  pc->mt = 16 * pc->mrep;
  pc->n_mblk = (cout + pc->mt - 1) / pc->mt;
  pc->wbytes = pack_conv_weights(w, cout, cin, ks, pc->mt, pc->cp, dtype, nullptr, &pc->nchunks,
                                 &pc->ksteps_full);
  std::vector<uint16_t> host(pc->wbytes / 2);
  pack_conv_weights(w, cout, cin, ks, pc->mt, pc->cp, dtype, host.data(), nullptr, nullptr);
  std::vector<float> hb((size_t)pc->n_mblk * pc->mt, 0.f);
  if (bias) for (int i = 0; i < cout; ++i) hb[i] = bias[i];
  SCP_REQUIRE(conv_zero_page() != nullptr, "conv: cannot allocate the zero page");  // create-time, not in the launch path
  SCP_CHECK_HIP(hipMalloc(&pc->d_w, pc->wbytes));
  SCP_CHECK_HIP(hipMalloc(&pc->d_bias, hb.size() * sizeof(float)));
  SCP_CHECK_HIP(hipMemcpy(pc->d_w, host.data(), pc->wbytes, hipMemcpyHostToDevice));
  SCP_CHECK_HIP(hipMemcpy(pc->d_bias, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
  return SCPOSE_OK;
}

void conv_free(PackedConv* pc) {
  if (pc->d_w) (void)hipFree(pc->d_w);
  if (pc->d_bias) (void)hipFree(pc->d_bias);
  pc->d_w = nullptr; pc->d_bias = nullptr;
}

int plane_stride_for(int stride, int halo_h, int halo_w) {
  int bytes = halo_h * halo_w * 16;
  if (stride == 1) return (bytes + 255) & ~255;       // q and q^1 planes: same bank phase
  bytes = (bytes + 31) & ~31;                          // stride 2: odd 16-B slot phase
  return bytes + 16;
}

size_t conv_lds_bytes(const PackedConv& pc, int nrep, int th, int tw) {
  (void)nrep;
  const int k2 = pc.ks / 2;
  const int hh = (th - 1) * pc.stride + 1 + 2 * k2, hw = (tw - 1) * pc.stride + 1 + 2 * k2;
  return 256 + (size_t)pc.ksteps_full * 4 * pc.mt * 16 +
         (size_t)pc.cp * plane_stride_for(pc.stride, hh, hw);
}

template <int T, int KS, int STRIDE, int MREP>
static int32_t launch_nrep(int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
  dim3 grid(L.total_blocks), block(256);
#define SCP_LAUNCH(NR)                                                                          \
  case NR: {                                                                                    \
    auto kern = conv_igemm_kernel<T, KS, STRIDE, MREP, NR>;                                      \
    static bool big_lds_enabled = false; /* once per instantiation, outside any graph capture */\
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

template <int T, int KS, int STRIDE>
static int32_t launch_mrep(int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
  switch (mrep) {
    case 1: return launch_nrep<T, KS, STRIDE, 1>(nrep, L, lds, st);
    case 2: return launch_nrep<T, KS, STRIDE, 2>(nrep, L, lds, st);
    case 3: return launch_nrep<T, KS, STRIDE, 3>(nrep, L, lds, st);
    case 4: return launch_nrep<T, KS, STRIDE, 4>(nrep, L, lds, st);
    case 6: return launch_nrep<T, KS, STRIDE, 6>(nrep, L, lds, st);
  }
  set_error("conv: mrep %d unsupported", mrep);
  return SCPOSE_E_INVALID;
}

template <int T>
static int32_t launch_ks(int ks, int stride, int mrep, int nrep, const ConvLaunch& L, size_t lds,
                         hipStream_t st) {
  if (ks == 3 && stride == 1) return launch_mrep<T, 3, 1>(mrep, nrep, L, lds, st);
  if (ks == 3 && stride == 2) return launch_mrep<T, 3, 2>(mrep, nrep, L, lds, st);
  if (ks == 1 && stride == 1) return launch_mrep<T, 1, 1>(mrep, nrep, L, lds, st);
  set_error("conv: k=%d stride=%d unsupported", ks, stride);
  return SCPOSE_E_INVALID;
}

int32_t conv_launch(const PackedConv& pc, const void* in, int N, int H, int W, const void* res,
                    int relu, int out_nchw_f32, void* out, hipStream_t stream) {
  SCP_REQUIRE(N > 0 && H > 0 && W > 0, "conv: bad shape N=%d H=%d W=%d", N, H, W);
  SCP_REQUIRE(out_nchw_f32 || pc.cout % 8 == 0, "conv: blocked output needs Cout%%8==0 (Cout=%d)", pc.cout);
  ConvLaunch L;
  L.in = in; L.wpk = pc.d_w; L.bias = pc.d_bias; L.res = res; L.out = out;
  L.N = N; L.H = H; L.W = W;
  L.Ho = (H - 1) / pc.stride + 1;  // k=3,p=1 or k=1,p=0
  L.Wo = (W - 1) / pc.stride + 1;
  L.cin_planes = pc.cin / 8;
  L.cout = pc.cout;
  int nrep;
  choose_tile(pc.ks, pc.stride, L.Ho, L.Wo, &nrep, &L.th, &L.tw);
  L.tiles_x = (L.Wo + L.tw - 1) / L.tw;
  L.tiles_y = (L.Ho + L.th - 1) / L.th;
  const int k2 = pc.ks / 2;
  L.halo_h = (L.th - 1) * pc.stride + 1 + 2 * k2;
  L.halo_w = (L.tw - 1) * pc.stride + 1 + 2 * k2;
  L.plane_stride = plane_stride_for(pc.stride, L.halo_h, L.halo_w);
  L.cp = pc.cp; L.nchunks = pc.nchunks; L.ksteps_full = pc.ksteps_full;
  L.n_mblk = pc.n_mblk;
  L.relu = relu; L.out_nchw_f32 = out_nchw_f32;
  L.total_blocks = N * L.tiles_x * L.tiles_y * pc.n_mblk;
  SCP_REQUIRE(L.halo_h * L.halo_w <= (pc.stride == 1 ? 2 : 3) * 256, "conv: halo %dx%d too large",
              L.halo_h, L.halo_w);
  SCP_REQUIRE((pc.ksteps_full + 1) * 4 <= 64, "conv: k-offset table overflow (%d ksteps)", pc.ksteps_full);
  static const bool force_v1 = getenv("SCPOSE_CONV_V1") != nullptr;   // A/B switch for development
  if (!force_v1) {
    bool fits = false;
    const int32_t rc = conv_launch_pipe(pc, L, nrep, stream, &fits);
    if (fits) return rc;
  }
  const size_t lds = conv_lds_bytes(pc, nrep, L.th, L.tw);
  SCP_REQUIRE(lds <= 160 * 1024, "conv: LDS %zu bytes exceeds 160 KiB", lds);
  if (pc.dtype == SCPOSE_DT_BF16)
    return launch_ks<0>(pc.ks, pc.stride, pc.mrep, nrep, L, lds, stream);
  return launch_ks<1>(pc.ks, pc.stride, pc.mrep, nrep, L, lds, stream);
}

}  // namespace scpose
