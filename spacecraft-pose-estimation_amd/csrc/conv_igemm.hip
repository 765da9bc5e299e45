// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 (CDNA4) on v_mfma_f32_16x16x32_{bf16,f16}:
// host side (tiling choice, weight packing, launch).  The kernel is in conv_pipe.hip.
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
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "conv_device.h"

namespace scpose {

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
  else if (planes <= 6 && mtiles == mr) lim = 6;   // whole K in one chunk, one Cout block: weights stay resident
  else if (mr >= 4) lim = 2;                        // weight-streaming 3x3: small chunks so that TWO workgroups fit a CU
  else lim = 4;
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

// MFMA accumulator row (4*(lane>>4) + reg) -> output channel within the 16-channel tile:
// lane groups q = 0,2 (lanes l, l+32) share plane 0 (channels 0-3 | 4-7), q = 1,3 plane 1.
static inline int conv_row_channel(int row) {
  const int q = row >> 2, reg = row & 3;
  return (q & 1) * 8 + (q >> 1) * 4 + reg;
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
            // MFMA row rr of a 16-row tile carries channel conv_row_channel(rr): lanes l and l+32
            // then hold the two halves of one 8-channel plane (16-byte epilogue stores)
            const int co = mb * mt + (r & ~15) + conv_row_channel(r & 15);
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
  std::vector<uint16_t> host;
  std::vector<float> hb;
  if (conv_m32_choose(cin, cout, ks, stride, &pc->mrep, &pc->wm, &pc->cp)) {   // 32x32x16 MFMA kernel
    pc->variant = 1;
    pc->mt = pc->mrep == kMrep48 ? 48 : 32 * pc->mrep * pc->wm;
    pc->n_mblk = (cout + pc->mt - 1) / pc->mt;
    pc->wbytes = pack_conv_weights_m32(w, cout, cin, ks, pc->mt, pc->cp, dtype, nullptr, &pc->nchunks, &pc->ksteps_full);
    host.resize(pc->wbytes / 2);
    pack_conv_weights_m32(w, cout, cin, ks, pc->mt, pc->cp, dtype, host.data(), nullptr, nullptr);
    hb.assign((size_t)pc->n_mblk * pc->mt, 0.f);
    if (bias) for (int co = 0; co < cout; ++co) hb[co] = bias[co];   // natural channel order
  } else {
    pc->variant = 0; pc->wm = 1;
    choose_mrep_cp(cin, cout, ks, stride, &pc->mrep, &pc->cp);
    pc->mt = 16 * pc->mrep;
    pc->n_mblk = (cout + pc->mt - 1) / pc->mt;
    pc->wbytes = pack_conv_weights(w, cout, cin, ks, pc->mt, pc->cp, dtype, nullptr, &pc->nchunks,
                                   &pc->ksteps_full);
    host.resize(pc->wbytes / 2);
    pack_conv_weights(w, cout, cin, ks, pc->mt, pc->cp, dtype, host.data(), nullptr, nullptr);
    hb.assign((size_t)pc->n_mblk * pc->mt, 0.f);
    if (bias)   // stored in packed (MFMA row) order, like the weight rows
      for (size_t pos = 0; pos < hb.size(); ++pos) {
        const int co = (int)(pos & ~(size_t)15) + conv_row_channel((int)(pos & 15));
        if (co < cout) hb[pos] = bias[co];
      }
  }
  if (ks == 1 && stride == 1 && cout % 8 == 0 && pc->variant == 0) {   // streaming 1x1 kernel (conv1x1.hip) when the weights fit LDS: two workgroups
                                                                        // per CU up to 64 KB, one above (384 -> 192 / 96 of the stage-4 fuse layers: 144 / 96 KB)
    const int planes = cin / 8, ksteps = (planes + 3) / 4, cpad = (cout + 63) / 64 * 64;
    const size_t wb = (size_t)ksteps * 4 * cpad * 16;
    const bool ks_ok = (ksteps >= 1 && ksteps <= 4) || ksteps == 6 || ksteps == 8 || ksteps == 12;   // Cin = 32..128, 192, 256, 384
    // (the one-workgroup-per-CU variant -- more than 80 KB of LDS -- is built for k-step counts 6 / 8 / 12 only: shallower layers with
    // that many weights, e.g. 128 -> 384, keep the conv_pipe path instead of launching a 2-per-CU variant past its LDS opt-in)
    const size_t lds1 = wb + (size_t)cpad * 4;
    if (ks_ok && lds1 <= (ksteps <= 4 ? 80 * 1024 : 156 * 1024)) {
      std::vector<uint16_t> h1(wb / 2);
      const size_t got = pack_conv_weights(w, cout, cin, 1, cpad, planes, dtype, h1.data(), nullptr, nullptr);
      SCP_REQUIRE(got == wb, "conv1x1: packed size %zu != %zu", got, wb);
      std::vector<float> b1(cpad, 0.f);
      if (bias)
        for (int pos = 0; pos < cpad; ++pos) {
          const int co = (pos & ~15) + conv_row_channel(pos & 15);
          if (co < cout) b1[pos] = bias[co];
        }
      SCP_CHECK_HIP(hipMalloc(&pc->d_w1, wb));
      SCP_CHECK_HIP(hipMalloc(&pc->d_b1, b1.size() * sizeof(float)));
      SCP_CHECK_HIP(hipMemcpy(pc->d_w1, h1.data(), wb, hipMemcpyHostToDevice));
      SCP_CHECK_HIP(hipMemcpy(pc->d_b1, b1.data(), b1.size() * sizeof(float), hipMemcpyHostToDevice));
      pc->w1_bytes = wb; pc->cout_pad1 = cpad;
    }
  }
  {   // 3x3 stride-2 layers with Cin = 32 / 48 / 64: register-weight kernel (conv_s2r.hip)
    static const char* e = dev_env("SCPOSE_S2R");
    int planes, nblk, g;
    if (ks == 3 && !(e && atoi(e) == 0) && conv_s2r_config(cin, cout, stride, &planes, &nblk, &g)) {
      std::vector<uint16_t> h2(conv_s2r_pack(w, cout, cin, dtype, nullptr) / 2);
      conv_s2r_pack(w, cout, cin, dtype, h2.data());
      std::vector<float> b2(cout);
      conv_s2r_pack_bias(bias, cout, b2.data());
      SCP_CHECK_HIP(hipMalloc(&pc->d_ws2, h2.size() * 2));
      SCP_CHECK_HIP(hipMalloc(&pc->d_bs2, b2.size() * sizeof(float)));
      SCP_CHECK_HIP(hipMemcpy(pc->d_ws2, h2.data(), h2.size() * 2, hipMemcpyHostToDevice));
      SCP_CHECK_HIP(hipMemcpy(pc->d_bs2, b2.data(), b2.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
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
  if (pc->d_w1) (void)hipFree(pc->d_w1);
  if (pc->d_b1) (void)hipFree(pc->d_b1);
  if (pc->d_ws2) (void)hipFree(pc->d_ws2);
  if (pc->d_bs2) (void)hipFree(pc->d_bs2);
  pc->d_w = nullptr; pc->d_bias = nullptr; pc->d_w1 = nullptr; pc->d_b1 = nullptr; pc->d_ws2 = nullptr; pc->d_bs2 = nullptr;
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

int32_t conv_launch(const PackedConv& pc, const void* in, int N, int H, int W, const void* res,
                    int relu, int out_nchw_f32, void* out, hipStream_t stream, const void* in2, int split_planes, int cu_share) {
  SCP_REQUIRE(!in2 || (pc.variant == 0 && pc.ks == 1 && split_planes > 0 && split_planes % pc.cp == 0 && split_planes < pc.cin / 8),
              "conv: a second input tensor needs a 1x1 layer whose K-chunks do not straddle the split (split=%d planes, cp=%d)", split_planes, pc.cp);
  SCP_REQUIRE(N > 0 && H > 0 && W > 0, "conv: bad shape N=%d H=%d W=%d", N, H, W);
  SCP_REQUIRE(out_nchw_f32 || pc.cout % 8 == 0, "conv: blocked output needs Cout%%8==0 (Cout=%d)", pc.cout);
  ConvLaunch L;
  L.in = in; L.in2 = in2; L.split_planes = in2 ? split_planes : 0; L.wpk = pc.d_w; L.bias = pc.d_bias; L.res = res; L.out = out;
  L.N = N; L.H = H; L.W = W;
  L.Ho = (H - 1) / pc.stride + 1;  // k=3,p=1 or k=1,p=0
  L.Wo = (W - 1) / pc.stride + 1;
  L.cin_planes = pc.cin / 8;
  L.cout = pc.cout;
  L.cu_share = cu_share;
  if (pc.d_ws2 && !res && !in2 && !out_nchw_f32 && (size_t)N * L.cin_planes * H * W * 16 < 0xfffffff0ull &&
      (size_t)N * (pc.cout / 8) * L.Ho * L.Wo * 16 < 0xfffffff0ull)
    return conv_s2r_launch(pc, in, N, H, W, relu, out, stream);
  if (pc.variant == 1) {
    SCP_REQUIRE(!out_nchw_f32, "conv m32: float32 NCHW output unsupported");
    L.relu = relu; L.out_nchw_f32 = 0;
    return conv_launch_m32(pc, L, stream);
  }
  {   // 1x1 layers whose weights fit LDS: barrier-free streaming kernel (needs buffer-addressable tensors)
    static const char* e1 = dev_env("SCPOSE_K1_STREAM");
    const size_t ib = (size_t)N * L.cin_planes * H * W * 16, ob = (size_t)N * (pc.cout / 8) * H * W * 16;
    if (pc.d_w1 && (!in2 || split_planes % 4 == 0) && !out_nchw_f32 && ib < 0xfffffff0ull && ob < 0xfffffff0ull && !(e1 && atoi(e1) == 0))
      return conv1x1_stream_launch(pc, in, N, H, W, res, relu, out, stream, in2, split_planes);
  }
  int nrep;
  choose_tile(pc.ks, pc.stride, L.Ho, L.Wo, &nrep, &L.th, &L.tw);
  const int k2 = pc.ks / 2;
  auto set_geometry = [&]() {
    L.tiles_x = (L.Wo + L.tw - 1) / L.tw;
    L.tiles_y = (L.Ho + L.th - 1) / L.th;
    L.halo_h = (L.th - 1) * pc.stride + 1 + 2 * k2;
    L.halo_w = (L.tw - 1) * pc.stride + 1 + 2 * k2;
    L.plane_stride = plane_stride_for(pc.stride, L.halo_h, L.halo_w);
    nrep = (L.th * L.tw + 63) / 64;
  };
  set_geometry();
  const bool resident = pc.nchunks == 1 && pc.n_mblk == 1;
  // Weight-resident layers are bound by HBM, not MFMA: prefer a tile small enough for TWO resident
  // workgroups per CU (one streams while the other computes) over a big tile with one.
  if (resident && 2 * conv_pipe_lds_bytes(pc, L.plane_stride, 1) > 160 * 1024 && L.th % 2 == 0 && L.th * L.tw >= 128) {
    const int th0 = L.th;
    L.th /= 2;
    set_geometry();
    if (2 * conv_pipe_lds_bytes(pc, L.plane_stride, 1) > 160 * 1024) { L.th = th0; set_geometry(); }
  }
  {   // development: SCPOSE_TILE="th,tw" overrides the tile of weight-resident 3x3 stride-1 layers
    static const char* e = dev_env("SCPOSE_TILE");
    int eth = 0, etw = 0;
    if (e && resident && pc.ks == 3 && pc.stride == 1 && sscanf(e, "%d,%d", &eth, &etw) == 2 && eth > 0 && etw > 0) {
      L.th = eth; L.tw = etw; set_geometry();
    }
  }
  // Weight-streaming 3x3 layers.  Measured on MI355X: a CU's LDS-DMA path moves ~16 B/clk and the
  // issuing wave stalls for it.  So (a) the bytes staged per MFMA must be small: one staged weight
  // chunk serves TWO pixel tiles; and (b) DMA issue must overlap MFMAs: the two tiles belong to two
  // wave groups of one 512-thread workgroup (two waves per SIMD, <= 256 registers each).
  int nt = 1, occ = 1, groups = 1;
  if (!resident && pc.ks == 3 && pc.stride == 1 && pc.mrep >= 4) {
    groups = 2;
    if (pc.mrep * nrep > 24) {   // 256-register budget of the two-group variant: at most 6 x 4 MFMA tiles per wave
      int th = L.th;
      while (th > 1 && (pc.mrep * ((th * L.tw + 63) / 64) > 24 || L.Ho % th != 0)) --th;
      if (pc.mrep * ((th * L.tw + 63) / 64) <= 24) { L.th = th; set_geometry(); }
    }
    if (conv_pipe_lds_bytes(pc, L.plane_stride, 2) > 160 * 1024 || pc.mrep * nrep > 24) groups = 1;
    if (groups == 1 && 2 * pc.mrep * nrep * 4 <= 224) nt = 2;
  }
  if (!resident && pc.ks == 3 && pc.stride == 2 && pc.mrep >= 4 && nrep <= 2) {
    static const char* e = dev_env("SCPOSE_S2_NT");
    nt = e ? atoi(e) : 2;   // weight chunk shared by 2 sequential pixel tiles (measured: 3 is no better than 1)
  }
  if (resident && 2 * conv_pipe_lds_bytes(pc, L.plane_stride, 1) <= 160 * 1024) occ = 2;
  // 1x1 layers are bound by their memory instructions (K is tiny): two workgroups per CU double the waves that
  // keep loads and stores in flight; halve the pixel tile until two of them fit the LDS and the register budget
  static const char* e11 = dev_env("SCPOSE_K1_OCC");
  static const char* e11p = dev_env("SCPOSE_K1_MINPIX");
  // (not with a single K-step, Cin <= 32: measured 124 us with two workgroups vs 82 us with one for 32->128 @192x192x16)
  if (pc.ks == 1 && !out_nchw_f32 && L.Ho * L.Wo >= (e11p ? atoi(e11p) : 4096) && !(e11 && atoi(e11) == 1) &&
      (pc.ksteps_full >= 2 || (e11 && atoi(e11) == 2))) {   // high-resolution maps only: measured slower on the small ones
    while (L.th % 2 == 0 && L.th * L.tw > 64 &&
           (2 * conv_pipe_lds_bytes(pc, L.plane_stride, 1) > 160 * 1024 || pc.mrep * nrep * 4 > 128)) {
      L.th /= 2;
      set_geometry();
    }
    if (2 * conv_pipe_lds_bytes(pc, L.plane_stride, 1) <= 160 * 1024 && pc.mrep * nrep * 4 <= 128) occ = 2;
  }
  {   // buffer-addressed input DMA when the input tensor(s) fit a 32-bit descriptor
    static const char* e = dev_env("SCPOSE_M32_BUF");
    const size_t plane = (size_t)N * H * W * 16;
    const size_t ib = plane * (in2 ? split_planes : L.cin_planes), ib2 = in2 ? plane * (L.cin_planes - split_planes) : 0;
    const bool fits = ib < 0xfffffff0ull && ib2 < 0xfffffff0ull && !(e && atoi(e) == 0);
    L.in_bytes = fits ? (uint32_t)ib : 0;
    L.out_bytes = fits ? (uint32_t)ib2 : 0;
  }
  L.cp = pc.cp; L.nchunks = pc.nchunks; L.ksteps_full = pc.ksteps_full;
  L.n_mblk = pc.n_mblk;
  L.relu = relu; L.out_nchw_f32 = out_nchw_f32;
  L.total_blocks = N * L.tiles_x * L.tiles_y * pc.n_mblk;
  SCP_REQUIRE(L.halo_h * L.halo_w <= (pc.stride == 1 ? 2 : 3) * 256, "conv: halo %dx%d too large",
              L.halo_h, L.halo_w);
  SCP_REQUIRE((pc.ksteps_full + 1) * 4 <= 64, "conv: k-offset table overflow (%d ksteps)", pc.ksteps_full);
  return conv_launch_pipe(pc, L, nrep, nt, occ, groups, stream);
}

}  // namespace scpose
