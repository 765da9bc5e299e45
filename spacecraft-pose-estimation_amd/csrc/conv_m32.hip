// Host side of the 32x32x16-MFMA convolution (conv_m32_kernel.h): eligibility, weight packing,
// tile search, launch.  See the kernel header for the design and the measurements behind it.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace scpose {

int32_t conv_m32_dispatch_bf16(int mr, int wm, int nr, int occ, const ConvLaunch& L, size_t lds, hipStream_t st);
int32_t conv_m32_dispatch_f16(int mr, int wm, int nr, int occ, const ConvLaunch& L, size_t lds, hipStream_t st);
int32_t conv_m32p_dispatch_bf16(int stride, int mr, int nr, int c16, int cw2, const ConvLaunch& L, size_t lds, hipStream_t st);
int32_t conv_m32p_dispatch_f16(int stride, int mr, int nr, int c16, int cw2, const ConvLaunch& L, size_t lds, hipStream_t st);

// kernel variants built (keep in step with m32_dispatch / m32p_dispatch): occ = resident workgroups
// per CU of the single-role kernel (1, 2), or 3 = the producer/consumer kernel (512 threads, one per CU);
// nb16 > 0: the producer/consumer kernel with 16x16x32 consumers and nb16 columns of 16 pixels per consumer wave
// (conv_m32p_kernel.h, C16) -- the ONLY family of the layers conv_m16_eligible() names, whatever the batch, so that a
// frame's sums are formed in one order; nr then only sizes the tile (nb16 * 16 = nr * 32 pixels)
struct M32Variant { int mr, wm, nr, occ, nb16; };
constexpr int kM16Cw2Default = 0;   // product default of SCPOSE_M16_CW2 (see m32p_two_consumer_waves)
static const M32Variant kVariants[] = {
  {3, 1, 3, 1, 0}, {3, 1, 2, 2, 0}, {3, 1, 1, 3, 0}, {3, 1, 2, 3, 0}, {3, 1, 3, 3, 0},
  {2, 1, 2, 1, 0}, {2, 1, 4, 1, 0}, {2, 1, 3, 2, 0}, {2, 1, 1, 3, 0}, {2, 1, 2, 3, 0}, {2, 1, 3, 3, 0},
  {3, 1, 1, 3, 2}, {3, 1, 2, 3, 4}, {3, 1, 3, 3, 5}, {3, 1, 3, 3, 6},
  {kMrep48, 1, 4, 3, 8},   // 48-row blocks, eight columns per consumer wave (512-pixel tile groups)
};

// 16x16x32 consumers: stride-1 3x3 layers with 96-row Cout blocks whose input is a whole number of 32-channel pairs of plane
// pairs (the consumers process 2-plane groups in pairs) and has at least the three 2-plane chunks the producer/consumer
// schedule needs: 96 -> 96, 192 -> 192, 384 -> 384 of HRNet-W48.  SCPOSE_M16=0 (development) keeps them on 32x32x16.
static bool conv_m16_eligible(const PackedConv& pc) {
  static const char* e = dev_env("SCPOSE_M16");
  if (pc.mrep == kMrep48) return true;   // (conv_m32_choose checked the layer; this family has no 32x32x16 form)
  return pc.ks == 3 && pc.stride == 1 && pc.mrep == 3 && pc.wm == 1 && pc.cp == 2 && (pc.cin / 16) % 2 == 0 && pc.cin / 16 >= 3 && !(e && atoi(e) == 0);
}

// Two consumer waves per SIMD (768-thread workgroups): built for 6 and 4 columns per tile group (3 / 2 per consumer wave).
// SCPOSE_M16_CW2 (development): 0 = never, 1 = every eligible layer, 2 = only layers with more than one Cout block (192 / 384 channels)
static int m32p_two_consumer_waves(const PackedConv& pc, int nb16, int cp) {
  static const char* e = dev_env("SCPOSE_M16_CW2");
  const int mode = e ? atoi(e) : kM16Cw2Default;
  if (mode == 0 || pc.mrep != 3 || pc.stride != 1 || !(nb16 == 6 || nb16 == 4)) return 0;
  (void)cp;
  return mode == 1 || pc.n_mblk > 1;
}

bool conv_m32_choose(int cin, int cout, int ks, int stride, int* mr, int* wm, int* cp) {
  static const char* e = dev_env("SCPOSE_M32");
  if (e && atoi(e) == 0) return false;
  static const char* e2 = dev_env("SCPOSE_M32_S2");
  if (ks != 3 || cin % 16 != 0) return false;
  // stride 2 (fuse down paths, transition): producer/consumer kernel only, which needs >= 3 K-chunks;
  // measured 1.4-1.7x faster than the 16x16x32 kernel on these input-heavy layers (SCPOSE_M32_S2=0 disables)
  if (stride != 1 && !(stride == 2 && (cout % 96 == 0 || cout % 64 == 0 || cout == 48) && cin / 16 >= 3 && !(e2 && atoi(e2) == 0))) return false;
  int m = 0, w = 1;
  if (cout % 96 == 0) m = 3;          // Cout blocks of 96 (HRNet-W48: 96, 192, 384)
  else if (cout % 64 == 0) m = 2;     // Cout blocks of 64 (HRNet-W32: 64, 128, 256; layer1)
  else if (cout == 48 && (stride == 2 || cin >= 96)) {
    // stride 2 (input-bound): one 64-row block with 16 rows of padding.  Stride 1 with a deep K (transition1, 256 -> 48: MFMA-bound):
    // a 48-row block on the 16x16x32 consumers' 3 x 8 form when K is a whole number of 32-channel pairs of plane pairs -- the 64-row
    // form spent a quarter of its MFMAs on the padding rows (SCPOSE_M48=0, development: the 64-row form)
    static const char* e48 = dev_env("SCPOSE_M48");
    m = (stride == 1 && (cin / 16) % 2 == 0 && cin / 16 >= 4 && !(e48 && atoi(e48) == 0)) ? kMrep48 : 2;
  }
  else return false;
  *mr = m; *wm = w;
  const int planes = cin / 8;
  // whole K in one chunk with resident weights when it fits comfortably, else stream 2-plane chunks
  const int mt = m == kMrep48 ? 48 : 32 * m * w;
  const size_t whole = (size_t)(planes / 2) * 9 * 2 * mt * 16;
  // (stride 2 runs on the producer/consumer kernel only, which needs >= 3 K-chunks: always stream 2-plane chunks there --
  // a single resident chunk, e.g. 48 -> 64, would pass create and then find no tiling at the first forward)
  *cp = (m != kMrep48 && stride == 1 && cout == mt && whole <= 60 * 1024 && planes <= 6) ? planes : 2;
  return true;
}

size_t pack_conv_weights_m32(const float* w, int cout, int cin, int ks, int mt, int cp, int dtype,
                             uint16_t* dst, int* nchunks_out, int* ksteps_full_out) {
  const int planes = cin / 8, kk = ks * ks;
  const int nchunks = (planes + cp - 1) / cp;
  const int ksteps_full = (cp / 2) * kk;
  const int n_mblk = (cout + mt - 1) / mt;
  const size_t chunk_elems = (size_t)ksteps_full * 2 * mt * 8;
  const size_t total = (size_t)n_mblk * nchunks * chunk_elems;
  if (nchunks_out) *nchunks_out = nchunks;
  if (ksteps_full_out) *ksteps_full_out = ksteps_full;
  if (!dst) return total * 2;
  memset(dst, 0, total * 2);
  for (int mb = 0; mb < n_mblk; ++mb)
    for (int c = 0; c < nchunks; ++c) {
      const int plane0 = c * cp;
      const int pl = (planes - plane0) < cp ? (planes - plane0) : cp;
      uint16_t* base = dst + ((size_t)mb * nchunks + c) * chunk_elems;
      for (int pp = 0; pp < pl / 2; ++pp)
        for (int tap = 0; tap < kk; ++tap)
          for (int kg = 0; kg < 2; ++kg) {
            const int st = pp * kk + tap;
            const int plane = plane0 + 2 * pp + kg;
            for (int r = 0; r < mt; ++r) {
              const int co = mb * mt + r;   // natural channel order
              if (co >= cout) continue;
              uint16_t* d = base + ((size_t)(st * 2 + kg) * mt + r) * 8;
              for (int j = 0; j < 8; ++j)
                d[j] = host_f32_to_16(w[((size_t)co * cin + plane * 8 + j) * kk + tap], dtype);
            }
          }
    }
  return total * 2;
}

// producer/consumer kernel, K-chunks of `cp` planes (a launch-time choice: the packed image of a 2*cp-plane chunk is the
// images of its two cp-plane chunks back to back, and K is accumulated in the same order whatever the chunking):
// weights stay resident (all chunks in LDS) when the layer has one Cout block and they fit
struct M32pChunking { int cp, nchunks, ksteps_full; };
static M32pChunking m32p_chunking(const PackedConv& pc, int cp) { return {cp, (pc.cin / 8) / cp, (cp / 2) * pc.ks * pc.ks}; }
// (pxcap = pixel slots of a tile group = rows of the retire buffer: 4 waves x nr x 32, or 4 x nb16 x 16)
static int m32p_wbufs(const PackedConv& pc, const M32pChunking& ck, int plane_stride, int pxcap) {
  const size_t lds_w = (size_t)ck.ksteps_full * 2 * pc.mt * 16;
  const size_t lds_bias = (((size_t)pc.n_mblk * pc.mt * 4) + 511) & ~(size_t)511;
  const size_t rest = lds_bias + 2 * (size_t)ck.cp * plane_stride + (size_t)(pc.mt / 8) * pxcap * 16;
  return (pc.n_mblk == 1 && ck.nchunks > 2 && rest + ck.nchunks * lds_w <= 160 * 1024) ? ck.nchunks : 2;
}
static size_t m32p_lds_bytes(const PackedConv& pc, const M32pChunking& ck, int plane_stride, int pxcap) {
  const size_t lds_w = (size_t)ck.ksteps_full * 2 * pc.mt * 16;
  const size_t lds_bias = (((size_t)pc.n_mblk * pc.mt * 4) + 511) & ~(size_t)511;
  return lds_bias + m32p_wbufs(pc, ck, plane_stride, pxcap) * lds_w + 2 * (size_t)ck.cp * plane_stride + (size_t)(pc.mt / 8) * pxcap * 16;
}

static size_t m32_lds_bytes(const PackedConv& pc, int plane_stride) {
  const bool resident = pc.nchunks == 1 && pc.n_mblk == 1;
  const size_t lds_w = (size_t)pc.ksteps_full * 2 * pc.mt * 16;
  const size_t lds_bias = (((size_t)pc.n_mblk * pc.mt * 4) + 511) & ~(size_t)511;
  return 1024 + lds_bias + (resident ? 1 : 2) * lds_w + 2 * (size_t)pc.cp * plane_stride;
}

int32_t conv_launch_m32(const PackedConv& pc, ConvLaunch& L, hipStream_t stream) {
  const int wn = 4 / pc.wm;
  const int k2 = pc.ks / 2;
  static const char* occ_env = dev_env("SCPOSE_M32_OCC");
  const int occ_only = occ_env ? atoi(occ_env) : 0;
  // tile search over the built variants: maximise useful MFMA columns, prefer two workgroups per CU
  // (one computes while the other is stalled in its memory instructions), then pixels per weight chunk
  double best = -1e30, best_p = 1e30;
  bool found = false, found_p = false;
  int b_th = 0, b_tw = 0, b_nseg = 0, b_nr = 0, b_ps = 0, b_occ = 1, b_cp = pc.cp, b_nb16 = 0;
  long b_items = 0;
  int p_th = 0, p_tw = 0, p_nseg = 0, p_nr = 0, p_ps = 0, p_cp = pc.cp, p_nb16 = 0;   // best producer/consumer candidate (cost model)
  const bool m16 = conv_m16_eligible(pc);
  static const char* cus_env = dev_env("SCPOSE_M32_CUS");   // development: size small layers for a share of the chip (concurrent lanes)
  const int cus_all = conv_device_cus();
  // a share of the chip only for layers that cannot fill it anyway (<= 64 k output pixels: chains of DMA round trips, whose
  // duration hardly depends on the CU count -- W32 256x256 batch 64 captured forward 3.39 -> 3.07 ms with half the chip each)
  const int share = cus_env && atoi(cus_env) > 0 ? atoi(cus_env) : L.cu_share;
  const int cus = (share > 0 && share < cus_all && (long)L.N * L.Ho * L.Wo <= 65536) ? share : cus_all;
  const int tw_cand[8] = {L.Wo, 64, 48, 32, 24, 16, 12, 8};
  // the search below depends on (layer, N, Ho, Wo) only: its result is remembered in the layer (a forward launches the
  // same shapes every time; small batches are launch-bound on the host)
  PackedConv::TileMemo& memo = pc.m32_memo[cus != cus_all];
  const bool memo_hit = memo.n == L.N && memo.ho == L.Ho && memo.wo == L.Wo && memo.cus == cus && !dev_env("SCPOSE_M32_OCC") && !dev_env("SCPOSE_M32_NR") && !dev_env("SCPOSE_M32_CPMUL") && !dev_env("SCPOSE_M32_CUS") && !dev_env("SCPOSE_M16_NB") && !dev_env("SCPOSE_M32_TILE");
  if (memo_hit) { found = true; b_th = memo.th; b_tw = memo.tw; b_nseg = memo.nseg; b_nr = memo.nr; b_ps = memo.ps; b_occ = memo.occ; b_cp = memo.cp; b_nb16 = memo.nb16; }
  for (const M32Variant& v : kVariants) {
    if (memo_hit) break;
    static const char* nr_env = dev_env("SCPOSE_M32_NR");   // development: restrict the search to one column count
    if (v.mr != pc.mrep || v.wm != pc.wm || (occ_only && v.occ != occ_only) || (nr_env && v.nr != atoi(nr_env))) continue;
    if (m16 != (v.nb16 > 0)) continue;   // the layer's kernel family does not depend on the batch (see kVariants)
    static const char* nb_env = dev_env("SCPOSE_M16_NB");   // development: restrict the search to one column count of the 16x16x32 consumers
    if (nb_env && v.nb16 > 0 && v.nb16 != atoi(nb_env)) continue;
    const int nr = v.nr;
    const int cap = v.nb16 > 0 ? wn * v.nb16 * 16 : wn * nr * 32;
    if (v.occ == 3 && (pc.nchunks < 3 || (pc.cin / 8) % pc.cp != 0)) continue;   // retire-buffer schedule needs >= 3 equal chunks
    if (pc.stride == 2 && v.occ != 3) continue;
    const int halo_cap = v.occ == 2 ? 512 : 1024;
    const size_t lds_cap = v.occ == 2 ? 80 * 1024 : 160 * 1024;
    static const char* tile_env = dev_env("SCPOSE_M32_TILE");   // development: "th,tw" restricts the producer/consumer tile search
    int f_th = 0, f_tw = 0;
    if (tile_env) sscanf(tile_env, "%d,%d", &f_th, &f_tw);
    for (int ti = 0; ti < 8; ++ti) {
      const int tw = tw_cand[ti];
      if (tw > L.Wo || tw > cap || (ti > 0 && tw >= L.Wo)) continue;
      if (f_tw > 0 && v.occ == 3 && tw != f_tw) continue;
      for (int th = 1; th <= L.Ho && th * tw <= cap; ++th) {
        if (f_th > 0 && v.occ == 3 && th != f_th) continue;
        const int hh = (th - 1) * pc.stride + 1 + 2 * k2, hw = (tw - 1) * pc.stride + 1 + 2 * k2;
        for (int nseg = 1; nseg <= 8; ++nseg) {
          if (nseg * th * tw > cap || nseg * hh * hw > halo_cap) break;
          const int ps = (nseg * hh * hw * 16 + 255) & ~255;
          if (v.occ != 3 && m32_lds_bytes(pc, ps) > lds_cap) break;
          const int tx = (L.Wo + tw - 1) / tw, ty = (L.Ho + th - 1) / th;
          const double eff = (double)L.Ho * L.Wo / ((double)tx * ty / nseg * cap);
          // measured preference: producer/consumer for 96-row blocks, two workgroups per CU for 64-row blocks
          const double pref = v.occ == 3 ? (v.mr == 3 ? 1.5 : 1.1) : v.occ == 2 ? 1.25 : 1.0;
          const long items = (((long)L.N * tx * ty + nseg - 1) / nseg) * pc.n_mblk;
          if (v.occ == 3) {
            // producer/consumer candidates are ranked by a cycle model fitted to the phase stamps:
            // stage = max(consumer MFMAs, producer bytes at ~13 B/clk) + barrier, but never shorter than the round trip of
            // the stage's LDS-DMA (small tiles: the stage count, not the MFMAs, sets the time -> deeper K-chunks);
            // items are quantised per CU
            bool any = false;
            static const char* mul_env = dev_env("SCPOSE_M32_CPMUL");   // development: cap the K-chunk depth multiplier
            const int mul_max = mul_env ? atoi(mul_env) : 4;
            for (int mul = 1; mul <= mul_max; mul *= 2) {
              const int cp = pc.cp * mul;
              if ((mul > 1 && pc.cp != 2) || (pc.cin / 8) % cp != 0) continue;
              const M32pChunking ck = m32p_chunking(pc, cp);
              if (ck.nchunks < 3 || m32p_lds_bytes(pc, ck, ps, cap) > lds_cap) continue;
              any = true;
              const double per_cu = (double)((items + cus - 1) / cus);
              const double mr_eff = v.mr == kMrep48 ? 1.5 : (double)v.mr;   // 32-row units of the Cout block
              const double mfma = mr_eff * (cap / 128.0) * (cp / 2) * pc.ks * pc.ks * 32.0 * 1.35;
              const bool res_w = m32p_wbufs(pc, ck, ps, cap) > 2;
              const double bytes = (res_w ? 0.0 : (double)ck.ksteps_full * 2 * pc.mt * 16) + (double)cp * ps +
                                   2.0 * pc.mt * (nseg * th * tw) * 2 / ck.nchunks;
              double stage = (mfma > bytes / 13.0 ? mfma : bytes / 13.0) + 700.0;
              if (stage < 2600.0) stage = 2600.0;
              // 16x16x32 consumers read B fragments as 16 consecutive pixel slots: with a tile width that is a multiple of 16 a column
              // never straddles two halo rows and its ds_read_b128 lane groups meet no bank twice (96 -> 96 @48 x 48: 8 x 48 instead of
              // 16 x 24 tiles, -1.2 % measured, profiles/round4_m16_tile_choices.txt) -- a tie-break, not a term of the model
              const double cost = per_cu * (ck.nchunks * stage + mr_eff * (cap / 128.0) * 16 * 25.0) * ((v.nb16 > 0 && tw % 16 != 0) ? 1.02 : 1.0);
              if (cost < best_p) { found_p = true; best_p = cost; p_th = th; p_tw = tw; p_nseg = nseg; p_nr = nr; p_ps = ps; p_cp = cp; p_nb16 = v.nb16; }
            }
            if (!any) break;
            continue;
          }
          const double score = eff * pref / (1.0 + 2.0 / (wn * nr)) - 0.02 * (double)(hh * hw) / (th * tw);
          if (score > best) { found = true; best = score; b_th = th; b_tw = tw; b_nseg = nseg; b_nr = nr; b_ps = ps; b_occ = v.occ; b_items = items; }
        }
      }
    }
  }
  // measured preference: producer/consumer for 96-row blocks and for stride 2, two workgroups per CU for 64-row blocks --
  // unless those would leave the chip under-filled (at most one round of items: small batches, deep branches), where the
  // layer is a chain of DMA round trips and the small-tile producer/consumer candidates are 1.7-2x faster
  // (W32 batch 64: 128->128 @16x16 30.8 -> 14.9 us, 256->256 @8x8 40.1 -> 23.0 us before deeper chunks)
  if (!memo_hit && found_p && (!found || pc.mrep == 3 || pc.mrep == kMrep48 || pc.stride == 2 || pc.cout == 48 || b_items <= (long)cus * b_occ)) {
    found = true; b_th = p_th; b_tw = p_tw; b_nseg = p_nseg; b_nr = p_nr; b_ps = p_ps; b_occ = 3; b_cp = p_cp; b_nb16 = p_nb16;
  }
  SCP_REQUIRE(found, "conv m32: no tiling for %dx%d output", L.Ho, L.Wo);
  memo.nb16 = b_nb16;
  memo.n = L.N; memo.ho = L.Ho; memo.wo = L.Wo; memo.th = b_th; memo.tw = b_tw; memo.nseg = b_nseg; memo.nr = b_nr; memo.ps = b_ps; memo.occ = b_occ; memo.cp = b_cp; memo.cus = cus;
  L.th = b_th; L.tw = b_tw; L.nt = b_nseg;
  L.tiles_x = (L.Wo + L.tw - 1) / L.tw;
  L.tiles_y = (L.Ho + L.th - 1) / L.th;
  L.halo_h = (L.th - 1) * pc.stride + 1 + 2 * k2;
  L.halo_w = (L.tw - 1) * pc.stride + 1 + 2 * k2;
  L.plane_stride = b_ps;
  const M32pChunking ck = b_occ == 3 ? m32p_chunking(pc, b_cp) : M32pChunking{pc.cp, pc.nchunks, pc.ksteps_full};
  L.cp = ck.cp; L.nchunks = ck.nchunks; L.ksteps_full = ck.ksteps_full; L.n_mblk = pc.n_mblk;
  L.lds_w = ck.ksteps_full * 2 * pc.mt * 16;
  L.lds_x = ck.cp * L.plane_stride;
  L.lds_bias = ((pc.n_mblk * pc.mt * 4) + 511) & ~511;
  const int pxcap = b_nb16 > 0 ? wn * b_nb16 * 16 : wn * b_nr * 32;
  L.nbuf_w = b_occ == 3 ? m32p_wbufs(pc, ck, L.plane_stride, pxcap) : (pc.nchunks == 1 && pc.n_mblk == 1) ? 1 : 2;
  L.nbuf_x = 2;
  L.groups = 1;
  const size_t lds = b_occ == 3 ? m32p_lds_bytes(pc, ck, L.plane_stride, pxcap) : m32_lds_bytes(pc, L.plane_stride);
  L.zero16 = conv_zero_page();
  SCP_REQUIRE(L.zero16, "conv: cannot allocate the zero page");
  L.tiles_total = L.N * L.tiles_x * L.tiles_y;
  { static const char* e = dev_env("SCPOSE_NST"); const int v = e ? atoi(e) : 0;   // producer/consumer kernel: stages that store the previous tile
    L.total_blocks = (b_occ == 3 && v > 0 && v <= pc.nchunks - 2) ? v : 0; }
  L.items_total = ((L.tiles_total + L.nt - 1) / L.nt) * pc.n_mblk;
  { static const char* e = dev_env("SCPOSE_DBG"); L.dbg = e ? atoi(e) : 0; }
  if (!kDevBuild) L.dbg &= 32;   // shipped library: only the host-side "print the tile choice" bit means anything (common.h: SCP_DBG)
  {   // producer/consumer kernel: buffer-addressed global traffic when both tensors fit a 32-bit descriptor
    static const char* e = dev_env("SCPOSE_M32_BUF");
    const size_t ib = (size_t)L.N * L.cin_planes * L.H * L.W * 16, ob = (size_t)L.N * ((pc.cout + 7) / 8) * L.Ho * L.Wo * 16;
    const bool fits = ib < 0xfffffff0ull && ob < 0xfffffff0ull && !(e && atoi(e) == 0);
    L.in_bytes = fits ? (uint32_t)ib : 0;
    L.out_bytes = fits ? (uint32_t)ob : 0;
  }
  L.dbg_buf = nullptr;
  if (L.dbg & 8) L.dbg_buf = conv_dbg_buffer(stream);
  if (L.dbg & 32)
    fprintf(stderr, "m32 %d->%d %dx%d N=%d: mr=%d nr=%d nb16=%d occ=%d cp=%d tile %dx%d nseg=%d halo %dx%d lds=%zu items=%d\n", pc.cin, pc.cout, L.Ho, L.Wo, L.N,
            pc.mrep, b_nr, b_nb16, b_occ, L.cp, L.th, L.tw, L.nt, L.halo_h, L.halo_w, lds, L.items_total);
  L.fd_npix = make_fastdiv(L.th * L.tw); L.fd_tw = make_fastdiv(L.tw);
  L.fd_hp = make_fastdiv(L.halo_h * L.halo_w); L.fd_halo_w = make_fastdiv(L.halo_w);
  L.fd_tiles_img = make_fastdiv(L.tiles_x * L.tiles_y); L.fd_tiles_x = make_fastdiv(L.tiles_x);
  L.fd_nmblk = make_fastdiv(pc.n_mblk);
  int grid = cus * (b_occ == 2 ? 2 : 1);
  if (grid > L.items_total) grid = L.items_total;
  L.items_per_wg = (L.items_total + grid - 1) / grid;
  L.grid = (L.items_total + L.items_per_wg - 1) / L.items_per_wg;
  conv_dbg_set_grid(L.grid);
  if (b_occ == 3) {
    // L.groups doubles as "K-chunks of weights held in producer registers" for the producer/consumer kernel
    static const char* wr_env = dev_env("SCPOSE_M32_WREG");
    // two consumer waves per SIMD (conv_m32p_kernel.h, CW2; round 6): a property of the LAYER (its Cout-block count), not of the batch, so
    // that a frame's sums are formed in one order whatever the batch -- the consumers' MFMA order per pixel is the same in both forms anyway
    const int cw2 = m32p_two_consumer_waves(pc, b_nb16, ck.cp);
    L.groups = (!cw2 && pc.n_mblk == 1 && ck.nchunks == 6 && ck.cp == 2 && pc.stride == 1 && pc.mrep == 3 && (b_nb16 > 0 ? b_nb16 == 6 : b_nr == 3) && !(wr_env && atoi(wr_env) == 0)) ? 6 : 1;
    // b_nb16 > 0: 16x16x32 consumers (conv_m32p_kernel.h, C16; round 4) -- see conv_m16_eligible
    if (pc.dtype == SCPOSE_DT_BF16) return conv_m32p_dispatch_bf16(pc.stride, pc.mrep, b_nr, b_nb16, cw2, L, lds, stream);
    return conv_m32p_dispatch_f16(pc.stride, pc.mrep, b_nr, b_nb16, cw2, L, lds, stream);
  }
  if (pc.dtype == SCPOSE_DT_BF16) return conv_m32_dispatch_bf16(pc.mrep, pc.wm, b_nr, b_occ, L, lds, stream);
  return conv_m32_dispatch_f16(pc.mrep, pc.wm, b_nr, b_occ, L, lds, stream);
}

}  // namespace scpose
