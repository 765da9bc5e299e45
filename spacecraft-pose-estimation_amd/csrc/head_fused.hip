// Fused tail of pose_hrnet: the last fuse sum, final_layer and (optionally) the heat-map decode in ONE pass over the
// branch-0 tensor.
//
//   y0   = ReLU( x0 + up2(c1) + up4(c2) + up8(c3) )   HighResolutionModule.forward of the last stage-4 module, whose fuse
//                                                      layer has one row (landmark_regression/lib/models/pose_hrnet.py:256-263,
//                                                      multi_scale_output=False :411-414); c_j = the 1x1 conv + BN outputs
//                                                      at 1/2^j resolution, read with (y >> j, x >> j) as fuse_sum does
//   hm   = final_layer(y0)                             1x1 convolution C -> J with bias (:323-329, :458)
//   pred = get_final_preds(hm)                         arg-max, quarter-pixel refinement, back-transform
//                                                      (lib/core/inference.py:18-79, lib/utils/transforms.py:49-110)
//
// Why: as three launches the tail moves y0 twice (fuse_sum writes 226 MB at batch 256 / 96x96, final_layer reads them:
// 110 + 70 us) and, when only the key points are wanted, writes and re-reads 104 MB of heat-maps (decode: 23 us on the
// side stream).  Here y0 exists only in registers: a lane sums the terms of ONE 16-byte (pixel, 8-channel) vector in
// fp32 in the reference's j order, applies ReLU, rounds to 16 bits -- the same rounding fuse_sum applies when it stores
// y0 -- and that register IS the B fragment of v_mfma_f32_16x16x32 for its pixel and k-group (k-step s, k-group q =
// channel plane 4 s + q); the A fragments are final_layer's weights (16 rows = joints, zero-padded).  Two MFMAs give a
// wave the J heat-map values of 16 pixels.  With TRACK every lane also keeps the running maximum (value, first index)
// of its four joints; a second, tiny kernel (one wave per frame) combines the partial maxima, recomputes the four
// neighbours of every maximum with the SAME instruction sequence (a heat-map value depends on nothing but its pixel, so
// the recomputed values are bit-identical to what the first kernel wrote or would have written) and finishes the decode
// with decode.hip's own code (decode_device.h).  Results: heat-maps and key points are bit-identical between the two
// entry points (scpose_hrnet_forward + scpose_decode vs scpose_hrnet_forward_decode).
#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // make_buf, load16_buf, BUF_OOB, pipe_fdiv, u32x4
#include "decode_device.h"

namespace scpose {

namespace {

constexpr int kMaxStrips = 16;

struct HeadFusedArgs {
  const void* term[4];
  uint32_t term_bytes[4];
  int32_t shift[4];
  const void* wfrag;     // [2 k-steps][4 k-groups][16 rows][8] 16-bit: row = joint, k-group q of k-step s = channel plane 4 s + q
  const float* bias;     // [16]
  int32_t N, planes, H, W, J;
  int32_t cols_per_img, cols_per_wg, strips;
  FastDiv fd_w;
  float* hm;             // N x J x H x W, or null
  float* part_v;         // [N][strips][16] partial maxima (TRACK), or null
  int32_t* part_i;
  // second kernel
  const float* center;
  const float* scale;
  float* preds;          // N x J x 3
  int32_t post_process;
};

// B fragment of pixel (y, x) of frame n for k-step s: ReLU(sum of the terms) of channel plane 4 s + q, rounded to 16 bits
template <typename T, int NT>
struct TermLoader {
  // base[s][k]: 16-byte slot of (frame n, plane 4 s + q, row 0, column 0) in term k, or -1 when the plane does not exist
  // (K is padded to 64 channels): loop-invariant per lane.  Map dimensions are < 2^24, so the row offset is a 24-bit multiply.
  int base[2][NT];
  __device__ __forceinline__ void init(const HeadFusedArgs& a, int n, int q) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int plane = 4 * s + q, sh = a.shift[k];
        base[s][k] = plane < a.planes ? (n * a.planes + plane) * (a.H >> sh) * (a.W >> sh) : -1;
      }
  }
  __device__ __forceinline__ void request(const HeadFusedArgs& a, int y, int x, bool valid, u32x4 (*v)[NT]) const {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int sh = a.shift[k];
        const uint32_t slot = (uint32_t)base[s][k] + __umul24((uint32_t)(y >> sh), (uint32_t)(a.W >> sh)) + (uint32_t)(x >> sh);
        const uint32_t off = (valid && base[s][k] >= 0) ? slot * 16u : BUF_OOB;          // off: reads zeros
        v[s][k] = load16_buf(make_buf(a.term[k], a.term_bytes[k]), off, 0u);             // descriptor: loop-invariant scalar work
      }
  }
  static __device__ __forceinline__ typename FragOf<T>::type fragment(const u32x4* v) {
    f32x2 s[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < NT; ++k) {     // the reference's j order, fp32 (elementwise.hip: fuse_sum); pairs: one v_pk_add_f32
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 t = {from_bits<T>((uint16_t)(v[k][e] & 0xffff)), from_bits<T>((uint16_t)(v[k][e] >> 16))};
        s[e] += t;
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(fmaxf(s[e][0], 0.f), fmaxf(s[e][1], 0.f));
    return __builtin_bit_cast(typename FragOf<T>::type, o);
  }
};

}  // namespace

template <int DT, int NT, bool TRACK>
__global__ __launch_bounds__(256) void head_fused_kernel(const HeadFusedArgs a) {
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  __shared__ float s_v[4][16];
  __shared__ int s_i[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int n = blockIdx.x / a.strips, strip = blockIdx.x - n * a.strips;
  const int HW = a.H * a.W;
  frag_t wa[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) wa[s] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(a.wfrag) + ((s * 4 + q) * 16 + r) * 16);
  const float4 b4 = *reinterpret_cast<const float4*>(a.bias + 4 * q);
  TermLoader<T, NT> L;
  L.init(a, n, q);
  // running maxima of this lane's four joints.  A lane visits its pixels in increasing index order, so "first occurrence wins"
  // is a strict comparison, and "NaN counts as the maximum, the first NaN wins" is !(v <= best) while best is not NaN: five
  // instructions per value instead of the general better().  A lane all of whose values are -inf never updates: it gets
  // its first pixel's index at the end.
  float bv[4];
  int bi[4];
  int first_pix = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < 4; ++i) { bv[i] = -__builtin_inff(); bi[i] = 0x7fffffff; }
  // heat-map stores through a buffer descriptor: rows past J, pixels past the map and "no heat-map buffer" (0 records) are
  // all out-of-range offsets the hardware drops -- no branches around the four stores of a column
  const buf_rsrc_t rs_hm = make_buf(a.hm, a.hm ? (uint32_t)((size_t)a.N * a.J * HW * 4) : 0u);
  uint32_t hm_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) hm_base[i] = (uint32_t)((n * a.J + min(4 * q + i, a.J - 1)) * HW);
  const int jrows = a.J - 4 * q;     // rows i < jrows of this lane are joints

  const int c_begin = strip * a.cols_per_wg;
  const int c_end = min(a.cols_per_img, c_begin + a.cols_per_wg);
  constexpr int U = 2;   // columns in flight per wave: 2 x 2 k-steps x NT 16-byte loads per lane
  // pixel of column u: advances by 4 * U columns = 64 * U pixels per iteration; (y, x) follow without a division
  constexpr int STEP = 64 * U;
  const int dy = pipe_fdiv(STEP, a.fd_w), dx = STEP - dy * a.W;
  int pix[U], py[U], px[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    pix[u] = (c_begin + wave + 4 * u) * 16 + r;
    py[u] = pipe_fdiv(pix[u], a.fd_w); px[u] = pix[u] - py[u] * a.W;
  }
  for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * U) {
    u32x4 v[U][2][NT];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = c0 + 4 * u < c_end && pix[u] < HW;
      L.request(a, py[u], px[u], ok[u], v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 acc = {b4.x, b4.y, b4.z, b4.w};
      acc = mfma16<T>(wa[0], L.fragment(v[u][0]), acc);
      acc = mfma16<T>(wa[1], L.fragment(v[u][1]), acc);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        store4_buf(rs_hm, (ok[u] && i < jrows) ? (hm_base[i] + (uint32_t)pix[u]) * 4u : BUF_OOB, __float_as_uint(acc[i]));
      if (TRACK && ok[u]) {
        first_pix = min(first_pix, pix[u]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool up = !(acc[i] <= bv[i]) && bv[i] == bv[i];
          bv[i] = up ? acc[i] : bv[i];
          bi[i] = up ? pix[u] : bi[i];
        }
      }
      pix[u] += STEP; px[u] += dx; py[u] += dy;
      if (px[u] >= a.W) { px[u] -= a.W; py[u] += 1; }
    }
  }
  if constexpr (TRACK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (bi[i] == 0x7fffffff) bi[i] = first_pix;   // nothing but -inf seen (or no pixel at all: stays 0x7fffffff)
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {   // the 16 lanes that share k-group q hold the same four joints
        const float ov = __shfl_xor(bv[i], off, 64);
        const int oi = __shfl_xor(bi[i], off, 64);
        if (better(ov, oi, bv[i], bi[i])) { bv[i] = ov; bi[i] = oi; }
      }
      if (r == 0) { s_v[wave][4 * q + i] = bv[i]; s_i[wave][4 * q + i] = bi[i]; }
    }
    __syncthreads();
    if (tid < 16) {
      float v = s_v[0][tid];
      int ix = s_i[0][tid];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (better(s_v[w][tid], s_i[w][tid], v, ix)) { v = s_v[w][tid]; ix = s_i[w][tid]; }
      a.part_v[(n * a.strips + strip) * 16 + tid] = v;
      a.part_i[(n * a.strips + strip) * 16 + tid] = ix;
    }
  }
}

// One wave per frame: maxima of the J maps from the strips' partial maxima, the four neighbours of each recomputed,
// quarter-pixel refinement and back-transform (decode_device.h).
template <int DT, int NT>
__global__ __launch_bounds__(64) void head_decode_kernel(const HeadFusedArgs a) {
  typedef typename DtOf<DT>::type T;
  typedef typename FragOf<T>::type frag_t;
  __shared__ int s_y[64], s_x[64], s_ok[64];
  __shared__ float s_nb[64];
  const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
  const int n = blockIdx.x;
  // lanes 0..15: joint = lane
  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  if (lane < 16) {
    float pv[kMaxStrips];
    int pi[kMaxStrips];
#pragma unroll
    for (int s = 0; s < kMaxStrips; ++s) {     // all loads first
      const bool on = s < a.strips;
      pv[s] = on ? a.part_v[(n * a.strips + s) * 16 + lane] : -__builtin_inff();
      pi[s] = on ? a.part_i[(n * a.strips + s) * 16 + lane] : 0x7fffffff;
    }
#pragma unroll
    for (int s = 0; s < kMaxStrips; ++s)
      if (better(pv[s], pi[s], bv, bi)) { bv = pv[s]; bi = pi[s]; }
  }
  float cx = 0.f, cy = 0.f;
  int px = 0, py = 0;
  bool refine = false;
  if (lane < a.J) {
    decode_coords(bv, bi, a.W, cx, cy);
    refine = a.post_process && decode_refines(cx, cy, a.H, a.W, px, py);
  }
  // entry e = 4 * joint + {0: right, 1: left, 2: below, 3: above}
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = 4 * lane + k;
    if (lane < 16) {
      s_y[e] = py + (k == 2 ? 1 : k == 3 ? -1 : 0);
      s_x[e] = px + (k == 0 ? 1 : k == 1 ? -1 : 0);
      s_ok[e] = refine ? 1 : 0;
    }
  }
  __syncthreads();
  frag_t wa[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) wa[s] = *reinterpret_cast<const frag_t*>(static_cast<const char*>(a.wfrag) + ((s * 4 + q) * 16 + r) * 16);
  const float4 b4 = *reinterpret_cast<const float4*>(a.bias + 4 * q);
  TermLoader<T, NT> L;
  L.init(a, n, q);
  u32x4 v[4][2][NT];
#pragma unroll
  for (int c = 0; c < 4; ++c) {     // column c = entries 16 c .. 16 c + 15 = joints 4 c .. 4 c + 3; every load before the first use
    const int e = 16 * c + r;
    L.request(a, s_y[e], s_x[e], s_ok[e] != 0, v[c]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = 16 * c + r;
    f32x4 acc = {b4.x, b4.y, b4.z, b4.w};
    acc = mfma16<T>(wa[0], L.fragment(v[c][0]), acc);
    acc = mfma16<T>(wa[1], L.fragment(v[c][1]), acc);
    // entry e belongs to joint 4 c + (r >> 2): row (r >> 2) of the rows 4 c .. 4 c + 3, which k-group q == c holds
    if (q == c) {
      const int i = r >> 2;
      s_nb[e] = i == 0 ? acc[0] : i == 1 ? acc[1] : i == 2 ? acc[2] : acc[3];
    }
  }
  __syncthreads();
  if (lane < a.J) {
    if (refine) decode_refine(s_nb[4 * lane], s_nb[4 * lane + 1], s_nb[4 * lane + 2], s_nb[4 * lane + 3], cx, cy);
    decode_to_image(cx, cy, bv, a.H, a.W, a.center + n * 2, a.scale + n * 2, a.preds + (size_t)(n * a.J + lane) * 3);
  }
}

// Host side ---------------------------------------------------------------------------------------------------------
void head_fused_pack(const float* w, const float* b, int J, int C, int dtype, uint16_t* wfrag /* 2*4*16*8 */, float* bias /* 16 */) {
  for (int s = 0; s < 2; ++s)
    for (int q = 0; q < 4; ++q)
      for (int r = 0; r < 16; ++r)
        for (int e = 0; e < 8; ++e) {
          const int ch = 8 * (4 * s + q) + e;
          wfrag[((s * 4 + q) * 16 + r) * 8 + e] = host_f32_to_16((r < J && ch < C) ? w[(size_t)r * C + ch] : 0.f, dtype);
        }
  for (int r = 0; r < 16; ++r) bias[r] = (r < J && b) ? b[r] : 0.f;
}

size_t head_fused_part_bytes(int n) { return (size_t)n * kMaxStrips * 16 * 4; }   // each of part_v / part_i

bool head_fused_supported(int nterms, int C, int J, int N, int H, int W) {
  if (nterms < 1 || nterms > 4 || C % 8 != 0 || C > 64 || J < 1 || J > 16) return false;
  return (size_t)N * (C / 8) * H * W * 16 < 0xfffffff0ull && (size_t)N * J * H * W * 4 < 0xfffffff0ull && (size_t)H * W < (1u << 24);   // 32-bit buffer offsets; float32 pixel indices
}

template <int DT, int NT>
static void launch_nt(const HeadFusedArgs& a, bool track, hipStream_t st) {
  if (track) hipLaunchKernelGGL((head_fused_kernel<DT, NT, true>), dim3(a.N * a.strips), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((head_fused_kernel<DT, NT, false>), dim3(a.N * a.strips), dim3(256), 0, st, a);
  if (track) hipLaunchKernelGGL((head_decode_kernel<DT, NT>), dim3(a.N), dim3(64), 0, st, a);
}

// terms / shifts: the fuse row (term k at 1 / 2^shift resolution); hm may be null when preds is given
int32_t head_fused_launch(const void* const* terms, const int32_t* shifts, int nterms, int N, int C, int H, int W, int J,
                          int dtype, const void* wfrag, const float* bias, float* hm, float* part_v, int32_t* part_i,
                          const float* center, const float* scale, int post_process, float* preds, hipStream_t stream) {
  SCP_REQUIRE(head_fused_supported(nterms, C, J, N, H, W), "head_fused: unsupported shape (terms %d, C %d, J %d)", nterms, C, J);
  SCP_REQUIRE(hm || preds, "head_fused: neither heat-maps nor key points requested");
  SCP_REQUIRE(!preds || (part_v && part_i && center && scale), "head_fused: decode needs center / scale and the partial-maxima scratch");
  HeadFusedArgs a{};
  for (int k = 0; k < nterms; ++k) {
    SCP_REQUIRE(shifts[k] >= 0 && (H >> shifts[k]) << shifts[k] == H && (W >> shifts[k]) << shifts[k] == W,
                "head_fused: term %d shift %d does not divide %dx%d", k, shifts[k], H, W);
    a.term[k] = terms[k]; a.shift[k] = shifts[k];
    a.term_bytes[k] = (uint32_t)((size_t)N * (C / 8) * (H >> shifts[k]) * (W >> shifts[k]) * 16);
  }
  a.wfrag = wfrag; a.bias = bias;
  a.N = N; a.planes = C / 8; a.H = H; a.W = W; a.J = J;
  a.cols_per_img = (H * W + 15) / 16;
  int strips = (1024 + N - 1) / N;
  if (strips > kMaxStrips) strips = kMaxStrips;
  if (strips > (a.cols_per_img + 7) / 8) strips = (a.cols_per_img + 7) / 8;
  if (strips < 1) strips = 1;
  a.strips = strips;
  a.cols_per_wg = (a.cols_per_img + strips - 1) / strips;
  a.fd_w = make_fastdiv((uint32_t)W);
  a.hm = hm; a.part_v = part_v; a.part_i = part_i;
  a.center = center; a.scale = scale; a.preds = preds; a.post_process = post_process;
  const bool track = preds != nullptr;
#define SCP_HF(DT) \
  switch (nterms) { case 1: launch_nt<DT, 1>(a, track, stream); break; case 2: launch_nt<DT, 2>(a, track, stream); break; \
                    case 3: launch_nt<DT, 3>(a, track, stream); break; default: launch_nt<DT, 4>(a, track, stream); break; }
  if (dtype == SCPOSE_DT_BF16) { SCP_HF(0) } else { SCP_HF(1) }
#undef SCP_HF
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
