// Streaming 1x1 convolution (+ folded BN bias, + residual, + ReLU) on the 16x16x32 MFMA.
//
// A 1x1 layer has no halo, so nothing needs staging: one 16-byte vector of the blocked activation layout
// [N][C/8][H][W][8] -- 8 consecutive channels of one pixel -- is exactly one lane's B fragment of
// v_mfma_f32_16x16x32 (lane = pixel column + 16 * k-group, k-group q of k-step s = plane 4s + q).  Each wave
// therefore owns 16*NG consecutive pixels of an image, loads their KSTEPS x NG fragments straight into registers
// (buffer-addressed: planes past Cin and pixels past the map read as zeros), and walks over the output channels 64 at
// a time with the layer's weights resident in LDS ([k-step][k-group][Cout padded to 64][8], the row order of
// conv_igemm.hip: conv_row_channel).  The epilogue pairs two pixel groups with v_permlane32_swap so that every lane
// ends up with the 8 channels of one (plane, pixel) and stores 16 bytes; residual vectors are fetched the same way.
// No barriers after the prologue, no LDS traffic besides the weight fragments: the kernel is a pure HBM stream with
// as many waves in flight as registers allow.  Replaces conv_pipe for the 1x1 layers of layer1 (Bottleneck conv1 /
// conv3, pose_hrnet.py:78-98), of the fuse up-paths (:197-208) and the tap maps of the hrnet_cms heads, whenever the
// packed weights fit LDS (two workgroups per CU up to 64 KB, one above: 384 -> 192 / 96).
#include <type_traits>

#include "common.h"
#include "conv_device.h"
#include "conv_pipe_kernel.h"   // buffer helpers, u32x4

namespace scpose {

struct Conv1Launch {
  const void* in; const void* w; const float* bias; const void* res; void* out;
  const void* in2;          // K-concatenated layer: planes >= split_planes come from this tensor (else null)
  int32_t split_planes;     // multiple of 4: a k-step never straddles the two tensors
  uint32_t in2_bytes;
  uint32_t in_bytes, out_bytes, w_bytes;
  int32_t N, HW, cin_planes, cout_planes, cout_pad, relu;
  int32_t blocks_per_img, total_blocks;
};

typedef __attribute__((ext_vector_type(8))) __bf16 c1_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 c1_f16x8;
typedef __attribute__((ext_vector_type(4))) float c1_f32x4;

template <typename T> struct C1Frag { typedef c1_bf16x8 type; };
template <> struct C1Frag<_Float16> { typedef c1_f16x8 type; };

template <typename T>
__device__ __forceinline__ c1_f32x4 c1_mfma(u32x4 a, u32x4 b, c1_f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename C1Frag<T>::type F;
  if constexpr (std::is_same<T, __bf16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(F, a), __builtin_bit_cast(F, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(F, a), __builtin_bit_cast(F, b), c, 0, 0, 0);
#else
  return c;
#endif
}

// load16_buf: conv_pipe_kernel.h

// OCC = workgroups (4 waves each) per CU the register budget is cut for: more waves in flight for the long-K layers,
// whose fragments are the bulk of the registers (KSTEPS * NG * 4 per lane)
template <typename T, int KSTEPS, int NG, int OCC>
__global__ __launch_bounds__(256, OCC) void conv1x1_stream_kernel(const Conv1Launch p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_l = reinterpret_cast<float*>(smem + p.w_bytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4, psel = q & 1, upper = lane >> 5;
  const buf_rsrc_t rs_in2 = make_buf(p.in2 ? p.in2 : p.in, p.in2 ? p.in2_bytes : p.in_bytes);
  const buf_rsrc_t rs_w = make_buf(p.w, p.w_bytes), rs_in = make_buf(p.in, p.in_bytes),
                   rs_res = make_buf(p.res ? p.res : p.out, p.out_bytes), rs_out = make_buf(p.out, p.out_bytes);
  for (uint32_t o = 0; o < p.w_bytes; o += 4096)
    if (o + (uint32_t)tid * 16u < p.w_bytes) dma16_buf(rs_w, (uint32_t)tid * 16u, o, smem + o + wave * 1024);
  for (int i = tid; i < p.cout_pad; i += 256) bias_l[i] = p.bias[i];
  __syncthreads();   // vmcnt(0) + barrier: weights and bias are in LDS

  constexpr int PXW = 16 * NG;
  const int HW = p.HW;
  for (int blk = blockIdx.x * 4 + wave; blk < p.total_blocks; blk += gridDim.x * 4) {
    const int img = blk / p.blocks_per_img, p0 = (blk - img * p.blocks_per_img) * PXW;
    // ---- B fragments of this wave's pixels: KSTEPS x NG 16-byte vectors per lane ----
    u32x4 b[KSTEPS][NG];
    const int planes1 = p.in2 ? p.split_planes : p.cin_planes;           // planes held by the first tensor
    const uint32_t in_img = (uint32_t)(img * planes1 * HW) * 16u;
    const uint32_t in2_img = (uint32_t)(img * (p.cin_planes - planes1) * HW) * 16u;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int plane = 4 * s + q;
      const bool second = p.in2 && 4 * s >= p.split_planes;              // wave-uniform
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int pix = p0 + g * 16 + r;
        const bool ok = plane < p.cin_planes && pix < HW;
        if (second) b[s][g] = load16_buf(rs_in2, ok ? (uint32_t)((plane - p.split_planes) * HW + pix) * 16u : BUF_OOB, in2_img);
        else b[s][g] = load16_buf(rs_in, ok ? (uint32_t)(plane * HW + pix) * 16u : BUF_OOB, in_img);
      }
    }
    // pixel this lane stores for pair gp: group 2*gp + upper, column r
    const uint32_t out_img = (uint32_t)(img * p.cout_planes * HW) * 16u;
    for (int pass = 0; pass < p.cout_pad / 64; ++pass) {
      // residual vectors of this pass (issued before the MFMAs, consumed after them)
      u32x4 rv[4][NG / 2];
      uint32_t ovo[4][NG / 2];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int plane = pass * 8 + 2 * m + psel;
#pragma unroll
        for (int gp = 0; gp < NG / 2; ++gp) {
          const int pix = p0 + (2 * gp + upper) * 16 + r;
          ovo[m][gp] = (plane < p.cout_planes && pix < HW) ? (uint32_t)(plane * HW + pix) * 16u : BUF_OOB;
          rv[m][gp] = p.res ? load16_buf(rs_res, ovo[m][gp], out_img) : u32x4{0u, 0u, 0u, 0u};
        }
      }
      c1_f32x4 acc[4][NG];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const c1_f32x4 bs = *reinterpret_cast<const c1_f32x4*>(bias_l + pass * 64 + m * 16 + 4 * q);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[m][g] = bs;
      }
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(smem + ((size_t)((s * 4 + q) * p.cout_pad + pass * 64 + m * 16 + r)) * 16);
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[m][g] = c1_mfma<T>(a, b[s][g], acc[m][g]);
        }
      // All MFMAs of the pass retire before the epilogue touches an accumulator.  hipcc's own wait states in front of
      // v_permlane32_swap were too short for the 8-pass 16x16x32 MFMA (lanes 12-15 of the last-written registers came
      // out stale, run-to-run different): 20 explicit wait states, with every accumulator as an operand so that
      // neither the MFMAs nor the epilogue can be moved across.
      if constexpr (NG == 4)
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                       "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),
                       "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
      else
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                       "+v"(acc[3][0]), "+v"(acc[3][1]));
      // ---- epilogue: pair pixel groups (2gp, 2gp+1): lower half-wave keeps group 2gp, upper half-wave group 2gp+1 ----
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int gp = 0; gp < NG / 2; ++gp) {
          uint32_t lo[4], hi[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { lo[j] = __float_as_uint(acc[m][2 * gp][j]); hi[j] = __float_as_uint(acc[m][2 * gp + 1][j]); }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(lo[j], hi[j], false, false);
            lo[j] = sw[0]; hi[j] = sw[1];
          }
#endif
          // now lo = channels 0-3, hi = channels 4-7 of (plane 2m + psel of this pass, this lane's pixel)
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = __uint_as_float(lo[j]); v[4 + j] = __uint_as_float(hi[j]); }
          const u32x4 rr = rv[m][gp];
          v[0] += from_bits<T>(rr[0] & 0xffff); v[1] += from_bits<T>(rr[0] >> 16);
          v[2] += from_bits<T>(rr[1] & 0xffff); v[3] += from_bits<T>(rr[1] >> 16);
          v[4] += from_bits<T>(rr[2] & 0xffff); v[5] += from_bits<T>(rr[2] >> 16);
          v[6] += from_bits<T>(rr[3] & 0xffff); v[7] += from_bits<T>(rr[3] >> 16);
          const uint32_t floor2 = p.relu ? 0u : 0x80008000u;
          u32x4 ov;
          ov[0] = relu2_16(pack2<T>(v[0], v[1]), floor2); ov[1] = relu2_16(pack2<T>(v[2], v[3]), floor2);
          ov[2] = relu2_16(pack2<T>(v[4], v[5]), floor2); ov[3] = relu2_16(pack2<T>(v[6], v[7]), floor2);
          store16_buf(rs_out, ovo[m][gp], out_img, ov);
        }
    }
  }
}

bool conv1x1_stream_eligible(const PackedConv& pc) { return pc.d_w1 != nullptr; }

int32_t conv1x1_stream_launch(const PackedConv& pc, const void* in, int N, int H, int W, const void* res, int relu,
                              void* out, hipStream_t stream, const void* in2, int split_planes) {
  Conv1Launch L;
  L.in2 = in2; L.split_planes = in2 ? split_planes : 0;
  L.in = in; L.w = pc.d_w1; L.bias = pc.d_b1; L.res = res; L.out = out;
  L.N = N; L.HW = H * W; L.cin_planes = pc.cin / 8; L.cout_planes = pc.cout / 8; L.cout_pad = pc.cout_pad1; L.relu = relu;
  L.in_bytes = (uint32_t)((size_t)N * (in2 ? split_planes : L.cin_planes) * L.HW * 16);
  L.in2_bytes = in2 ? (uint32_t)((size_t)N * (L.cin_planes - split_planes) * L.HW * 16) : 0;
  L.out_bytes = (uint32_t)((size_t)N * L.cout_planes * L.HW * 16);
  L.w_bytes = (uint32_t)pc.w1_bytes;
  const int ksteps = (L.cin_planes + 3) / 4;
  // register budget: fragments KSTEPS*NG*4 + accumulators 16*NG + residual vectors 8*NG per lane: 64 pixels per wave
  // up to Cin = 128, 32 pixels per wave for Cin = 192 / 256 / 384; two workgroups per CU; every variant is free of
  // scratch spills.
  const int ng = ksteps <= 4 ? 4 : 2;
  L.blocks_per_img = (L.HW + 16 * ng - 1) / (16 * ng);
  L.total_blocks = N * L.blocks_per_img;
  const size_t lds = pc.w1_bytes + (size_t)pc.cout_pad1 * 4;
  const int cus = conv_device_cus();
  int grid = (L.total_blocks + 3) / 4;
  // weights above 80 KB (384 -> 192, 384 -> 96: round 4) leave room for one workgroup per CU; its four waves then get the
  // whole register file (launch_bounds 256, 1)
  const int occ = lds > 80 * 1024 ? 1 : 2;
  if (grid > cus * occ) grid = cus * occ;
  if (grid < 1) grid = 1;
  const bool bf = pc.dtype == SCPOSE_DT_BF16;
#define C1_LAUNCH(KS, NGV, OC)                                                                                              \
  do {                                                                                                                  \
    if (bf) {                                                                                                           \
      static LdsOptIn set_b;                                                                                            \
      { const int32_t rc_ = lds_opt_in(reinterpret_cast<const void*>(conv1x1_stream_kernel<__bf16, KS, NGV, OC>), (OC) == 1 ? 160 * 1024 : 80 * 1024, &set_b); if (rc_ != SCPOSE_OK) return rc_; } \
      hipLaunchKernelGGL((conv1x1_stream_kernel<__bf16, KS, NGV, OC>), dim3(grid), dim3(256), lds, stream, L);             \
    } else {                                                                                                            \
      static LdsOptIn set_f;                                                                                            \
      { const int32_t rc_ = lds_opt_in(reinterpret_cast<const void*>(conv1x1_stream_kernel<_Float16, KS, NGV, OC>), (OC) == 1 ? 160 * 1024 : 80 * 1024, &set_f); if (rc_ != SCPOSE_OK) return rc_; } \
      hipLaunchKernelGGL((conv1x1_stream_kernel<_Float16, KS, NGV, OC>), dim3(grid), dim3(256), lds, stream, L);           \
    }                                                                                                                   \
  } while (0)
  switch (ksteps) {
    case 1: C1_LAUNCH(1, 4, 2); break;
    case 2: C1_LAUNCH(2, 4, 2); break;
    case 3: C1_LAUNCH(3, 4, 2); break;
    case 4: C1_LAUNCH(4, 4, 2); break;
    case 6: if (occ == 1) C1_LAUNCH(6, 2, 1); else C1_LAUNCH(6, 2, 2); break;
    case 8: if (occ == 1) C1_LAUNCH(8, 2, 1); else C1_LAUNCH(8, 2, 2); break;
    case 12: if (occ == 1) C1_LAUNCH(12, 2, 1); else C1_LAUNCH(12, 2, 2); break;
    default:
      set_error("conv1x1: %d k-steps unsupported", ksteps);
      return SCPOSE_E_INVALID;
  }
#undef C1_LAUNCH
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
