// Stem conv1: 3 -> 64 channels, 3x3, stride 2, pad 1, folded BN + ReLU, f32 arithmetic.
//
// Replaces conv1/bn1/relu of the reference forward
// (landmark_regression/lib/models/pose_hrnet.py:282-284, :426-428) and, for the uint8 input
// format, the ToTensor()+Normalize(mean,std) of landmark_regression/tools/test.py:106-114
// (x/255 - mean)/std fused into the load.
//
// K = 27 is too thin for MFMA; the layer is 0.13 % of the network's FLOPs and is bound by
// its 64-channel output write, so it runs on the f32 VALU: one thread per output pixel, all
// 64 output channels (two per packed-f32 FMA), weights through the scalar cache (wave-uniform addresses -> s_load).
// Output is the blocked [N][8][H/2][W/2][8] 16-bit tensor the MFMA convolutions consume.
#include "common.h"

namespace scpose {

template <int DT> struct StemDt { typedef __bf16 type; };
template <> struct StemDt<1> { typedef _Float16 type; };

template <typename T> __device__ __forceinline__ uint16_t stem_bits(float f) {
  T t = (T)f;
  return __builtin_bit_cast(uint16_t, t);
}

template <int DT, int FMT>
__global__ __launch_bounds__(256) void stem_conv1_kernel(const void* __restrict__ in,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ mean_std,
                                                         int N, int H, int W, void* __restrict__ out) {
  typedef typename StemDt<DT>::type T;
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)N * Ho * Wo;
  // uint8 input: ToTensor + Normalize have only 3 x 256 possible results; every block tabulates them once with the
  // reference's operations ((u/255 - mean)/std, two IEEE divisions each) instead of dividing 54 times per thread
  __shared__ float lut[FMT == SCPOSE_IN_U8_NHWC ? 768 : 1];
  if constexpr (FMT == SCPOSE_IN_U8_NHWC) {
    for (int e = threadIdx.x; e < 768; e += 256) {
      const int c = e >> 8;
      lut[e] = ((float)(e & 255) / 255.0f - mean_std[c]) / mean_std[3 + c];
    }
    __syncthreads();
  }
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ox = (int)(gid % Wo);
  const int oy = (int)((gid / Wo) % Ho);
  const int n = (int)(gid / ((size_t)Wo * Ho));

  float x[27];  // [c][ky][kx], matching OIHW weight order
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy + ky - 1;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox + kx - 1;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        if (ok) {
          if constexpr (FMT == SCPOSE_IN_F32_NCHW) {
            v = static_cast<const float*>(in)[(((size_t)n * 3 + c) * H + iy) * W + ix];
          } else {
            // ToTensor: u/255 ; Normalize: (t - mean)/std   (same op order as torchvision), tabulated above
            v = lut[c * 256 + static_cast<const uint8_t*>(in)[(((size_t)n * H + iy) * W + ix) * 3 + c]];
          }
        }
        x[c * 9 + ky * 3 + kx] = v;
      }
    }
  }

  const size_t plane = (size_t)Ho * Wo;
  char* obase = static_cast<char*>(out) + (((size_t)n * 8) * plane + (size_t)oy * Wo + ox) * 16;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll 1
  for (int cg = 0; cg < 8; ++cg) {
    // weights of this 8-channel group as [k][8] (hrnet.cpp repacks them): channel pairs are adjacent, so one packed
    // FMA (v_pk_fma_f32, two f32 lanes per instruction) advances two output channels; per channel the 27 fused
    // multiply-adds still run in k order from the bias, i.e. the same bits as a scalar fmaf chain
    const f32x2* wg = reinterpret_cast<const f32x2*>(w + cg * (27 * 8));
    f32x2 a2[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) a2[jp] = f32x2{bias[cg * 8 + 2 * jp], bias[cg * 8 + 2 * jp + 1]};
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const f32x2 xk = f32x2{x[k], x[k]};
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) a2[jp] = __builtin_elementwise_fma(xk, wg[k * 4 + jp], a2[jp]);
    }
    float acc[8];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) { acc[2 * jp] = fmaxf(a2[jp].x, 0.f); acc[2 * jp + 1] = fmaxf(a2[jp].y, 0.f); }
    uint4 o;
    o.x = (uint32_t)stem_bits<T>(acc[0]) | ((uint32_t)stem_bits<T>(acc[1]) << 16);
    o.y = (uint32_t)stem_bits<T>(acc[2]) | ((uint32_t)stem_bits<T>(acc[3]) << 16);
    o.z = (uint32_t)stem_bits<T>(acc[4]) | ((uint32_t)stem_bits<T>(acc[5]) << 16);
    o.w = (uint32_t)stem_bits<T>(acc[6]) | ((uint32_t)stem_bits<T>(acc[7]) << 16);
    *reinterpret_cast<uint4*>(obase + (size_t)cg * plane * 16) = o;
  }
}

int32_t stem_launch(const void* in, int in_fmt, const float* w_folded, const float* bias,
                    const float* mean_std, int N, int H, int W, int dtype, void* out,
                    hipStream_t stream) {
  SCP_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H=%d W=%d must be even", H, W);
  SCP_REQUIRE(in_fmt == SCPOSE_IN_F32_NCHW || in_fmt == SCPOSE_IN_U8_NHWC, "stem: input format %d", in_fmt);
  SCP_REQUIRE(in_fmt == SCPOSE_IN_F32_NCHW || mean_std, "stem: u8 input needs mean/std");
  const size_t total = (size_t)N * (H / 2) * (W / 2);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  const bool bf = dtype == SCPOSE_DT_BF16;
  if (in_fmt == SCPOSE_IN_F32_NCHW) {
    if (bf) hipLaunchKernelGGL((stem_conv1_kernel<0, SCPOSE_IN_F32_NCHW>), grid, block, 0, stream, in, w_folded, bias, mean_std, N, H, W, out);
    else hipLaunchKernelGGL((stem_conv1_kernel<1, SCPOSE_IN_F32_NCHW>), grid, block, 0, stream, in, w_folded, bias, mean_std, N, H, W, out);
  } else {
    if (bf) hipLaunchKernelGGL((stem_conv1_kernel<0, SCPOSE_IN_U8_NHWC>), grid, block, 0, stream, in, w_folded, bias, mean_std, N, H, W, out);
    else hipLaunchKernelGGL((stem_conv1_kernel<1, SCPOSE_IN_U8_NHWC>), grid, block, 0, stream, in, w_folded, bias, mean_std, N, H, W, out);
  }
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
