// Output heads of the hrnet_cms family (landmark_regression/lib/models/hrnet_cms.py:353-419, :551-557;
// hrnet_cms_384.py the same with k3 s2): per branch b
//     x_b = Conv2d(32 -> J, 1x1)(ConvTranspose2d(C_b -> 32, k, stride s, padding 1, output_padding 1)(y_b))
//           + bilinear_x2(x_{b+1})                                   (align_corners=False; no term for b = 3)
// There is no non-linearity between the two layers, so hrnet.cpp folds them into ONE transposed convolution
// C_b -> J.  A transposed convolution is a 1x1 convolution per kernel tap followed by a scatter; the 1x1
// convolutions of all k*k taps run as one MFMA convolution C_b -> k*k*CPT channels (the "tap map", channel
// = tap * CPT + joint, CPT = 8 or 16, blocked 16-bit layout like every activation), and this kernel is the
// scatter turned around into a gather: an output pixel (oy, ox) receives tap (ky, kx) of input pixel
// (iy, ix) iff oy = s*iy - 1 + ky and ox = s*ix - 1 + kx; with k <= 2s that is at most 2 x 2 taps.
// HBM-bound: one thread per output pixel reads <= 4 (8) 16-byte tap vectors and the 4 bilinear neighbours
// of the coarser level, and writes J floats (each a coalesced plane row across the wave).
#include "common.h"

namespace scpose {

struct HeadArgs {
  const void* taps;    // [N][K*K*CPT/8][H][W][8] 16-bit
  const float* bias;   // [J] folded bias
  const float* prev;   // [N][J][S*H/2][S*W/2] f32 or null
  float* out;          // [N][J][S*H][S*W] f32
  int N, J, H, W, K, S, cpt;
};

template <typename T> __device__ __forceinline__ void head_add8(float* acc, const uint4 v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[2 * i] += (float)__builtin_bit_cast(T, (uint16_t)(w[i] & 0xffff));
    acc[2 * i + 1] += (float)__builtin_bit_cast(T, (uint16_t)(w[i] >> 16));
  }
}

// torch upsample_bilinear2d, align_corners=False, scale_factor=2: src = 0.5*(dst + 0.5) - 0.5 clamped at 0
__device__ __forceinline__ void head_lerp(int dst, int in_size, int* i0, int* i1, float* l1) {
  float src = 0.5f * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  *i0 = (int)src;
  *i1 = *i0 + (*i0 < in_size - 1 ? 1 : 0);
  *l1 = src - (float)*i0;
}

template <typename T>
__global__ __launch_bounds__(256) void head_gather_kernel(const HeadArgs a) {
  const int OH = a.S * a.H, OW = a.S * a.W;
  const size_t total = (size_t)a.N * OH * OW;
  const int halves = a.cpt / 8;
  const size_t plane = (size_t)a.H * a.W;            // 16-byte vectors per tap-map plane
  for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
    const int ox = (int)(gid % OW);
    size_t t = gid / OW;
    const int oy = (int)(t % OH);
    const size_t n = t / OH;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const uint4* tp = static_cast<const uint4*>(a.taps) + n * (size_t)(a.K * a.K * halves) * plane;
    const int iy0 = (oy + 1) / a.S, ky0 = (oy + 1) % a.S;
    const int ix0 = (ox + 1) / a.S, kx0 = (ox + 1) % a.S;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int iy = iy0 - dy, ky = ky0 + dy * a.S;
      if (ky >= a.K || iy < 0 || iy >= a.H) continue;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int ix = ix0 - dx, kx = kx0 + dx * a.S;
        if (kx >= a.K || ix < 0 || ix >= a.W) continue;
        const uint4* p = tp + (size_t)((ky * a.K + kx) * halves) * plane + (size_t)iy * a.W + ix;
        head_add8<T>(acc, p[0]);
        if (halves == 2) head_add8<T>(acc + 8, p[plane]);
      }
    }
    int py0 = 0, py1 = 0, px0 = 0, px1 = 0;
    float ly = 0.f, lx = 0.f;
    const int PH = OH / 2, PW = OW / 2;
    if (a.prev) {
      head_lerp(oy, PH, &py0, &py1, &ly);
      head_lerp(ox, PW, &px0, &px1, &lx);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j >= a.J) break;
      float v = acc[j] + a.bias[j];
      if (a.prev) {
        const float* q = a.prev + (n * a.J + j) * (size_t)PH * PW;
        const float top = (1.f - lx) * q[(size_t)py0 * PW + px0] + lx * q[(size_t)py0 * PW + px1];
        const float bot = (1.f - lx) * q[(size_t)py1 * PW + px0] + lx * q[(size_t)py1 * PW + px1];
        v += (1.f - ly) * top + ly * bot;
      }
      a.out[((n * a.J + j) * OH + oy) * (size_t)OW + ox] = v;
    }
  }
}

int32_t head_gather_launch(const void* taps, const float* bias, const float* prev, int N, int J, int H, int W,
                           int K, int S, int dtype, float* out, hipStream_t stream) {
  SCP_REQUIRE(J > 0 && J <= 16, "head: NUM_JOINTS=%d (1..16)", J);
  SCP_REQUIRE((K == 5 && S == 4) || (K == 3 && S == 2), "head: kernel %d stride %d (5/4 or 3/2)", K, S);
  SCP_REQUIRE(N > 0 && H > 0 && W > 0, "head: bad shape N=%d H=%d W=%d", N, H, W);
  SCP_REQUIRE(!prev || ((S * H) % 2 == 0 && (S * W) % 2 == 0), "head: odd output size");
  HeadArgs a{taps, bias, prev, out, N, J, H, W, K, S, J <= 8 ? 8 : 16};
  const size_t total = (size_t)N * S * H * S * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (dtype == SCPOSE_DT_BF16)
    hipLaunchKernelGGL(head_gather_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(head_gather_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

// Ensemble mean of validate_cv (landmark_regression/lib/core/function.py:530-536): output = sum of the models'
// heat-maps, then / len(models).  One call per added model: acc = (acc + x) / div, with div = 1 for all but the
// last model (x / 1 is exact), so the arithmetic and its order are the reference's.
__global__ __launch_bounds__(256) void heatmap_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ x,
                                                                 float div, size_t count) {
  for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < count; gid += (size_t)gridDim.x * 256)
    acc[gid] = (acc[gid] + x[gid]) / div;
}

int32_t heatmap_accumulate_launch(float* acc, const float* x, float div, size_t count, hipStream_t stream) {
  if (count == 0) return SCPOSE_OK;
  size_t blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(heatmap_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, acc, x, div, count);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
