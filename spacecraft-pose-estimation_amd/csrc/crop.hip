// Crop pre-processing: per-sample affine bilinear warp of a full frame to the network input size.
//
// Replaces cv2.warpAffine(data_numpy, trans, (W, H), flags=cv2.INTER_LINEAR) of
// landmark_regression/lib/dataset/JointsDataset.py:191-195 (trans = get_affine_transform(c, s, 0,
// image_size), lib/utils/transforms.py:57-89) and the optional BGR->RGB swap of :149-150.
// The arithmetic is OpenCV's fixed-point path for uint8 images (3.4 imgwarp.cpp: hal::warpAffine, WarpAffineInvoker,
// remapBilinear<FixedPtCast<int, uchar, 15>>), the same restatement as utils/transforms.py:warp_affine_bilinear:
//   X = (round((M1*y + M2)*1024) + 16 + round(M0*x*1024)) >> 5      source x in 1/32 px (likewise Y)
//   pixel = (S00*(32-a)(32-b)*32 + S01*a(32-b)*32 + S10*(32-a)b*32 + S11*a*b*32 + 16384) >> 15,   a = X & 31, b = Y & 31
// with taps outside the frame counting as 0 (BORDER_CONSTANT) and M = the inverse map the host derives exactly as
// cv::warpAffine does (utils/transforms.py:invert_affine_cv).  cv2 is not available to compare with ("parity
// unpinned"); the kernel is bit-exact against the NumPy restatement and the scalar oracle (oracle/warp_ref.py).
// One thread per output pixel; the 3 channels of a tap are 3 adjacent bytes; output is the uint8
// NHWC tensor the stem kernel consumes (its ToTensor/Normalize is fused there).
#include "common.h"

#pragma clang fp contract(off)

namespace scpose {

__global__ __launch_bounds__(256) void crop_warp_kernel(const uint8_t* __restrict__ frames,
                                                        const int64_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ hw,
                                                        const double* __restrict__ minv, int N, int oh,
                                                        int ow, int swap_rb, uint8_t* __restrict__ out,
                                                        const int32_t* __restrict__ roi) {
  const size_t total = (size_t)N * oh * ow;
  for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
    const int x = (int)(gid % ow);
    const size_t t = gid / ow;
    const int y = (int)(t % oh);
    const int n = (int)(t / oh);
    const double* m = minv + (size_t)n * 6;
    const int sh = hw[2 * n], sw = hw[2 * n + 1];
    const uint8_t* src = frames + offsets[n];
    // roi != null: only the window [ry, ry + rh) x [rx, rx + rw) of frame n is stored (the loader ships the part of the frame the
    // warp can touch instead of 1920 x 1200 x 3 bytes per frame); coordinates, weights and the frame border stay those of the whole frame
    const int rx = roi ? roi[4 * n] : 0, ry = roi ? roi[4 * n + 1] : 0, rw = roi ? roi[4 * n + 2] : sw, rh = roi ? roi[4 * n + 3] : sh;
    const double xs = (double)x, ys = (double)y;
    auto sat_int = [](double v) -> long long {      // cv::saturate_cast<int>(double): cvRound + clamp to int32
      v = rint(v);
      return (long long)(v < -2147483648.0 ? -2147483648.0 : (v > 2147483647.0 ? 2147483647.0 : v));
    };
    const long long X = (sat_int((m[1] * ys + m[2]) * 1024.0) + 16 + sat_int(m[0] * xs * 1024.0)) >> 5;
    const long long Y = (sat_int((m[4] * ys + m[5]) * 1024.0) + 16 + sat_int(m[3] * xs * 1024.0)) >> 5;
    long long x0 = X >> 5, y0 = Y >> 5;
    x0 = x0 < -32768 ? -32768 : (x0 > 32767 ? 32767 : x0);      // saturate_cast<short>
    y0 = y0 < -32768 ? -32768 : (y0 > 32767 ? 32767 : y0);
    const int a = (int)(X & 31), b = (int)(Y & 31);
    const int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
    uint8_t res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      auto tap = [&](long long yy, long long xx) -> int {
        const bool ok = yy >= 0 && yy < sh && xx >= 0 && xx < sw && yy >= ry && yy < ry + rh && xx >= rx && xx < rx + rw;
        return ok ? (int)src[((size_t)(yy - ry) * rw + (size_t)(xx - rx)) * 3 + c] : 0;
      };
      const int v = (tap(y0, x0) * w00 + tap(y0, x0 + 1) * w01 + tap(y0 + 1, x0) * w10 + tap(y0 + 1, x0 + 1) * w11 + 16384) >> 15;
      res[c] = (uint8_t)(v > 255 ? 255 : v);
    }
    uint8_t* o = out + gid * 3;
    o[0] = swap_rb ? res[2] : res[0];
    o[1] = res[1];
    o[2] = swap_rb ? res[0] : res[2];
  }
}

int32_t crop_warp_launch(const uint8_t* frames, const int64_t* offsets, const int32_t* hw, const double* minv,
                         int N, int oh, int ow, int swap_rb, uint8_t* out, hipStream_t stream, const int32_t* roi) {
  const size_t total = (size_t)N * oh * ow;
  if (total == 0) return SCPOSE_OK;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(crop_warp_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, frames, offsets, hw, minv, N, oh,
                     ow, swap_rb, out, roi);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
