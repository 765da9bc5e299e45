// Heatmap decode: one 256-thread workgroup per (frame, joint) map.
//
// Replaces get_max_preds (landmark_regression/lib/core/inference.py:18-46), the quarter-pixel
// refinement and back-transform of get_final_preds (:49-79) with transform_preds /
// get_affine_transform(inv=1) / affine_transform (lib/utils/transforms.py:49-110), and the
// [x, y, maxval] row assembly of validate() (lib/core/function.py:392-393).
//
// HBM-bound: each map (H*W f32) is read exactly once with 16 B/lane loads; every thread keeps a
// running (value, first index) pair, waves combine with 6 xor-shuffles, the workgroup's four waves through LDS.  Tie / NaN rules are
// numpy's: first occurrence wins, NaN counts as the maximum (first NaN wins).
#include "common.h"
#include "decode_device.h"   // better(), decode_coords / decode_refines / decode_refine / decode_to_image (shared with head_fused.hip)

namespace scpose {

struct DecodeArgs {
  const float* hm;
  const float* center;
  const float* scale;
  float* preds_xyc;  // N*J*3 or null
  float* coords;     // N*J*2 or null (get_max_preds coords, heatmap px, masked)
  float* maxvals;    // N*J or null
  int N, J, H, W;
  int post_process;
};

// One 256-thread workgroup per map: every thread issues all of its 16-byte loads before it compares anything (a 96 x 96
// map is 9 loads per thread; with one wave per map and a load-compare loop the kernel kept 1 KB in flight per wave and
// ran at 0.56 TB/s), reduces over its wave with xor-shuffles and over the four waves through LDS.
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int map = blockIdx.x;
  const int HW = a.H * a.W;
  const float* m = a.hm + (size_t)map * HW;

  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  if ((HW & 3) == 0) {
    constexpr int U = 4;                                  // loads in flight per thread and round
    for (int i0 = tid * 4; i0 < HW; i0 += 1024 * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 1024;
        v[u] = i < HW ? *reinterpret_cast<const float4*>(m + i) : make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 1024;
        if (i < HW) {
          if (better(v[u].x, i, bv, bi)) { bv = v[u].x; bi = i; }
          if (better(v[u].y, i + 1, bv, bi)) { bv = v[u].y; bi = i + 1; }
          if (better(v[u].z, i + 2, bv, bi)) { bv = v[u].z; bi = i + 2; }
          if (better(v[u].w, i + 3, bv, bi)) { bv = v[u].w; bi = i + 3; }
        }
      }
    }
  } else {
    for (int i = tid; i < HW; i += 256) {
      const float v = m[i];
      if (better(v, i, bv, bi)) { bv = v; bi = i; }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
  __syncthreads();
  if (tid != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }

  float cx, cy;
  decode_coords(bv, bi, a.W, cx, cy);
  if (a.maxvals) a.maxvals[map] = bv;
  if (a.coords) { a.coords[map * 2] = cx; a.coords[map * 2 + 1] = cy; }
  if (!a.preds_xyc) return;
  int px, py;
  if (a.post_process && decode_refines(cx, cy, a.H, a.W, px, py))
    decode_refine(m[py * a.W + px + 1], m[py * a.W + px - 1], m[(py + 1) * a.W + px], m[(py - 1) * a.W + px], cx, cy);
  const int n = map / a.J;
  decode_to_image(cx, cy, bv, a.H, a.W, a.center + n * 2, a.scale + n * 2, a.preds_xyc + (size_t)map * 3);
}

int32_t decode_launch(const float* hm, int N, int J, int H, int W, const float* center,
                      const float* scale, int post_process, float* preds_xyc, float* coords,
                      float* maxvals, hipStream_t stream) {
  SCP_REQUIRE(N >= 0 && J > 0 && H > 0 && W > 0, "decode: bad shape N=%d J=%d H=%d W=%d", N, J, H, W);
  SCP_REQUIRE((size_t)H * W < (1u << 24), "decode: H*W=%zu not exactly representable in float32 (reference :37)", (size_t)H * W);
  SCP_REQUIRE(!preds_xyc || (center && scale), "decode: center/scale required for image-space output");
  if (N == 0) return SCPOSE_OK;
  DecodeArgs a{hm, center, scale, preds_xyc, coords, maxvals, N, J, H, W, post_process};
  hipLaunchKernelGGL(decode_kernel, dim3(N * J), dim3(256), 0, stream, a);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
