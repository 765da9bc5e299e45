// Device-side pieces of the heat-map decode shared by decode.hip (decode of a heat-map tensor in HBM) and head_fused.hip
// (decode inside the network's last kernel): the arg-max ordering of get_max_preds and everything get_final_preds does
// once the maximum of a map is known.
#pragma once
#include "common.h"

namespace scpose {

// np.argmax order (landmark_regression/lib/core/inference.py:30-31): the first occurrence of the maximum wins and
// NaN counts as the maximum (the first NaN wins)
__device__ __forceinline__ bool better(float v1, int i1, float v2, int i2) {
  const bool n1 = v1 != v1, n2 = v2 != v2;
  if (n1 || n2) return (n1 && n2) ? (i1 < i2) : n1;
  return v1 > v2 || (v1 == v2 && i1 < i2);
}

// inference.py:37-45 -- idx as float32, x = idx % W, y = floor(idx / W), masked when !(max > 0)
__device__ __forceinline__ void decode_coords(float bv, int bi, int W, float& cx, float& cy) {
  cx = (float)(bi % W); cy = (float)(bi / W);
  if (!(bv > 0.0f)) { cx = 0.f; cy = 0.f; }
}

// inference.py:56-69: the quarter-pixel refinement looks at the four neighbours of (px, py) only when
// 1 < px < W - 1 and 1 < py < H - 1 (strict inequalities)
__device__ __forceinline__ bool decode_refines(float cx, float cy, int H, int W, int& px, int& py) {
  px = (int)floorf(cx + 0.5f); py = (int)floorf(cy + 0.5f);
  return 1 < px && px < W - 1 && 1 < py && py < H - 1;
}

// sign(right - left) * 0.25, sign(below - above) * 0.25 with sign(0) = 0 and NaN propagating, as np.sign does
__device__ __forceinline__ void decode_refine(float right, float left, float below, float above, float& cx, float& cy) {
  const float dx = right - left, dy = below - above;
  const float sx = dx != dx ? dx : (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f));
  const float sy = dy != dy ? dy : (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f));
  cx += sx * 0.25f;
  cy += sy * 0.25f;
}

// transforms.py:57-89 with rot = 0, inv = 1, output_size = (W, H).  The three float32 point pairs the reference hands
// to cv2.getAffineTransform are rebuilt with the same float32 roundings; the affine map they define is then solved in
// closed form in float64:   dst: (W/2,H/2) (W/2,H/2-W/2) (0,H/2-W/2)  ->  src: (cx,cy) (cx,s1y) (s2x,s1y)
__device__ __forceinline__ void decode_to_image(float cx, float cy, float bv, int H, int W, const float* center, const float* scale,
                                                float* o) {
  const float ccx = center[0], ccy = center[1];
  const float src_w = __fmul_rn(scale[0], 200.0f);              // scale_tmp[0]
  const float s1y = __fadd_rn(ccy, -0.5f * src_w);              // src[1,1]
  const float d = __fsub_rn(ccy, s1y);                          // direct[1] of get_3rd_point
  const float s2x = __fsub_rn(ccx, d);                          // src[2,0]
  const double half_w = 0.5 * (double)W, half_h = 0.5 * (double)H;
  const double a00 = ((double)ccx - (double)s2x) / half_w;
  const double a11 = ((double)ccy - (double)s1y) / half_w;
  const double xi = (double)ccx + a00 * ((double)cx - half_w);
  const double yi = (double)ccy + a11 * ((double)cy - half_h);
  o[0] = (float)xi;
  o[1] = (float)yi;
  o[2] = bv;
}

}  // namespace scpose
