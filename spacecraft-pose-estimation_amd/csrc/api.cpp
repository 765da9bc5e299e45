// extern "C" surface (include/scpose.h) over the kernels: decode, PnP, single-layer entry points.
// The HRNet entry points live next to the plan in hrnet.cpp.
#include "common.h"
#include <new>

using namespace scpose;

struct scpose_conv { PackedConv pc; };

extern "C" int32_t scpose_abi_version(void) { return SCPOSE_ABI_VERSION; }
extern "C" int32_t scpose_is_dev_build(void) { return scpose::kDevBuild ? 1 : 0; }
extern "C" const char* scpose_last_error(void) { return last_error(); }

extern "C" int32_t scpose_decode(const float* heatmaps, int32_t n, int32_t j, int32_t h, int32_t w,
                                 const float* center, const float* scale, int32_t post_process,
                                 float* preds_xyc, void* stream) {
  if (n == 0) return SCPOSE_OK;  /* empty batch: nothing to do, pointers may be null */
  SCP_REQUIRE(heatmaps && preds_xyc, "decode: null argument");
  return decode_launch(heatmaps, n, j, h, w, center, scale, post_process, preds_xyc, nullptr,
                       nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_crop_warp(const uint8_t* frames, const int64_t* offsets, const int32_t* frame_hw,
                                    const double* minv, int32_t n, int32_t out_h, int32_t out_w, int32_t swap_rb,
                                    uint8_t* crops, void* stream) {
  if (n == 0) return SCPOSE_OK;
  SCP_REQUIRE(frames && offsets && frame_hw && minv && crops, "crop_warp: null argument");
  SCP_REQUIRE(out_h > 0 && out_w > 0, "crop_warp: bad output size %dx%d", out_h, out_w);
  return crop_warp_launch(frames, offsets, frame_hw, minv, n, out_h, out_w, swap_rb, crops,
                          static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_crop_warp_roi(const uint8_t* windows, const int64_t* offsets, const int32_t* frame_hw, const int32_t* roi_xywh,
                                        const double* minv, int32_t n, int32_t out_h, int32_t out_w, int32_t swap_rb,
                                        uint8_t* crops, void* stream) {
  if (n == 0) return SCPOSE_OK;
  SCP_REQUIRE(windows && offsets && frame_hw && roi_xywh && minv && crops, "crop_warp_roi: null argument");
  SCP_REQUIRE(out_h > 0 && out_w > 0, "crop_warp_roi: bad output size %dx%d", out_h, out_w);
  return crop_warp_launch(windows, offsets, frame_hw, minv, n, out_h, out_w, swap_rb, crops,
                          static_cast<hipStream_t>(stream), roi_xywh);
}

extern "C" int32_t scpose_flip_merge(const float* a, const float* b, const int32_t* perm, int32_t n, int32_t j,
                                     int32_t h, int32_t w, int32_t shift, float* out, void* stream) {
  if (n == 0) return SCPOSE_OK;
  SCP_REQUIRE(a && b && perm && out, "flip_merge: null argument");
  SCP_REQUIRE(j > 0 && h > 0 && w > 0, "flip_merge: bad shape J=%d H=%d W=%d", j, h, w);
  return flip_merge_launch(a, b, perm, n, j, h, w, shift, out, static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_heatmap_accumulate(float* acc, const float* x, float div, int64_t count, void* stream) {
  if (count == 0) return SCPOSE_OK;
  SCP_REQUIRE(acc && x && count > 0, "heatmap_accumulate: null argument");
  SCP_REQUIRE(div > 0.f, "heatmap_accumulate: div=%f", (double)div);
  return heatmap_accumulate_launch(acc, x, div, (size_t)count, static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_max_preds(const float* heatmaps, int32_t n, int32_t j, int32_t h,
                                    int32_t w, float* coords, float* maxvals, void* stream) {
  if (n == 0) return SCPOSE_OK;
  SCP_REQUIRE(heatmaps && coords && maxvals, "max_preds: null argument");
  return decode_launch(heatmaps, n, j, h, w, nullptr, nullptr, 0, nullptr, coords, maxvals,
                       static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_pnp_epnp_ransac(const float* kp_xyc, const double* landmarks,
                                          const double* K, const double* dist, int32_t n, int32_t j,
                                          double conf_thr0, int32_t min_pts, double thr_decay,
                                          int32_t thr_iters, int32_t max_iters, double reproj_err,
                                          double confidence, double* rot, double* tvec,
                                          double* rvec, int32_t* status, void* stream) {
  if (n == 0) return SCPOSE_OK;
  SCP_REQUIRE(kp_xyc && landmarks && K && rot && tvec && status, "pnp: null argument");
  return pnp_launch(kp_xyc, landmarks, K, dist, n, j, conf_thr0, min_pts, thr_decay, thr_iters,
                    max_iters, reproj_err, confidence, rot, tvec, rvec, status,
                    static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_pnp_epnp_ransac_rows(const float* kp_xyc, const double* landmarks, const double* K,
                                               const double* dist, int32_t n, int32_t j, double conf_thr0,
                                               int32_t min_pts, double thr_decay, int32_t thr_iters,
                                               int32_t max_iters, double reproj_err, double confidence,
                                               double* rows, void* stream) {
  if (n == 0) return SCPOSE_OK;
  SCP_REQUIRE(kp_xyc && landmarks && K && rows, "pnp_rows: null argument");
  return pnp_launch(kp_xyc, landmarks, K, dist, n, j, conf_thr0, min_pts, thr_decay, thr_iters,
                    max_iters, reproj_err, confidence, nullptr, nullptr, nullptr, nullptr,
                    static_cast<hipStream_t>(stream), rows);
}

extern "C" int32_t scpose_conv_create(const float* weight, const float* bias, int32_t cout,
                                      int32_t cin, int32_t ksize, int32_t stride, int32_t dtype,
                                      scpose_conv_t* out) {
  SCP_REQUIRE(weight && out, "conv_create: null argument");
  scpose_conv* c = new (std::nothrow) scpose_conv();
  if (!c) { set_error("conv_create: out of host memory"); return SCPOSE_E_NOMEM; }
  const int32_t rc = conv_upload(weight, bias, cout, cin, ksize, stride, dtype, &c->pc);
  if (rc != SCPOSE_OK) { conv_free(&c->pc); delete c; return rc; }
  *out = c;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_conv_destroy(scpose_conv_t c) {
  if (!c) return SCPOSE_OK;
  conv_free(&c->pc);
  delete c;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_conv_forward(scpose_conv_t c, const void* in, int32_t n, int32_t h,
                                       int32_t w, const void* residual, int32_t relu,
                                       int32_t out_nchw_f32, void* out, void* stream) {
  SCP_REQUIRE(c && in && out, "conv_forward: null argument");
  return conv_launch(c->pc, in, n, h, w, residual, relu, out_nchw_f32, out,
                     static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_basic_block_forward(scpose_conv_t conv1, scpose_conv_t conv2, const void* in, int32_t n,
                                              int32_t h, int32_t w, void* out, void* stream) {
  SCP_REQUIRE(conv1 && conv2 && in && out, "basic_block_forward: null argument");
  SCP_REQUIRE(block_fusable(conv1->pc, conv2->pc), "basic_block_forward: not a fusable pair (3x3 stride-1 C->C->C, C = 32 or 48)");
  return block_launch(conv1->pc, conv2->pc, in, n, h, w, out, static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_fuse_sum(const void* const* terms, const int32_t* shifts, int32_t nterms,
                                   int32_t n, int32_t c, int32_t h, int32_t w, int32_t dtype,
                                   void* out, void* stream) {
  SCP_REQUIRE(terms && shifts && out, "fuse_sum: null argument");
  return fuse_sum_launch(terms, shifts, nterms, n, c, h, w, dtype, out,
                         static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_nchw_f32_to_blocked(const float* src, int32_t n, int32_t c, int32_t h,
                                              int32_t w, int32_t dtype, void* dst, void* stream) {
  SCP_REQUIRE(src && dst, "nchw_f32_to_blocked: null argument");
  return nchw_to_blocked_launch(src, n, c, h, w, dtype, dst, static_cast<hipStream_t>(stream));
}

extern "C" int32_t scpose_blocked_to_nchw_f32(const void* src, int32_t n, int32_t c, int32_t h,
                                              int32_t w, int32_t dtype, float* dst, void* stream) {
  SCP_REQUIRE(src && dst, "blocked_to_nchw_f32: null argument");
  return blocked_to_nchw_launch(src, n, c, h, w, dtype, dst, static_cast<hipStream_t>(stream));
}
