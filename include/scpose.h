/*
 * scpose.h -- C ABI of the MI355X (gfx950) HRNet -> heatmap decode -> EPnP/RANSAC library.
 *
 * The reference (mohsij/spacecraft-pose-estimation) has no FFI for this path; the path sits
 * behind Python call signatures.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference repo root).  INTEGRATION.md shows the ctypes
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns int32 status: 0 = ok, <0 = SCPOSE_E_*; the message of the last
 *     failure on the calling thread is returned by scpose_last_error().  Nothing throws.
 *   - OWNERSHIP: the host (PyTorch) allocates and owns every device buffer passed in
 *     (inputs, outputs, workspace).  The library owns only the opaque handle and the packed
 *     weights it uploads at create time (freed by scpose_hrnet_destroy).
 *   - STREAMS: every launch function takes a hipStream_t as void*; nothing synchronises
 *     internally and nothing allocates in a launch function, with one exception: the first
 *     launch of a kernel on a device opts that kernel into large LDS (hipFuncSetAttribute) and,
 *     for 64-bit-addressed tensors, allocates a 256-byte zero page -- both memoised PER DEVICE.
 *     Run one forward eagerly on a device before capturing launches into a hipGraph there
 *     (scpose_hrnet_graph_* below does so itself).
 *   - THREADING: a handle is bound to the device current at create time and is not
 *     thread-safe; distinct handles are independent.  The only process-wide state is the
 *     per-device memoisation above (idempotent, indexed by device id) and the thread-local
 *     error message.
 *   - ONE STREAM PER HANDLE AT A TIME: besides its read-only weights a handle owns a few words
 *     of mutable device state -- the dynamic tile queues of its persistent kernels (16 words
 *     per layer, self-resetting at the end of each launch).  All launches that use one handle
 *     -- eager forwards AND replays of graphs captured from it -- must therefore be ordered
 *     with respect to each other (same stream, or event dependencies); two forwards of one
 *     handle in flight at once would claim tiles from the same queue.  Every other buffer a
 *     forward writes lies in the caller's workspace.  For concurrent forwards create one
 *     handle per stream.
 *   - "blocked" activation layout used between layers: [N][C/8][H][W][8] 16-bit elements
 *     (bf16 or f16), C a multiple of 8.
 *   - NO CPU ENTRY POINTS: this library has no host implementation of any function below and no fallback -- every
 *     function needs a gfx950 device and fails with SCPOSE_E_HIP without one.  (SURVEY.md section 8b sketched
 *     `scpose_cpu_*` twins for the CPU baseline; they are deliberately not part of the ABI: the CPU restatement of
 *     the path is test infrastructure and lives under oracle/ -- hrnet_ref.py, decode_ref.py, pnp_ref.c, warp_ref.py --
 *     where only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; tests/test_abi.py checks
 *     that nothing under the package, the CLIs or bench.py's measured path imports it.)
 */
#ifndef SCPOSE_H
#define SCPOSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCPOSE_ABI_VERSION 7

enum {
  SCPOSE_OK = 0,
  SCPOSE_E_INVALID = -1,   /* bad argument / unsupported shape */
  SCPOSE_E_MISSING = -2,   /* a required checkpoint tensor was not supplied */
  SCPOSE_E_HIP = -3,       /* a HIP runtime call failed */
  SCPOSE_E_WORKSPACE = -4, /* workspace too small */
  SCPOSE_E_NOMEM = -5
};

/* 16-bit storage / MFMA operand type of the network (accumulation is always fp32). */
enum { SCPOSE_DT_BF16 = 0, SCPOSE_DT_F16 = 1 };

/* output head = cfg.MODEL.NAME (landmark_regression/lib/models/):
 *   FINAL_LAYER  pose_hrnet.py:323-329   final_layer conv on branch 0, heat-maps H/4 x W/4
 *   CMS          hrnet_cms.py:353-419, :551-557   four ConvTranspose2d(k5,s4)+Conv2d heads summed coarse-to-fine
 *                with bilinear x2 upsampling, heat-maps H x W ("equal_to_image")
 *   CMS_384      hrnet_cms_384.py (same, k3 s2), heat-maps H/2 x W/2 ("4x") */
enum { SCPOSE_HEAD_FINAL_LAYER = 0, SCPOSE_HEAD_CMS = 1, SCPOSE_HEAD_CMS_384 = 2 };

/* input formats of scpose_hrnet_forward */
enum {
  SCPOSE_IN_F32_NCHW = 0,  /* normalised float32 N x 3 x H x W: what the reference module's
                              forward(x) receives (landmark_regression/tools/test.py:106-114) */
  SCPOSE_IN_U8_NHWC = 1    /* raw uint8 N x H x W x 3 RGB crop; ToTensor + Normalize(mean,std)
                              of tools/test.py:106-108 is fused into the stem kernel */
};

int32_t scpose_abi_version(void);
const char* scpose_last_error(void);
/* 0 for the shipped library (libscpose_hip.so: no instrumentation or ablation path compiled into any kernel), 1 for the
 * development build of the same sources (libscpose_hip_dev.so, -DSCPOSE_DEV_BUILD), which tools_dev/ loads with SCPOSE_DEV=1. */
int32_t scpose_is_dev_build(void);

/* ------------------------------------------------------------------------------------------
 * HRNet.  Replaces models.pose_hrnet.get_pose_net(cfg, is_train=False) + load_state_dict +
 * PoseHighResolutionNet.forward
 * (landmark_regression/lib/models/pose_hrnet.py:274-331, :425-460, :495-501;
 *  call sites landmark_regression/tools/test.py:84-98, lib/core/function.py:341).
 * ---------------------------------------------------------------------------------------- */
typedef struct scpose_hrnet_desc {
  int32_t num_joints;          /* cfg.MODEL.NUM_JOINTS */
  int32_t final_conv_kernel;   /* cfg.MODEL.EXTRA.FINAL_CONV_KERNEL (1 or 3) */
  int32_t num_stages;          /* always 3 (STAGE2..STAGE4) */
  int32_t num_modules[3];      /* EXTRA.STAGEk.NUM_MODULES */
  int32_t num_branches[3];     /* EXTRA.STAGEk.NUM_BRANCHES (2,3,4) */
  int32_t num_blocks[3][4];    /* EXTRA.STAGEk.NUM_BLOCKS (blocks per branch) */
  int32_t num_channels[3][4];  /* EXTRA.STAGEk.NUM_CHANNELS (the block's planes: a BOTTLENECK branch carries 4x as many channels) */
  int32_t dtype;               /* SCPOSE_DT_* */
  float mean[3], std[3];       /* Normalize() constants for SCPOSE_IN_U8_NHWC */
  int32_t head;                /* SCPOSE_HEAD_*: which member of the model family (cfg.MODEL.NAME) */
  int32_t block[3];            /* EXTRA.STAGEk.BLOCK (blocks_dict, pose_hrnet.py:266-269): 0 BASIC, 1 BOTTLENECK (expansion 4) */
} scpose_hrnet_desc;

typedef struct scpose_hrnet* scpose_hrnet_t;

/* names/ptrs/numels: the checkpoint's state_dict as HOST float32 arrays (conv weights OIHW),
 * keyed exactly as the reference module's state_dict ("conv1.weight", "bn1.running_var",
 * "stage3.2.fuse_layers.1.0.0.0.weight", "final_layer.bias", ...).  Unknown keys are ignored
 * (num_batches_tracked etc.); a missing required key fails with SCPOSE_E_MISSING unless
 * allow_missing != 0, in which case the tensor takes the reference constructor's default
 * (strict=False behaviour of tools/test.py:90: BN -> identity, conv -> zeros).
 * BatchNorm (eval, eps 1e-5) is folded into the conv weights/bias here and the result is
 * packed for the MFMA kernels and uploaded. */
int32_t scpose_hrnet_create(const scpose_hrnet_desc* desc, const char* const* names,
                            const float* const* ptrs, const int64_t* numels, int32_t count,
                            int32_t allow_missing, scpose_hrnet_t* out);
int32_t scpose_hrnet_destroy(scpose_hrnet_t h);

/* bytes of device workspace scpose_hrnet_forward needs for a batch of n frames of h x w. */
int32_t scpose_hrnet_workspace_bytes(scpose_hrnet_t h, int32_t n, int32_t height, int32_t width,
                                     size_t* bytes);
/* heat-map size scpose_hrnet_forward writes for an input of height x width: H/4 (pose_hrnet, the
 * cfg.MODEL.HEATMAP_SIZE of the shipped YAMLs), H (hrnet_cms) or H/2 (hrnet_cms_384). */
int32_t scpose_hrnet_heatmap_size(scpose_hrnet_t h, int32_t height, int32_t width, int32_t* out_h,
                                  int32_t* out_w);
/* number of kernel launches of one forward and total conv FLOPs (2*MAC) per frame. */
int32_t scpose_hrnet_stats(scpose_hrnet_t h, int32_t height, int32_t width, int32_t* launches,
                           double* flops_per_frame, double* act_bytes_per_frame);

/* in: device pointer in in_fmt; heatmaps: device float32 N x J x H/4 x W/4 (scpose_hrnet_heatmap_size for the
 * hrnet_cms heads) (NCHW, raw scores,
 * exactly what the reference forward returns).  H and W must be multiples of 32. */
int32_t scpose_hrnet_forward(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n,
                             int32_t height, int32_t width, float* heatmaps, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Forward + decode in one call: the key points of validate() / the pose exporter without a heat-map round trip through HBM.
 * Replaces the pair model(input) -> get_final_preds(output, center, scale) of landmark_regression/lib/core/function.py:376-393
 * (lib/core/inference.py:18-79, lib/utils/transforms.py:49-110); arguments as scpose_hrnet_forward and scpose_decode.
 * For pose_hrnet with FINAL_CONV_KERNEL == 1 the last fuse sum, final_layer and the decode run as one pass over the
 * branch-0 tensor (head_fused.hip): `heatmaps` may then be NULL and nothing is written for them; when it is given it
 * receives exactly what scpose_hrnet_forward writes.  Other heads (3x3 final layer, hrnet_cms) need `heatmaps` and run
 * forward + scpose_decode back to back on `stream`.  preds_xyc is bit-identical to scpose_decode(scpose_hrnet_forward(...)). */
int32_t scpose_hrnet_tail_fused(scpose_hrnet_t h, int32_t n, int32_t height, int32_t width, int32_t* fused);   /* 1: heatmaps may be NULL */
int32_t scpose_hrnet_forward_decode(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                    int32_t width, const float* center, const float* scale, int32_t post_process,
                                    float* preds_xyc, float* heatmaps, void* workspace, size_t workspace_bytes,
                                    void* stream);

/* Captured forward.  The launch list of one forward for a FIXED (input buffer, batch shape, heat-map buffer, workspace)
 * is recorded once into a hipGraph and replayed with one call: for small batches the host-side launch cost of the
 * ~280 kernels disappears, and with concurrent != 0 the ops that do not depend on each other -- the branches of a
 * HighResolutionModule (pose_hrnet.py:247-253), the rows of its fuse layer (:254-265), the transition convolutions
 * (:333-372) -- are recorded on parallel graph branches, so that the small-grid kernels of the low-resolution
 * branches fill the CUs the high-resolution ones leave idle.  concurrent == 2 puts only the fuse rows and the transition
 * convolutions side by side (many short HBM-bound launches) and runs the branches one after the other: the choice for
 * batches whose branch kernels each fill the chip (W48 384x384 batch 256: -0.3 ms; with concurrent == 1 the MFMA-bound
 * branch kernels contend and the forward gets slower).  Results are bit-identical to scpose_hrnet_forward.
 * The concurrent memory plan keeps a tensor alive until every op that may run beside its last reader has finished:
 * size the workspace with scpose_hrnet_graph_workspace_bytes (>= scpose_hrnet_workspace_bytes).
 * create runs one eager forward on an internal stream (it needs valid input in `in`) and synchronises it; launch
 * only enqueues.  The caller refills `in` and reads `heatmaps` in stream order around scpose_hrnet_graph_launch. */
typedef struct scpose_hrnet_graph* scpose_hrnet_graph_t;
int32_t scpose_hrnet_graph_workspace_bytes(scpose_hrnet_t h, int32_t n, int32_t height, int32_t width, size_t* bytes);
int32_t scpose_hrnet_graph_create(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                  int32_t width, float* heatmaps, void* workspace, size_t workspace_bytes,
                                  int32_t concurrent, scpose_hrnet_graph_t* out);
/* the captured form of scpose_hrnet_forward_decode (center / scale / preds_xyc / heatmaps are baked in like `in`) */
int32_t scpose_hrnet_graph_create_decode(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                         int32_t width, const float* center, const float* scale, int32_t post_process,
                                         float* preds_xyc, float* heatmaps, void* workspace, size_t workspace_bytes,
                                         int32_t concurrent, scpose_hrnet_graph_t* out);
int32_t scpose_hrnet_graph_launch(scpose_hrnet_graph_t g, void* stream);
int32_t scpose_hrnet_graph_nodes(scpose_hrnet_graph_t g, int32_t* nodes);   /* kernel + dependency nodes captured */
int32_t scpose_hrnet_graph_destroy(scpose_hrnet_graph_t g);

/* Unit-level parity hook: runs the forward up to and including the op that produces the named intermediate tensor
 * and writes it as float32 N x C x h x w (converted from the 16-bit blocked layout).  Names follow the forward of
 * pose_hrnet.py:425-460: "stem1" (:426-428), "stem2" (:429-431), "layer1" (:432), "stage<S>.<M>.out0" = y_list[0]
 * after module M of stage S (:247-265).  With out == NULL only *channels / *out_h / *out_w are filled (shape query). */
int32_t scpose_hrnet_tap_names(scpose_hrnet_t h, char* buf, int32_t cap);   /* comma-separated names this handle offers ("stem1"
                                                                              exists only when the stem runs as two layers) */
int32_t scpose_hrnet_forward_tap(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                 int32_t width, const char* tap, float* out, int32_t* channels, int32_t* out_h,
                                 int32_t* out_w, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement hooks (bench.py): the same forward with a HIP event recorded on `stream` before
 * every launch and after the last one, then per-launch milliseconds, algorithmic FLOPs and
 * bytes per frame and a kernel signature {kind, 10*ksize+stride | nterms, Cin, Cout} with kind
 * 0 stem conv1 alone (two-layer stem), 1 convolution, 2 fuse sum, 3 fused BasicBlock, 4 head gather
 * (hrnet_cms), 5 fused stem (conv1 + conv2), 6 fused Bottleneck, 7 fused tail (last fuse sum + final_layer; the fuse op it absorbs
 * reports no work).  profile_read blocks on the last event.  Call with ms == NULL to get *count. */
int32_t scpose_hrnet_forward_profiled(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n,
                                      int32_t height, int32_t width, float* heatmaps,
                                      void* workspace, size_t workspace_bytes, void* stream);
/* (ABI 6) the same hook for scpose_hrnet_forward_decode: the launch list of the key-point path (fused tail with the decode
 * inside, no heat-map written when `heatmaps` is NULL) -- what bench.py's timed steps replay -- with per-launch events */
int32_t scpose_hrnet_forward_decode_profiled(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                             int32_t width, const float* center, const float* scale, int32_t post_process,
                                             float* preds_xyc, float* heatmaps, void* workspace, size_t workspace_bytes,
                                             void* stream);
int32_t scpose_hrnet_profile_read(scpose_hrnet_t h, int32_t height, int32_t width, int32_t cap,
                                  float* ms, double* flops_per_frame, double* bytes_per_frame,
                                  int32_t* sig, int32_t* count);

/* ------------------------------------------------------------------------------------------
 * Heatmap decode.  Replaces get_max_preds + get_final_preds + transform_preds
 * (landmark_regression/lib/core/inference.py:18-79, lib/utils/transforms.py:49-110) and the
 * all_preds assembly of validate() (lib/core/function.py:389-393).
 *   heatmaps  device f32 N x J x H x W
 *   center    device f32 N x 2, scale device f32 N x 2 (meta['center'], meta['scale'])
 *   preds_xyc device f32 N x J x 3 = [x_img, y_img, maxval]
 * post_process = cfg.TEST.POST_PROCESS.  One wavefront per (n, j) map.
 * ---------------------------------------------------------------------------------------- */
int32_t scpose_decode(const float* heatmaps, int32_t n, int32_t j, int32_t h, int32_t w,
                      const float* center, const float* scale, int32_t post_process,
                      float* preds_xyc, void* stream);
/* get_max_preds alone: coords device f32 N x J x 2 (heatmap px), maxvals device f32 N x J. */
int32_t scpose_max_preds(const float* heatmaps, int32_t n, int32_t j, int32_t h, int32_t w,
                         float* coords, float* maxvals, void* stream);

/* Crop pre-processing.  Replaces, per sample, cv2.warpAffine(frame, get_affine_transform(c, s, 0,
 * IMAGE_SIZE), IMAGE_SIZE, flags=INTER_LINEAR) and the COLOR_RGB channel swap of
 * landmark_regression/lib/dataset/JointsDataset.py:134-150, :191-195.
 *   frames    device u8: the samples' full frames (H_i x W_i x 3, any sizes) packed back to back
 *   offsets   device i64 N: byte offset of sample i's frame inside `frames`
 *   frame_hw  device i32 N x 2: (H_i, W_i)
 *   minv      device f64 N x 6: row-major 2x3 INVERSE of the reference's `trans` (crop pixel -> frame pixel),
 *             inverted the way cv::warpAffine does it (OpenCV 3.4 imgwarp.cpp; utils/transforms.py:invert_affine_cv)
 *   crops     device u8 N x out_h x out_w x 3 = the SCPOSE_IN_U8_NHWC input of scpose_hrnet_forward
 * swap_rb != 0 exchanges channels 0 and 2 (BGR frame -> RGB crop).  Border value 0.  Arithmetic: OpenCV's
 * fixed-point uint8 path (coordinates quantised to 1/32 px, integer weights scaled by 2^15, (sum + 2^14) >> 15). */
int32_t scpose_crop_warp(const uint8_t* frames, const int64_t* offsets, const int32_t* frame_hw,
                         const double* minv, int32_t n, int32_t out_h, int32_t out_w, int32_t swap_rb,
                         uint8_t* crops, void* stream);
/* (ABI 7) The same warp when only a WINDOW of every frame is resident: windows = packed uint8 (roi_h x roi_w x 3) blocks at
 * offsets[i], roi_xywh = device i32 N x 4 [x0, y0, w, h] of window i inside its frame, frame_hw = the FULL frame's size (the
 * border rule is the frame's).  The caller guarantees that every tap the warp reads inside the frame lies inside the window
 * (dataset/JointsDataset.py computes it from the affine); taps outside a window read 0.  Same crops, bit for bit, for a
 * fraction of the host-to-device bytes (a 1920 x 1200 frame is 6.9 MB, the window of a 300 px target 0.5 MB). */
int32_t scpose_crop_warp_roi(const uint8_t* windows, const int64_t* offsets, const int32_t* frame_hw,
                             const int32_t* roi_xywh, const double* minv, int32_t n, int32_t out_h,
                             int32_t out_w, int32_t swap_rb, uint8_t* crops, void* stream);

/* Flip test (cfg.TEST.FLIP_TEST, lib/core/function.py:347-366): out = (a + flip_back(b)) * 0.5 where b
 * is the forward of the x-flipped input; flip_back (lib/utils/transforms.py:15-29) mirrors b in x and
 * swaps the joints of each flip pair; shift != 0 applies the TEST.SHIFT_HEATMAP column shift (:361-363).
 *   a, b, out  device f32 N x J x H x W (out may alias a)
 *   perm       device i32 J: perm[j] = partner joint of j (j itself when unpaired) */
int32_t scpose_flip_merge(const float* a, const float* b, const int32_t* perm, int32_t n, int32_t j,
                          int32_t h, int32_t w, int32_t shift, float* out, void* stream);

/* Ensemble mean of validate_cv (landmark_regression/lib/core/function.py:530-536,
 * tools/test_cv_ensemble.py:84-98): acc = (acc + x) / div over count float32 values.  Call once per
 * additional model with div = 1, and with div = number of models for the last one. */
int32_t scpose_heatmap_accumulate(float* acc, const float* x, float div, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched PnP.  Replaces, per frame, the confidence filter + cv2.solvePnPRansac(...,
 * flags=SOLVEPNP_EPNP, iterationsCount, reprojectionError) + cv2.Rodrigues of
 * pose_estimation/export_predicted_poses_real.py:186-203.  One wavefront per frame, fp64.
 *   kp_xyc     device f32 N x J x 3 (x, y, confidence) -- rows of pred.mat
 *   landmarks  device f64 J x 3, K device f64 3x3 row-major, dist device f64[5] (k1,k2,p1,p2,k3)
 *   conf_thr0 / min_pts / thr_decay / thr_iters: the threshold loop of :188-197
 *              (0.95, 15, 0.8, 100 in the reference)
 *   rot        device f64 N x 9 row-major rotation matrix (= cv2.Rodrigues(rvec)[0])
 *   tvec       device f64 N x 3
 *   rvec       device f64 N x 3 (may be NULL)
 *   status     device i32 N: >=0 number of RANSAC inliers; <0 failure code
 *              (-1: fewer than 4 usable points [the reference raises], -2: RANSAC found no model, -4: the final solve is not
 *              finite [cv2 would return NaN]); every failure writes the identity rotation and a zero translation.
 *              Exactly four usable points: as in OpenCV 3.4, no RANSAC -- the P3P kernel on the first three points, the
 *              fourth picks among its up to four poses (status 4, or -2 when P3P finds none); exactly five: one EPnP.
 * ---------------------------------------------------------------------------------------- */
int32_t scpose_pnp_epnp_ransac(const float* kp_xyc, const double* landmarks, const double* K,
                               const double* dist, int32_t n, int32_t j, double conf_thr0,
                               int32_t min_pts, double thr_decay, int32_t thr_iters,
                               int32_t max_iters, double reproj_err, double confidence,
                               double* rot, double* tvec, double* rvec, int32_t* status,
                               void* stream);
/* (ABI 7) Same solve, one output: rows = device f64 N x 13, row i = [R (9, row-major), t (3), (double)status] of frame i --
 * the record export_predicted_poses_real.py:224-226 builds per frame, laid out as the block a rank all-gathers (SURVEY.md
 * section 8e) and copies to the host, so a step needs no assembly launches between the solve and the collective. */
int32_t scpose_pnp_epnp_ransac_rows(const float* kp_xyc, const double* landmarks, const double* K,
                                    const double* dist, int32_t n, int32_t j, double conf_thr0,
                                    int32_t min_pts, double thr_decay, int32_t thr_iters,
                                    int32_t max_iters, double reproj_err, double confidence,
                                    double* rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Single-layer entry points (unit-level parity of the kernels the forward is made of).
 * ---------------------------------------------------------------------------------------- */
typedef struct scpose_conv* scpose_conv_t;
/* weight: host f32 OIHW (already BN-folded or plain), bias: host f32[cout] or NULL. */
int32_t scpose_conv_create(const float* weight, const float* bias, int32_t cout, int32_t cin,
                           int32_t ksize, int32_t stride, int32_t dtype, scpose_conv_t* out);
int32_t scpose_conv_destroy(scpose_conv_t c);
/* in/out/residual: blocked 16-bit device tensors; out_nchw_f32 != 0 writes float32 NCHW
 * (cout channels) instead.  y = [relu]( conv(x) + bias [+ residual] ). */
int32_t scpose_conv_forward(scpose_conv_t c, const void* in, int32_t n, int32_t h, int32_t w,
                            const void* residual, int32_t relu, int32_t out_nchw_f32, void* out,
                            void* stream);
/* Fused BasicBlock (pose_hrnet.py:41-57): out = relu(conv2(relu(conv1(x))) + x) in one kernel, for two 3x3 /
 * stride-1 / C -> C convolutions created with scpose_conv_create (C = 32 or 48).  in/out blocked C x h x w.
 * Returns SCPOSE_E_INVALID when the pair is not fusable (then run the two scpose_conv_forward calls). */
int32_t scpose_basic_block_forward(scpose_conv_t conv1, scpose_conv_t conv2, const void* in, int32_t n,
                                   int32_t h, int32_t w, void* out, void* stream);
/* out = relu(sum_t upsample_nearest(term_t, 2^shift_t)); all blocked, out is c x h x w. */
int32_t scpose_fuse_sum(const void* const* terms, const int32_t* shifts, int32_t nterms,
                        int32_t n, int32_t c, int32_t h, int32_t w, int32_t dtype, void* out,
                        void* stream);
/* layout converters: float32 NCHW <-> blocked 16-bit (c multiple of 8). */
int32_t scpose_nchw_f32_to_blocked(const float* src, int32_t n, int32_t c, int32_t h, int32_t w,
                                   int32_t dtype, void* dst, void* stream);
int32_t scpose_blocked_to_nchw_f32(const void* src, int32_t n, int32_t c, int32_t h, int32_t w,
                                   int32_t dtype, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCPOSE_H */
