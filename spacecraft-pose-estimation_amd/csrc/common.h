// Internal declarations shared by the gfx950 kernels and the host-side plan.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <string.h>

#include "../../include/scpose.h"

namespace scpose {

// ---- error plumbing (thread-local message, int status; nothing throws across the ABI) ----
void set_error(const char* fmt, ...);
const char* dev_env(const char* name);   // getenv() for development switches; null unless SCPOSE_DEV=1 (util.cpp)
const char* last_error();

#define SCP_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ::scpose::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                          __FILE__, __LINE__);                                      \
      return SCPOSE_E_HIP;                                                          \
    }                                                                               \
  } while (0)

#define SCP_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::scpose::set_error(__VA_ARGS__);   \
      return SCPOSE_E_INVALID;            \
    }                                     \
  } while (0)

// Development instrumentation and ablation switches (phase stamps, "skip the MFMA loop" and the like) exist only in
// the DEVELOPMENT build of the library (make dev -> libscpose_hip_dev.so, -DSCPOSE_DEV_BUILD, loaded by tools_dev/ through
// SCPOSE_DEV=1).  In the shipped libscpose_hip.so these fold to constants at compile time: its kernels contain no
// switchable wrong-results path and never read the dbg fields of their launch descriptors.
#ifdef SCPOSE_DEV_BUILD
#define SCP_DBG(p, bits) ((p).dbg & (bits))
#define SCP_DBG_BUF(p) ((p).dbg_buf)
#define SCP_DEV_ONLY(x) (x)
constexpr bool kDevBuild = true;
#else
#define SCP_DBG(p, bits) (0)
#define SCP_DBG_BUF(p) (static_cast<unsigned long long*>(nullptr))
#define SCP_DEV_ONLY(x) (0)
constexpr bool kDevBuild = false;
#endif

// Kernels that use more than 64 KB of dynamic LDS must be opted in with hipFuncSetAttribute, which applies to the
// device that is current at the call.  One memo per kernel instantiation, indexed by device: the attribute is set the
// first time that instantiation is launched on each device (idempotent per-device memoisation like conv_zero_page();
// never a per-process flag, which would leave a second device of the same process without the opt-in).
struct LdsOptIn {
  bool done[16] = {false};
};
int32_t lds_opt_in(const void* kernel, int bytes, LdsOptIn* memo);

// Exact unsigned division by a launch-time constant (n < 2^31): q = (mulhi(n, mul) + n*add) >> shift.
struct FastDiv {
  uint32_t mul, shift, add, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f; f.d = d ? d : 1;
  uint32_t s = 0;
  while ((2u << s) <= f.d) ++s;                 // s = floor(log2 d)
  if ((f.d & (f.d - 1)) == 0) { f.mul = 0; f.add = 1; f.shift = s; }
  else { f.mul = (uint32_t)((((uint64_t)1 << (32 + s)) / f.d) + 1); f.add = 0; f.shift = s; }
  return f;
}

// ---- 3x3 / 1x1 implicit-GEMM convolution ---------------------------------------------------
// Device-side description of one convolution launch.  All tensors are "blocked":
// [N][C/8][H][W][8] 16-bit elements.
struct ConvLaunch {
  const void* in;      // blocked input  N x Cin x H x W
  const void* in2;     // optional second input tensor supplying the planes >= split_planes (K-concatenated 1x1 conv), else null
  int32_t split_planes;
  const void* wpk;     // packed weights (see pack_conv_weights)
  const float* bias;   // f32 [n_mblk*MT] (zero padded)
  const void* res;     // optional blocked residual, shape of out
  void* out;           // blocked output, or f32 NCHW when out_nchw_f32
  int32_t N, H, W, Ho, Wo;
  int32_t cin_planes;  // Cin/8
  int32_t cout;        // real Cout (for store masks)
  int32_t th, tw;      // output tile (th*tw <= 64*NREP)
  int32_t tiles_x, tiles_y;
  int32_t halo_w, halo_h;
  int32_t plane_stride;  // LDS bytes per staged input plane
  int32_t cp;            // planes per K-chunk (even)
  int32_t nchunks;
  int32_t ksteps_full;   // k-steps (32 deep) of a full chunk
  int32_t n_mblk;        // Cout blocks of MT
  int32_t relu;
  int32_t out_nchw_f32;
  int32_t total_blocks;
  // pipelined (persistent) variant
  const void* zero16;    // >= 16 zero bytes in global memory (source of padding pixels for LDS-DMA)
  int32_t items_total;   // (tile, Cout block) work items
  int32_t items_per_wg;
  int32_t grid;          // persistent workgroups
  int32_t lds_w, lds_x;  // bytes of one weight-chunk / input-chunk LDS buffer
  int32_t lds_bias;      // bytes reserved for the bias vector in LDS
  int32_t nbuf_w, nbuf_x;  // LDS buffers: weights 1 (resident) or 2, inputs 2 or 3
  int32_t tiles_total;   // N * tiles_x * tiles_y
  int32_t nt;            // pixel tiles per wave group processed one after the other on one staged weight chunk
  int32_t groups;        // wave groups (256 threads each) working on different tiles in parallel (1 or 2)
  FastDiv fd_npix, fd_tw, fd_hp, fd_halo_w, fd_tiles_img, fd_tiles_x, fd_nmblk;   // conv_m32: divisions by launch constants
  uint32_t in_bytes, out_bytes;   // sizes of the input / output (= residual) tensors when both are < 4 GiB
                                  // (buffer-addressed global traffic, conv_pipe_kernel.h: dma16_buf), else 0 (64-bit
                                  // addressing).  conv_pipe: out_bytes = size of in2 (K-concatenated second input).
  unsigned long long* dbg_buf;  // development build only (SCP_DBG_BUF): per-workgroup phase cycle sums (dbg & 8), else null
  int32_t dbg;           // development build only (SCP_DBG): ablation bits 1 skip MFMA loop, 2 skip epilogue, 4 skip input DMA;
                         // the shipped library's kernels never read either field
  int32_t cu_share;      // host only: CUs to size the layer for (0 = the whole chip)
};

// Per-layer choice of kernel variant + tiling.
struct ConvConfig {
  int ks, stride;      // 1|3, 1|2
  int mrep, nrep;      // 16x16 MFMA tiles per wave along Cout / pixels
  int th, tw;          // output tile
  int cp;              // input planes (8 channels each) per K chunk
};

struct PackedConv {
  int cin, cout, ks, stride, dtype;
  int mt;              // cout block (16*mrep)
  int n_mblk;
  int cp, nchunks, ksteps_full;
  size_t wbytes;       // packed weight bytes
  void* d_w = nullptr;     // device packed weights
  float* d_bias = nullptr; // device bias (n_mblk*mt)
  int mrep;
  int variant = 0;     // 0: 16x16x32 MFMA kernels (conv_pipe / conv_stag), 1: 32x32x16 kernel (conv_m32)
  int wm = 1;          // variant 1: waves along Cout (mt = 32*mrep*wm)
  // streaming 1x1 kernel (conv1x1.hip): weights [k-step][k-group][cout_pad1][8] and bias in MFMA row order, when eligible
  void* d_w1 = nullptr;
  float* d_b1 = nullptr;
  size_t w1_bytes = 0;
  int cout_pad1 = 0;
  // tiling chosen by conv_launch_m32 for the last (N, Ho, Wo) seen (the search is a function of those and the layer only)
  struct TileMemo { int n = -1, ho = 0, wo = 0, th = 0, tw = 0, nseg = 0, nr = 0, ps = 0, occ = 0, cp = 0, cus = 0, nb16 = 0; };
  mutable TileMemo m32_memo[2];   // [0] whole chip, [1] a share of it (concurrent lanes)
  // register-weight stride-2 kernel (conv_s2r.hip): weights [k-step][cout block][k-group][row][8], bias in MFMA row order
  void* d_ws2 = nullptr;
  float* d_bs2 = nullptr;
};
bool conv_s2r_config(int cin, int cout, int stride, int* planes, int* nblk, int* g);
size_t conv_s2r_pack(const float* w, int cout, int cin, int dtype, uint16_t* dst);
void conv_s2r_pack_bias(const float* bias, int cout, float* dst);
int32_t conv_s2r_launch(const PackedConv& pc, const void* in, int N, int H, int W, int relu, void* out, hipStream_t stream);
bool conv1x1_stream_eligible(const PackedConv& pc);
int32_t conv1x1_stream_launch(const PackedConv& pc, const void* in, int N, int H, int W, const void* res, int relu,
                              void* out, hipStream_t stream, const void* in2 = nullptr, int split_planes = 0);

// Pick (mrep, cp) for a layer independent of the spatial size; tiles are chosen per launch.
void choose_mrep_cp(int cin, int cout, int ks, int stride, int* mrep, int* cp);
// Choose spatial tiling for an output map.
void choose_tile(int ks, int stride, int Ho, int Wo, int* nrep, int* th, int* tw);

// Host: fold-free packing of f32 OIHW weights into the kernel's LDS image order.
// Returns bytes; fills `dst` (16-bit words) when non-null.
size_t pack_conv_weights(const float* w, int cout, int cin, int ks, int mt, int cp, int dtype,
                         uint16_t* dst, int* nchunks, int* ksteps_full);

int32_t conv_upload(const float* w, const float* bias, int cout, int cin, int ks, int stride,
                    int dtype, PackedConv* pc);
void conv_free(PackedConv* pc);
// cu_share > 0: the caller runs other layers beside this one (concurrent lanes of the captured forward); a small 3x3 layer
// then sizes its grid and tiles for that many CUs instead of the whole chip (same results: tiling never changes a pixel's sum)
int32_t conv_launch(const PackedConv& pc, const void* in, int N, int H, int W, const void* res,
                    int relu, int out_nchw_f32, void* out, hipStream_t stream,
                    const void* in2 = nullptr, int split_planes = 0, int cu_share = 0);
size_t conv_lds_bytes(const PackedConv& pc, int nrep, int th, int tw);
int plane_stride_for(int stride, int halo_h, int halo_w);
// software-pipelined persistent kernel (conv_pipe_kernel.h); nt = pixel tiles per work item
int32_t conv_launch_pipe(const PackedConv& pc, ConvLaunch& L, int nrep, int nt, int occ, int groups, hipStream_t stream);
size_t conv_pipe_lds_bytes(const PackedConv& pc, int plane_stride, int groups);
const void* conv_zero_page();   // lazily allocated 256 zero bytes on the current device
// fused BasicBlock (conv_block_kernel.h): out = relu(conv2(relu(conv1(x))) + x) for C -> C -> C 3x3 layers
bool block_fusable(const PackedConv& c1, const PackedConv& c2);
int32_t block_launch(const PackedConv& c1, const PackedConv& c2, const void* in, int N, int H, int W, void* out,
                     hipStream_t stream);
int conv_device_cus();
// branch chain (conv_chain.hip): the BasicBlocks of one low-resolution branch in one launch, a frame per workgroup, activations in LDS
bool conv_chain_channels(int C);                 // channel counts with a chain kernel (the weights are packed at create time)
bool conv_chain_supported(int C, int H, int W);  // ... and the map sizes it runs at (decided per forward)
size_t conv_chain_pack(const float* w, int nconv, int C, int dtype, uint16_t* dst);   // nconv folded OIHW 3x3 weights back to back; bytes
int32_t conv_chain_launch(const void* in, void* out, const void* wpk, const float* bias, int nconv, int N, int C, int H, int W,
                          int dtype, uint32_t* sched, hipStream_t stream);
unsigned long long* conv_dbg_buffer(hipStream_t stream);   // development instrumentation
void conv_dbg_set_grid(int grid);
// PackedConv::mrep value of the 48-row Cout block of the producer/consumer kernel's 3 x 8 form (conv_m32p_kernel.h, M16 = 3): one and
// a half 32-row units have no integer; mt = 48
constexpr int kMrep48 = 15;
// 32x32x16-MFMA kernel (conv_m32_kernel.h): layer eligibility + variant, packing, launch
bool conv_m32_choose(int cin, int cout, int ks, int stride, int* mr, int* wm, int* cp);
size_t pack_conv_weights_m32(const float* w, int cout, int cin, int ks, int mt, int cp, int dtype,
                             uint16_t* dst, int* nchunks, int* ksteps_full);
int32_t conv_launch_m32(const PackedConv& pc, ConvLaunch& L, hipStream_t stream);   // SCPOSE_E_UNSUPPORTED-free: returns 1 if the shape has no good tiling (caller falls back)

// ---- stem: 3 -> 64, 3x3 stride 2 from f32 NCHW or u8 NHWC ----------------------------------
int32_t stem_launch(const void* in, int in_fmt, const float* w_folded /*dev [8][27][8]: channel group, tap, channel*/,
                    const float* bias /*dev [64]*/, const float* mean_std /*dev [6] or null*/,
                    int N, int H, int W, int dtype, void* out, hipStream_t stream);

// fused stem (stem_fused.hip): conv1 + bn1 + relu + conv2 + bn2 + relu in one launch, both on MFMA
void stem_fused_pack(const float* w1, const float* b1, const float* w2, const float* b2, int dtype, std::vector<uint16_t>* pw1,
                     std::vector<uint16_t>* pw2, std::vector<float>* pb1, std::vector<float>* pb2);
int32_t stem_fused_launch(const void* in, int in_fmt, const void* w1, const void* w2, const float* b1, const float* b2,
                          const float* mean_std, int N, int H, int W, int dtype, void* out,
                          uint32_t* sched /*2 zero-initialised device words: dynamic tile queue (conv_device.h: tile_claim)*/, hipStream_t stream);

// fused layer1 Bottleneck (bottleneck.hip): conv1 1x1 Cin->64, conv2 3x3 64->64, conv3 1x1 64->256 + residual in one launch;
// Cin = 256: identity residual; Cin = 64: the first Bottleneck, its `downsample` projection (wds, bds) folded into conv3
bool bottleneck_fusable(int cin, int cmid, int cout);
void bottleneck_pack(const float* w1, const float* w2, const float* w3, const float* wds, const float* b1, const float* b2, const float* b3,
                     const float* bds, int cin, int dtype,
                     std::vector<uint16_t>* pw1, std::vector<uint16_t>* pw2, std::vector<uint16_t>* pw3, std::vector<float>* pb);
int32_t bottleneck_launch(const void* in, const void* w1, const void* w2, const void* w3, const float* bias, int N, int H, int W,
                          int cin, int dtype, void* out, uint32_t* sched /*11 zero-initialised device words: per-XCD tile queues [0..7], finished workgroups [8], device-wide queue of the first Bottleneck [9..10]*/,
                          hipStream_t stream);

// fuse row 0 + the first hop of every down path from branch 0 of a HighResolutionModule, one pass over branch 0 (fuse_down.hip)
struct FuseDownPacked {
  void* d_w = nullptr;     // conv_s2r_pack of the concatenated first-hop convolutions [2 c0 | c0 | c0][c0][3][3]
  float* d_b = nullptr;
  int c0 = 0, nb = 0, dtype = 0;
};
bool fuse_down_supported(int nb, int c0, int c1);
int32_t fuse_down_upload(const float* w, const float* bias, int nb, int c0, int dtype, FuseDownPacked* fd);
void fuse_down_free(FuseDownPacked* fd);
int32_t fuse_down_launch(const FuseDownPacked& fd, const void* x0, int N, int H, int W, const void* const* terms, void* y,
                         void* const* outs, hipStream_t stream);

// ---- elementwise ---------------------------------------------------------------------------
int32_t fuse_sum_launch(const void* const* terms, const int32_t* shifts, int nterms, int N, int C,
                        int H, int W, int dtype, void* out, hipStream_t stream);
int32_t crop_warp_launch(const uint8_t* frames, const int64_t* offsets, const int32_t* hw, const double* minv,
                         int N, int oh, int ow, int swap_rb, uint8_t* out, hipStream_t stream, const int32_t* roi = nullptr);
int32_t head_gather_launch(const void* taps, const float* bias, const float* prev, int N, int J, int H, int W,
                           int K, int S, int dtype, float* out, hipStream_t stream);
int32_t heatmap_accumulate_launch(float* acc, const float* x, float div, size_t count, hipStream_t stream);
int32_t flip_merge_launch(const float* a, const float* b, const int32_t* perm, int N, int J, int H, int W, int shift,
                          float* out, hipStream_t stream);
int32_t nchw_to_blocked_launch(const float* src, int N, int C, int H, int W, int dtype, void* dst,
                               hipStream_t stream);
int32_t blocked_to_nchw_launch(const void* src, int N, int C, int H, int W, int dtype, float* dst,
                               hipStream_t stream);

// ---- decode / pnp --------------------------------------------------------------------------
// head_fused.hip: last fuse sum + final_layer (+ decode) in one pass (pose_hrnet.py:256-263, :458; lib/core/inference.py:18-79)
void head_fused_pack(const float* w, const float* b, int J, int C, int dtype, uint16_t* wfrag, float* bias);
size_t head_fused_part_bytes(int n);
bool head_fused_supported(int nterms, int C, int J, int N, int H, int W);
int32_t head_fused_launch(const void* const* terms, const int32_t* shifts, int nterms, int N, int C, int H, int W, int J,
                          int dtype, const void* wfrag, const float* bias, float* hm, float* part_v, int32_t* part_i,
                          const float* center, const float* scale, int post_process, float* preds, hipStream_t stream);
int32_t decode_launch(const float* hm, int N, int J, int H, int W, const float* center,
                      const float* scale, int post_process, float* preds_xyc, float* coords,
                      float* maxvals, hipStream_t stream);
int32_t pnp_launch(const float* kp_xyc, const double* landmarks, const double* K,
                   const double* dist, int N, int J, double conf_thr0, int min_pts, double thr_decay,
                   int thr_iters, int max_iters, double reproj_err, double confidence, double* rot,
                   double* tvec, double* rvec, int32_t* status, hipStream_t stream, double* rows = nullptr);

// f32 -> 16-bit storage on the host (round to nearest even), matching the device casts.
uint16_t host_f32_to_16(float f, int dtype);
float host_16_to_f32(uint16_t v, int dtype);

}  // namespace scpose
