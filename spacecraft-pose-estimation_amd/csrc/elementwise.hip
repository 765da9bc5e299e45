// HBM-bound elementwise kernels on the blocked [N][C/8][H][W][8] 16-bit layout.
//
// fuse_sum: y_i = ReLU( sum_j term_ij ) of HighResolutionModule.forward
//   (landmark_regression/lib/models/pose_hrnet.py:256-263).  A term is either a tensor at the
//   output resolution (x_i itself, or the output of the stride-2 3x3 chain :211-239) or the
//   1x1-conv+BN result at a LOWER resolution that nn.Upsample(scale_factor=2^s, 'nearest')
//   (:206) would enlarge -- read here with (y>>s, x>>s) instead of materialising the upsample.
//   Terms are summed in fp32 in the reference's j order, then ReLU, then one 16-bit rounding.
// One thread = one 16-byte (pixel, 8-channel) vector: 16 B/lane coalesced loads and stores.
#include "common.h"

namespace scpose {

struct FuseArgs {
  const void* term[4];
  int shift[4];
  int nterms;
  int N, planes, H, W;
  void* out;
};

template <typename T> __device__ __forceinline__ float ew_from(uint32_t bits16) {
  return (float)__builtin_bit_cast(T, (uint16_t)bits16);
}
template <typename T> __device__ __forceinline__ uint32_t ew_to(float f) {
  T t = (T)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, t);
}

template <int DT> struct EwDt { typedef __bf16 type; };
template <> struct EwDt<1> { typedef _Float16 type; };

template <int DT>
__global__ __launch_bounds__(256) void fuse_sum_kernel(const FuseArgs a) {
  typedef typename EwDt<DT>::type T;
  const size_t total = (size_t)a.N * a.planes * a.H * a.W;
  for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total;
       gid += (size_t)gridDim.x * 256) {
    const int x = (int)(gid % a.W);
    size_t t = gid / a.W;
    const int y = (int)(t % a.H);
    const size_t np = t / a.H;  // n * planes + plane
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < a.nterms) {
        const int sh = a.shift[k];
        const int h = a.H >> sh, w = a.W >> sh;
        const uint4 v = *reinterpret_cast<const uint4*>(
            static_cast<const char*>(a.term[k]) + ((np * h + (y >> sh)) * w + (x >> sh)) * 16);
        s[0] += ew_from<T>(v.x & 0xffff); s[1] += ew_from<T>(v.x >> 16);
        s[2] += ew_from<T>(v.y & 0xffff); s[3] += ew_from<T>(v.y >> 16);
        s[4] += ew_from<T>(v.z & 0xffff); s[5] += ew_from<T>(v.z >> 16);
        s[6] += ew_from<T>(v.w & 0xffff); s[7] += ew_from<T>(v.w >> 16);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = fmaxf(s[j], 0.f);
    uint4 o;
    o.x = ew_to<T>(s[0]) | (ew_to<T>(s[1]) << 16);
    o.y = ew_to<T>(s[2]) | (ew_to<T>(s[3]) << 16);
    o.z = ew_to<T>(s[4]) | (ew_to<T>(s[5]) << 16);
    o.w = ew_to<T>(s[6]) | (ew_to<T>(s[7]) << 16);
    *reinterpret_cast<uint4*>(static_cast<char*>(a.out) + gid * 16) = o;
  }
}

int32_t fuse_sum_launch(const void* const* terms, const int32_t* shifts, int nterms, int N, int C,
                        int H, int W, int dtype, void* out, hipStream_t stream) {
  SCP_REQUIRE(nterms >= 1 && nterms <= 4, "fuse_sum: %d terms (1..4)", nterms);
  SCP_REQUIRE(C % 8 == 0, "fuse_sum: C=%d must be a multiple of 8", C);
  FuseArgs a;
  for (int k = 0; k < 4; ++k) { a.term[k] = k < nterms ? terms[k] : nullptr; a.shift[k] = k < nterms ? shifts[k] : 0; }
  for (int k = 0; k < nterms; ++k)
    SCP_REQUIRE(shifts[k] >= 0 && (H >> shifts[k]) << shifts[k] == H && (W >> shifts[k]) << shifts[k] == W,
                "fuse_sum: term %d shift %d does not divide %dx%d", k, shifts[k], H, W);
  a.nterms = nterms; a.N = N; a.planes = C / 8; a.H = H; a.W = W; a.out = out;
  const size_t total = (size_t)N * a.planes * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU
  if (dtype == SCPOSE_DT_BF16)
    hipLaunchKernelGGL(fuse_sum_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(fuse_sum_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

// ---- layout converters (test / debugging plumbing, also the module's generic input path) ----
template <int DT>
__global__ void nchw_to_blocked_kernel(const float* __restrict__ src, int N, int C, int H, int W,
                                       void* __restrict__ dst) {
  typedef typename EwDt<DT>::type T;
  const size_t HW = (size_t)H * W;
  const size_t total = (size_t)N * (C / 8) * HW;
  for (size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = gid % HW;
    const size_t np = gid / HW;
    const size_t n = np / (C / 8), pl = np % (C / 8);
    uint32_t b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = ew_to<T>(src[(n * C + pl * 8 + j) * HW + pix]);
    uint4 o = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
    *reinterpret_cast<uint4*>(static_cast<char*>(dst) + gid * 16) = o;
  }
}

template <int DT>
__global__ void blocked_to_nchw_kernel(const void* __restrict__ src, int N, int C, int H, int W,
                                       float* __restrict__ dst) {
  typedef typename EwDt<DT>::type T;
  const size_t HW = (size_t)H * W;
  const size_t total = (size_t)N * (C / 8) * HW;
  for (size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = gid % HW;
    const size_t np = gid / HW;
    const size_t n = np / (C / 8), pl = np % (C / 8);
    const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const char*>(src) + gid * 16);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dst[(n * C + pl * 8 + j) * HW + pix] = ew_from<T>((w[j >> 1] >> ((j & 1) * 16)) & 0xffff);
  }
}

// ---- flip-test merge (TEST.FLIP_TEST): out = (a + flip_back(b)) * 0.5 -------------------------
// landmark_regression/lib/core/function.py:347-366: b is the network output for the horizontally
// flipped input; flip_back (lib/utils/transforms.py:15-29) mirrors it in x and swaps the joints of
// each flip pair; with TEST.SHIFT_HEATMAP columns 1.. take the value of the column to their left
// (column 0 keeps its own).  fp32, same operation order as the reference: (a + b') * 0.5.
__global__ __launch_bounds__(256) void flip_merge_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const int32_t* __restrict__ perm, int N, int J, int H, int W,
                                                         int shift, float* __restrict__ out) {
  const size_t total = (size_t)N * J * H * W;
  for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
    const int x = (int)(gid % W);
    size_t t = gid / W;
    const int y = (int)(t % H);
    t /= H;
    const int j = (int)(t % J);
    const size_t n = t / J;
    const int xs = (shift && x > 0) ? x - 1 : x;          // column of the flipped-back map that lands on x
    const float bv = b[((n * J + perm[j]) * H + y) * W + (W - 1 - xs)];
    out[gid] = (a[gid] + bv) * 0.5f;
  }
}

int32_t flip_merge_launch(const float* a, const float* b, const int32_t* perm, int N, int J, int H, int W, int shift,
                          float* out, hipStream_t stream) {
  const size_t total = (size_t)N * J * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return SCPOSE_OK;
  hipLaunchKernelGGL(flip_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, b, perm, N, J, H, W, shift, out);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

static unsigned ew_grid(size_t total) {
  size_t b = (total + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

int32_t nchw_to_blocked_launch(const float* src, int N, int C, int H, int W, int dtype, void* dst,
                               hipStream_t stream) {
  SCP_REQUIRE(C % 8 == 0, "layout: C=%d must be a multiple of 8", C);
  const size_t total = (size_t)N * (C / 8) * H * W;
  if (dtype == SCPOSE_DT_BF16)
    hipLaunchKernelGGL(nchw_to_blocked_kernel<0>, dim3(ew_grid(total)), dim3(256), 0, stream, src, N, C, H, W, dst);
  else
    hipLaunchKernelGGL(nchw_to_blocked_kernel<1>, dim3(ew_grid(total)), dim3(256), 0, stream, src, N, C, H, W, dst);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

int32_t blocked_to_nchw_launch(const void* src, int N, int C, int H, int W, int dtype, float* dst,
                               hipStream_t stream) {
  SCP_REQUIRE(C % 8 == 0, "layout: C=%d must be a multiple of 8", C);
  const size_t total = (size_t)N * (C / 8) * H * W;
  if (dtype == SCPOSE_DT_BF16)
    hipLaunchKernelGGL(blocked_to_nchw_kernel<0>, dim3(ew_grid(total)), dim3(256), 0, stream, src, N, C, H, W, dst);
  else
    hipLaunchKernelGGL(blocked_to_nchw_kernel<1>, dim3(ew_grid(total)), dim3(256), 0, stream, src, N, C, H, W, dst);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
