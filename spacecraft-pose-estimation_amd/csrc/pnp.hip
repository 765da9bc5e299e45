// Batched EPnP + RANSAC: one wavefront per frame, fp64.
//
// Replaces the per-frame serial CPU loop of pose_estimation/export_predicted_poses_real.py
// :177-203: confidence filter (:186-197), cv2.solvePnPRansac(obj, img, K, dist,
// flags=SOLVEPNP_EPNP, iterationsCount, reprojectionError) (:199-201) and cv2.Rodrigues (:203).
// cv2 is opencv-python 3.4.11.41 (environment.yml:37); its algorithm (solvepnp.cpp
// solvePnPRansac/PnPRansacCallback, ptsetreg.cpp RANSACPointSetRegistrator::run, epnp.cpp,
// calibration.cpp Rodrigues/projectPoints, undistort.cpp undistortPoints, lapack.cpp
// JacobiSVDImpl_) is implemented here directly for the GPU.
//
// Mapping to the wavefront:
//   * lane j < J  : per-landmark work (confidence filter, undistortion, compaction by ballot);
//   * lane k      : RANSAC hypothesis k of the current batch of <= 64 iterations.  The minimal
//     5-point subsets are drawn from ONE cv::RNG stream in iteration order (all lanes step the
//     generator together), every lane solves its own 5-point EPnP and counts inliers, then the
//     batch is scanned in iteration order with OpenCV's acceptance rule and adaptive iteration
//     count -- exactly the sequential loop's result, with the iterations' latency overlapped;
//   * the 12x12 M^T M of each lane lives in LDS ([entry][lane], conflict-free 8-byte access);
//   * the final EPnP over the inliers is wave-uniform.
// Latency-bound by design (a few hundred microseconds per wave, all frames in parallel); it is
// <1 % of the HRNet forward at the benchmark sizes.
#include <float.h>

#include "common.h"

#pragma clang fp contract(off)   // separate mul/add like the scalar CPU path it is checked against

namespace scpose {

static constexpr int kMaxJ = 64;
// Lanes that own a 12x12 work matrix in LDS (144 doubles each): the hypotheses of a RANSAC batch plus the speculative
// final fits.  32 instead of 64 halves the workgroup's LDS to 36 KB + 72 B per landmark, so four one-wave workgroups share a CU instead of two
// and a batch of frames takes half as many CUs away from the convolution kernels of the next forward (bench.py runs the
// PnP stage beside it); the ordered replay below is independent of the batch width, so the results do not change.
static constexpr int kPW = 32;

struct PnpArgs {
  const float* kp;          // N x J x 3
  const double* landmarks;  // J x 3
  const double* K;          // 3x3
  const double* dist;       // 5 or null
  double* rot;              // N x 9
  double* tvec;             // N x 3
  double* rvec;             // N x 3 or null
  int32_t* status;          // N
  double* rows;             // N x 13 [R (9, row-major), t (3), status] or null: when set, the only output written (rot / tvec / status may be null)
  int N, J;
  double conf_thr0, thr_decay;
  int min_pts, thr_iters, max_iters;
  double reproj_err, confidence;
  int dbg_no_spec;          // development (SCPOSE_PNP_SPEC=0): ignore the speculative final fits
};

struct Cam { double fx, fy, cx, cy, k[5]; };

// ---- cv::RNG ---------------------------------------------------------------------------------
struct CvRng {
  unsigned long long state;
  __device__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---- libm-free elementary functions -----------------------------------------------------------
// hypot / log / integer pow as fixed sequences of IEEE-754 operations (+ - * / sqrt are correctly rounded on the device
// and on the host, and contraction is off), so that the C oracle, which runs the same sequences (oracle/pnp_ref.c), gets
// the same bits.  The device's and glibc's own hypot / log / pow differ in their last bits, and a 1-ulp difference in a
// Jacobi rotation is amplified without bound when three of the four smallest eigenvalues of M^T M nearly coincide
// (five nearly coplanar points: 2.6e-4 rad between the two implementations in round 2).
__device__ __forceinline__ double det_hypot(double a, double b) {
  a = fabs(a); b = fabs(b);
  const double hi = a > b ? a : b, lo = a > b ? b : a;
  if (hi == 0.) return 0.;
  const double r = lo / hi;
  return hi * sqrt(1. + r * r);
}
__device__ double det_log(double x) {   // x > 0 and finite; ~1e-16 relative
  unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int e = (int)((u >> 52) & 0x7ff);
  if (e == 0) { x *= 18014398509481984.; u = __builtin_bit_cast(unsigned long long, x); e = (int)((u >> 52) & 0x7ff) - 54; }   // subnormal: * 2^54
  e -= 1023;
  double m = __builtin_bit_cast(double, (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);   // [1, 2)
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }                                               // [sqrt(1/2), sqrt(2)]
  const double f = m - 1., sq = f / (2. + f), z = sq * sq;
  double pz = 1. / 27.;                                                                            // atanh series, 14 terms
  pz = pz * z + 1. / 25.; pz = pz * z + 1. / 23.; pz = pz * z + 1. / 21.; pz = pz * z + 1. / 19.;
  pz = pz * z + 1. / 17.; pz = pz * z + 1. / 15.; pz = pz * z + 1. / 13.; pz = pz * z + 1. / 11.;
  pz = pz * z + 1. / 9.; pz = pz * z + 1. / 7.; pz = pz * z + 1. / 5.; pz = pz * z + 1. / 3.;
  pz = pz * z + 1.;
  return (double)e * 0.6931471805599453 + 2. * sq * pz;
}
__device__ __forceinline__ double det_powi(double x, int n) {   // n >= 0, left-to-right products
  double r = 1.;
  for (int i = 0; i < n; i++) r = r * x;
  return r;
}

// ---- one-sided Jacobi SVD on small private matrices (JacobiSVDImpl_<double>) -----------------
// At: N rows of length M (columns of the M x N input).  Out: rows of At = left singular
// vectors, W descending, Vt rows = right singular vectors.
template <int M, int N>
__device__ void jacobi_small(double* At, double* W, double* Vt) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  for (int i = 0; i < N; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
    W[i] = sd;
    for (int k = 0; k < N; k++) Vt[i * N + k] = 0;
    Vt[i * N + i] = 1;
  }
  const int max_iter = M > 30 ? M : 30;
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < N - 1; i++)
      for (int j = i + 1; j < N; j++) {
        double* Ai = At + i * M; double* Aj = At + j * M;
        double a = W[i], p = 0, b = W[j], c, s;
        for (int k = 0; k < M; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = det_hypot(p, beta);
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < M; k++) {
          const double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        double* Vi = Vt + i * N; double* Vj = Vt + j * N;
        for (int k = 0; k < N; k++) {
          const double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0; Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < N; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < N - 1; i++) {
    int j = i;
    for (int k = i + 1; k < N; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      for (int k = 0; k < M; k++) { t = At[i * M + k]; At[i * M + k] = At[j * M + k]; At[j * M + k] = t; }
      for (int k = 0; k < N; k++) { t = Vt[i * N + k]; Vt[i * N + k] = Vt[j * N + k]; Vt[j * N + k] = t; }
    }
  }
  CvRng rng{0x12345678ULL};
  for (int i = 0; i < N; i++) {
    double sd = W[i];
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {  // exactly-zero singular value
      const double val0 = 1. / M;
      for (int k = 0; k < M; k++) At[i * M + k] = (rng.next() & 256) != 0 ? val0 : -val0;
      for (int it = 0; it < 2; it++)
        for (int j = 0; j < i; j++) {
          double asum = 0;
          sd = 0;
          for (int k = 0; k < M; k++) sd += At[i * M + k] * At[j * M + k];
          for (int k = 0; k < M; k++) {
            const double t = At[i * M + k] - sd * At[j * M + k];
            At[i * M + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < M; k++) At[i * M + k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
      sd = sqrt(sd);
    }
    const double s = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < M; k++) At[i * M + k] *= s;
  }
}

// x = pinv(A) b  (cvSolve / cvInvert with CV_SVD: SVBkSb with threshold 2*eps*sum(w))
template <int M, int N, int NB>
__device__ void svd_solve(const double* A, const double* b, double* x) {
  double At[N * M], W[N], Vt[N * N];
  for (int i = 0; i < N; i++)
    for (int k = 0; k < M; k++) At[i * M + k] = A[k * N + i];
  jacobi_small<M, N>(At, W, Vt);
  double thr = 0;
  for (int i = 0; i < N; i++) thr += W[i];
  thr *= DBL_EPSILON * 2;
  for (int j = 0; j < N * NB; j++) x[j] = 0;
  for (int i = 0; i < N; i++) {
    double wi = W[i];
    if (fabs(wi) <= thr) continue;
    wi = 1 / wi;
    for (int j = 0; j < NB; j++) {
      double s = 0;
      for (int k = 0; k < M; k++) s += At[i * M + k] * b[k * NB + j];
      s *= wi;
      for (int k = 0; k < N; k++) x[k * NB + j] += s * Vt[i * N + k];
    }
  }
}

// 12x12 one-sided Jacobi on the lane's LDS matrix; only U^T (rows) and the ordering are needed
// (the V accumulation of JacobiSVDImpl_ does not influence U or W).
#define UT(r, c) ut[((r) * 12 + (c)) * kPW]
__device__ void jacobi12(double* ut /* = base + lane */) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  double W[12];
  for (int i = 0; i < 12; i++) {
    double sd = 0;
    for (int k = 0; k < 12; k++) sd += UT(i, k) * UT(i, k);
    W[i] = sd;
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < 11; i++)
      for (int j = i + 1; j < 12; j++) {
        double ai[12], aj[12];
        double a = W[i], p = 0, b = W[j], c, s;
#pragma unroll
        for (int k = 0; k < 12; k++) { ai[k] = UT(i, k); aj[k] = UT(j, k); }
#pragma unroll
        for (int k = 0; k < 12; k++) p += ai[k] * aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = det_hypot(p, beta);
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
          const double t0 = c * ai[k] + s * aj[k], t1 = -s * ai[k] + c * aj[k];
          UT(i, k) = t0; UT(j, k) = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
      }
    if (!changed) break;
  }
  for (int i = 0; i < 12; i++) {
    double sd = 0;
    for (int k = 0; k < 12; k++) sd += UT(i, k) * UT(i, k);
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < 11; i++) {
    int j = i;
    for (int k = i + 1; k < 12; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      for (int k = 0; k < 12; k++) { t = UT(i, k); UT(i, k) = UT(j, k); UT(j, k) = t; }
    }
  }
  CvRng rng{0x12345678ULL};
  for (int i = 0; i < 12; i++) {
    double sd = W[i];
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / 12;
      for (int k = 0; k < 12; k++) UT(i, k) = (rng.next() & 256) != 0 ? val0 : -val0;
      for (int it = 0; it < 2; it++)
        for (int j = 0; j < i; j++) {
          double asum = 0;
          sd = 0;
          for (int k = 0; k < 12; k++) sd += UT(i, k) * UT(j, k);
          for (int k = 0; k < 12; k++) {
            const double t = UT(i, k) - sd * UT(j, k);
            UT(i, k) = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < 12; k++) UT(i, k) *= asum;
        }
      sd = 0;
      for (int k = 0; k < 12; k++) sd += UT(i, k) * UT(i, k);
      sd = sqrt(sd);
    }
    const double s = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < 12; k++) UT(i, k) *= s;
  }
}

// ---- calib3d pieces --------------------------------------------------------------------------
__device__ void rodrigues_vec2mat(const double r[3], double R[9]) {
  const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) {
    for (int k = 0; k < 9; k++) R[k] = 0;
    R[0] = R[4] = R[8] = 1;
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
  const double rx = r[0] * it, ry = r[1] * it, rz = r[2] * it;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  const double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  for (int k = 0; k < 9; k++) R[k] = c1 * rrt[k] + s * rxm[k];
  R[0] += c; R[4] += c; R[8] += c;
}

__device__ void rodrigues_mat2vec(const double Rin[9], double r[3]) {
  double At[9], W[3], Vt[9], R[9];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) At[i * 3 + k] = Rin[k * 3 + i];
  jacobi_small<3, 3>(At, W, Vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double a = 0;
      for (int k = 0; k < 3; k++) a += At[k * 3 + i] * Vt[k * 3 + j];
      R[i * 3 + j] = a;
    }
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
    double t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
    t = (R[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
    t = (R[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
    if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
    theta /= sqrt(rx * rx + ry * ry + rz * rz);
    r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
  }
}

__device__ void project_point(const Cam& cam, const double R[9], const double t[3], const double* X,
                              double* u, double* v) {
  double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  z = z ? 1. / z : 1;
  x *= z; y *= z;
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  const double cdist = 1 + cam.k[0] * r2 + cam.k[1] * r4 + cam.k[4] * r6;
  const double xd = x * cdist + cam.k[2] * a1 + cam.k[3] * a2;
  const double yd = y * cdist + cam.k[2] * a3 + cam.k[3] * a1;
  *u = xd * cam.fx + cam.cx;
  *v = yd * cam.fy + cam.cy;
}

__device__ void undistort_point(const Cam& cam, double u, double v, double* xo, double* yo) {
  const double x0 = (u - cam.cx) * (1. / cam.fx), y0 = (v - cam.cy) * (1. / cam.fy);
  double x = x0, y = y0;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((cam.k[4] * r2 + cam.k[1]) * r2 + cam.k[0]) * r2);
    if (icdist < 0) { x = x0; y = y0; break; }
    const double dx = 2 * cam.k[2] * x * y + cam.k[3] * (r2 + 2 * x * x);
    const double dy = cam.k[2] * (r2 + 2 * y * y) + 2 * cam.k[3] * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  *xo = x; *yo = y;
}

// ---- point set of one EPnP call: 5 packed indices (draw order) or an inlier bit mask ----------
struct PtSet {
  unsigned long long mask;
  unsigned packed;
  int n;
  bool hyp;
};
struct PtIter {
  unsigned long long m; unsigned packed; bool hyp;
  __device__ explicit PtIter(const PtSet& s) : m(s.mask), packed(s.packed), hyp(s.hyp) {}
  __device__ int next() {
    if (hyp) { const int i = packed & 63; packed >>= 6; return i; }
    const int i = __builtin_ctzll(m); m &= m - 1; return i;
  }
};

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double dist2(const double* a, const double* b) {
  return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

struct EpnpCtx {
  const double* obj;  // LDS: compacted object points (float32-rounded) [n][3]
  const double* us;   // LDS: undistorted*f+c image points [n][2]
  double uc, vc, fu, fv;
  double cws[4][3];
  double cc_inv[9];
};

__device__ __forceinline__ void alphas_of(const EpnpCtx& e, const double* pi, double a[4]) {
  for (int j = 0; j < 3; j++)
    a[1 + j] = e.cc_inv[3 * j] * (pi[0] - e.cws[0][0]) + e.cc_inv[3 * j + 1] * (pi[1] - e.cws[0][1]) +
               e.cc_inv[3 * j + 2] * (pi[2] - e.cws[0][2]);
  a[0] = 1.0f - a[1] - a[2] - a[3];
}

// qr_solve of epnp.cpp for the 6x4 Gauss-Newton system (row-scan quirk of `eta` kept)
__device__ void qr_solve_6x4(double* pA, double* pb, double* pX) {
  constexpr int nr = 6, nc = 4;
  double A1[4], A2[4];
  double* ppAkk = pA;
  for (int k = 0; k < nc; k++) {
    double* ppAik1 = ppAkk; double eta = fabs(*ppAik1);
    for (int i = k + 1; i < nr; i++) {
      const double elt = fabs(*ppAik1);
      if (eta < elt) eta = elt;
      ppAik1 += nc;
    }
    if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
    double* ppAik2 = ppAkk; double sum2 = 0.0; const double inv_eta = 1. / eta;
    for (int i = k; i < nr; i++) { *ppAik2 *= inv_eta; sum2 += *ppAik2 * *ppAik2; ppAik2 += nc; }
    double sigma = sqrt(sum2);
    if (*ppAkk < 0) sigma = -sigma;
    *ppAkk += sigma;
    A1[k] = sigma * *ppAkk;
    A2[k] = -eta * sigma;
    for (int j = k + 1; j < nc; j++) {
      double* ppAik = ppAkk; double sum = 0;
      for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
      const double tau = sum / A1[k];
      ppAik = ppAkk;
      for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
    }
    ppAkk += nc + 1;
  }
  double* ppAjj = pA;
  for (int j = 0; j < nc; j++) {
    double* ppAij = ppAjj; double tau = 0;
    for (int i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
    tau /= A1[j];
    ppAij = ppAjj;
    for (int i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
    ppAjj += nc + 1;
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
  for (int i = nc - 2; i >= 0; i--) {
    double* ppAij = pA + i * nc + (i + 1); double sum = 0;
    for (int j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
    pX[i] = (pb[i] - sum) / A2[i];
  }
}

__device__ void gauss_newton(const double* L, const double* rho, double betas[4]) {
  for (int k = 0; k < 5; k++) {
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
      const double* rl = L + i * 10;
      double* ra = a + i * 4;
      ra[0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
      ra[1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
      ra[2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
      ra[3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
      b[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                       rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                       rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                       rl[9] * betas[3] * betas[3]);
    }
    qr_solve_6x4(a, b, x);
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
}

// compute_ccs + compute_pcs + solve_for_sign + estimate_R_and_t + reprojection_error
__device__ double compute_R_and_t(const EpnpCtx& e, const PtSet& ps, const double* ut, const double* betas,
                                  double R[9], double t[3]) {
  double ccs[4][3];
  for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * UT(11 - i, 3 * j + k);
  auto pc_of = [&](const double* a, double pc[3]) {
    for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
  };
  {  // solve_for_sign: depth of the first point
    PtIter it(ps);
    double a[4], pc[3];
    alphas_of(e, e.obj + 3 * it.next(), a);
    pc_of(a, pc);
    if (pc[2] < 0.0)
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
  }
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  {
    PtIter it(ps);
    for (int i = 0; i < ps.n; i++) {
      const double* pw = e.obj + 3 * it.next();
      double a[4], pc[3];
      alphas_of(e, pw, a); pc_of(a, pc);
      for (int j = 0; j < 3; j++) { pc0[j] += pc[j]; pw0[j] += pw[j]; }
    }
  }
  for (int j = 0; j < 3; j++) { pc0[j] /= ps.n; pw0[j] /= ps.n; }
  double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    PtIter it(ps);
    for (int i = 0; i < ps.n; i++) {
      const double* pw = e.obj + 3 * it.next();
      double a[4], pc[3];
      alphas_of(e, pw, a); pc_of(a, pc);
      for (int j = 0; j < 3; j++) {
        abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
        abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
        abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
      }
    }
  }
  double At[9], W[3], Vt[9];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) At[i * 3 + k] = abt[k * 3 + i];
  jacobi_small<3, 3>(At, W, Vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += At[k * 3 + i] * Vt[k * 3 + j];
      R[i * 3 + j] = s;
    }
  const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] -
                     R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
  if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
  t[0] = pc0[0] - dot3(R, pw0);
  t[1] = pc0[1] - dot3(R + 3, pw0);
  t[2] = pc0[2] - dot3(R + 6, pw0);
  double sum2 = 0.0;
  {
    PtIter it(ps);
    for (int i = 0; i < ps.n; i++) {
      const int idx = it.next();
      const double* pw = e.obj + 3 * idx;
      const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1];
      const double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
      const double ue = e.uc + e.fu * Xc * inv_Zc, ve = e.vc + e.fv * Yc * inv_Zc;
      const double u = e.us[2 * idx], v = e.us[2 * idx + 1];
      sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
  }
  return sum2 / ps.n;
}

// epnp::compute_pose followed by Rodrigues(R) (solvePnP, SOLVEPNP_EPNP branch)
__device__ void solve_epnp(EpnpCtx& e, const PtSet& ps, double* ut, double rvec[3], double tvec[3]) {
  const int n = ps.n;
  // choose_control_points
  e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
  { PtIter it(ps); for (int i = 0; i < n; i++) { const double* p = e.obj + 3 * it.next(); for (int j = 0; j < 3; j++) e.cws[0][j] += p[j]; } }
  for (int j = 0; j < 3; j++) e.cws[0][j] /= n;
  {
    double At[9], dc[3], vt[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0;
        PtIter it(ps);
        for (int i = 0; i < n; i++) { const double* p = e.obj + 3 * it.next(); s += (p[a] - e.cws[0][a]) * (p[b] - e.cws[0][b]); }
        At[b * 3 + a] = s;   // At = transpose (the matrix is symmetric)
      }
    jacobi_small<3, 3>(At, dc, vt);
    for (int i = 1; i < 4; i++) {
      const double k = sqrt(dc[i - 1] / n);
      for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * At[3 * (i - 1) + j];
    }
  }
  {  // compute_barycentric_coordinates: cvInvert(CC, CC_inv, CV_SVD)
    double cc[9];
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
    svd_solve<3, 3, 3>(cc, eye, e.cc_inv);
  }
  // M^T M accumulated row pair by row pair (same summation order as cvMulTransposed over M)
  for (int a = 0; a < 12; a++)
    for (int b = 0; b < 12; b++) UT(a, b) = 0;
  {
    PtIter it(ps);
    for (int i = 0; i < n; i++) {
      const int idx = it.next();
      double al[4];
      alphas_of(e, e.obj + 3 * idx, al);
      const double u = e.us[2 * idx], v = e.us[2 * idx + 1];
      double m1[12], m2[12];
      for (int a = 0; a < 4; a++) {
        m1[3 * a] = al[a] * e.fu; m1[3 * a + 1] = 0.0; m1[3 * a + 2] = al[a] * (e.uc - u);
        m2[3 * a] = 0.0; m2[3 * a + 1] = al[a] * e.fv; m2[3 * a + 2] = al[a] * (e.vc - v);
      }
      for (int a = 0; a < 12; a++)
        for (int b = a; b < 12; b++) {
          double s = UT(a, b);
          s += m1[a] * m1[b];
          s += m2[a] * m2[b];
          UT(a, b) = s;
        }
    }
    for (int a = 0; a < 12; a++)
      for (int b = 0; b < a; b++) UT(a, b) = UT(b, a);
  }
  jacobi12(ut);

  double L[60], rho[6];
  {  // compute_L_6x10
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = UT(11 - i, 3 * a) - UT(11 - i, 3 * b);
        dv[i][j][1] = UT(11 - i, 3 * a + 1) - UT(11 - i, 3 * b + 1);
        dv[i][j][2] = UT(11 - i, 3 * a + 2) - UT(11 - i, 3 * b + 2);
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = L + 10 * i;
      row[0] = dot3(dv[0][i], dv[0][i]);
      row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
      row[2] = dot3(dv[1][i], dv[1][i]);
      row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
      row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
      row[5] = dot3(dv[2][i], dv[2][i]);
      row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
      row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
      row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
      row[9] = dot3(dv[3][i], dv[3][i]);
    }
  }
  rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
  rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);

  double betas[4], bestR[9], bestt[3], best_err = 0;
  for (int N = 1; N <= 3; N++) {
    if (N == 1) {  // find_betas_approx_1: columns {0,1,3,6}
      double l[24], b4[4];
      for (int i = 0; i < 6; i++) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
      svd_solve<6, 4, 1>(l, rho, b4);
      if (b4[0] < 0) { betas[0] = sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
      else { betas[0] = sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0]; }
    } else if (N == 2) {  // columns {0,1,2}
      double l[18], b3[3];
      for (int i = 0; i < 6; i++) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
      svd_solve<6, 3, 1>(l, rho, b3);
      if (b3[0] < 0) { betas[0] = sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
      else { betas[0] = sqrt(b3[0]); betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
      if (b3[1] < 0) betas[0] = -betas[0];
      betas[2] = 0.0; betas[3] = 0.0;
    } else {  // columns {0,1,2,3,4}
      double l[30], b5[5];
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 5; j++) l[5 * i + j] = L[10 * i + j];
      svd_solve<6, 5, 1>(l, rho, b5);
      if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
      else { betas[0] = sqrt(b5[0]); betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
      if (b5[1] < 0) betas[0] = -betas[0];
      betas[2] = b5[3] / betas[0];
      betas[3] = 0.0;
    }
    gauss_newton(L, rho, betas);
    double R[9], t[3];
    const double err = compute_R_and_t(e, ps, ut, betas, R, t);
    if (N == 1 || err < best_err) {   // rep_errors[2] < [1], then [3] < [N]: strict '<' keeps the earlier one
      best_err = err;
      for (int k = 0; k < 9; k++) bestR[k] = R[k];
      for (int k = 0; k < 3; k++) bestt[k] = t[k];
    }
  }
  for (int k = 0; k < 3; k++) tvec[k] = bestt[k];
  rodrigues_mat2vec(bestR, rvec);
}
#undef UT

// ------------------------------------------------------------------------------------------------------------------
// Exactly four usable landmarks: cv2.solvePnPRansac switches its kernel (OpenCV 3.4 solvepnp.cpp: `npoints == 4` ->
// model_points = 4, SOLVEPNP_P3P, and model_points == npoints -> solvePnP directly, no RANSAC), i.e. p3p.cpp: Gao's P3P
// on the first three correspondences -- a quartic in x = |PA| / |PC| (polynom_solver.cpp) -- gives up to four poses, each
// completed by Horn's quaternion alignment (p3p::align, jacobi_4x4); the fourth point picks the one it reprojects best.
// Here the quartic is solved once per wave and its (up to four) real roots go to lanes 0-3, which build, align and score
// their pose side by side; an ordered scan over those lanes replays the sequential "first minimum wins" choice.
// Reference call site: pose_estimation/export_predicted_poses_real.py:199-201.
// ------------------------------------------------------------------------------------------------------------------
__device__ int solve_deg2(double a, double b, double c, double& x1, double& x2) {
  const double delta = b * b - 4 * a * c;
  if (delta < 0) return 0;
  const double inv_2a = 0.5 / a;
  if (delta == 0) { x1 = -b * inv_2a; x2 = x1; return 1; }
  const double sq = sqrt(delta);
  x1 = (-b + sq) * inv_2a; x2 = (-b - sq) * inv_2a;
  return 2;
}
__device__ int solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
  if (a == 0) {
    if (b == 0) {
      if (c == 0) return 0;
      x0 = -d / c;
      return 1;
    }
    x2 = 0;
    return solve_deg2(b, c, d, x0, x1);
  }
  const double inv_a = 1. / a, b_a = inv_a * b, b_a2 = b_a * b_a, c_a = inv_a * c, d_a = inv_a * d;
  const double Q = (3 * c_a - b_a2) / 9, R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
  const double Q3 = Q * Q * Q, D = Q3 + R * R, b_a_3 = (1. / 3.) * b_a;
  if (Q == 0) {
    if (R == 0) { x0 = x1 = x2 = -b_a_3; return 3; }
    x0 = pow(2 * R, 1 / 3.0) - b_a_3;
    return 1;
  }
  if (D <= 0) {
    const double theta = acos(R / sqrt(-Q3)), sqrt_Q = sqrt(-Q);
    x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
    x1 = 2 * sqrt_Q * cos((theta + 2 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
    x2 = 2 * sqrt_Q * cos((theta + 4 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
    return 3;
  }
  const double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
  const double BD = (AD == 0) ? 0 : -Q / AD;
  x0 = AD + BD - b_a_3;
  return 1;
}
__device__ int solve_deg4(double a, double b, double c, double d, double e, double x[4]) {
  if (a == 0) { x[3] = 0; return solve_deg3(b, c, d, e, x[0], x[1], x[2]); }
  const double inv_a = 1. / a;
  b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
  const double b2 = b * b, bc = b * c, b3 = b2 * b;
  double r0, r1, r2;
  if (solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2) == 0) return 0;
  const double R2 = 0.25 * b2 - c + r0;
  if (R2 < 0) return 0;
  const double R = sqrt(R2), inv_R = 1. / R;
  double D2, E2;
  if (R < 10E-12) {
    const double temp = r0 * r0 - 4 * e;
    if (temp < 0) D2 = E2 = -1;
    else {
      const double st = sqrt(temp);
      D2 = 0.75 * b2 - 2 * c + 2 * st;
      E2 = D2 - 4 * st;
    }
  } else {
    const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
    D2 = u + v; E2 = u - v;
  }
  const double b_4 = 0.25 * b, R_2 = 0.5 * R;
  int nb = 0;
  if (D2 >= 0) {
    const double D = sqrt(D2);
    nb = 2;
    x[0] = R_2 + 0.5 * D - b_4;
    x[1] = x[0] - D;
  }
  if (E2 >= 0) {
    const double E = sqrt(E2);
    if (nb == 0) { x[0] = -R_2 + 0.5 * E - b_4; x[1] = x[0] - E; nb = 2; }
    else { x[2] = -R_2 + 0.5 * E - b_4; x[3] = x[2] - E; nb = 4; }
  }
  return nb;
}

// p3p::jacobi_4x4 (cyclic Jacobi, symmetric 4 x 4, upper triangle used and destroyed)
__device__ void jacobi_4x4(double* A, double* D, double* U) {
  double B[4], Z[4];
  for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1. : 0.;
  B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
  for (int i = 0; i < 4; i++) { D[i] = B[i]; Z[i] = 0; }
  for (int iter = 0; iter < 50; iter++) {
    const double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
    if (sum == 0.0) return;
    const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
    for (int i = 0; i < 3; i++)
      for (int j = i + 1; j < 4; j++) {
        double& Aij_ref = A[4 * i + j];
        const double Aij = Aij_ref, eps_machine = 100.0 * fabs(Aij);
        if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) {
          Aij_ref = 0.0;
        } else if (fabs(Aij) > tresh) {
          double hh = D[j] - D[i], t;
          if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
          else {
            const double theta = 0.5 * hh / Aij;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          hh = t * Aij;
          Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
          Aij_ref = 0.0;
          const double c = 1.0 / sqrt(1 + t * t), sn = t * c, tau = sn / (1.0 + c);
          auto rot = [&](double& g_ref, double& h_ref) {
            const double g = g_ref, h = h_ref;
            g_ref = g - sn * (h + g * tau);
            h_ref = h + sn * (g - h * tau);
          };
          for (int k = 0; k <= i - 1; k++) rot(A[k * 4 + i], A[k * 4 + j]);
          for (int k = i + 1; k <= j - 1; k++) rot(A[i * 4 + k], A[k * 4 + j]);
          for (int k = j + 1; k < 4; k++) rot(A[i * 4 + k], A[j * 4 + k]);
          for (int k = 0; k < 4; k++) rot(U[k * 4 + i], U[k * 4 + j]);
        }
      }
    for (int i = 0; i < 4; i++) { B[i] += Z[i]; D[i] = B[i]; Z[i] = 0; }
  }
}

// p3p::align: the rigid motion that takes the three world points X (rows) onto the camera-frame points M
__device__ void p3p_align(const double M[3][3], const double* X, double R[9], double T[3]) {
  double Cs[3], Ce[3], sm[9], Qs[16], evs[4], U[16];
  for (int i = 0; i < 3; i++) { Ce[i] = (M[0][i] + M[1][i] + M[2][i]) / 3; Cs[i] = (X[i] + X[3 + i] + X[6 + i]) / 3; }
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) sm[i * 3 + j] = (X[i] * M[0][j] + X[3 + i] * M[1][j] + X[6 + i] * M[2][j]) / 3 - Ce[j] * Cs[i];
  Qs[0] = sm[0] + sm[4] + sm[8];
  Qs[5] = sm[0] - sm[4] - sm[8];
  Qs[10] = sm[4] - sm[8] - sm[0];
  Qs[15] = sm[8] - sm[0] - sm[4];
  Qs[4] = Qs[1] = sm[5] - sm[7];
  Qs[8] = Qs[2] = sm[6] - sm[2];
  Qs[12] = Qs[3] = sm[1] - sm[3];
  Qs[9] = Qs[6] = sm[3] + sm[1];
  Qs[13] = Qs[7] = sm[6] + sm[2];
  Qs[14] = Qs[11] = sm[7] + sm[5];
  jacobi_4x4(Qs, evs, U);
  int i_ev = 0;
  double ev_max = evs[0];
  for (int i = 1; i < 4; i++)
    if (evs[i] > ev_max) { ev_max = evs[i]; i_ev = i; }
  double q[4];
  for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
  const double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
  const double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3], q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
  R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q1_2 - q0_3); R[2] = 2. * (q1_3 + q0_2);
  R[3] = 2. * (q1_2 + q0_3); R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q2_3 - q0_1);
  R[6] = 2. * (q1_3 - q0_2); R[7] = 2. * (q2_3 + q0_1); R[8] = q02 + q32 - q12 - q22;
  for (int i = 0; i < 3; i++) T[i] = Ce[i] - (R[3 * i] * Cs[0] + R[3 * i + 1] * Cs[1] + R[3 * i + 2] * Cs[2]);
}

// obj: 4 x 3 (float32-rounded landmarks), upx: 4 x 2 undistorted image points, float32-rounded, mapped back to pixels
// (p3p::extract_points).  Returns true and (rvec, tvec) wave-uniformly, or false when the first three points admit no pose.
__device__ bool solve_p3p(const Cam& cam, const double* obj, const double* upx, int lane, double rvec[3], double tvec[3]) {
  const double inv_fx = 1. / cam.fx, inv_fy = 1. / cam.fy, cx_fx = cam.cx / cam.fx, cy_fy = cam.cy / cam.fy;
  double m[3][3];
  for (int i = 0; i < 3; i++) {
    const double u = inv_fx * upx[2 * i] - cx_fx, v = inv_fy * upx[2 * i + 1] - cy_fy;
    const double mk = 1. / sqrt(u * u + v * v + 1);
    m[i][0] = u * mk; m[i][1] = v * mk; m[i][2] = mk;
  }
  auto dist3 = [&](int i, int j) {
    const double dx = obj[3 * i] - obj[3 * j], dy = obj[3 * i + 1] - obj[3 * j + 1], dz = obj[3 * i + 2] - obj[3 * j + 2];
    return sqrt(dx * dx + dy * dy + dz * dz);
  };
  const double d0 = dist3(1, 2), d1 = dist3(0, 2), d2 = dist3(0, 1);
  const double c0 = m[1][0] * m[2][0] + m[1][1] * m[2][1] + m[1][2] * m[2][2];
  const double c1 = m[0][0] * m[2][0] + m[0][1] * m[2][1] + m[0][2] * m[2][2];
  const double c2 = m[0][0] * m[1][0] + m[0][1] * m[1][1] + m[0][2] * m[1][2];
  // p3p::solve_for_lengths, wave-uniform part: the quartic
  const double p = c0 * 2, q = c1 * 2, r = c2 * 2;
  const double inv_d22 = 1. / (d2 * d2), a = inv_d22 * (d0 * d0), b = inv_d22 * (d1 * d1);
  const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, pr = p * r, pqr = q * pr;
  if (p2 + q2 + r2 - pqr - 1 == 0) return false;
  const double ab = a * b, a_2 = 2 * a;
  const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
  if (A == 0) return false;
  const double a_4 = 4 * a;
  const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
  const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
  const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
  const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
  const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
  const double b0 = b * temp * temp;
  if (b0 == 0) return false;
  double roots[4] = {0, 0, 0, 0};
  const int nroots = solve_deg4(A, B, C, D, E, roots);
  if (nroots == 0) return false;
  // lane i < nroots: root i -> lengths -> alignment -> reprojection of the fourth point
  const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q, inv_b0 = 1. / b0;
  const int li = lane & 3;
  const double x = li == 0 ? roots[0] : li == 1 ? roots[1] : li == 2 ? roots[2] : roots[3];
  bool valid = lane < nroots && x > 0;
  double Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Tl[3] = {0, 0, 0}, reproj = 0;
  if (valid) {
    const double x2 = x * x;
    const double b1 =
        ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
        (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
          (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
         (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
          pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
         2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
         p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    valid = b1 > 0;
    if (valid) {
      const double y = inv_b0 * b1, v = x2 + y * y - x * y * r;
      valid = v > 0;
      if (valid) {
        const double Z = d2 / sqrt(v), len[3] = {x * Z, y * Z, Z};
        double M[3][3];
        for (int j = 0; j < 3; j++) { M[j][0] = len[j] * m[j][0]; M[j][1] = len[j] * m[j][1]; M[j][2] = len[j] * m[j][2]; }
        p3p_align(M, obj, Rl, Tl);
        const double X3 = obj[9], Y3 = obj[10], Z3 = obj[11];
        const double X3p = Rl[0] * X3 + Rl[1] * Y3 + Rl[2] * Z3 + Tl[0];
        const double Y3p = Rl[3] * X3 + Rl[4] * Y3 + Rl[5] * Z3 + Tl[1];
        const double Z3p = Rl[6] * X3 + Rl[7] * Y3 + Rl[8] * Z3 + Tl[2];
        const double du = cam.cx + cam.fx * X3p / Z3p - upx[6], dv = cam.cy + cam.fy * Y3p / Z3p - upx[7];
        reproj = du * du + dv * dv;
      }
    }
  }
  // the sequential choice of p3p::solve: first solution, then any later one with a strictly smaller reprojection error
  const unsigned long long vm = __ballot(valid) & 0xfULL;
  if (vm == 0ULL) return false;
  int ns = -1;
  double min_reproj = 0;
  for (int i = 0; i < 4; i++) {
    if (!((vm >> i) & 1ULL)) continue;
    const double ri = __shfl(reproj, i, 64);
    if (ns < 0 || min_reproj > ri) { ns = i; min_reproj = ri; }
  }
  double Rb[9];
  for (int k = 0; k < 9; k++) Rb[k] = __shfl(Rl[k], ns, 64);
  for (int k = 0; k < 3; k++) tvec[k] = __shfl(Tl[k], ns, 64);
  rodrigues_mat2vec(Rb, rvec);
  return true;
}

__device__ int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
  ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
  double num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
  double denom = 1. - det_powi(1. - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = det_log(num);
  denom = det_log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// One wave per frame, a few frames per workgroup (blockDim = 64 x frames; the waves share nothing but the CU).  A wave of
// this kernel owns a whole SIMD (512 registers: the spills of the 12 x 12 solves live in AGPRs), so a CU that holds even one
// frame cannot take a convolution workgroup of the next forward, which bench.py runs beside this kernel.
__global__ __launch_bounds__(256) void pnp_kernel(const PnpArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem_all[];
  const int J = a.J;                         // the per-point arrays are sized by the launch's J (pnp_launch), not by kMaxJ
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* smem = smem_all + (size_t)wv * (144 * kPW + J * 9);
  double* s_ut = smem;                       // 144 x kPW
  double* s_obj = s_ut + 144 * kPW;          // J x 3
  double* s_u32 = s_obj + J * 3;             // J x 2  (undistorted, float32-rounded, * f + c)
  double* s_u64 = s_u32 + J * 2;             // J x 2
  double* s_img = s_u64 + J * 2;             // J x 2  raw float32 image points

  const int lane = threadIdx.x & 63;
  const int frame_raw = blockIdx.x * (blockDim.x >> 6) + wv;
  const bool frame_valid = frame_raw < a.N;
  const int frame = frame_valid ? frame_raw : a.N - 1;   // surplus waves of the last workgroup redo the last frame up to the barrier
  Cam cam;
  cam.fx = a.K[0]; cam.fy = a.K[4]; cam.cx = a.K[2]; cam.cy = a.K[5];
  for (int i = 0; i < 5; i++) cam.k[i] = a.dist ? a.dist[i] : 0.0;

  // ---- confidence filter (export_predicted_poses_real.py:186-197) ----
  const float* kp = a.kp + (size_t)frame * J * 3;
  const float conf = lane < J ? kp[lane * 3 + 2] : -__builtin_inff();
  double thr64 = a.conf_thr0;
  unsigned long long sel;
  for (int it = 0;;) {
    const float thr = (float)thr64;
    sel = __ballot(lane < J && conf > thr);
    if (__popcll(sel) >= a.min_pts) break;
    thr64 *= a.thr_decay;
    if (++it >= a.thr_iters) { sel = __ballot(lane < J && conf > (float)thr64); break; }
  }
  const int n = __popcll(sel);
  if ((sel >> lane) & 1ULL) {   // compact in landmark order; float32 roundings of solvePnPRansac's convertTo
    const int pos = __popcll(sel & ((1ULL << lane) - 1ULL));
    for (int k = 0; k < 3; k++) s_obj[pos * 3 + k] = (double)(float)a.landmarks[lane * 3 + k];
    const double u = (double)kp[lane * 3], v = (double)kp[lane * 3 + 1];
    double x, y;
    undistort_point(cam, u, v, &x, &y);
    s_u64[pos * 2] = x * cam.fx + cam.cx; s_u64[pos * 2 + 1] = y * cam.fy + cam.cy;
    s_u32[pos * 2] = (double)(float)x * cam.fx + cam.cx; s_u32[pos * 2 + 1] = (double)(float)y * cam.fy + cam.cy;
    s_img[pos * 2] = u; s_img[pos * 2 + 1] = v;
  }
  __syncthreads();                           // (the only barrier: each wave's LDS staging is visible to its own lanes)
  if (!frame_valid) return;

  double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
  int status;
  EpnpCtx e;
  e.obj = s_obj; e.uc = cam.cx; e.vc = cam.cy; e.fu = cam.fx; e.fv = cam.fy;
  double* ut = s_ut + (lane & (kPW - 1));    // wave-uniform solves: lanes l and l + kPW write the same values to the same slots
  const int model_points = 5;

  if (n < 4) {
    status = -1;
  } else if (n == 4) {
    // OpenCV switches to its P3P kernel for exactly four points and skips RANSAC (solve_p3p above); no pose -> -2 like a failed RANSAC
    status = solve_p3p(cam, s_obj, s_u32, lane, rvec, tvec) ? 4 : -2;
  } else if (n == model_points) {
    PtSet ps{(1ULL << n) - 1ULL, 0u, n, false};
    e.us = s_u32;
    solve_epnp(e, ps, ut, rvec, tvec);
    status = n;
  } else {
    int niters = a.max_iters > 1 ? a.max_iters : 1;
    int max_good = 0;
    unsigned long long best_mask = 0;
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    CvRng rng{~0ULL};
    const float t2 = (float)(a.reproj_err * a.reproj_err);
    // Speculative final fit.  solvePnPRansac ends with one more EPnP over the inlier set (export_predicted_poses_real.py:199,
    // OpenCV solvepnp.cpp), a serial tail as long as a whole hypothesis batch.  The inlier set is almost always "every
    // point" or "every point but one", so in the FIRST batch the top n + 1 lanes solve exactly those sets (same
    // function, same points in the same order, s_u64 image points) beside the 5-point hypotheses of the other lanes.
    // If RANSAC ends on one of them, that lane's pose IS the final fit, bit for bit; otherwise the fit runs as before.
    // The hypothesis stream is unchanged: batches only group consecutive iterations, and the ordered scan below
    // replays the sequential accept / RANSACUpdateNumIters rule whatever the group sizes are.
    const unsigned long long full_mask = (n >= 64) ? ~0ULL : ((1ULL << n) - 1ULL);
    const int nspec = n + 1 <= kPW / 2 ? n + 1 : 1;
    const int sidx = (kPW - 1) - lane;                            // 0: full set, i: full set without point i - 1
    const bool spec_lane = sidx >= 0 && sidx < nspec;
    const unsigned long long spec_mask = full_mask & ~(sidx > 0 && spec_lane ? 1ULL << (sidx - 1) : 0ULL);
    double spec_r[3] = {0, 0, 0}, spec_t[3] = {0, 0, 0};
    for (int iter0 = 0; iter0 < niters;) {
      const bool first = iter0 == 0;
      const int nb = min(first ? kPW - nspec : kPW, niters - iter0);
      unsigned my_packed = 0;
      for (int k = 0; k < nb; k++) {   // getSubset: duplicate-free draws, one shared RNG stream
        unsigned packed = 0;
        int idx[5];
        for (int i = 0; i < model_points; i++) {
          int idx_i;
          for (;;) {
            idx_i = idx[i] = rng.uniform(0, n);
            int j = 0;
            for (; j < i; j++) if (idx_i == idx[j]) break;
            if (j == i) break;
          }
          packed |= (unsigned)idx_i << (6 * i);
        }
        if (k == lane) my_packed = packed;
      }
      double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
      unsigned long long mask = 0;
      int good = 0;
      const bool hyp_lane = lane < nb, spec_now = first && spec_lane;
      if (hyp_lane || spec_now) {   // ONE call site: hypothesis and speculative lanes run side by side, not one after the other
        PtSet ps{spec_now ? spec_mask : 0ULL, my_packed, spec_now ? (int)__popcll(spec_mask) : model_points, !spec_now};
        e.us = spec_now ? s_u64 : s_u32;
        solve_epnp(e, ps, ut, r, t);
      }
      if (spec_now) { for (int k = 0; k < 3; k++) { spec_r[k] = r[k]; spec_t[k] = t[k]; } }
      if (hyp_lane) {
        double R[9];
        rodrigues_vec2mat(r, R);
        for (int i = 0; i < n; i++) {   // PnPRansacCallback::computeError + findInliers
          double u, v;
          project_point(cam, R, t, s_obj + 3 * i, &u, &v);
          const float pu = (float)u, pv = (float)v;
          const float dx = (float)s_img[2 * i] - pu, dy = (float)s_img[2 * i + 1] - pv;
          const float err = (float)((double)dx * dx + (double)dy * dy);
          if (err <= t2) { mask |= 1ULL << i; good++; }
        }
      }
      // scan the batch in iteration order with OpenCV's acceptance rule
      int best_lane = -1;
      for (int k = 0; k < nb && iter0 + k < niters; k++) {
        const int g = __shfl(good, k, 64);
        if (g > max(max_good, model_points - 1)) {
          max_good = g;
          best_lane = k;
          niters = ransac_update_num_iters(a.confidence, (double)(n - g) / n, model_points, niters);
        }
      }
      if (best_lane >= 0) {
        best_mask = __shfl(mask, best_lane, 64);
        for (int k = 0; k < 3; k++) { best_r[k] = __shfl(r[k], best_lane, 64); best_t[k] = __shfl(t[k], best_lane, 64); }
      }
      iter0 += nb;
    }
    if (max_good <= 0) {
      status = -2;
    } else {
      const unsigned long long hit = __ballot(spec_lane && spec_mask == best_mask);
      if (hit != 0ULL && !SCP_DEV_ONLY(a.dbg_no_spec)) {   // the final fit has already been computed by a speculative lane
        const int src = __builtin_ctzll(hit);
        for (int k = 0; k < 3; k++) { rvec[k] = __shfl(spec_r[k], src, 64); tvec[k] = __shfl(spec_t[k], src, 64); }
        status = (int)__popcll(best_mask);
      } else {
        PtSet ps{best_mask, 0u, (int)__popcll(best_mask), false};
        e.us = s_u64;
        solve_epnp(e, ps, ut, rvec, tvec);   // wave-uniform
        status = ps.n;
      }
    }
  }

  if (lane == 0) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    // a degenerate final solve (non-finite rvec / tvec) is reported, not written: status -4, identity / zero pose like the other
    // failures -- one such frame must not put NaN into opencv_poses.json
    if (status > 0 && !(isfinite(rvec[0]) && isfinite(rvec[1]) && isfinite(rvec[2]) && isfinite(tvec[0]) && isfinite(tvec[1]) && isfinite(tvec[2]))) status = -4;
    if (status > 0) rodrigues_vec2mat(rvec, R);
    else { rvec[0] = rvec[1] = rvec[2] = 0; tvec[0] = tvec[1] = tvec[2] = 0; }
    if (a.rows) {   // the block a rank all-gathers / copies to the host: one row per frame, no assembly launches
      double* row = a.rows + (size_t)frame * 13;
      for (int k = 0; k < 9; k++) row[k] = R[k];
      for (int k = 0; k < 3; k++) row[9 + k] = tvec[k];
      row[12] = (double)status;
    } else {
      for (int k = 0; k < 9; k++) a.rot[(size_t)frame * 9 + k] = R[k];
      for (int k = 0; k < 3; k++) a.tvec[(size_t)frame * 3 + k] = tvec[k];
      a.status[frame] = status;
    }
    if (a.rvec) for (int k = 0; k < 3; k++) a.rvec[(size_t)frame * 3 + k] = rvec[k];
  }
}

int32_t pnp_launch(const float* kp_xyc, const double* landmarks, const double* K, const double* dist,
                   int N, int J, double conf_thr0, int min_pts, double thr_decay, int thr_iters,
                   int max_iters, double reproj_err, double confidence, double* rot, double* tvec,
                   double* rvec, int32_t* status, hipStream_t stream, double* rows) {
  SCP_REQUIRE(J >= 1 && J <= kMaxJ, "pnp: J=%d landmarks (1..%d)", J, kMaxJ);
  SCP_REQUIRE(confidence > 0 && confidence < 1, "pnp: confidence %g must be in (0,1)", confidence);
  SCP_REQUIRE(N >= 0, "pnp: N=%d", N);
  if (N == 0) return SCPOSE_OK;
  PnpArgs a{kp_xyc, landmarks, K, dist, rot, tvec, rvec, status, rows, N, J, conf_thr0, thr_decay, min_pts, thr_iters,
            max_iters, reproj_err, confidence, 0};
  { static const char* e = dev_env("SCPOSE_PNP_SPEC"); a.dbg_no_spec = (kDevBuild && e && atoi(e) == 0) ? 1 : 0; }
  // 36 864 B of work matrices + 72 B per landmark and frame (37 656 B at J = 11).  Two frames per workgroup: measured alone
  // (256 frames, 10 % outliers) 0.77 ms against 0.80 ms for one and 1.02 ms for four frames per workgroup (four waves
  // contend for one CU's LDS); beside the next forward all three cost the step the same 0.45-0.5 ms
  const size_t per_frame = (size_t)(144 * kPW + J * 9) * sizeof(double);
  int fpw = 2;
  { static const char* e = dev_env("SCPOSE_PNP_FPW"); if (kDevBuild && e && atoi(e) >= 1 && atoi(e) <= 4 && atoi(e) * per_frame <= 160 * 1024) fpw = atoi(e); }   // development: frames per workgroup
  const size_t lds = per_frame * fpw;
  static LdsOptIn big_lds;   // per device (common.h); opted in once for the whole LDS
  { const int32_t rc = lds_opt_in(reinterpret_cast<const void*>(pnp_kernel), 160 * 1024, &big_lds); if (rc != SCPOSE_OK) return rc; }
  hipLaunchKernelGGL(pnp_kernel, dim3((N + fpw - 1) / fpw), dim3(64 * fpw), lds, stream, a);
  SCP_CHECK_HIP(hipGetLastError());
  return SCPOSE_OK;
}

}  // namespace scpose
