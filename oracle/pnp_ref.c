/*
 * pnp_ref.c -- scalar C restatement of the reference's per-frame pose solve (TEST ORACLE,
 * not product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it).
 *
 * Path restated: pose_estimation/export_predicted_poses_real.py:186-203
 *     confidence filter (:186-197)  ->  cv2.solvePnPRansac(obj[mask], img[mask], K, dist,
 *     flags=SOLVEPNP_EPNP, iterationsCount=10000, reprojectionError=15.0) (:199-201)
 *     ->  cv2.Rodrigues(rvec) (:203)
 *
 * PARITY UNPINNED.  cv2 is opencv-python==3.4.11.41 (environment.yml:37,
 * landmark_regression/requirements.txt:2): third party, not vendored under /root/reference and
 * not installable in this image, and the reference holds no test vector for this call.  The
 * code below restates the published OpenCV 3.4 algorithm (modules/calib3d/src/solvepnp.cpp
 * solvePnPRansac + PnPRansacCallback, ptsetreg.cpp RANSACPointSetRegistrator, epnp.cpp,
 * calibration.cpp cvRodrigues2/cvProjectPoints2, imgproc/src/undistort.cpp undistortPoints,
 * core/src/lapack.cpp JacobiSVDImpl_/SVBkSb, core RNG) from knowledge of that source; it is
 * anchored only on analytic known-answer cases (tests/test_oracle_pnp.py).
 *
 * Notable OpenCV behaviours kept:
 *   - float64 object points are converted to float32 on entry to solvePnPRansac; image points
 *     arrive as float32; inside RANSAC everything is float32 in, float64 arithmetic;
 *   - minimal sample = 5 points for EPnP; with exactly 4 usable points solvePnPRansac calls solvePnP(SOLVEPNP_P3P)
 *     directly (p3p.cpp: Gao's P3P on the first three points, the fourth picks among up to four poses);
 *   - RNG is cv::RNG((uint64)-1) (MWC, coefficient 4164903690), uniform(0,count) = next % count;
 *   - undistortPoints: 5 fixed-point iterations; RANSAC error = squared pixel distance of the
 *     float32-rounded projection (with distortion), threshold 15^2, compared in float32;
 *   - a model is accepted when goodCount > max(best, 4); niters = RANSACUpdateNumIters(...);
 *   - final pose = EPnP on all inliers of the best model (float64 copies of the float32 data),
 *     no LM refinement; rvec = Rodrigues(R) and the caller's R = Rodrigues(rvec).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXPTS 64

/* ------------------------------------------------------------------------------------------ */
/* core: RNG, one-sided Jacobi SVD (JacobiSVDImpl_<double>), SVD back-substitution            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t state; } cvrng;
static unsigned rng_next(cvrng* r) {
  r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
  return (unsigned)r->state;
}
static int rng_uniform(cvrng* r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

/* At: n rows of length m (the COLUMNS of the m x n matrix A, m >= n); on return rows of At
 * are the left singular vectors u_i (length m), W the singular values (descending), Vt rows the
 * right singular vectors.  Mirrors lapack.cpp JacobiSVDImpl_ (eps = 10*DBL_EPSILON). */
/* hypot / log / integer pow as fixed sequences of IEEE-754 operations (this file is compiled with -ffp-contract=off;
 * + - * / sqrt are correctly rounded everywhere), instead of libm's: OpenCV calls hypot (lapack.cpp JacobiSVDImpl_),
 * log and pow (ptsetreg.cpp RANSACUpdateNumIters), whose last bits differ between C libraries.  The HIP kernel runs the
 * same sequences (csrc/pnp.hip), so the two agree bit for bit through the Jacobi sweeps; the difference to libm's values
 * is <= 2 ulp and only matters where the problem is ill-conditioned in the first place. */
static double det_hypot(double a, double b) {
  double hi, lo, r;
  a = fabs(a); b = fabs(b);
  hi = a > b ? a : b; lo = a > b ? b : a;
  if (hi == 0.) return 0.;
  r = lo / hi;
  return hi * sqrt(1. + r * r);
}
static double det_log(double x) {
  unsigned long long u;
  int e;
  double m, f, sq, z, pz;
  memcpy(&u, &x, 8);
  e = (int)((u >> 52) & 0x7ff);
  if (e == 0) { x *= 18014398509481984.; memcpy(&u, &x, 8); e = (int)((u >> 52) & 0x7ff) - 54; }
  e -= 1023;
  u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  memcpy(&m, &u, 8);
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  f = m - 1.; sq = f / (2. + f); z = sq * sq;
  pz = 1. / 27.;
  pz = pz * z + 1. / 25.; pz = pz * z + 1. / 23.; pz = pz * z + 1. / 21.; pz = pz * z + 1. / 19.;
  pz = pz * z + 1. / 17.; pz = pz * z + 1. / 15.; pz = pz * z + 1. / 13.; pz = pz * z + 1. / 11.;
  pz = pz * z + 1. / 9.; pz = pz * z + 1. / 7.; pz = pz * z + 1. / 5.; pz = pz * z + 1. / 3.;
  pz = pz * z + 1.;
  return (double)e * 0.6931471805599453 + 2. * sq * pz;
}
static double det_powi(double x, int n) {
  double r = 1.;
  int i;
  for (i = 0; i < n; i++) r = r * x;
  return r;
}

static void jacobi_svd(double* At, double* W, double* Vt, int m, int n) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  int i, j, k, iter, max_iter = m > 30 ? m : 30;
  for (i = 0; i < n; i++) {
    double sd = 0;
    for (k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
    W[i] = sd;
    for (k = 0; k < n; k++) Vt[i * n + k] = 0;
    Vt[i * n + i] = 1;
  }
  for (iter = 0; iter < max_iter; iter++) {
    int changed = 0;
    for (i = 0; i < n - 1; i++)
      for (j = i + 1; j < n; j++) {
        double *Ai = At + i * m, *Aj = At + j * m;
        double a = W[i], p = 0, b = W[j], c, s;
        for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        {
          double beta = a - b, gamma = det_hypot(p, beta);
          if (beta < 0) {
            double delta = (gamma - beta) * 0.5;
            s = sqrt(delta / gamma);
            c = p / (gamma * s * 2);
          } else {
            c = sqrt((gamma + beta) / (gamma * 2));
            s = p / (gamma * c * 2);
          }
        }
        a = b = 0;
        for (k = 0; k < m; k++) {
          double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = 1;
        {
          double *Vi = Vt + i * n, *Vj = Vt + j * n;
          for (k = 0; k < n; k++) {
            double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0; Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (i = 0; i < n; i++) {
    double sd = 0;
    for (k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
    W[i] = sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    for (k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      for (k = 0; k < m; k++) { t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
      for (k = 0; k < n; k++) { t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
    }
  }
  {
    cvrng rng = {0x12345678};
    for (i = 0; i < n; i++) {
      double sd = W[i], s;
      int ii;
      for (ii = 0; ii < 100 && sd <= minval; ii++) {
        /* zero singular value: random vector orthogonalised against the previous u's */
        const double val0 = 1. / m;
        for (k = 0; k < m; k++) {
          double val = (rng_next(&rng) & 256) != 0 ? val0 : -val0;
          At[i * m + k] = val;
        }
        for (iter = 0; iter < 2; iter++)
          for (j = 0; j < i; j++) {
            double asum = 0;
            sd = 0;
            for (k = 0; k < m; k++) sd += At[i * m + k] * At[j * m + k];
            for (k = 0; k < m; k++) {
              double t = At[i * m + k] - sd * At[j * m + k];
              At[i * m + k] = t;
              asum += fabs(t);
            }
            asum = asum > eps * 100 ? 1 / asum : 0;
            for (k = 0; k < m; k++) At[i * m + k] *= asum;
          }
        sd = 0;
        for (k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
        sd = sqrt(sd);
      }
      s = sd > minval ? 1 / sd : 0.;
      for (k = 0; k < m; k++) At[i * m + k] *= s;
    }
  }
}

/* x = pinv(A) b for A m x n (row-major, m >= n), b m x nb, x n x nb: cvSolve(.., CV_SVD) / SVBkSb */
static void svd_solve(const double* A, int m, int n, const double* b, int nb, double* x) {
  double At[12 * 12], W[12], Vt[12 * 12];
  int i, j, k;
  double thr = 0;
  for (i = 0; i < n; i++)
    for (k = 0; k < m; k++) At[i * m + k] = A[k * n + i];
  jacobi_svd(At, W, Vt, m, n);
  for (i = 0; i < n; i++) thr += W[i];
  thr *= DBL_EPSILON * 2;
  for (j = 0; j < n * nb; j++) x[j] = 0;
  for (i = 0; i < n; i++) {
    double wi = W[i];
    if (fabs(wi) <= thr) continue;
    wi = 1 / wi;
    for (j = 0; j < nb; j++) {
      double s = 0;
      for (k = 0; k < m; k++) s += At[i * m + k] * b[k * nb + j];
      s *= wi;
      for (k = 0; k < n; k++) x[k * nb + j] += s * Vt[i * n + k];
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* calib3d: Rodrigues, projectPoints, undistortPoints                                          */
/* ------------------------------------------------------------------------------------------ */
static void rodrigues_vec2mat(const double r[3], double R[9]) {
  double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) {
    memset(R, 0, 9 * sizeof(double));
    R[0] = R[4] = R[8] = 1;
    return;
  }
  {
    double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    double rx = r[0] * it, ry = r[1] * it, rz = r[2] * it;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    int k;
    for (k = 0; k < 9; k++) R[k] = c1 * rrt[k] + s * rxm[k];
    R[0] += c; R[4] += c; R[8] += c;
  }
}

static void rodrigues_mat2vec(const double Rin[9], double r[3]) {
  double At[9], W[3], Vt[9], R[9];
  int i, j, k;
  double rx, ry, rz, s, c, theta;
  for (i = 0; i < 3; i++)
    for (k = 0; k < 3; k++) At[i * 3 + k] = Rin[k * 3 + i];
  jacobi_svd(At, W, Vt, 3, 3);
  for (i = 0; i < 3; i++)   /* R = U * Vt, U[:,k] = At row k */
    for (j = 0; j < 3; j++) {
      double a = 0;
      for (k = 0; k < 3; k++) a += At[k * 3 + i] * Vt[k * 3 + j];
      R[i * 3 + j] = a;
    }
  rx = R[7] - R[5]; ry = R[2] - R[6]; rz = R[3] - R[1];
  s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  theta = acos(c);
  if (s < 1e-5) {
    double t;
    if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
    t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
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

typedef struct { double fx, fy, cx, cy, k[5]; } camera;

static void project_point(const camera* cam, const double R[9], const double t[3], const double X[3],
                          double* u, double* v) {
  double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  double r2, r4, r6, a1, a2, a3, cdist, xd, yd;
  const double* k = cam->k;
  z = z ? 1. / z : 1;
  x *= z; y *= z;
  r2 = x * x + y * y; r4 = r2 * r2; r6 = r4 * r2;
  a1 = 2 * x * y; a2 = r2 + 2 * x * x; a3 = r2 + 2 * y * y;
  cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
  xd = x * cdist + k[2] * a1 + k[3] * a2;   /* icdist2 = 1 (k4..k6 = 0), no thin-prism / tilt */
  yd = y * cdist + k[2] * a3 + k[3] * a1;
  *u = xd * cam->fx + cam->cx;
  *v = yd * cam->fy + cam->cy;
}

static void undistort_point(const camera* cam, double u, double v, double* xo, double* yo) {
  const double* k = cam->k;
  double x0, y0, x, y;
  int j;
  x0 = x = (u - cam->cx) * (1. / cam->fx);
  y0 = y = (v - cam->cy) * (1. / cam->fy);
  for (j = 0; j < 5; j++) {
    double r2 = x * x + y * y;
    double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
    double dx, dy;
    if (icdist < 0) { x = x0; y = y0; break; }
    dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
    dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  *xo = x; *yo = y;
}

/* ------------------------------------------------------------------------------------------ */
/* epnp.cpp                                                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  double uc, vc, fu, fv;
  int n;
  double pws[3 * MAXPTS], us[2 * MAXPTS], alphas[4 * MAXPTS], pcs[3 * MAXPTS];
  double cws[4][3], ccs[4][3];
} epnp;

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* a, const double* b) {
  return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

static void choose_control_points(epnp* e) {
  int i, j, n = e->n;
  double pw0tpw0[9] = {0}, dc[3], vt[9];
  e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
  for (i = 0; i < n; i++)
    for (j = 0; j < 3; j++) e->cws[0][j] += e->pws[3 * i + j];
  for (j = 0; j < 3; j++) e->cws[0][j] /= n;
  {  /* cvMulTransposed(PW0, PW0tPW0, 1): PW0^T * PW0 */
    int a, b;
    for (a = 0; a < 3; a++)
      for (b = 0; b < 3; b++) {
        double s = 0;
        for (i = 0; i < n; i++) s += (e->pws[3 * i + a] - e->cws[0][a]) * (e->pws[3 * i + b] - e->cws[0][b]);
        pw0tpw0[3 * a + b] = s;
      }
  }
  {  /* cvSVD(.., U_T): rows of uct = left singular vectors */
    double At[9];
    for (i = 0; i < 3; i++)
      for (j = 0; j < 3; j++) At[i * 3 + j] = pw0tpw0[j * 3 + i];
    jacobi_svd(At, dc, vt, 3, 3);
    for (i = 1; i < 4; i++) {
      double k = sqrt(dc[i - 1] / n);
      for (j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * At[3 * (i - 1) + j];
    }
  }
}

static void compute_barycentric_coordinates(epnp* e) {
  double cc[9], cc_inv[9], eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  int i, j;
  for (i = 0; i < 3; i++)
    for (j = 1; j < 4; j++) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
  svd_solve(cc, 3, 3, eye, 3, cc_inv); /* cvInvert(CC, CC_inv, CV_SVD) */
  for (i = 0; i < e->n; i++) {
    const double* pi = e->pws + 3 * i;
    double* a = e->alphas + 4 * i;
    for (j = 0; j < 3; j++)
      a[1 + j] = cc_inv[3 * j] * (pi[0] - e->cws[0][0]) + cc_inv[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                 cc_inv[3 * j + 2] * (pi[2] - e->cws[0][2]);
    a[0] = 1.0f - a[1] - a[2] - a[3];
  }
}

static void compute_L_6x10(const double* ut, double* l) {
  const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
  double dv[4][6][3];
  int i, j;
  for (i = 0; i < 4; i++) {
    int a = 0, b = 1;
    for (j = 0; j < 6; j++) {
      dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
      dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
      dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
      b++;
      if (b > 3) { a++; b = a + 1; }
    }
  }
  for (i = 0; i < 6; i++) {
    double* row = l + 10 * i;
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

static void compute_rho(const epnp* e, double* rho) {
  rho[0] = dist2(e->cws[0], e->cws[1]); rho[1] = dist2(e->cws[0], e->cws[2]);
  rho[2] = dist2(e->cws[0], e->cws[3]); rho[3] = dist2(e->cws[1], e->cws[2]);
  rho[4] = dist2(e->cws[1], e->cws[3]); rho[5] = dist2(e->cws[2], e->cws[3]);
}

static void find_betas_approx_1(const double* L, const double* rho, double* betas) {
  double l[6 * 4], b4[4];
  int i;
  for (i = 0; i < 6; i++) {
    l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6];
  }
  svd_solve(l, 6, 4, rho, 1, b4);
  if (b4[0] < 0) {
    betas[0] = sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0];
  } else {
    betas[0] = sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0];
  }
}

static void find_betas_approx_2(const double* L, const double* rho, double* betas) {
  double l[6 * 3], b3[3];
  int i;
  for (i = 0; i < 6; i++) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
  svd_solve(l, 6, 3, rho, 1, b3);
  if (b3[0] < 0) {
    betas[0] = sqrt(-b3[0]);
    betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
  } else {
    betas[0] = sqrt(b3[0]);
    betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
  }
  if (b3[1] < 0) betas[0] = -betas[0];
  betas[2] = 0.0; betas[3] = 0.0;
}

static void find_betas_approx_3(const double* L, const double* rho, double* betas) {
  double l[6 * 5], b5[5];
  int i, j;
  for (i = 0; i < 6; i++)
    for (j = 0; j < 5; j++) l[5 * i + j] = L[10 * i + j];
  svd_solve(l, 6, 5, rho, 1, b5);
  if (b5[0] < 0) {
    betas[0] = sqrt(-b5[0]);
    betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
  } else {
    betas[0] = sqrt(b5[0]);
    betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
  }
  if (b5[1] < 0) betas[0] = -betas[0];
  betas[2] = b5[3] / betas[0];
  betas[3] = 0.0;
}

/* Householder QR solve of epnp::qr_solve, including its row-scan quirk (eta looks at rows
 * k..nr-2 only) -- eta is only a scale factor, so the quirk changes rounding, not results. */
static void qr_solve(double* pA, double* pb, double* pX, int nr, int nc) {
  double A1[6], A2[6];
  double* ppAkk = pA;
  int i, j, k;
  for (k = 0; k < nc; k++) {
    double *ppAik1 = ppAkk, eta = fabs(*ppAik1);
    for (i = k + 1; i < nr; i++) {
      double elt = fabs(*ppAik1);
      if (eta < elt) eta = elt;
      ppAik1 += nc;
    }
    if (eta == 0) {
      A1[k] = A2[k] = 0.0;
      return;
    } else {
      double *ppAik2 = ppAkk, sum2 = 0.0, inv_eta = 1. / eta, sigma;
      for (i = k; i < nr; i++) {
        *ppAik2 *= inv_eta;
        sum2 += *ppAik2 * *ppAik2;
        ppAik2 += nc;
      }
      sigma = sqrt(sum2);
      if (*ppAkk < 0) sigma = -sigma;
      *ppAkk += sigma;
      A1[k] = sigma * *ppAkk;
      A2[k] = -eta * sigma;
      for (j = k + 1; j < nc; j++) {
        double *ppAik = ppAkk, sum = 0, tau;
        for (i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
        tau = sum / A1[k];
        ppAik = ppAkk;
        for (i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
      }
    }
    ppAkk += nc + 1;
  }
  {
    double* ppAjj = pA;
    for (j = 0; j < nc; j++) {
      double *ppAij = ppAjj, tau = 0;
      for (i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
      tau /= A1[j];
      ppAij = ppAjj;
      for (i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
      ppAjj += nc + 1;
    }
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
  for (i = nc - 2; i >= 0; i--) {
    double *ppAij = pA + i * nc + (i + 1), sum = 0;
    for (j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
    pX[i] = (pb[i] - sum) / A2[i];
  }
}

static void gauss_newton(const double* L, const double* rho, double betas[4]) {
  int k, i;
  for (k = 0; k < 5; k++) {
    double a[6 * 4], b[6], x[4] = {0, 0, 0, 0};
    for (i = 0; i < 6; i++) {
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
    qr_solve(a, b, x, 6, 4);
    for (i = 0; i < 4; i++) betas[i] += x[i];
  }
}

static void estimate_R_and_t(epnp* e, double R[3][3], double t[3]) {
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0}, abt[9] = {0}, At[9], W[3], Vt[9];
  int i, j, k, n = e->n;
  for (i = 0; i < n; i++)
    for (j = 0; j < 3; j++) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
  for (j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
  for (i = 0; i < n; i++) {
    const double *pc = e->pcs + 3 * i, *pw = e->pws + 3 * i;
    for (j = 0; j < 3; j++) {
      abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
      abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
      abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
    }
  }
  for (i = 0; i < 3; i++)
    for (k = 0; k < 3; k++) At[i * 3 + k] = abt[k * 3 + i];
  jacobi_svd(At, W, Vt, 3, 3);   /* U[:,k] = At row k, V[:,k] = Vt row k */
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += At[k * 3 + i] * Vt[k * 3 + j];   /* dot(U row i, V row j) */
      R[i][j] = s;
    }
  {
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
  }
  t[0] = pc0[0] - dot3(R[0], pw0);
  t[1] = pc0[1] - dot3(R[1], pw0);
  t[2] = pc0[2] - dot3(R[2], pw0);
}

static double reprojection_error(const epnp* e, double R[3][3], const double t[3]) {
  double sum2 = 0.0;
  int i;
  for (i = 0; i < e->n; i++) {
    const double* pw = e->pws + 3 * i;
    double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
    double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
    double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
    double u = e->us[2 * i], v = e->us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / e->n;
}

static double compute_R_and_t(epnp* e, const double* ut, const double* betas, double R[3][3], double t[3]) {
  int i, j, k;
  for (i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0f;
  for (i = 0; i < 4; i++) {
    const double* v = ut + 12 * (11 - i);
    for (j = 0; j < 4; j++)
      for (k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
  }
  for (i = 0; i < e->n; i++) {
    const double* a = e->alphas + 4 * i;
    double* pc = e->pcs + 3 * i;
    for (j = 0; j < 3; j++)
      pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
  }
  if (e->pcs[2] < 0.0) {   /* solve_for_sign */
    for (i = 0; i < 4; i++)
      for (j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
    for (i = 0; i < e->n; i++) { e->pcs[3 * i] = -e->pcs[3 * i]; e->pcs[3 * i + 1] = -e->pcs[3 * i + 1]; e->pcs[3 * i + 2] = -e->pcs[3 * i + 2]; }
  }
  estimate_R_and_t(e, R, t);
  return reprojection_error(e, R, t);
}

static void epnp_compute_pose(epnp* e, double Rout[9], double tout[3]) {
  double M[2 * MAXPTS * 12], mtm[144], d[12], ut[144], vt[144];
  double l_6x10[60], rho[6], Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
  int i, a, b, N;
  choose_control_points(e);
  compute_barycentric_coordinates(e);
  for (i = 0; i < e->n; i++) {   /* fill_M */
    const double* as = e->alphas + 4 * i;
    double *M1 = M + 2 * i * 12, *M2 = M1 + 12, u = e->us[2 * i], v = e->us[2 * i + 1];
    for (a = 0; a < 4; a++) {
      M1[3 * a] = as[a] * e->fu; M1[3 * a + 1] = 0.0; M1[3 * a + 2] = as[a] * (e->uc - u);
      M2[3 * a] = 0.0; M2[3 * a + 1] = as[a] * e->fv; M2[3 * a + 2] = as[a] * (e->vc - v);
    }
  }
  for (a = 0; a < 12; a++)   /* cvMulTransposed(M, MtM, 1) */
    for (b = 0; b < 12; b++) {
      double s = 0;
      for (i = 0; i < 2 * e->n; i++) s += M[i * 12 + a] * M[i * 12 + b];
      mtm[a * 12 + b] = s;
    }
  for (a = 0; a < 12; a++)   /* cvSVD(MtM, D, Ut, 0, U_T): rows of ut = left singular vectors */
    for (b = 0; b < 12; b++) ut[a * 12 + b] = mtm[b * 12 + a];
  jacobi_svd(ut, d, vt, 12, 12);

  compute_L_6x10(ut, l_6x10);
  compute_rho(e, rho);

  find_betas_approx_1(l_6x10, rho, Betas[1]);
  gauss_newton(l_6x10, rho, Betas[1]);
  rep[1] = compute_R_and_t(e, ut, Betas[1], Rs[1], ts[1]);
  find_betas_approx_2(l_6x10, rho, Betas[2]);
  gauss_newton(l_6x10, rho, Betas[2]);
  rep[2] = compute_R_and_t(e, ut, Betas[2], Rs[2], ts[2]);
  find_betas_approx_3(l_6x10, rho, Betas[3]);
  gauss_newton(l_6x10, rho, Betas[3]);
  rep[3] = compute_R_and_t(e, ut, Betas[3], Rs[3], ts[3]);

  N = 1;
  if (rep[2] < rep[1]) N = 2;
  if (rep[3] < rep[N]) N = 3;
  memcpy(Rout, Rs[N], 9 * sizeof(double));
  memcpy(tout, ts[N], 3 * sizeof(double));
}

/* solvePnP(.., SOLVEPNP_EPNP): undistort -> epnp -> Rodrigues.  obj/img hold the values the
 * caller's Mats hold (float32-representable); img_is_f32 says whether the undistorted points
 * are rounded to float32 (true inside RANSAC, false for the final float64 call). */
static void solve_pnp_epnp(const camera* cam, const double* obj, const double* img, int n, int img_is_f32,
                           double rvec[3], double tvec[3]) {
  epnp e;
  double R[9];
  int i;
  e.uc = cam->cx; e.vc = cam->cy; e.fu = cam->fx; e.fv = cam->fy;
  e.n = n;
  for (i = 0; i < n; i++) {
    double x, y;
    undistort_point(cam, img[2 * i], img[2 * i + 1], &x, &y);
    if (img_is_f32) { x = (double)(float)x; y = (double)(float)y; }
    e.pws[3 * i] = obj[3 * i]; e.pws[3 * i + 1] = obj[3 * i + 1]; e.pws[3 * i + 2] = obj[3 * i + 2];
    e.us[2 * i] = x * e.fu + e.uc;
    e.us[2 * i + 1] = y * e.fv + e.vc;
  }
  epnp_compute_pose(&e, R, tvec);
  rodrigues_mat2vec(R, rvec);
}

/* ------------------------------------------------------------------------------------------ */
/* p3p.cpp + polynom_solver.cpp: the kernel solvePnPRansac switches to for exactly four points  */
/* ------------------------------------------------------------------------------------------ */
/* OpenCV 3.4 solvepnp.cpp, solvePnPRansac: `else if (npoints == 4) { model_points = 4; ransac_kernel_method = SOLVEPNP_P3P; }`
 * followed by `if (model_points == npoints) return solvePnP(opoints, ipoints, K, dist, rvec, tvec, false, ransac_kernel_method)`:
 * with four usable landmarks there is no RANSAC at all; solvePnP(SOLVEPNP_P3P) undistorts the (float32) image points,
 * p3p::extract_points maps them back to pixels (x * fx + cx), p3p::solve finds up to four poses from the first three
 * correspondences (X.S. Gao et al., "Complete solution classification for the perspective-three-point problem", PAMI 2003:
 * solve_for_lengths -> quartic, then Horn's quaternion alignment) and keeps the one that reprojects the fourth point best. */
static int solve_deg2(double a, double b, double c, double* x1, double* x2) {
  double delta = b * b - 4 * a * c, inv_2a, sqrt_delta;
  if (delta < 0) return 0;
  inv_2a = 0.5 / a;
  if (delta == 0) { *x1 = -b * inv_2a; *x2 = *x1; return 1; }
  sqrt_delta = sqrt(delta);
  *x1 = (-b + sqrt_delta) * inv_2a;
  *x2 = (-b - sqrt_delta) * inv_2a;
  return 2;
}

static int solve_deg3(double a, double b, double c, double d, double* x0, double* x1, double* x2) {
  double inv_a, b_a, b_a2, c_a, d_a, Q, R, Q3, D, b_a_3, AD, BD;
  if (a == 0) {
    if (b == 0) {
      if (c == 0) return 0;
      *x0 = -d / c;
      return 1;
    }
    *x2 = 0;
    return solve_deg2(b, c, d, x0, x1);
  }
  inv_a = 1. / a;
  b_a = inv_a * b; b_a2 = b_a * b_a;
  c_a = inv_a * c;
  d_a = inv_a * d;
  Q = (3 * c_a - b_a2) / 9;
  R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
  Q3 = Q * Q * Q;
  D = Q3 + R * R;
  b_a_3 = (1. / 3.) * b_a;
  if (Q == 0) {
    if (R == 0) { *x0 = *x1 = *x2 = -b_a_3; return 3; }
    *x0 = pow(2 * R, 1 / 3.0) - b_a_3;
    return 1;
  }
  if (D <= 0) {   /* three real roots */
    double theta = acos(R / sqrt(-Q3));
    double sqrt_Q = sqrt(-Q);
    *x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
    *x1 = 2 * sqrt_Q * cos((theta + 2 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
    *x2 = 2 * sqrt_Q * cos((theta + 4 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
    return 3;
  }
  AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
  BD = (AD == 0) ? 0 : -Q / AD;
  *x0 = AD + BD - b_a_3;
  return 1;
}

static int solve_deg4(double a, double b, double c, double d, double e, double* x0, double* x1, double* x2, double* x3) {
  double inv_a, b2, bc, b3, r0, r1, r2, R2, R, inv_R, D2, E2, b_4, R_2;
  int n, nb_real_roots = 0;
  if (a == 0) { *x3 = 0; return solve_deg3(b, c, d, e, x0, x1, x2); }
  inv_a = 1. / a;
  b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
  b2 = b * b; bc = b * c; b3 = b2 * b;
  n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
  if (n == 0) return 0;
  R2 = 0.25 * b2 - c + r0;
  if (R2 < 0) return 0;
  R = sqrt(R2);
  inv_R = 1. / R;
  if (R < 10E-12) {
    double temp = r0 * r0 - 4 * e;
    if (temp < 0) D2 = E2 = -1;
    else {
      double sqrt_temp = sqrt(temp);
      D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
      E2 = D2 - 4 * sqrt_temp;
    }
  } else {
    double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
    D2 = u + v;
    E2 = u - v;
  }
  b_4 = 0.25 * b; R_2 = 0.5 * R;
  if (D2 >= 0) {
    double D = sqrt(D2), D_2 = 0.5 * D;
    nb_real_roots = 2;
    *x0 = R_2 + D_2 - b_4;
    *x1 = *x0 - D;
  }
  if (E2 >= 0) {
    double E = sqrt(E2), E_2 = 0.5 * E;
    if (nb_real_roots == 0) {
      *x0 = -R_2 + E_2 - b_4;
      *x1 = *x0 - E;
      nb_real_roots = 2;
    } else {
      *x2 = -R_2 + E_2 - b_4;
      *x3 = *x2 - E;
      nb_real_roots = 4;
    }
  }
  return nb_real_roots;
}

/* p3p::jacobi_4x4: cyclic Jacobi on the symmetric 4 x 4 A (upper triangle used and destroyed); D eigenvalues, U eigenvectors in columns */
static int jacobi_4x4(double* A, double* D, double* U) {
  double B[4], Z[4];
  int iter, i, j, k;
  for (i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1. : 0.;
  B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
  memcpy(D, B, 4 * sizeof(double));
  memset(Z, 0, 4 * sizeof(double));
  for (iter = 0; iter < 50; iter++) {
    double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
    double tresh;
    if (sum == 0.0) return 1;
    tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
    for (i = 0; i < 3; i++) {
      double* pAij = A + 5 * i + 1;
      for (j = i + 1; j < 4; j++) {
        double Aij = *pAij;
        double eps_machine = 100.0 * fabs(Aij);
        if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
          *pAij = 0.0;
        else if (fabs(Aij) > tresh) {
          double hh = D[j] - D[i], t, c, s, tau;
          if (fabs(hh) + eps_machine == fabs(hh))
            t = Aij / hh;
          else {
            double theta = 0.5 * hh / Aij;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          hh = t * Aij;
          Z[i] -= hh; Z[j] += hh;
          D[i] -= hh; D[j] += hh;
          *pAij = 0.0;
          c = 1.0 / sqrt(1 + t * t);
          s = t * c;
          tau = s / (1.0 + c);
          for (k = 0; k <= i - 1; k++) {
            double g = A[k * 4 + i], h = A[k * 4 + j];
            A[k * 4 + i] = g - s * (h + g * tau);
            A[k * 4 + j] = h + s * (g - h * tau);
          }
          for (k = i + 1; k <= j - 1; k++) {
            double g = A[i * 4 + k], h = A[k * 4 + j];
            A[i * 4 + k] = g - s * (h + g * tau);
            A[k * 4 + j] = h + s * (g - h * tau);
          }
          for (k = j + 1; k < 4; k++) {
            double g = A[i * 4 + k], h = A[j * 4 + k];
            A[i * 4 + k] = g - s * (h + g * tau);
            A[j * 4 + k] = h + s * (g - h * tau);
          }
          for (k = 0; k < 4; k++) {
            double g = U[k * 4 + i], h = U[k * 4 + j];
            U[k * 4 + i] = g - s * (h + g * tau);
            U[k * 4 + j] = h + s * (g - h * tau);
          }
        }
        pAij++;
      }
    }
    for (i = 0; i < 4; i++) B[i] += Z[i];
    memcpy(D, B, 4 * sizeof(double));
    memset(Z, 0, 4 * sizeof(double));
  }
  return 0;
}

/* p3p::align: rigid motion taking the three world points X onto the camera-frame points M_end (Horn, unit quaternion) */
static void p3p_align(double M_end[3][3], const double* X /* 3 x 3, row = point */, double R[3][3], double T[3]) {
  double C_start[3], C_end[3], s[9], Qs[16], evs[4], U[16], q[4], ev_max;
  double q02, q12, q22, q32, q0_1, q0_2, q0_3, q1_2, q1_3, q2_3;
  int i, j, i_ev = 0;
  for (i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
  for (i = 0; i < 3; i++) C_start[i] = (X[i] + X[3 + i] + X[6 + i]) / 3;
  for (j = 0; j < 3; j++)
    for (i = 0; i < 3; i++)
      s[i * 3 + j] = (X[i] * M_end[0][j] + X[3 + i] * M_end[1][j] + X[6 + i] * M_end[2][j]) / 3 - C_end[j] * C_start[i];
  Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
  Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
  Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
  Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
  Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
  Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
  Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
  Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
  Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
  Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
  jacobi_4x4(Qs, evs, U);
  ev_max = evs[0];
  for (i = 1; i < 4; i++)
    if (evs[i] > ev_max) ev_max = evs[i_ev = i];
  for (i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
  q02 = q[0] * q[0]; q12 = q[1] * q[1]; q22 = q[2] * q[2]; q32 = q[3] * q[3];
  q0_1 = q[0] * q[1]; q0_2 = q[0] * q[2]; q0_3 = q[0] * q[3];
  q1_2 = q[1] * q[2]; q1_3 = q[1] * q[3];
  q2_3 = q[2] * q[3];
  R[0][0] = q02 + q12 - q22 - q32; R[0][1] = 2. * (q1_2 - q0_3); R[0][2] = 2. * (q1_3 + q0_2);
  R[1][0] = 2. * (q1_2 + q0_3); R[1][1] = q02 + q22 - q12 - q32; R[1][2] = 2. * (q2_3 - q0_1);
  R[2][0] = 2. * (q1_3 - q0_2); R[2][1] = 2. * (q2_3 + q0_1); R[2][2] = q02 + q32 - q12 - q22;
  for (i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
}

/* p3p::solve_for_lengths: |PA|, |PB|, |PC| from the pairwise distances |BC| |AC| |AB| and the cosines of the angles at P */
static int p3p_solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3]) {
  double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
  double inv_d22 = 1. / (distances[2] * distances[2]);
  double a = inv_d22 * (distances[0] * distances[0]);
  double b = inv_d22 * (distances[1] * distances[1]);
  double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
  double pr = p * r, pqr = q * pr;
  double ab, a_2, A, a_4, B, C, D, E, temp, b0, real_roots[4], r3, pr2, r3q, inv_b0;
  int n, i, nb_solutions = 0;
  if (p2 + q2 + r2 - pqr - 1 == 0) return 0;   /* reality condition (the four points should not be coplanar) */
  ab = a * b; a_2 = 2 * a;
  A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
  if (A == 0) return 0;
  a_4 = 4 * a;
  B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
  C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
  D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
  E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
  temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
  b0 = b * temp * temp;
  if (b0 == 0) return 0;
  n = solve_deg4(A, B, C, D, E, &real_roots[0], &real_roots[1], &real_roots[2], &real_roots[3]);
  if (n == 0) return 0;
  r3 = r2 * r; pr2 = p * r2; r3q = r3 * q;
  inv_b0 = 1. / b0;
  for (i = 0; i < n; i++) {
    double x = real_roots[i], x2, b1, y, v, Z;
    if (x <= 0) continue;
    x2 = x * x;
    b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
         (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
           (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
          (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
           pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
          2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
          p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    if (b1 <= 0) continue;
    y = inv_b0 * b1;
    v = x2 + y * y - x * y * r;
    if (v <= 0) continue;
    Z = distances[2] / sqrt(v);
    lengths[nb_solutions][0] = x * Z;
    lengths[nb_solutions][1] = y * Z;
    lengths[nb_solutions][2] = Z;
    nb_solutions++;
  }
  return nb_solutions;
}

/* solvePnP(.., SOLVEPNP_P3P) on exactly four correspondences.  obj: 4 x 3 (float32-representable), img: 4 x 2 raw float32 image
 * points.  Returns 1 and (rvec, tvec), or 0 when the first three points admit no pose. */
static int solve_pnp_p3p(const camera* cam, const double* obj, const double* img, double rvec[3], double tvec[3]) {
  const double inv_fx = 1. / cam->fx, inv_fy = 1. / cam->fy, cx_fx = cam->cx / cam->fx, cy_fy = cam->cy / cam->fy;
  double mu[4], mv[4], m[3][3], distances[3], cosines[3], lengths[4][3], Rs[4][3][3], ts[4][3], min_reproj = 0, R9[9];
  int i, j, n, ns = 0, nb = 0;
  for (i = 0; i < 4; i++) {   /* undistortPoints writes CV_32FC2; p3p::extract_points maps back to pixels in double */
    double x, y;
    undistort_point(cam, img[2 * i], img[2 * i + 1], &x, &y);
    mu[i] = (double)(float)x * cam->fx + cam->cx;
    mv[i] = (double)(float)y * cam->fy + cam->cy;
  }
  for (i = 0; i < 3; i++) {   /* unit bearing vectors of the first three points */
    double u = inv_fx * mu[i] - cx_fx, v = inv_fy * mv[i] - cy_fy;
    double norm = sqrt(u * u + v * v + 1), mk = 1. / norm;
    m[i][0] = u * mk; m[i][1] = v * mk; m[i][2] = mk;
  }
  distances[0] = sqrt((obj[3] - obj[6]) * (obj[3] - obj[6]) + (obj[4] - obj[7]) * (obj[4] - obj[7]) + (obj[5] - obj[8]) * (obj[5] - obj[8]));
  distances[1] = sqrt((obj[0] - obj[6]) * (obj[0] - obj[6]) + (obj[1] - obj[7]) * (obj[1] - obj[7]) + (obj[2] - obj[8]) * (obj[2] - obj[8]));
  distances[2] = sqrt((obj[0] - obj[3]) * (obj[0] - obj[3]) + (obj[1] - obj[4]) * (obj[1] - obj[4]) + (obj[2] - obj[5]) * (obj[2] - obj[5]));
  cosines[0] = m[1][0] * m[2][0] + m[1][1] * m[2][1] + m[1][2] * m[2][2];
  cosines[1] = m[0][0] * m[2][0] + m[0][1] * m[2][1] + m[0][2] * m[2][2];
  cosines[2] = m[0][0] * m[1][0] + m[0][1] * m[1][1] + m[0][2] * m[1][2];
  n = p3p_solve_for_lengths(lengths, distances, cosines);
  for (i = 0; i < n; i++) {
    double M_orig[3][3];
    for (j = 0; j < 3; j++) { M_orig[j][0] = lengths[i][j] * m[j][0]; M_orig[j][1] = lengths[i][j] * m[j][1]; M_orig[j][2] = lengths[i][j] * m[j][2]; }
    p3p_align(M_orig, obj, Rs[nb], ts[nb]);
    nb++;
  }
  if (nb == 0) return 0;
  for (i = 0; i < nb; i++) {   /* the fourth point picks the solution */
    const double X3 = obj[9], Y3 = obj[10], Z3 = obj[11];
    double X3p = Rs[i][0][0] * X3 + Rs[i][0][1] * Y3 + Rs[i][0][2] * Z3 + ts[i][0];
    double Y3p = Rs[i][1][0] * X3 + Rs[i][1][1] * Y3 + Rs[i][1][2] * Z3 + ts[i][1];
    double Z3p = Rs[i][2][0] * X3 + Rs[i][2][1] * Y3 + Rs[i][2][2] * Z3 + ts[i][2];
    double mu3p = cam->cx + cam->fx * X3p / Z3p;
    double mv3p = cam->cy + cam->fy * Y3p / Z3p;
    double reproj = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
    if (i == 0 || min_reproj > reproj) { ns = i; min_reproj = reproj; }
  }
  for (i = 0; i < 3; i++) {
    for (j = 0; j < 3; j++) R9[3 * i + j] = Rs[ns][i][j];
    tvec[i] = ts[ns][i];
  }
  rodrigues_mat2vec(R9, rvec);
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* ptsetreg.cpp RANSAC + solvepnp.cpp solvePnPRansac                                           */
/* ------------------------------------------------------------------------------------------ */
static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
  double num, denom;
  p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
  ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
  num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
  denom = 1. - det_powi(1. - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = det_log(num);
  denom = det_log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

static int find_inliers(const camera* cam, const double* obj, const double* img, int n, const double rvec[3],
                        const double tvec[3], double thresh, unsigned char* mask) {
  double R[9];
  const float t = (float)(thresh * thresh);
  int i, nz = 0;
  rodrigues_vec2mat(rvec, R);
  for (i = 0; i < n; i++) {
    double u, v;
    float pu, pv, dx, dy, err;
    project_point(cam, R, tvec, obj + 3 * i, &u, &v);
    pu = (float)u; pv = (float)v;                      /* projpoints is CV_32F */
    dx = (float)img[2 * i] - pu; dy = (float)img[2 * i + 1] - pv;
    err = (float)((double)dx * dx + (double)dy * dy);  /* norm(Matx21f, NORM_L2SQR) -> float */
    mask[i] = (unsigned char)(err <= t);
    nz += mask[i];
  }
  return nz;
}

/* Returns number of inliers (>0) on success, -2 when RANSAC (or P3P, n = 4) finds no model, -1 when n < 4. */
static int solve_pnp_ransac(const camera* cam, const double* obj64, const float* img32, int n, int max_iters,
                            double reproj_err, double confidence, double rvec[3], double tvec[3],
                            int* iters_run) {
  double obj[3 * MAXPTS], img[2 * MAXPTS];
  unsigned char mask[MAXPTS], best_mask[MAXPTS];
  double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
  const int model_points = 5;
  int i, iter, niters = max_iters > 1 ? max_iters : 1, max_good = 0, count = n;
  cvrng rng = {(uint64_t)-1};
  if (iters_run) *iters_run = 0;
  if (n < 4 || n > MAXPTS) return -1;
  for (i = 0; i < 3 * n; i++) obj[i] = (double)(float)obj64[i];   /* convertTo(CV_32F) */
  for (i = 0; i < 2 * n; i++) img[i] = (double)img32[i];
  if (n == 4) {   /* model_points = npoints = 4: solvePnP(SOLVEPNP_P3P) directly, no RANSAC (see solve_pnp_p3p) */
    if (!solve_pnp_p3p(cam, obj, img, rvec, tvec)) { memset(rvec, 0, 24); memset(tvec, 0, 24); return -2; }
    return n;
  }
  if (n == model_points) {
    solve_pnp_epnp(cam, obj, img, n, 1, rvec, tvec);
    return n;
  }
  for (iter = 0; iter < niters; iter++) {
    int idx[5], good;
    double so[15], si[10], r[3], t[3];
    int ii = 0, j, attempts = 0;
    /* getSubset: duplicate-free draws (checkSubset is the default "true") */
    for (; ii < model_points && attempts < 10000;) {
      int idx_i;
      for (;;) {
        idx_i = idx[ii] = rng_uniform(&rng, 0, count);
        for (j = 0; j < ii; j++)
          if (idx_i == idx[j]) break;
        if (j == ii) break;
      }
      memcpy(so + 3 * ii, obj + 3 * idx_i, 3 * sizeof(double));
      memcpy(si + 2 * ii, img + 2 * idx_i, 2 * sizeof(double));
      ii++;
    }
    solve_pnp_epnp(cam, so, si, model_points, 1, r, t);
    good = find_inliers(cam, obj, img, count, r, t, reproj_err, mask);
    if (iters_run) *iters_run = iter + 1;
    if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
      memcpy(best_mask, mask, count);
      memcpy(best_r, r, sizeof(r)); memcpy(best_t, t, sizeof(t));
      max_good = good;
      niters = ransac_update_num_iters(confidence, (double)(count - good) / count, model_points, niters);
    }
  }
  if (max_good <= 0) { memset(rvec, 0, 24); memset(tvec, 0, 24); return -2; }
  {
    double io[3 * MAXPTS], ii2[2 * MAXPTS];
    int m = 0;
    for (i = 0; i < count; i++)
      if (best_mask[i]) {
        memcpy(io + 3 * m, obj + 3 * i, 3 * sizeof(double));
        memcpy(ii2 + 2 * m, img + 2 * i, 2 * sizeof(double));
        m++;
      }
    solve_pnp_epnp(cam, io, ii2, m, 0, rvec, tvec);
    return m;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* export_predicted_poses_real.py:186-203 for one frame / a batch                              */
/* ------------------------------------------------------------------------------------------ */
int pnp_ref_frame(const float* kp_xyc, const double* landmarks, const double* K, const double* dist, int J,
                  double conf_thr0, int min_pts, double thr_decay, int thr_iters, int max_iters,
                  double reproj_err, double confidence, double* R9, double* t3, double* rvec3, int* iters_run) {
  camera cam;
  double obj[3 * MAXPTS], rvec[3], tvec[3];
  float img[2 * MAXPTS];
  double thr64 = conf_thr0;   /* python float: multiplied in float64 ... */
  float thr;                  /* ... compared against the float32 scores as float32 */
  int i, n, it = 0, status;
  if (J > MAXPTS) return -1;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (i = 0; i < 5; i++) cam.k[i] = dist ? dist[i] : 0.0;
  /* threshold loop: float32 comparisons, numpy semantics (python float threshold vs float32 array:
     the array is compared as float32 against the threshold rounded to float32) */
  for (;;) {
    thr = (float)thr64;
    n = 0;
    for (i = 0; i < J; i++) n += kp_xyc[3 * i + 2] > thr;
    if (n >= min_pts) break;
    thr64 *= thr_decay;
    thr = (float)thr64;
    it++;
    if (it >= thr_iters) break;
  }
  n = 0;
  for (i = 0; i < J; i++)
    if (kp_xyc[3 * i + 2] > thr) {
      obj[3 * n] = landmarks[3 * i]; obj[3 * n + 1] = landmarks[3 * i + 1]; obj[3 * n + 2] = landmarks[3 * i + 2];
      img[2 * n] = kp_xyc[3 * i]; img[2 * n + 1] = kp_xyc[3 * i + 1];
      n++;
    }
  status = solve_pnp_ransac(&cam, obj, img, n, max_iters, reproj_err, confidence, rvec, tvec, iters_run);
  if (status < 0) {
    memset(R9, 0, 9 * sizeof(double)); R9[0] = R9[4] = R9[8] = 1;
    memset(t3, 0, 3 * sizeof(double));
    if (rvec3) memset(rvec3, 0, 3 * sizeof(double));
    return status;
  }
  rodrigues_vec2mat(rvec, R9);
  memcpy(t3, tvec, sizeof(tvec));
  if (rvec3) memcpy(rvec3, rvec, sizeof(rvec));
  return status;
}

void pnp_ref_batch(const float* kp_xyc, const double* landmarks, const double* K, const double* dist, int N, int J,
                   double conf_thr0, int min_pts, double thr_decay, int thr_iters, int max_iters, double reproj_err,
                   double confidence, double* R, double* t, double* rvec, int* status, int* iters_run) {
  int i;
  for (i = 0; i < N; i++)
    status[i] = pnp_ref_frame(kp_xyc + (size_t)i * J * 3, landmarks, K, dist, J, conf_thr0, min_pts, thr_decay,
                              thr_iters, max_iters, reproj_err, confidence, R + 9 * i, t + 3 * i,
                              rvec ? rvec + 3 * i : 0, iters_run ? iters_run + i : 0);
}

/* exposed pieces for unit tests of the restatement */
int pnp_ref_p3p(const double* K, const double* dist, const double* obj, const double* img, double* rvec, double* tvec) {
  camera cam;
  int i;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (i = 0; i < 5; i++) cam.k[i] = dist ? dist[i] : 0.0;
  return solve_pnp_p3p(&cam, obj, img, rvec, tvec);
}
void pnp_ref_epnp(const double* K, const double* dist, const double* obj, const double* img, int n, double* rvec, double* tvec) {
  camera cam;
  int i;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (i = 0; i < 5; i++) cam.k[i] = dist ? dist[i] : 0.0;
  solve_pnp_epnp(&cam, obj, img, n, 0, rvec, tvec);
}
void pnp_ref_rodrigues(const double* in, int in_is_matrix, double* out) {
  if (in_is_matrix) rodrigues_mat2vec(in, out); else rodrigues_vec2mat(in, out);
}
void pnp_ref_project(const double* K, const double* dist, const double* R, const double* t, const double* obj, int n, double* uv) {
  camera cam;
  int i;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (i = 0; i < 5; i++) cam.k[i] = dist ? dist[i] : 0.0;
  for (i = 0; i < n; i++) project_point(&cam, R, t, obj + 3 * i, uv + 2 * i, uv + 2 * i + 1);
}
void pnp_ref_undistort(const double* K, const double* dist, const double* uv, int n, double* xy) {
  camera cam;
  int i;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (i = 0; i < 5; i++) cam.k[i] = dist ? dist[i] : 0.0;
  for (i = 0; i < n; i++) undistort_point(&cam, uv[2 * i], uv[2 * i + 1], xy + 2 * i, xy + 2 * i + 1);
}
unsigned pnp_ref_rng_draws(int count, int n, int* out) {   /* first n uniform(0,count) of RNG((uint64)-1) */
  cvrng r = {(uint64_t)-1};
  int i;
  for (i = 0; i < n; i++) out[i] = rng_uniform(&r, 0, count);
  return (unsigned)r.state;
}
