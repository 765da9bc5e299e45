"""Third-party cross-checks of the UNPINNED pieces of the PnP oracle.

cv2 (opencv-python==3.4.11.41) cannot be had in this image, so oracle/pnp_ref.c restates OpenCV's solvePnPRansac / Rodrigues internals
from knowledge of that source.  What CAN be done here: hold every linear-algebra block of the restatement against an INDEPENDENT
implementation -- numpy.linalg, numpy.roots, scipy.spatial.transform.Rotation, scipy.optimize -- on the matrices the path actually meets
(the fixture frames' 12 x 12 MtM, the 6 x 4 Gauss-Newton systems, the P3P quartics).  That replaces "from memory" by "agrees with a third
party" for: the one-sided Jacobi SVD and its back-substitution, the eigenvectors EPnP takes from MtM, the Householder QR solve, the
P3P polynomial solvers and 4 x 4 Jacobi eigen-solver, Rodrigues both ways (incl. theta -> 0 and theta -> pi), and undistortPoints'
five fixed-point iterations against a converged inverse of the reference's own project().  The RANSAC driver, the beta cases and the
sign / selection logic remain restated-from-knowledge (oracle/pnp_ref.c header lists both sets)."""
import ctypes
from ctypes import c_double, c_int, c_void_p

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

from oracle import pnp_ref as P


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _svd(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    U = np.zeros((n, m)); W = np.zeros(n); Vt = np.zeros((n, n))
    P.lib().pnp_ref_test_svd(_p(A), c_int(m), c_int(n), _p(U), _p(W), _p(Vt))
    return U, W, Vt


@pytest.mark.parametrize("shape", [(3, 3), (6, 4), (6, 5), (12, 12), (22, 12), (10, 3)])
def test_jacobi_svd_agrees_with_numpy(shape):
    """JacobiSVDImpl_ restated (oracle/pnp_ref.c: jacobi_svd) against numpy.linalg.svd (LAPACK gesdd): singular values to 1e-13
    relative, A = U^T diag(W) Vt reconstructed to 1e-13, orthonormal factors; also on rank-deficient and ill-conditioned input."""
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    m, n = shape
    for case in range(6):
        A = rng.standard_normal((m, n))
        if case == 4:                      # rank n - 1
            A[:, -1] = A[:, 0] * 2.0 - A[:, 1]
        if case == 5:                      # condition number 1e10
            u, s, vt = np.linalg.svd(A, full_matrices=False)
            A = (u * np.logspace(0, -10, n)) @ vt
        U, W, Vt = _svd(A)
        s_np = np.linalg.svd(A, compute_uv=False)
        assert np.all(np.diff(W) <= 1e-300) and np.allclose(W, s_np, rtol=1e-12, atol=1e-13 * s_np[0])
        assert np.allclose(U.T @ np.diag(W) @ Vt, A, atol=1e-13 * max(1.0, s_np[0]))
        assert np.allclose(Vt @ Vt.T, np.eye(n), atol=1e-13)
        keep = W > 1e-9 * W[0]
        assert np.allclose((U @ U.T)[np.ix_(keep, keep)], np.eye(int(keep.sum())), atol=1e-12)


def test_svd_solve_is_the_least_squares_solution():
    """cvSolve(.., DECOMP_SVD) / SVBkSb restated (svd_solve: estimate_R_and_t's 3 x 3 and the beta systems) against numpy.linalg.lstsq."""
    rng = np.random.default_rng(5)
    for m, n, nb in ((6, 4, 1), (6, 3, 1), (6, 5, 1), (3, 3, 3), (12, 12, 2)):
        A = rng.standard_normal((m, n)); b = rng.standard_normal((m, nb))
        x = np.zeros((n, nb))
        P.lib().pnp_ref_test_svd_solve(_p(np.ascontiguousarray(A)), c_int(m), c_int(n), _p(np.ascontiguousarray(b)), c_int(nb), _p(x))
        assert np.allclose(x, np.linalg.lstsq(A, b, rcond=None)[0], atol=1e-12)


def test_qr_solve_is_the_least_squares_solution():
    """epnp::qr_solve restated (Householder QR with OpenCV's row-scan quirk; the 6 x 4 system of every Gauss-Newton step) against
    numpy.linalg.lstsq, on random systems and on systems scaled over twelve decades (eta is only a scale factor)."""
    rng = np.random.default_rng(6)
    for k in range(40):
        A = rng.standard_normal((6, 4)) * 10.0 ** rng.integers(-6, 6); b = rng.standard_normal(6) * 10.0 ** rng.integers(-3, 3)
        x = np.zeros(4)
        P.lib().pnp_ref_test_qr_solve(_p(A.copy()), _p(b.copy()), _p(x), c_int(6), c_int(4))
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.allclose(x, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max()), k


def test_mtm_eigenvectors_agree_with_numpy_eigh_on_the_fixture_frames():
    """EPnP's null-space basis: the four eigenvectors of MtM with the smallest eigenvalues, taken as the last rows of the U^T of
    cvSVD(MtM) (epnp.cpp compute_pose).  On 32 seeded frames (11 landmarks, the SPEED+ camera): eigenvalues = numpy.linalg.eigh's to
    1e-9 of the largest, and each of the four vectors lies in eigh's corresponding eigenspace (|cos| > 1 - 1e-9 where the eigenvalue is
    simple) -- the subspace they span is the same to 1e-8 in any case."""
    rng = np.random.default_rng(7)
    kps, Rs, ts = P.synth_keypoints(32, rng, noise_px=0.5, outlier_frac=0.0)
    P.lib().pnp_ref_test_mtm.restype = None
    for i in range(32):
        us = np.ascontiguousarray(kps[i, :, :2], dtype=np.float64)
        mtm = np.zeros(144); d = np.zeros(12); ut = np.zeros(144)
        P.lib().pnp_ref_test_mtm(_p(np.ascontiguousarray(P.CAMERA_K)), _p(np.ascontiguousarray(P.LANDMARKS)), _p(us), c_int(11), _p(mtm), _p(d), _p(ut))
        M = mtm.reshape(12, 12); Ut = ut.reshape(12, 12)
        assert np.allclose(M, M.T, rtol=1e-12, atol=1e-9)
        w, v = np.linalg.eigh((M + M.T) / 2)           # ascending
        assert np.allclose(d[::-1], w, atol=1e-9 * w[-1])
        small_o, small_np = Ut[8:12][::-1].T, v[:, :4]    # 12 x 4 each, ascending eigenvalue
        # same 4-dimensional subspace: principal angles ~ 0
        sv = np.linalg.svd(small_o.T @ small_np, compute_uv=False)
        assert sv.min() > 1 - 1e-8, (i, sv)
        for k in range(4):
            gap = min(abs(w[k] - w[j]) for j in range(12) if j != k)
            if gap > 1e-6 * w[-1]:
                assert abs(small_o[:, k] @ small_np[:, k]) > 1 - 1e-9, (i, k)


def test_p3p_polynomial_solvers_agree_with_numpy_roots():
    """polynom_solver.cpp restated (solve_deg2 / 3 / 4: Ferrari + Cardano in closed form) against numpy.roots (companion-matrix
    eigenvalues): every real root numpy finds with multiplicity one is among the solver's to 1e-7, and every root the solver returns
    is a root (|p(x)| small against the polynomial's scale)."""
    rng = np.random.default_rng(8)
    P.lib().pnp_ref_test_poly.restype = c_int
    for deg in (2, 3, 4):
        for k in range(200):
            r_true = rng.uniform(-3, 3, deg)
            if deg == 4 and k % 3 == 0:      # two real + a complex pair
                c = np.poly([r_true[0], r_true[1], complex(r_true[2], 0.5 + abs(r_true[3])), complex(r_true[2], -0.5 - abs(r_true[3]))]).real
            else:
                c = np.poly(r_true)
            c = c * rng.uniform(0.5, 2.0)
            roots = np.zeros(4)
            n = P.lib().pnp_ref_test_poly(c_int(deg), _p(np.ascontiguousarray(c, dtype=np.float64)), _p(roots))
            got = np.sort(roots[:n])
            ref = np.roots(c)
            ref_real = np.sort(ref[np.abs(ref.imag) < 1e-9].real)
            for x in got:
                assert abs(np.polyval(c, x)) <= 1e-7 * np.abs(c).max() * max(1.0, abs(x)) ** deg, (deg, k, x)
            sep = min([abs(a - b) for i, a in enumerate(ref) for b in ref[i + 1:]] + [1.0])
            if sep > 1e-3:                       # well-separated roots: all of numpy's real roots are found
                assert len(got) == len(ref_real) and np.allclose(got, ref_real, atol=1e-7), (deg, k, got, ref_real)


def test_p3p_jacobi_4x4_agrees_with_numpy_eigh():
    """p3p::jacobi_4x4 restated (Horn's quaternion alignment takes the eigenvector of the largest eigenvalue) against numpy.linalg.eigh."""
    rng = np.random.default_rng(9)
    P.lib().pnp_ref_test_jacobi4.restype = c_int
    for k in range(50):
        B = rng.standard_normal((4, 4)); A = B + B.T
        D = np.zeros(4); U = np.zeros(16)
        assert P.lib().pnp_ref_test_jacobi4(_p(A.copy().ravel()), _p(D), _p(U)) == 1
        w, v = np.linalg.eigh(A)
        order = np.argsort(D)
        assert np.allclose(D[order], w, atol=1e-12)
        Um = U.reshape(4, 4)[:, order]
        for j in range(4):
            assert abs(abs(Um[:, j] @ v[:, j]) - 1) < 1e-10


def test_rodrigues_agrees_with_scipy_rotation():
    """cvRodrigues2 restated, both directions, against scipy.spatial.transform.Rotation: random rotations, theta -> 0 (1e-3 .. 1e-12, 0) and
    theta -> pi (pi - 1e-3 .. pi - 1e-9, pi), every axis orientation.  vec -> mat to 2e-14; mat -> vec to 1e-9 away from 0 and pi (below sin(theta) = 1e-5 OpenCV returns the zero vector: kept), and up to the
    sign ambiguity r ~ -r at theta = pi (compared as matrices there: 1e-7, the conditioning of the problem, not of the code)."""
    rng = np.random.default_rng(10)
    axes = rng.standard_normal((40, 3)); axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    thetas = np.concatenate([rng.uniform(0.05, np.pi - 0.05, 20), [1e-3, 1e-5, 1e-8, 1e-12, 0.0], np.pi - np.array([1e-3, 1e-5, 1e-7, 1e-9, 0.0])])
    for a in axes:
        for th in thetas:
            r = a * th
            Rm = P.rodrigues(r)
            Rs = Rotation.from_rotvec(r).as_matrix()
            assert np.abs(Rm - Rs).max() <= 2e-14, (th, np.abs(Rm - Rs).max())
            back = P.rodrigues(Rs.ravel())
            if th <= 1.0000001e-5:      # cvRodrigues2: sin(theta) < 1e-5 with cos(theta) > 0 returns the ZERO vector (calibration.cpp) -- kept
                assert np.all(back == 0.0), (th, back)
            elif th < np.pi - 1e-4:
                assert np.abs(back - r).max() <= 1e-9, (th, back, r)
            assert np.abs(P.rodrigues(back) - Rs).max() <= (2.1e-5 if np.sin(th) < 1.0000001e-5 else 1e-7), th   # the round trip as a rotation; inside OpenCV's sin(theta) < 1e-5 cut (either end) the off-diagonal 2 sin(theta) is dropped


def test_undistort_five_iterations_against_a_converged_inverse_of_the_reference_projection():
    """undistortPoints restated (5 fixed-point iterations, OpenCV 3.4) against a converged inverse (scipy least_squares, xtol 1e-15) of the
    distortion model the REFERENCE's own project() applies (oracle project_numpy, pinned to speed_plus_utils/utils.py:108-139 by
    tests/test_camera_golden.py).  States the truncation error of the five iterations over the 1920 x 1200 image: 9.6e-11 in normalised
    coordinates (2.9e-7 px) at the corners (measured; asserted <= 1e-9) -- far below the float32 rounding (6e-5 px at 1000 px) the
    undistorted points get in OpenCV, so the fixed iteration count is not a source of deviation for this camera."""
    K, dist = P.CAMERA_K, P.CAMERA_DIST
    k1, k2, p1, p2, k3 = dist

    def distort(xy):
        x, y = xy
        r2 = x * x + y * y
        c = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
        return np.array([x * c + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * c + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y])
    us, vs = np.meshgrid(np.linspace(0, 1920, 9), np.linspace(0, 1200, 7))
    uv = np.stack([us.ravel(), vs.ravel()], 1)
    got = P.undistort(uv)
    worst = 0.0
    for (u, v), g in zip(uv, got):
        target = np.array([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]])
        sol = least_squares(lambda q: distort(q) - target, target, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert np.abs(distort(sol.x) - target).max() < 1e-13
        worst = max(worst, float(np.abs(g - sol.x).max()))
        # and the reference's own projection maps the converged inverse back onto the pixel
        pix = P.project_numpy(np.eye(3), np.zeros(3), np.array([[sol.x[0], sol.x[1], 1.0]]))[0]
        assert np.abs(pix - np.array([u, v])).max() < 1e-9
    print("undistortPoints, 5 iterations: max |x - converged inverse| = %.2e (normalised) = %.2e px over the image" % (worst, worst * K[0, 0]))
    assert worst <= 1e-9
    centre = P.undistort(np.array([[960.0 + 100, 600.0 - 80]]))[0]
    t = np.array([100 / K[0, 0], -80 / K[1, 1]])
    sol = least_squares(lambda q: distort(q) - t, t, xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert np.abs(centre - sol.x).max() < 1e-10
