"""Independent evidence for the EPnP+RANSAC path (cv2 is absent here, so neither the C oracle nor the HIP kernel can be
pinned to cv2.solvePnPRansac itself): a SciPy Levenberg-Marquardt minimiser of the reprojection error, written against
the camera model only (pinhole + k1,k2,p1,p2,k3; pose_estimation/export_predicted_poses_real.py:104-121), shares no
code with oracle/pnp_ref.c or csrc/pnp.hip.  What it can establish:

  * the returned pose explains its inliers: reprojection RMS on the inlier set is within a stated factor of the RMS of
    the least-squares optimum over the same points (EPnP is algebraic, not ML, so equality is not expected);
  * the returned pose lies within the noise floor of that optimum and of the generating pose;
  * it is never much worse than the ground-truth pose on the same points.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


def project(rvec, t, X, K, dist):
    R = Rotation.from_rotvec(rvec).as_matrix()
    pc = X @ R.T + t
    x0, y0 = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    r2 = x0 * x0 + y0 * y0
    cd = 1 + dist[0] * r2 + dist[1] * r2 ** 2 + dist[4] * r2 ** 3
    x1 = x0 * cd + 2 * dist[2] * x0 * y0 + dist[3] * (r2 + 2 * x0 * x0)
    y1 = y0 * cd + dist[2] * (r2 + 2 * y0 * y0) + 2 * dist[3] * x0 * y0
    return np.stack([K[0, 0] * x1 + K[0, 2], K[1, 1] * y1 + K[1, 2]], 1)


def rms(R, t, X, uv, K, dist):
    d = project(Rotation.from_matrix(R).as_rotvec(), t, X, K, dist) - uv
    return float(np.sqrt((d ** 2).sum(1).mean()))


def inlier_mask(R, t, X, uv, K, dist, thr=15.0):
    d = project(Rotation.from_matrix(R).as_rotvec(), t, X, K, dist) - uv
    return (d ** 2).sum(1) <= thr * thr


def refine(R0, t0, X, uv, K, dist):
    """Least-squares pose on (X, uv) started at (R0, t0)."""
    x0 = np.concatenate([Rotation.from_matrix(R0).as_rotvec(), t0])
    sol = least_squares(lambda p: (project(p[:3], p[3:], X, K, dist) - uv).ravel(), x0, method="lm", xtol=1e-12, ftol=1e-12)
    return Rotation.from_rotvec(sol.x[:3]).as_matrix(), sol.x[3:]


def angle(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def audit(kp, R_est, t_est, status, R_gt, t_gt, X, K, dist):
    """Per-frame statistics of an estimator's output against the least-squares optimum on its own inlier set.
    Returns dict of arrays over the frames with status >= 4."""
    out = {k: [] for k in ("ratio_ls", "ratio_gt", "ang_ls", "ang_gt", "t_ls", "t_gt", "n_inl", "rms")}
    for i in range(len(kp)):
        if status[i] < 4:
            continue
        uv = kp[i, :, :2].astype(np.float64)
        m = inlier_mask(R_est[i], t_est[i], X, uv, K, dist)
        Rl, tl = refine(R_est[i], t_est[i], X[m], uv[m], K, dist)
        r_est, r_ls = rms(R_est[i], t_est[i], X[m], uv[m], K, dist), rms(Rl, tl, X[m], uv[m], K, dist)
        out["rms"].append(r_est)
        out["ratio_ls"].append(r_est / max(r_ls, 1e-9))
        out["ratio_gt"].append(r_est / max(rms(R_gt[i], t_gt[i], X[m], uv[m], K, dist), 1e-9))
        out["ang_ls"].append(angle(R_est[i], Rl)); out["ang_gt"].append(angle(R_est[i], R_gt[i]))
        out["t_ls"].append(np.linalg.norm(t_est[i] - tl) / np.linalg.norm(tl))
        out["t_gt"].append(np.linalg.norm(t_est[i] - t_gt[i]) / np.linalg.norm(t_gt[i]))
        out["n_inl"].append(int(m.sum()))
    return {k: np.array(v) for k, v in out.items()}
