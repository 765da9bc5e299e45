"""Independent cross-check of the EPnP+RANSAC restatement (oracle/pnp_ref.c) -- and, in the GPU test at the bottom, of
the HIP kernel -- against a SciPy least-squares reprojection minimiser (tests/pnp_independent.py).

cv2.solvePnPRansac (pose_estimation/export_predicted_poses_real.py:199-203) cannot be run here (no OpenCV in the image,
SURVEY.md section 8c), so this is the next best evidence: an estimator that shares no code with either implementation,
applied to the inlier set each one reports.  Bounds below were set from the measured distributions (1 px noise,
10 % / 30 % outliers, 256 frames: RMS ratio to the optimum median 1.02, max 1.28; angle to the optimum median 1.2e-3 rad)
with head-room, not tuned to pass."""
import numpy as np
import pytest

from oracle import pnp_ref as P
import pnp_independent as I

RMS_VS_OPTIMUM_MAX, RMS_VS_OPTIMUM_MEDIAN = 1.5, 1.08     # returned pose vs least-squares optimum on its inliers
RMS_VS_TRUTH_MAX = 1.3                                     # never much worse than the generating pose
ANG_VS_OPTIMUM_P95, T_VS_OPTIMUM_P95 = 1.5e-2, 6e-3        # within the noise floor of the optimum (rad, relative)


def check_against_least_squares(kp, R, t, status, Rs, ts, landmarks=P.LANDMARKS, expect_outliers=0):
    j = kp.shape[1]
    assert (status >= j - expect_outliers).all(), "RANSAC lost true inliers: %s" % np.unique(status)
    assert (t[:, 2] > 0).all(), "a pose behind the camera survived the EPnP sign fix"
    a = I.audit(kp, R, t, status, Rs, ts, landmarks, P.CAMERA_K, P.CAMERA_DIST)
    assert np.array_equal(a["n_inl"], status[status >= 4]), "reported inlier count != points within 15 px of the returned pose"
    assert a["ratio_ls"].max() <= RMS_VS_OPTIMUM_MAX and np.median(a["ratio_ls"]) <= RMS_VS_OPTIMUM_MEDIAN, \
        "reprojection RMS vs least-squares optimum: median %.3f max %.3f" % (np.median(a["ratio_ls"]), a["ratio_ls"].max())
    assert a["ratio_gt"].max() <= RMS_VS_TRUTH_MAX
    assert np.percentile(a["ang_ls"], 95) <= ANG_VS_OPTIMUM_P95 and np.percentile(a["t_ls"], 95) <= T_VS_OPTIMUM_P95
    return a


@pytest.mark.parametrize("outliers", [0.1, 0.3])
def test_oracle_pose_is_near_the_least_squares_optimum(outliers):
    rng = np.random.default_rng(11)
    kp, Rs, ts = P.synth_keypoints(256, rng, 1.0, outliers)
    o = P.solve_batch(kp)
    a = check_against_least_squares(kp, o["R"], o["t"], o["status"], Rs, ts, expect_outliers=int(round(outliers * 11)))
    print("outliers %.1f: RMS/optimum median %.3f max %.3f; angle to optimum median %.2e rad" % (
        outliers, np.median(a["ratio_ls"]), a["ratio_ls"].max(), np.median(a["ang_ls"])))


def near_coplanar_subset(n, rng):
    """Confidences admit exactly landmarks 0-3 (z = 0.160 m, a plane to 0.5 mm) + landmark 8 (z = 0.090): the direct
    5-point EPnP on an almost planar configuration (the threshold loop ends at 0.95 * 0.8^100 = 1.9e-10)."""
    kp, Rs, ts = P.synth_keypoints(n, rng, 0.5, 0.0)
    keep = [0, 1, 2, 3, 8]
    kp[:, :, 2] = 1e-11
    kp[:, keep, 2] = 1.0
    return kp, Rs, ts, keep


def test_oracle_degenerate_and_sign_cases():
    rng = np.random.default_rng(5)
    kp, Rs, ts, keep = near_coplanar_subset(64, rng)
    o = P.solve_batch(kp)
    assert (o["status"] == 5).all() and (o["t"][:, 2] > 0).all()
    ang = P.rot_angle(o["R"], Rs)
    assert np.median(ang) < 1e-2                         # 0.5 px noise on 5 nearly coplanar points
    for i in range(0, 64, 7):                            # the pose explains the five points it was computed from
        assert I.rms(o["R"][i], o["t"][i], P.LANDMARKS[keep], kp[i, keep, :2].astype(np.float64), P.CAMERA_K, P.CAMERA_DIST) < 3.0
    # An exactly planar target.  OpenCV's EPnP has no planar branch and solvePnPRansac returns the pose of the FINAL
    # EPnP over all RANSAC inliers (not the winning 5-point hypothesis), so on a planar inlier set the reported inlier
    # count can be 11 while the returned pose is poor: reference behaviour, restated as is.  What must hold: a status
    # is always reported, rotations stay orthonormal, and the run is deterministic.
    X = P.LANDMARKS.copy(); X[:, 2] = 0.0
    kpp, Rp, tp = P.synth_keypoints(64, rng, 0.5, 0.0, landmarks=X)
    op = P.solve_batch(kpp, landmarks=X)
    ok = op["status"] > 0
    assert ok.any() and np.abs(np.einsum("nij,nkj->nik", op["R"][ok], op["R"][ok]) - np.eye(3)).max() < 1e-9
    again = P.solve_batch(kpp, landmarks=X)
    assert np.array_equal(again["status"], op["status"]) and np.array_equal(again["R"], op["R"])


@pytest.mark.gpu
@pytest.mark.parametrize("outliers", [0.1, 0.3])
def test_hip_pose_is_near_the_least_squares_optimum(gpu_ops, outliers):
    """The same audit on the HIP kernel's own output (not via the oracle), BASELINE batch of 256 frames."""
    import torch
    rng = np.random.default_rng(11)
    kp, Rs, ts = P.synth_keypoints(256, rng, 1.0, outliers)
    rot, tv, st = gpu_ops.pnp_epnp_ransac(torch.from_numpy(kp).cuda(), torch.from_numpy(P.LANDMARKS).cuda(),
                                          torch.from_numpy(P.CAMERA_K).cuda(), torch.from_numpy(P.CAMERA_DIST).cuda())
    check_against_least_squares(kp, rot.cpu().numpy(), tv.cpu().numpy(), st.cpu().numpy(), Rs, ts,
                                expect_outliers=int(round(outliers * 11)))


@pytest.mark.gpu
def test_hip_degenerate_cases_equal_the_oracle(gpu_ops):
    """Near-coplanar 5-point subsets and an exactly planar target: status identical to the C oracle; poses equal to the
    north-star tolerance wherever the configuration is well conditioned (near-coplanar case), and equal statuses plus
    valid rotations in the degenerate planar case (where the 12x12 null space is not unique)."""
    import torch
    rng = np.random.default_rng(5)
    kp, Rs, ts, keep = near_coplanar_subset(64, rng)
    ref = P.solve_batch(kp)
    dev = lambda a: torch.from_numpy(a).cuda()
    rot, tv, st = gpu_ops.pnp_epnp_ransac(dev(kp), dev(P.LANDMARKS), dev(P.CAMERA_K), dev(P.CAMERA_DIST))
    rot, tv, st = rot.cpu().numpy(), tv.cpu().numpy(), st.cpu().numpy()
    assert np.array_equal(st, ref["status"]) and (tv[:, 2] > 0).all()
    # Five nearly coplanar points are an ill-conditioned EPnP (three of the four smallest eigenvalues of M^T M nearly
    # coincide): any last-bit difference in a Jacobi rotation is amplified (round 2: 2.6e-4 rad on 1 of 64 frames, from
    # hypot() of the device's vs glibc's libm).  Both implementations now run the same IEEE operation sequences for hypot /
    # log / pow (det_hypot, det_log, det_powi), so the north-star bar holds on every frame of this set as well.
    ang = P.rot_angle(rot, ref["R"])
    terr = np.linalg.norm(tv - ref["t"], axis=1) / np.linalg.norm(ref["t"], axis=1)
    print("near-coplanar: rotation max %.3g median %.3g rad, translation max %.3g median %.3g" % (ang.max(), np.median(ang), terr.max(), np.median(terr)))
    assert ang.max() <= 1e-4, "near-coplanar: max %.3g median %.3g rad" % (ang.max(), np.median(ang))
    assert terr.max() <= 1e-4
    X = P.LANDMARKS.copy(); X[:, 2] = 0.0
    kpp, Rp, tp = P.synth_keypoints(64, rng, 0.5, 0.0, landmarks=X)
    refp = P.solve_batch(kpp, landmarks=X)
    rot, tv, st = gpu_ops.pnp_epnp_ransac(dev(kpp), dev(X), dev(P.CAMERA_K), dev(P.CAMERA_DIST))
    rot, st = rot.cpu().numpy(), st.cpu().numpy()
    agree = np.mean(st == refp["status"])
    print("planar target: statuses agree on %.1f %% of the frames" % (100 * agree))
    assert agree == 1.0, "planar target: HIP and oracle statuses agree on %.0f %% of the frames" % (100 * agree)
    ok = st > 0
    assert np.abs(np.einsum("nij,nkj->nik", rot[ok], rot[ok]) - np.eye(3)).max() < 1e-9
