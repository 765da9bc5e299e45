"""Known-answer tests of the C restatement of OpenCV's EPnP+RANSAC (oracle/pnp_ref.c).
cv2 is absent from this image and from /root/reference (parity UNPINNED, see the file header):
these tests anchor the restatement on analytic facts instead -- exact projections recover the
generating pose, Rodrigues is an involution, undistortion inverts the distortion model, the
RNG follows the published MWC recurrence, RANSAC rejects gross outliers."""
import numpy as np
import pytest

from oracle import pnp_ref as P


def test_rng_follows_opencv_mwc_recurrence():
    state = (1 << 64) - 1
    want = []
    for _ in range(12):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        want.append((state & 0xFFFFFFFF) % 11)
    assert list(P.rng_draws(11, 12)) == want


def test_rodrigues_roundtrip_and_special_cases():
    rng = np.random.default_rng(0)
    for _ in range(50):
        R = P.random_rotation(rng)
        rv = P.rodrigues(R)
        assert np.abs(P.rodrigues(rv) - R).max() < 1e-12
        assert abs(np.linalg.norm(rv) - P.rot_angle(R, np.eye(3))) < 1e-7
    assert np.array_equal(P.rodrigues(np.zeros(3)), np.eye(3))
    Rpi = np.diag([1.0, -1.0, -1.0])                       # theta = pi about x: the s < 1e-5 branch
    assert np.abs(P.rodrigues(P.rodrigues(Rpi)) - Rpi).max() < 1e-12


def test_projection_and_undistortion_are_inverse():
    rng = np.random.default_rng(1)
    R = P.random_rotation(rng, 40)
    t = np.array([0.3, -0.2, 5.0])
    uv = P.project(R, t)
    assert np.abs(uv - P.project_numpy(R, t, P.LANDMARKS)).max() < 1e-9
    pc = P.LANDMARKS @ R.T + t
    assert np.abs(P.undistort(uv) - pc[:, :2] / pc[:, 2:]).max() < 1e-9


@pytest.mark.parametrize("npts", [5, 6, 11])
def test_epnp_recovers_exact_pose(npts):
    rng = np.random.default_rng(npts)
    for _ in range(20):
        R = P.random_rotation(rng)
        t = np.array([rng.uniform(-1, 1), rng.uniform(-0.6, 0.6), rng.uniform(3, 10)])
        obj = P.LANDMARKS[:npts]
        rv, tv = P.epnp(obj, P.project_numpy(R, t, obj))
        assert P.rot_angle(P.rodrigues(rv), R) < 1e-6
        assert np.linalg.norm(tv - t) / np.linalg.norm(t) < 1e-6


def test_ransac_clean_noisy_and_outliers():
    rng = np.random.default_rng(7)
    kp, Rs, ts = P.synth_keypoints(64, rng, 0.0, 0.0)
    o = P.solve_batch(kp)
    assert (o["status"] == 11).all() and (o["iters"] == 1).all()       # all-inlier model -> niters becomes 0
    assert P.rot_angle(o["R"], Rs).max() < 5e-6                          # float32 keypoints limit this
    kp, Rs, ts = P.synth_keypoints(64, rng, 1.0, 0.3)                    # 3 of 11 landmarks replaced
    o = P.solve_batch(kp)
    assert (o["status"] >= 8).all()
    assert np.median(P.rot_angle(o["R"], Rs)) < 2e-2
    assert (o["iters"] < 200).all()


def test_confidence_threshold_loop():
    rng = np.random.default_rng(3)
    kp, _, _ = P.synth_keypoints(4, rng, 0.5, 0.0)
    kp[0, :, 2] = 0.0                  # J = 11 < 15: loop runs 100 times, threshold 0.95*0.8^100 = 1.9e-10
    kp[1, 5:, 2] = 1e-11               # below the final threshold -> 5 points: direct EPnP
    kp[2, :, 2] = 1e-9                 # above it: all 11 kept
    o = P.solve_batch(kp)
    assert list(o["status"][:3]) == [-1, 5, 11]
    assert abs(0.95 * 0.8 ** 100 - 1.935e-10) < 1e-12


def test_p3p_recovers_exact_pose_from_four_points():
    """Exactly four usable landmarks: solvePnPRansac -> solvePnP(SOLVEPNP_P3P) (pnp_ref.c: solve_pnp_p3p).  Up to four poses fit the
    first three points, the fourth picks one.  Gao's closed-form P3P is ill-conditioned when x = |PA| / |PC| ~ 1 -- a 0.7 m target at
    3-10 m: the quartic's roots nearly coincide and the pose misses even its own three points (for cv2 as for this restatement) -- so
    the known answer is asserted where the returned pose reprojects its first three points to < 1e-2 px (about two thirds of the
    trials), and in the median over all of them."""
    rng = np.random.default_rng(44)
    sets = [[0, 1, 2, 4], [8, 9, 10, 3], [0, 5, 6, 10], [4, 1, 7, 9]]
    ang, terr, self3 = [], [], []
    n_ok = 0
    for k in range(200):
        R = P.random_rotation(rng)
        t = np.array([rng.uniform(-1, 1), rng.uniform(-0.6, 0.6), rng.uniform(3, 10)])
        obj = P.LANDMARKS[sets[k % 4]].astype(np.float32).astype(np.float64)
        img = P.project_numpy(R, t, obj).astype(np.float32).astype(np.float64)
        ok, rv, tv = P.p3p(obj, img)
        if not ok:
            continue
        n_ok += 1
        uv = P.project(P.rodrigues(rv), tv, obj)
        self3.append(np.abs(uv[:3] - img[:3]).max())
        ang.append(P.rot_angle(P.rodrigues(rv), R))
        terr.append(np.linalg.norm(tv - t) / np.linalg.norm(t))
    ang, terr, self3 = np.array(ang), np.array(terr), np.array(self3)
    well = self3 < 1e-2
    assert n_ok >= 190 and well.mean() > 0.5
    assert np.median(ang) < 1e-4 and np.median(terr) < 1e-4
    assert ang[well].max() < 2e-3 and terr[well].max() < 2e-3


def test_four_confident_landmarks_go_through_p3p():
    """export_predicted_poses_real.py:186-201 with only four scores above the final threshold: status 4 and the P3P pose."""
    rng = np.random.default_rng(45)
    kp, Rs, ts = P.synth_keypoints(8, rng, 0.0, 0.0)
    keep = [0, 2, 5, 9]
    kp[:, :, 2] = 1e-11
    kp[:, keep, 2] = 0.99
    o = P.solve_batch(kp)
    assert (o["status"] == 4).all()
    assert np.median(P.rot_angle(o["R"], Rs)) < 1e-4            # (the tail is P3P's conditioning: see the test above)
    assert np.median(np.linalg.norm(o["t"] - ts, axis=1) / np.linalg.norm(ts, axis=1)) < 1e-4
    # the pose equals the direct P3P call on those four correspondences
    ok, rv, tv = P.p3p(P.LANDMARKS[keep].astype(np.float32).astype(np.float64), kp[0, keep, :2].astype(np.float64))
    assert ok and np.abs(rv - o["rvec"][0]).max() < 1e-12 and np.abs(tv - o["t"][0]).max() < 1e-12
