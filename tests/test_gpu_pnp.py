"""Parity of the one-wavefront-per-frame EPnP+RANSAC kernel (csrc/pnp.hip, C ABI
scpose_pnp_epnp_ransac) against the scalar C oracle (oracle/pnp_ref.c).

Tolerance (BASELINE.json north_star): rotation <= 1e-4 rad (geodesic angle of R_ref^T R),
translation <= 1e-4 relative (|dt|/|t|); inlier counts (status) must be identical.
The oracle itself is "parity unpinned" w.r.t. cv2 (see oracle/pnp_ref.c); both are also
checked against the generating ground-truth pose."""
import numpy as np
import pytest
import torch

from oracle import pnp_ref as P

pytestmark = pytest.mark.gpu

ROT_TOL, T_TOL = 1e-4, 1e-4


def _gpu(gpu_ops, kp, landmarks=P.LANDMARKS, **kw):
    rot, tv, st, rv = gpu_ops.pnp_epnp_ransac(torch.from_numpy(kp).cuda(), torch.from_numpy(landmarks).cuda(),
                                              torch.from_numpy(P.CAMERA_K).cuda(), torch.from_numpy(P.CAMERA_DIST).cuda(),
                                              want_rvec=True, **kw)
    return rot.cpu().numpy(), tv.cpu().numpy(), st.cpu().numpy(), rv.cpu().numpy()


def _check(gpu, ref):
    rot, tv, st, rv = gpu
    assert np.array_equal(st, ref["status"]), "inlier counts differ: %s vs %s" % (st[:16], ref["status"][:16])
    ok = st > 0
    ang = P.rot_angle(rot[ok], ref["R"][ok])
    terr = np.linalg.norm(tv[ok] - ref["t"][ok], axis=1) / np.linalg.norm(ref["t"][ok], axis=1)
    assert ang.max() <= ROT_TOL, "rotation differs by %.3g rad" % ang.max()
    assert terr.max() <= T_TOL, "translation differs by %.3g (relative)" % terr.max()
    # orthonormal output
    eye = np.einsum("nij,nkj->nik", rot[ok], rot[ok])
    assert np.abs(eye - np.eye(3)).max() < 1e-12
    return ang.max(), terr.max()


@pytest.mark.parametrize("noise,outliers", [(0.0, 0.0), (1.0, 0.0), (1.0, 0.1), (1.0, 0.3), (3.0, 0.3)])
def test_pnp_matches_oracle(gpu_ops, noise, outliers):
    rng = np.random.default_rng(int(noise * 10 + outliers * 100))
    kp, Rs, ts = P.synth_keypoints(128, rng, noise, outliers)
    ref = P.solve_batch(kp)
    gpu = _gpu(gpu_ops, kp)
    a, t = _check(gpu, ref)
    print("noise %.1f outliers %.1f: max rot diff %.2e rad, max t diff %.2e, mean RANSAC iters %.1f" % (
        noise, outliers, a, t, ref["iters"].mean()))
    if noise == 0.0:   # known answer: exact projections recover the generating pose
        assert P.rot_angle(gpu[0], Rs).max() < 5e-6
        assert (np.linalg.norm(gpu[1] - ts, axis=1) / np.linalg.norm(ts, axis=1)).max() < 5e-6


def test_pnp_confidence_filter_and_failures(gpu_ops):
    rng = np.random.default_rng(5)
    kp, _, _ = P.synth_keypoints(8, rng, 0.5, 0.0)
    kp[0, :, 2] = 0.0                     # nothing passes (conf > ~2e-10 false): <4 points -> -1
    kp[1, 3:, 2] = 0.0                    # 3 points -> -1
    kp[2, 4:, 2] = -1.0                   # exactly 4 points -> OpenCV's P3P kernel, no RANSAC: status 4
    kp[3, 5:, 2] = 0.0                    # exactly 5 points: direct EPnP, no RANSAC
    kp[4, 6:, 2] = 1e-12                  # below the final threshold 0.95*0.8^100 = 1.9e-10 -> 6 points
    kp[5, :, 2] = 1e-9                    # just above it: all 11
    kp[6, 0, 2] = np.float32(0.95 * 0.8 ** 100)   # boundary: strict '>' on float32
    ref = P.solve_batch(kp)
    gpu = _gpu(gpu_ops, kp)
    assert list(ref["status"][:4]) == [-1, -1, 4, 5]
    _check(gpu, ref)
    assert np.array_equal(gpu[0][0], np.eye(3)) and np.all(gpu[1][0] == 0)


def test_pnp_exactly_four_points_take_the_p3p_branch(gpu_ops):
    """export_predicted_poses_real.py:199-201 with four usable landmarks: cv2.solvePnPRansac switches to solvePnP(SOLVEPNP_P3P)
    (OpenCV 3.4 solvepnp.cpp; p3p.cpp, polynom_solver.cpp) -- Gao's P3P on the first three points, the fourth picks among up to
    four poses.  The kernel spreads the quartic's roots over lanes 0-3 (csrc/pnp.hip: solve_p3p); the oracle runs them one after
    the other.  Noise-free and noisy key points, four landmark subsets (point order and geometry vary).
    P3P on a 0.7 m target at 3-10 m is ill-conditioned in a third of the cases (x = |PA| / |PC| ~ 1: near-multiple roots of the
    quartic -- for cv2 as for this restatement: the returned pose then misses its own first three points by > 1e-2 px), so the
    known-answer check against the generating pose is made on the well-conditioned frames only; kernel = oracle is asserted
    on ALL frames (measured: 4e-8 rad, bit-identical translations, the ill-conditioned ones included)."""
    rng = np.random.default_rng(46)
    subsets = [[0, 2, 5, 9], [1, 3, 4, 10], [8, 6, 0, 7], [2, 9, 4, 3]]
    n = 256
    kp, Rs, ts = P.synth_keypoints(n, rng, 0.0, 0.0)
    kp[n // 2:, :, :2] += rng.normal(0, 0.5, (n - n // 2, 11, 2)).astype(np.float32)
    kp[:, :, 2] = 1e-11
    for i in range(n):
        kp[i, subsets[i % 4], 2] = 0.99
    ref = P.solve_batch(kp)
    gpu = _gpu(gpu_ops, kp)
    assert set(ref["status"].tolist()) <= {4, -2} and (ref["status"] == 4).mean() > 0.9
    a, t = _check(gpu, ref)                                # identical status, rotation <= 1e-4 rad, translation <= 1e-4 on every frame
    ok = ref["status"] == 4
    well = np.zeros(n, dtype=bool)
    for i in np.nonzero(ok)[0]:
        sub = sorted(subsets[i % 4])                       # the filter keeps landmark order
        uv = P.project(ref["R"][i], ref["t"][i], P.LANDMARKS[sub].astype(np.float32).astype(np.float64))
        well[i] = np.abs(uv[:3] - kp[i, sub[:3], :2]).max() < 1e-2
    print("P3P: %d frames with a pose (%d well-conditioned); kernel vs oracle: max rot diff %.2e rad, t %.2e" % (ok.sum(), well.sum(), a, t))
    assert well.sum() >= 0.5 * ok.sum()
    clean = well[: n // 2]                                 # noise-free + well-conditioned: the generating pose comes back
    assert P.rot_angle(gpu[0][: n // 2][clean], Rs[: n // 2][clean]).max() < 1e-3


def test_pnp_threshold_loop_with_24_landmarks(gpu_ops):
    """J = 24 (Hubble default, evaluate_pipeline.py:44): the loop stops at the first threshold
    that admits >= 15 points, so low-confidence landmarks are really dropped."""
    rng = np.random.default_rng(9)
    lm = rng.uniform(-1, 1, (24, 3))
    kp, Rs, ts = P.synth_keypoints(32, rng, 0.5, 0.0, landmarks=lm)
    kp[:, :, 2] = rng.uniform(0.2, 1.0, (32, 24)).astype(np.float32)
    bad = kp[:, :, 2] < 0.5
    kp[bad, 0] += 300.0                   # low-confidence points are gross outliers
    ref = P.solve_batch(kp, landmarks=lm)
    gpu = _gpu(gpu_ops, kp, landmarks=lm)
    _check(gpu, ref)


def test_pnp_full_batch_properties(gpu_ops):
    """BASELINE config B batch (256 frames): every clean frame keeps all 11 inliers and the
    reprojection of the recovered pose stays within the RANSAC gate."""
    rng = np.random.default_rng(21)
    kp, Rs, ts = P.synth_keypoints(256, rng, 1.0, 0.0)
    rot, tv, st, _ = _gpu(gpu_ops, kp)
    assert (st == 11).all()
    for i in range(0, 256, 17):
        uv = P.project_numpy(rot[i], tv[i], P.LANDMARKS)
        assert np.abs(uv - kp[i, :, :2]).max() < 15.0


def test_pnp_empty_batch_and_bad_args(gpu_ops):
    rot, tv, st = gpu_ops.pnp_epnp_ransac(torch.zeros(0, 11, 3).cuda(), torch.from_numpy(P.LANDMARKS).cuda(),
                                          torch.from_numpy(P.CAMERA_K).cuda(), torch.from_numpy(P.CAMERA_DIST).cuda())
    assert rot.shape == (0, 3, 3) and st.shape == (0,)
    with pytest.raises(gpu_ops.nat.NativeError, match="landmarks"):
        gpu_ops.pnp_epnp_ransac(torch.zeros(1, 65, 3).cuda(), torch.zeros(65, 3, dtype=torch.float64).cuda(),
                                torch.from_numpy(P.CAMERA_K).cuda(), None)


@pytest.mark.parametrize("j,spoil", [(11, False), (11, True), (24, False), (24, True)])
def test_pnp_filter_and_projection_against_the_reference_camera_fixture(gpu_ops, j, spoil):
    """The kernel's confidence filter and camera model against vectors made by the REFERENCE's own code
    (tests/golden/camera_reference_outputs.npz: export_predicted_poses_real.py:186-197 executed line for line on seeded scores;
    speed_plus_utils/utils.py:108-139 project): key points are exact reference-model projections under the fixture's poses, so the
    frame's answer is known -- status = the number of landmarks the reference's loop admits (P3P for four, -1 below), the pose =
    the generating one; with `spoil` the landmarks the loop drops are 300 px off, so admitting one would show in the pose."""
    from test_camera_golden import G, frames_from_fixture, expected_status
    lm = G["landmarks"] if j == 11 else np.random.default_rng(24).uniform(-0.6, 0.6, (24, 3))
    kp, Rs, ts, count = frames_from_fixture(j, lm, spoil)
    rot, tv, st, _ = _gpu(gpu_ops, kp, landmarks=np.ascontiguousarray(lm))
    p3p = count == 4
    assert np.array_equal(st[~p3p], expected_status(count)[~p3p]) and set(st[p3p].tolist()) <= {4, -2}
    ok = st >= 5
    assert P.rot_angle(rot[ok], Rs[ok]).max() < 2e-5
    assert (np.linalg.norm(tv[ok] - ts[ok], axis=1) / np.linalg.norm(ts[ok], axis=1)).max() < 2e-5
    _check((rot, tv, st, None), P.solve_batch(kp, landmarks=lm))


def test_pnp_rows_entry_point_writes_the_gather_block(gpu_ops):
    """scpose_pnp_epnp_ransac_rows (ABI 7): the same solve, written as one (N, 13) [R, t, status] row per frame -- bit-identical
    to the three-output entry point, failures (identity / zero / negative status) included."""
    rng = np.random.default_rng(77)
    kp, _, _ = P.synth_keypoints(64, rng, 1.0, 0.2)
    kp[3, :, 2] = 0.0; kp[9, 4:, 2] = -1.0          # a frame with no usable landmark, one with exactly four
    rot, tv, st, _ = _gpu(gpu_ops, kp)
    rows = torch.full((64, 13), float("nan"), dtype=torch.float64, device="cuda")
    out = gpu_ops.pnp_epnp_ransac(torch.from_numpy(kp).cuda(), torch.from_numpy(P.LANDMARKS).cuda(), torch.from_numpy(P.CAMERA_K).cuda(),
                                  torch.from_numpy(P.CAMERA_DIST).cuda(), rows=rows)
    assert out is rows
    r = rows.cpu().numpy()
    assert np.array_equal(r[:, :9], rot.reshape(64, 9)) and np.array_equal(r[:, 9:12], tv) and np.array_equal(r[:, 12], st.astype(np.float64))
    with pytest.raises(gpu_ops.nat.NativeError, match="rows"):
        gpu_ops.pnp_epnp_ransac(torch.from_numpy(kp).cuda(), torch.from_numpy(P.LANDMARKS).cuda(), torch.from_numpy(P.CAMERA_K).cuda(),
                                torch.from_numpy(P.CAMERA_DIST).cuda(), rows=torch.zeros(64, 12, dtype=torch.float64, device="cuda"))
