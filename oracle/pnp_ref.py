"""ctypes front end of oracle/pnp_ref.c plus the synthetic pose / keypoint generator
(TEST ORACLE, not product -- see pnp_ref.c for the pinning status: camera model and confidence filter pinned to the
reference's own code, cv2.solvePnPRansac internals unpinned)."""
import ctypes
import os
import subprocess
from ctypes import c_double, c_int, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libpnp_ref.so")
_lib = None

# 11 Tango landmarks [m] and the SPEED+ camera of the reference's fixtures
# (object_detection/speed_plus_utils/landmarks.csv:2-12, calibration.json:1-24): data, not code.
LANDMARKS = np.array([
    [0.36940446496009827, -0.3845726549625397, 0.16007566452026367],
    [0.36786314845085144, 0.3836139440536499, 0.16053038835525513],
    [-0.36881211400032043, 0.38277047872543335, 0.16048267483711243],
    [-0.36801040172576904, -0.3831963539123535, 0.16058564186096191],
    [0.36815810203552246, -0.26237574219703674, -0.16152474284172058],
    [0.36859363317489624, 0.30254653096199036, -0.15993139147758484],
    [-0.36717548966407776, 0.30379965901374817, -0.1599225401878357],
    [-0.3663908839225769, -0.2586885094642639, -0.1586388796567917],
    [0.30565211176872253, -0.5800656676292419, 0.08969831466674805],
    [0.5425941348075867, 0.48880907893180847, 0.09245043992996216],
    [-0.5449637770652771, 0.48740869760513306, 0.09220433235168457]], dtype=np.float64)
CAMERA_K = np.array([[2988.5795163815555, 0, 960], [0, 2988.3401159176124, 600], [0, 0, 1]], dtype=np.float64)
CAMERA_DIST = np.array([-0.22383016606510672, 0.51409797089106379, -0.00066499611998340662,
                        -0.00021404771667484594, -0.13124227429077406], dtype=np.float64)


def build():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pnp_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.pnp_ref_batch.restype = None
        _lib.pnp_ref_batch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_double,
                                       c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def solve_batch(kp_xyc, landmarks=LANDMARKS, K=CAMERA_K, dist=CAMERA_DIST, conf_thr0=0.95, min_pts=15, thr_decay=0.8,
                thr_iters=100, max_iters=10000, reproj_err=15.0, confidence=0.99):
    """export_predicted_poses_real.py:186-203 for every frame, serially.  Returns dict with
    R (N,3,3), t (N,3), rvec (N,3), status (N,), iters (N,)."""
    kp = np.ascontiguousarray(kp_xyc, dtype=np.float32)
    n, j, _ = kp.shape
    lm = np.ascontiguousarray(landmarks, dtype=np.float64)
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    dc = np.ascontiguousarray(dist if dist is not None else np.zeros(5), dtype=np.float64)
    R = np.zeros((n, 3, 3)); t = np.zeros((n, 3)); rv = np.zeros((n, 3))
    st = np.zeros(n, dtype=np.int32); it = np.zeros(n, dtype=np.int32)
    lib().pnp_ref_batch(_p(kp), _p(lm), _p(Kc), _p(dc), n, j, conf_thr0, min_pts, thr_decay, thr_iters, max_iters,
                        reproj_err, confidence, _p(R), _p(t), _p(rv), _p(st), _p(it))
    return {"R": R, "t": t, "rvec": rv, "status": st, "iters": it}


def epnp(obj, img, K=CAMERA_K, dist=CAMERA_DIST):
    l = lib()
    obj = np.ascontiguousarray(obj, dtype=np.float64); img = np.ascontiguousarray(img, dtype=np.float64)
    rv = np.zeros(3); tv = np.zeros(3)
    l.pnp_ref_epnp(_p(np.ascontiguousarray(K, dtype=np.float64)), _p(np.ascontiguousarray(dist, dtype=np.float64)),
                   _p(obj), _p(img), c_int(len(obj)), _p(rv), _p(tv))
    return rv, tv


def p3p(obj4, img4, K=CAMERA_K, dist=CAMERA_DIST):
    """solvePnP(SOLVEPNP_P3P) on exactly four correspondences (pnp_ref.c: solve_pnp_p3p); returns (ok, rvec, tvec)."""
    obj = np.ascontiguousarray(obj4, dtype=np.float64); img = np.ascontiguousarray(img4, dtype=np.float64)
    assert obj.shape == (4, 3) and img.shape == (4, 2)
    rv = np.zeros(3); tv = np.zeros(3)
    lib().pnp_ref_p3p.restype = c_int
    ok = lib().pnp_ref_p3p(_p(np.ascontiguousarray(K, dtype=np.float64)), _p(np.ascontiguousarray(dist, dtype=np.float64)),
                           _p(obj), _p(img), _p(rv), _p(tv))
    return bool(ok), rv, tv


def rodrigues(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(3 if x.size == 9 else 9)
    lib().pnp_ref_rodrigues(_p(x), c_int(1 if x.size == 9 else 0), _p(out))
    return out if x.size == 9 else out.reshape(3, 3)


def project(R, t, obj=LANDMARKS, K=CAMERA_K, dist=CAMERA_DIST):
    obj = np.ascontiguousarray(obj, dtype=np.float64)
    uv = np.zeros((len(obj), 2))
    lib().pnp_ref_project(_p(np.ascontiguousarray(K, dtype=np.float64)), _p(np.ascontiguousarray(dist, dtype=np.float64)),
                          _p(np.ascontiguousarray(R, dtype=np.float64)), _p(np.ascontiguousarray(t, dtype=np.float64)),
                          _p(obj), c_int(len(obj)), _p(uv))
    return uv


def undistort(uv, K=CAMERA_K, dist=CAMERA_DIST):
    uv = np.ascontiguousarray(uv, dtype=np.float64)
    xy = np.zeros_like(uv)
    lib().pnp_ref_undistort(_p(np.ascontiguousarray(K, dtype=np.float64)), _p(np.ascontiguousarray(dist, dtype=np.float64)),
                            _p(uv), c_int(len(uv)), _p(xy))
    return xy


def conf_mask(conf, conf_thr0=0.95, min_pts=15, thr_decay=0.8, thr_iters=100):
    """export_predicted_poses_real.py:186-197 for one frame's scores: the boolean mask of the landmarks that reach the solver."""
    c = np.ascontiguousarray(conf, dtype=np.float32)
    mask = np.zeros(len(c), dtype=np.uint8)
    lib().pnp_ref_conf_mask.restype = c_int
    lib().pnp_ref_conf_mask(_p(c), c_int(len(c)), c_double(conf_thr0), c_int(min_pts), c_double(thr_decay), c_int(thr_iters), _p(mask))
    return mask.astype(bool)


def rng_draws(count, n):
    out = np.zeros(n, dtype=np.int32)
    lib().pnp_ref_rng_draws.restype = ctypes.c_uint
    lib().pnp_ref_rng_draws(c_int(count), c_int(n), _p(out))
    return out


# ------------------------------------------------------------------ synthetic data
def project_numpy(R, t, X, K=CAMERA_K, dist=CAMERA_DIST):
    """Pinhole + (k1,k2,p1,p2,k3) model of export_predicted_poses_real.py:104-121 in NumPy
    (independent of the C code: used to generate ground truth)."""
    pc = X @ R.T + t
    x0, y0 = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    r2 = x0 * x0 + y0 * y0
    cd = 1 + dist[0] * r2 + dist[1] * r2 * r2 + dist[4] * r2 * r2 * r2
    x1 = x0 * cd + dist[2] * 2 * x0 * y0 + dist[3] * (r2 + 2 * x0 * x0)
    y1 = y0 * cd + dist[2] * (r2 + 2 * y0 * y0) + dist[3] * 2 * x0 * y0
    return np.stack([K[0, 0] * x1 + K[0, 2], K[1, 1] * y1 + K[1, 2]], 1)


def random_rotation(rng, max_deg=180.0):
    axis = rng.standard_normal(3); axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(-max_deg, max_deg))
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def synth_keypoints(n, rng, noise_px=1.0, outlier_frac=0.0, landmarks=LANDMARKS, K=CAMERA_K, dist=CAMERA_DIST,
                    width=1920, height=1200):
    """Seeded poses (|t| in [3,10] m, target inside the image), projected landmarks + N(0,noise) px,
    conf = 1, a fraction of landmarks replaced by uniform image points.  Returns kp (n,J,3) f32, R, t."""
    j = len(landmarks)
    kp = np.zeros((n, j, 3), dtype=np.float32)
    Rs = np.zeros((n, 3, 3)); ts = np.zeros((n, 3))
    for i in range(n):
        while True:
            R = random_rotation(rng)
            z = rng.uniform(3.0, 10.0)
            t = np.array([rng.uniform(-0.25, 0.25) * z, rng.uniform(-0.15, 0.15) * z, z])
            uv = project_numpy(R, t, landmarks, K, dist)
            if (uv[:, 0] > 0).all() and (uv[:, 0] < width).all() and (uv[:, 1] > 0).all() and (uv[:, 1] < height).all():
                break
        uv = uv + rng.standard_normal(uv.shape) * noise_px
        nout = int(round(outlier_frac * j))
        if nout:
            idx = rng.choice(j, nout, replace=False)
            uv[idx, 0] = rng.uniform(0, width, nout)
            uv[idx, 1] = rng.uniform(0, height, nout)
        kp[i, :, :2] = uv
        kp[i, :, 2] = 1.0
        Rs[i], ts[i] = R, t
    return kp, Rs, ts


def rot_angle(Ra, Rb):
    """Geodesic angle [rad] between rotation matrices (batch)."""
    tr = np.einsum("...ij,...ij->...", Ra, Rb)
    return np.arccos(np.clip((tr - 1) / 2, -1, 1))
