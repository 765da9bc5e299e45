"""NumPy restatement of the reference heatmap decode (TEST ORACLE, not product).

Follows /root/reference/landmark_regression:
  get_max_preds ............ lib/core/inference.py:18-46
  get_final_preds .......... lib/core/inference.py:49-79 (incl. its per-(n,p) Python loop)
  transform_preds .......... lib/utils/transforms.py:49-54
  get_affine_transform ..... lib/utils/transforms.py:57-89 (3-point construction, float32 points)
  affine_transform ......... lib/utils/transforms.py:92-95
  get_3rd_point / get_dir .. lib/utils/transforms.py:98-110
cv2.getAffineTransform (third party, absent here) is restated as the 6x6 linear solve it
performs in float64 on the three float32 point pairs.
"""
import math

import numpy as np


def get_max_preds(batch_heatmaps):
    assert isinstance(batch_heatmaps, np.ndarray) and batch_heatmaps.ndim == 4
    n, j, _, width = batch_heatmaps.shape
    flat = batch_heatmaps.reshape((n, j, -1))
    idx = np.argmax(flat, 2).reshape((n, j, 1))
    maxvals = np.amax(flat, 2).reshape((n, j, 1))
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % width
    preds[:, :, 1] = np.floor(preds[:, :, 1] / width)
    mask = np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    preds *= mask
    return preds, maxvals


def _affine_from_3pts(src, dst):
    """The map cv2.getAffineTransform(src, dst) returns: 2x3 float64 M with M @ [x,y,1] = dst."""
    a = np.zeros((6, 6), dtype=np.float64)
    b = np.zeros(6, dtype=np.float64)
    for i in range(3):
        a[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        a[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def _third(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    scale = np.asarray(scale, dtype=np.float32)
    center = np.asarray(center, dtype=np.float32)
    scale_tmp = scale * np.float32(200.0)
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = [0.0, float(src_w) * -0.5]
    src_dir = [p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs]
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2, :] = _third(src[0, :], src[1, :])
    dst[2, :] = _third(dst[0, :], dst[1, :])
    return _affine_from_3pts(dst, src) if inv else _affine_from_3pts(src, dst)


def transform_preds(coords, center, scale, output_size):
    target = np.zeros(coords.shape)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        target[p, 0:2] = np.dot(trans, np.array([coords[p, 0], coords[p, 1], 1.0]))
    return target


def get_final_preds(post_process, batch_heatmaps, center, scale):
    coords, maxvals = get_max_preds(batch_heatmaps)
    hh, hw = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    if post_process:
        for n in range(coords.shape[0]):
            for p in range(coords.shape[1]):
                hm = batch_heatmaps[n][p]
                px = int(math.floor(coords[n][p][0] + 0.5))
                py = int(math.floor(coords[n][p][1] + 0.5))
                if 1 < px < hw - 1 and 1 < py < hh - 1:
                    diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(coords.shape[0]):
        preds[i] = transform_preds(coords[i], center[i], scale[i], [hw, hh])
    return preds, maxvals


def decode_xyc(post_process, batch_heatmaps, center, scale):
    """validate()'s all_preds rows (lib/core/function.py:392-393): (N,J,3) f32 [x,y,maxval]."""
    preds, maxvals = get_final_preds(post_process, batch_heatmaps, center, scale)
    out = np.zeros((preds.shape[0], preds.shape[1], 3), dtype=np.float32)
    out[:, :, 0:2] = preds[:, :, 0:2]
    out[:, :, 2:3] = maxvals
    return out


def gaussian_heatmaps(n, j, h, w, rng, sigma=2.0, amp=1.0):
    """Synthetic peaked maps in the style of JointsDataset.generate_target
    (lib/dataset/JointsDataset.py:264-332): gaussian with peak `amp` at an integer centre."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = np.zeros((n, j, h, w), dtype=np.float32)
    cx = rng.integers(0, w, size=(n, j))
    cy = rng.integers(0, h, size=(n, j))
    for a in range(n):
        for b in range(j):
            hm[a, b] = amp * np.exp(-((xs - cx[a, b]) ** 2 + (ys - cy[a, b]) ** 2) / (2 * sigma ** 2))
    return hm, cx, cy
