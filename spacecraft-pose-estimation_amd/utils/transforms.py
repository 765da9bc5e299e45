"""Host-side crop geometry of the reference (landmark_regression/lib/utils/transforms.py):
get_affine_transform (:57-89), affine_transform (:92-95), get_3rd_point (:98-100), get_dir
(:103-110), flip_back (:15-29), plus the pieces that replace third-party calls on this image
(no cv2 / torchvision): the 3-point affine solve, a bilinear warp, ToTensor/Normalize/Compose.
"""
import numpy as np
import torch


def affine_from_3pts(src, dst):
    """cv2.getAffineTransform(src, dst): the 2x3 float64 map taking the three src points to dst."""
    a = np.zeros((6, 6), dtype=np.float64)
    b = np.zeros(6, dtype=np.float64)
    for i in range(3):
        a[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        a[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale])
    scale_tmp = np.asarray(scale, dtype=np.float32) * np.float32(200.0)
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, float(src_w) * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    return affine_from_3pts(dst, src) if inv else affine_from_3pts(src, dst)


def affine_transform(pt, t):
    return np.dot(t, np.array([pt[0], pt[1], 1.0]).T)[:2]


def warp_affine_bilinear(img, trans, out_wh):
    """cv2.warpAffine(img, trans, (W, H), flags=INTER_LINEAR) with constant-0 border, restated in
    NumPy: dst(x, y) = bilinear(src, M^-1 [x, y, 1]).  (OpenCV interpolates with 5-bit fixed-point
    weights; this float version differs from it by at most ~1 grey level -- parity unpinned.)"""
    w, h = int(out_wh[0]), int(out_wh[1])
    m = np.vstack([np.asarray(trans, dtype=np.float64), [0, 0, 1]])
    minv = np.linalg.inv(m)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    sx = minv[0, 0] * xs + minv[0, 1] * ys + minv[0, 2]
    sy = minv[1, 0] * xs + minv[1, 1] * ys + minv[1, 2]
    x0 = np.floor(sx).astype(np.int64); y0 = np.floor(sy).astype(np.int64)
    fx = (sx - x0)[..., None]; fy = (sy - y0)[..., None]
    src = img.astype(np.float32)
    if src.ndim == 2:
        src = src[..., None]
    hh, ww = src.shape[:2]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < hh) & (xx >= 0) & (xx < ww)
        v = src[np.clip(yy, 0, hh - 1), np.clip(xx, 0, ww - 1)]
        return v * ok[..., None]
    out = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) +
           tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def flip_back(output_flipped, matched_parts):
    assert output_flipped.ndim == 4, "output_flipped should be [batch_size, num_joints, height, width]"
    output_flipped = output_flipped[:, :, :, ::-1]
    for pair in matched_parts:
        tmp = output_flipped[:, pair[0], :, :].copy()
        output_flipped[:, pair[0], :, :] = output_flipped[:, pair[1], :, :]
        output_flipped[:, pair[1], :, :] = tmp
    return output_flipped


# ---- minimal stand-ins for torchvision.transforms used by tools/test.py:106-114 ----
class ToTensor:
    def __call__(self, pic):
        arr = np.ascontiguousarray(pic)
        t = torch.from_numpy(arr)
        if t.ndim == 2:
            t = t[:, :, None]
        t = t.permute(2, 0, 1).contiguous()
        return t.float().div(255) if t.dtype == torch.uint8 else t


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
