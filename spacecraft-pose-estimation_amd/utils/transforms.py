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


def invert_affine_cv(trans):
    """The inverse map cv2.warpAffine derives from a forward 2x3 matrix (OpenCV 3.4 imgwarp.cpp, cv::warpAffine,
    branch !WARP_INVERSE_MAP), in its operation order: returns the six doubles [a11 a12 b1 a21 a22 b2]."""
    m = np.asarray(trans, dtype=np.float64).reshape(6).copy()
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0] = a11; m[1] *= -d; m[3] *= -d; m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2] = b1; m[5] = b2
    return m


def _sat_int(v):
    """cv::saturate_cast<int>(double): round half to even (cvRound), clamp to int32."""
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_bilinear(img, trans, out_wh):
    """cv2.warpAffine(img, trans, (W, H), flags=INTER_LINEAR), constant-0 border, for uint8 images: the FIXED-POINT
    algorithm OpenCV runs (3.4 imgwarp.cpp: hal::warpAffine + WarpAffineInvoker + remapBilinear with
    FixedPtCast<int, uchar, 15>), restated in NumPy:
      * source coordinates in 1/1024 px: X = round(M0*x*1024) + round((M1*y + M2)*1024) + 16, then >> 5, i.e. quantised
        to 1/32 px (INTER_BITS = 5) with the +16 as rounding offset;
      * integer bilinear weights (32-a)(32-b)*32 ... a*b*32 for the 5-bit fractions a, b (they sum to 32768; OpenCV's
        table entry for a = b = 0 is [32767, 0, 0, 1], which yields the same pixel as [32768, 0, 0, 0]);
      * pixel = (sum of the four taps * weights + 16384) >> 15, taps outside the frame count as 0.
    cv2 is not available to compare with (parity unpinned); the anchors are the algorithm's known answers
    (identity, integer shifts, half-pixel averages round up) in tests/."""
    w, h = int(out_wh[0]), int(out_wh[1])
    m = invert_affine_cv(trans)
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = _sat_int(m[0] * xs * 1024.0)
    bdelta = _sat_int(m[3] * xs * 1024.0)
    x0 = _sat_int((m[1] * ys + m[2]) * 1024.0) + 16
    y0 = _sat_int((m[4] * ys + m[5]) * 1024.0) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767); sy = np.clip(Y >> 5, -32768, 32767)
    a = (X & 31)[..., None]; b = (Y & 31)[..., None]
    src = img if img.ndim == 3 else img[..., None]
    hh, ww = src.shape[:2]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < hh) & (xx >= 0) & (xx < ww)
        v = src[np.clip(yy, 0, hh - 1), np.clip(xx, 0, ww - 1)].astype(np.int64)
        return v * ok[..., None]
    acc = (tap(sy, sx) * ((32 - a) * (32 - b) * 32) + tap(sy, sx + 1) * (a * (32 - b) * 32) +
           tap(sy + 1, sx) * ((32 - a) * b * 32) + tap(sy + 1, sx + 1) * (a * b * 32))
    out = np.clip((acc + 16384) >> 15, 0, 255).astype(np.uint8)
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
