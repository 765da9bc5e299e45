"""cv2.warpAffine(src, M, (W, H), flags=cv2.INTER_LINEAR) for uint8 images, constant-0 border (TEST ORACLE, not product).

The reference warps every crop with it (landmark_regression/lib/dataset/JointsDataset.py:191-195).  cv2 is the
third-party wheel opencv-python==3.4.11.41 (environment.yml:37), absent from /root/reference and from this image, so
this file restates the published algorithm of OpenCV 3.4 modules/imgproc/src/imgwarp.cpp, one scalar step per line:
  cv::warpAffine            -- inversion of the forward matrix (branch !WARP_INVERSE_MAP)
  hal::warpAffine           -- adelta / bdelta tables, AB_BITS = 10
  WarpAffineInvoker         -- X0 / Y0 per row, round_delta = 1024/32/2, coordinates with INTER_BITS = 5 fraction bits
  initInterTab2D            -- BilinearTab_i, the 32x32 table of four short weights (scale 2^15) incl. its fix-up step
  remapBilinear<FixedPtCast<int, uchar, 15>>  -- inlier / border-constant handling and the final rounding shift
PARITY UNPINNED: no cv2 here to run, and the reference holds no vector at this boundary.  Pure-Python loops: small cases.
CROSS-CHECKED against a third party (round 6, tests/test_oracle_warp_crosscheck.py): within 0.58 grey levels of scipy.ndimage.affine_transform
(order=1) on every interior pixel of smooth images -- the geometry, conventions and border rule agree with an independent bilinear
resampler; the fixed-point details (1/32 px coordinates, the weight table and its fix-up) remain restated from knowledge.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS


def _cv_round(v):
    return int(np.rint(v))                      # cvRound: to nearest, ties to even


def _sat_int(v):
    return max(-2147483648, min(2147483647, _cv_round(v)))


def _sat_short(v):
    return max(-32768, min(32767, int(v)))


def bilinear_tab_i():
    """BilinearTab_i of initInterTab2D(INTER_LINEAR, fixpt=true): [32*32][4] short weights."""
    tab1 = [(np.float32(1.0) - np.float32(i) / np.float32(INTER_TAB_SIZE), np.float32(i) / np.float32(INTER_TAB_SIZE))
            for i in range(INTER_TAB_SIZE)]
    flat = [0] * (INTER_TAB_SIZE * INTER_TAB_SIZE * 4 + 8)       # static storage: entries not yet written read as 0
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            base = (i * INTER_TAB_SIZE + j) * 4
            isum = 0
            for k1 in range(2):
                vy = tab1[i][k1]
                for k2 in range(2):
                    v = np.float32(vy * tab1[j][k2])
                    q = _sat_short(_cv_round(float(v) * INTER_REMAP_COEF_SCALE))
                    flat[base + k1 * 2 + k2] = q
                    isum += q
            if isum != INTER_REMAP_COEF_SCALE:       # only i = j = 0 (1.0 * 32768 saturates to 32767)
                diff = isum - INTER_REMAP_COEF_SCALE
                ksize2 = 1
                Mk1 = Mk2 = mk1 = mk2 = ksize2
                for k1 in range(ksize2, ksize2 + 2):
                    for k2 in range(ksize2, ksize2 + 2):
                        if flat[base + k1 * 2 + k2] < flat[base + mk1 * 2 + mk2]:
                            mk1, mk2 = k1, k2
                        elif flat[base + k1 * 2 + k2] > flat[base + Mk1 * 2 + Mk2]:
                            Mk1, Mk2 = k1, k2
                if diff < 0:
                    flat[base + Mk1 * 2 + Mk2] = _sat_short(flat[base + Mk1 * 2 + Mk2] - diff)
                else:
                    flat[base + mk1 * 2 + mk2] = _sat_short(flat[base + mk1 * 2 + mk2] - diff)
    return [flat[k * 4:k * 4 + 4] for k in range(INTER_TAB_SIZE * INTER_TAB_SIZE)]


_WTAB = None


def invert(M0):
    M = [float(v) for v in np.asarray(M0, dtype=np.float64).reshape(6)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D
    M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2] = b1; M[5] = b2
    return M


def warp_affine_linear_u8(src, M0, dsize):
    """src HxW or HxWxC uint8; M0 forward 2x3; dsize (W, H)."""
    global _WTAB
    if _WTAB is None:
        _WTAB = bilinear_tab_i()
    img = src if src.ndim == 3 else src[..., None]
    sh, sw, cn = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    M = invert(M0)
    adelta = [_sat_int(M[0] * x * AB_SCALE) for x in range(dw)]
    bdelta = [_sat_int(M[3] * x * AB_SCALE) for x in range(dw)]
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    dst = np.zeros((dh, dw, cn), dtype=np.uint8)
    width1, height1 = max(sw - 1, 0), max(sh - 1, 0)
    for y in range(dh):
        X0 = _sat_int((M[1] * y + M[2]) * AB_SCALE) + round_delta
        Y0 = _sat_int((M[4] * y + M[5]) * AB_SCALE) + round_delta
        for x in range(dw):
            X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS)
            Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS)
            sx, sy = _sat_short(X >> INTER_BITS), _sat_short(Y >> INTER_BITS)
            w = _WTAB[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]
            if 0 <= sx < width1 and 0 <= sy < height1:
                for k in range(cn):
                    t = (int(img[sy, sx, k]) * w[0] + int(img[sy, sx + 1, k]) * w[1] +
                         int(img[sy + 1, sx, k]) * w[2] + int(img[sy + 1, sx + 1, k]) * w[3])
                    dst[y, x, k] = max(0, min(255, (t + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS))
            elif sx >= sw or sx + 1 < 0 or sy >= sh or sy + 1 < 0:
                dst[y, x, :] = 0                                   # BORDER_CONSTANT, borderValue = 0
            else:
                for k in range(cn):
                    def at(yy, xx):
                        return int(img[yy, xx, k]) if 0 <= xx < sw and 0 <= yy < sh else 0
                    t = at(sy, sx) * w[0] + at(sy, sx + 1) * w[1] + at(sy + 1, sx) * w[2] + at(sy + 1, sx + 1) * w[3]
                    dst[y, x, k] = max(0, min(255, (t + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS))
    return dst if src.ndim == 3 else dst[..., 0]
