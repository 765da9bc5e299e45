"""Thin torch-facing wrappers over the C ABI (device memory, streams: plumbing only).

Every function here launches HIP kernels from libscpose_hip.so on the current torch
stream; tensors must live on a ROCm device.  Nothing falls back to eager PyTorch.
"""
import ctypes
from ctypes import c_int32, c_int64, c_void_p, c_char_p, c_size_t, c_double

import numpy as np
import torch

from . import _native as nat

_TORCH_DT = {nat.DT_BF16: torch.bfloat16, nat.DT_F16: torch.float16}
_DT_OF = {"bf16": nat.DT_BF16, "f16": nat.DT_F16, "fp16": nat.DT_F16, "bfloat16": nat.DT_BF16,
          "float16": nat.DT_F16}


def dtype_code(dtype):
    if isinstance(dtype, int):
        return dtype
    return _DT_OF[str(dtype).replace("torch.", "")]


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise nat.NativeError("scpose ops need device tensors (got %s); there is no CPU path" % t.device)


# ------------------------------------------------------------------ layout
def to_blocked(x, dtype="bf16"):
    """float32 NCHW (C % 8 == 0) -> blocked [N][C/8][H][W][8] 16-bit tensor."""
    _need_cuda(x)
    dt = dtype_code(dtype)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    out = torch.empty((n, c // 8, h, w, 8), dtype=_TORCH_DT[dt], device=x.device)
    nat.check(nat.lib().scpose_nchw_f32_to_blocked(_ptr(x), n, c, h, w, dt, _ptr(out), _stream()), "nchw_f32_to_blocked")
    return out


def from_blocked(xb):
    """blocked 16-bit tensor -> float32 NCHW."""
    _need_cuda(xb)
    n, cg, h, w, _ = xb.shape
    dt = nat.DT_BF16 if xb.dtype == torch.bfloat16 else nat.DT_F16
    out = torch.empty((n, cg * 8, h, w), dtype=torch.float32, device=xb.device)
    nat.check(nat.lib().scpose_blocked_to_nchw_f32(_ptr(xb), n, cg * 8, h, w, dt, _ptr(out), _stream()), "blocked_to_nchw_f32")
    return out


# ------------------------------------------------------------------ single conv layer
class Conv:
    """One (BN-folded) convolution on the MFMA kernel: y = [relu](conv(x) + b [+ res])."""

    def __init__(self, weight, bias=None, stride=1, dtype="bf16"):
        w = weight.detach().float().cpu().contiguous()
        self.cout, self.cin, self.ks, _ = w.shape
        self.stride = stride
        self.dt = dtype_code(dtype)
        b = bias.detach().float().cpu().contiguous() if bias is not None else None
        h = c_void_p()
        nat.check(nat.lib().scpose_conv_create(c_void_p(w.data_ptr()), c_void_p(b.data_ptr()) if b is not None else c_void_p(0),
                                               self.cout, self.cin, self.ks, stride, self.dt, ctypes.byref(h)), "conv_create")
        self._h = h

    def __call__(self, xb, residual=None, relu=False, out_nchw_f32=False):
        _need_cuda(xb, residual)
        n, cg, h, w, _ = xb.shape
        assert cg * 8 == self.cin, (cg * 8, self.cin)
        ho, wo = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        if out_nchw_f32:
            out = torch.empty((n, self.cout, ho, wo), dtype=torch.float32, device=xb.device)
        else:
            out = torch.empty((n, self.cout // 8, ho, wo, 8), dtype=xb.dtype, device=xb.device)
        nat.check(nat.lib().scpose_conv_forward(self._h, _ptr(xb), n, h, w, _ptr(residual), int(relu), int(out_nchw_f32),
                                                _ptr(out), _stream()), "conv_forward")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().scpose_conv_destroy(self._h)
                self._h = None
        except Exception:
            pass


def fuse_sum(terms, shifts, out_hw):
    """relu(sum_t nearest_upsample(term_t, 2**shift_t)); terms blocked, out at (H, W)."""
    _need_cuda(*terms)
    h, w = out_hw
    n, cg = terms[0].shape[0], terms[0].shape[1]
    dt = nat.DT_BF16 if terms[0].dtype == torch.bfloat16 else nat.DT_F16
    out = torch.empty((n, cg, h, w, 8), dtype=terms[0].dtype, device=terms[0].device)
    tp = (c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    sh = (c_int32 * len(terms))(*shifts)
    nat.check(nat.lib().scpose_fuse_sum(tp, sh, len(terms), n, cg * 8, h, w, dt, _ptr(out), _stream()), "fuse_sum")
    return out


# ------------------------------------------------------------------ decode
def decode(heatmaps, center, scale, post_process=True):
    """heatmaps (N,J,H,W) f32 device; center/scale (N,2) f32 device -> (N,J,3) [x_img,y_img,maxval]."""
    _need_cuda(heatmaps, center, scale)
    hm = heatmaps.contiguous().float()
    n, j, h, w = hm.shape
    c = center.contiguous().float()
    s = scale.contiguous().float()
    out = torch.empty((n, j, 3), dtype=torch.float32, device=hm.device)
    nat.check(nat.lib().scpose_decode(_ptr(hm), n, j, h, w, _ptr(c), _ptr(s), int(bool(post_process)), _ptr(out), _stream()), "decode")
    return out


def max_preds(heatmaps):
    _need_cuda(heatmaps)
    hm = heatmaps.contiguous().float()
    n, j, h, w = hm.shape
    coords = torch.empty((n, j, 2), dtype=torch.float32, device=hm.device)
    maxvals = torch.empty((n, j, 1), dtype=torch.float32, device=hm.device)
    nat.check(nat.lib().scpose_max_preds(_ptr(hm), n, j, h, w, _ptr(coords), _ptr(maxvals), _stream()), "max_preds")
    return coords, maxvals


def crop_warp(frames, trans, out_wh, swap_rb=False, device=None, roi=None, frame_hw=None):
    """Batched cv2.warpAffine(frame_i, trans_i, (W, H), INTER_LINEAR) on the device.
    frames: list of HxWx3 uint8 arrays/tensors (any sizes); trans: (N,2,3) forward affines as returned by
    get_affine_transform (frame -> crop).  Returns uint8 (N, H, W, 3) device crops.
    roi (N,4) [x0, y0, w, h] + frame_hw (N,2): frames[i] is only that WINDOW of a frame of size frame_hw[i] (warp_window below
    computes one that contains every tap); same crops bit for bit (scpose_crop_warp_roi)."""
    import numpy as np
    dev = device or torch.device("cuda", torch.cuda.current_device())
    packed = isinstance(frames, dict)      # pack_frames(): one flat uint8 tensor + offsets + sizes (what the data-loader workers send)
    if not packed:
        frames = pack_frames(frames)
    n = int(frames["hw"].shape[0])
    w, h = int(out_wh[0]), int(out_wh[1])
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    if n == 0:
        return out
    offs = frames["offsets"].tolist(); hw = [tuple(v) for v in frames["hw"].tolist()]
    buf = frames["flat"].to(dev, non_blocking=True)
    from .utils.transforms import invert_affine_cv
    minv = np.zeros((n, 6), dtype=np.float64)
    for i in range(n):
        minv[i] = invert_affine_cv(trans[i])            # the inverse map exactly as cv::warpAffine derives it
    offs_d = torch.tensor(offs, dtype=torch.int64, device=dev)
    minv_d = torch.from_numpy(minv).to(dev)
    if roi is not None:
        roi = np.asarray(roi, dtype=np.int32).reshape(n, 4); fhw = np.asarray(frame_hw, dtype=np.int32).reshape(n, 2)
        for i in range(n):
            if (int(roi[i, 3]), int(roi[i, 2])) != hw[i]:
                raise ValueError("crop_warp: window %d is %s but roi says %dx%d" % (i, hw[i], roi[i, 3], roi[i, 2]))
        fhw_d = torch.from_numpy(fhw).to(dev); roi_d = torch.from_numpy(roi).to(dev)      # (named: both must outlive the launch call)
        nat.check(nat.lib().scpose_crop_warp_roi(_ptr(buf), _ptr(offs_d), _ptr(fhw_d), _ptr(roi_d),
                                                 _ptr(minv_d), n, h, w, int(bool(swap_rb)), _ptr(out), _stream()), "crop_warp_roi")
        return out
    hw_d = torch.tensor(hw, dtype=torch.int32, device=dev)
    nat.check(nat.lib().scpose_crop_warp(_ptr(buf), _ptr(offs_d), _ptr(hw_d), _ptr(minv_d), n, h, w, int(bool(swap_rb)),
                                         _ptr(out), _stream()), "crop_warp")
    return out


def pack_frames(frames):
    """list of HxWx3 uint8 arrays / tensors (any sizes) -> {"flat": all bytes back to back, "offsets": int64 (n,), "hw": int32 (n, 2)}:
    the form scpose_crop_warp[_roi] takes.  The data-loader workers call it (dataset.collate_device_crop), so a batch crosses the
    process boundary as one tensor instead of one per frame."""
    import numpy as np
    flat, offs, hw, o = [], [], [], 0
    for f in frames:
        t = f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("crop_warp: frames must be HxWx3 uint8")
        flat.append(t.reshape(-1)); offs.append(o); hw.append((t.shape[0], t.shape[1])); o += t.numel()
    return {"flat": torch.cat(flat) if flat else torch.zeros(0, dtype=torch.uint8), "offsets": torch.tensor(offs, dtype=torch.int64),
            "hw": torch.tensor(hw, dtype=torch.int32).reshape(-1, 2)}


def warp_window(trans, out_wh, frame_hw):
    """[x0, y0, w, h]: a window of the frame that contains every pixel crop_warp reads for this affine (host side, NumPy).  The
    corners of the output grid go through the inverse map exactly as the kernel derives it; the fixed-point coordinates deviate
    from the real ones by < 1 px and a bilinear tap reads (x, x + 1): two pixels of margin, three on the far side; clipped to the frame
    (w or h may be 0: nothing of the frame is read)."""
    import numpy as np
    from .utils.transforms import invert_affine_cv
    m = np.asarray(invert_affine_cv(trans), dtype=np.float64).reshape(2, 3)
    w, h = int(out_wh[0]), int(out_wh[1])
    cx = np.array([0.0, w - 1.0, 0.0, w - 1.0]); cy = np.array([0.0, 0.0, h - 1.0, h - 1.0])
    sx = m[0, 0] * cx + m[0, 1] * cy + m[0, 2]; sy = m[1, 0] * cx + m[1, 1] * cy + m[1, 2]
    fh, fw = int(frame_hw[0]), int(frame_hw[1])
    x0 = min(max(int(np.floor(sx.min())) - 2, 0), fw); x1 = min(max(int(np.ceil(sx.max())) + 3, 0), fw)
    y0 = min(max(int(np.floor(sy.min())) - 2, 0), fh); y1 = min(max(int(np.ceil(sy.max())) + 3, 0), fh)
    return [x0, y0, max(x1 - x0, 0), max(y1 - y0, 0)]


def basic_block(conv1, conv2, xb):
    """Fused BasicBlock relu(conv2(relu(conv1(x))) + x) on a blocked tensor (two Conv objects, 3x3 stride 1, C -> C)."""
    _need_cuda(xb)
    n, planes, h, w, _ = xb.shape
    out = torch.empty_like(xb)
    nat.check(nat.lib().scpose_basic_block_forward(conv1._h, conv2._h, _ptr(xb), n, h, w, _ptr(out), _stream()),
              "basic_block_forward")
    return out


def flip_merge(out, out_flipped, flip_pairs, shift):
    """(out + flip_back(out_flipped)) * 0.5 of the flip test (lib/core/function.py:347-366), on the device.
    out / out_flipped: (N,J,H,W) f32 device heatmaps of the frame and of its x-flipped copy."""
    _need_cuda(out, out_flipped)
    a, b = out.contiguous().float(), out_flipped.contiguous().float()
    n, j, h, w = a.shape
    perm = list(range(j))
    for p0, p1 in flip_pairs:
        perm[p0], perm[p1] = p1, p0
    perm_d = torch.tensor(perm, dtype=torch.int32, device=a.device)
    res = torch.empty_like(a)
    nat.check(nat.lib().scpose_flip_merge(_ptr(a), _ptr(b), _ptr(perm_d), n, j, h, w, int(bool(shift)), _ptr(res),
                                          _stream()), "flip_merge")
    return res


def heatmap_accumulate(acc, x, div=1.0):
    """In place acc = (acc + x) / div on the device: the ensemble sum / mean of validate_cv
    (lib/core/function.py:530-536).  acc, x: float32 device tensors of the same shape."""
    _need_cuda(acc, x)
    if acc.shape != x.shape or acc.dtype != torch.float32 or x.dtype != torch.float32:
        raise nat.NativeError("heatmap_accumulate: float32 tensors of one shape expected")
    if not acc.is_contiguous():
        raise nat.NativeError("heatmap_accumulate: acc must be contiguous (it is updated in place)")
    nat.check(nat.lib().scpose_heatmap_accumulate(_ptr(acc), _ptr(x.contiguous()), float(div), acc.numel(), _stream()),
              "heatmap_accumulate")
    return acc


# ------------------------------------------------------------------ PnP
def pnp_epnp_ransac(kp_xyc, landmarks, K, dist, conf_thr0=0.95, min_pts=15, thr_decay=0.8, thr_iters=100,
                    max_iters=10000, reproj_err=15.0, confidence=0.99, want_rvec=False, rows=None):
    """kp_xyc (N,J,3) f32 device; landmarks (J,3), K (3,3), dist (5,) f64 device.
    Returns rot (N,3,3) f64, tvec (N,3) f64, status (N,) i32 [, rvec (N,3)].
    rows: a contiguous (N,13) f64 device buffer -> the kernel writes [R (9), t (3), status] per frame straight into it
    (scpose_pnp_epnp_ransac_rows) and `rows` is returned: the block a rank all-gathers and copies to the host."""
    _need_cuda(kp_xyc, landmarks, K, dist)
    kp = kp_xyc.contiguous().float()
    n, j, _ = kp.shape
    dev = kp.device
    lm = landmarks.contiguous().double()
    if lm.dim() != 2 or tuple(lm.shape) != (j, 3):
        raise nat.NativeError("pnp_epnp_ransac: landmarks must be (%d, 3) to match the keypoints, got %s" % (j, tuple(lm.shape)))
    Kd = K.contiguous().double()
    if Kd.numel() != 9:
        raise nat.NativeError("pnp_epnp_ransac: K must be a 3x3 camera matrix, got %s" % (tuple(K.shape),))
    # distortion vector: the kernel reads exactly (k1, k2, p1, p2, k3).  cv2 also accepts 4 coefficients (k3 = 0);
    # the 8/12/14-coefficient rational / thin-prism / tilted models are not implemented and must not be truncated.
    dd = dist.contiguous().double().reshape(-1) if dist is not None else torch.zeros(5, dtype=torch.float64, device=dev)
    if dd.numel() == 4:
        dd = torch.cat([dd, torch.zeros(1, dtype=torch.float64, device=dd.device)])
    elif dd.numel() != 5:
        raise nat.NativeError("pnp_epnp_ransac: %d distortion coefficients; supported are 4 or 5 (k1,k2,p1,p2[,k3])" % dd.numel())
    if rows is not None:
        if want_rvec:     # the gather block has no rvec column: refuse instead of silently returning one tensor
            raise nat.NativeError("pnp_epnp_ransac: rows=... excludes want_rvec (the (N, 13) block is [R, t, status]); call without rows for rvec")
        if tuple(rows.shape) != (n, 13) or rows.dtype != torch.float64 or not rows.is_contiguous() or rows.device != dev:
            raise nat.NativeError("pnp_epnp_ransac: rows must be a contiguous float64 (%d, 13) tensor on %s" % (n, dev))
        nat.check(nat.lib().scpose_pnp_epnp_ransac_rows(_ptr(kp), _ptr(lm), _ptr(Kd), _ptr(dd), n, j, conf_thr0, min_pts, thr_decay,
                                                        thr_iters, max_iters, reproj_err, confidence, _ptr(rows), _stream()),
                  "pnp_epnp_ransac_rows")
        return rows
    rot = torch.empty((n, 3, 3), dtype=torch.float64, device=dev)
    tv = torch.empty((n, 3), dtype=torch.float64, device=dev)
    rv = torch.empty((n, 3), dtype=torch.float64, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    nat.check(nat.lib().scpose_pnp_epnp_ransac(_ptr(kp), _ptr(lm), _ptr(Kd), _ptr(dd), n, j, conf_thr0, min_pts, thr_decay,
                                               thr_iters, max_iters, reproj_err, confidence, _ptr(rot), _ptr(tv), _ptr(rv),
                                               _ptr(st), _stream()), "pnp_epnp_ransac")
    return (rot, tv, st, rv) if want_rvec else (rot, tv, st)


# ------------------------------------------------------------------ HRNet engine
HEAD_CODES = {"pose_hrnet": 0, "hrnet_cms": 1, "hrnet_cms_384": 2}     # include/scpose.h SCPOSE_HEAD_*


def desc_from_cfg(cfg, dtype="bf16", mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """cfg: yacs-like node or plain dict with MODEL.NUM_JOINTS, MODEL.EXTRA.STAGE{2,3,4} and (optionally)
    MODEL.NAME in pose_hrnet / hrnet_cms / hrnet_cms_384 (the modules of landmark_regression/lib/models/)."""
    model = cfg["MODEL"]
    extra = model["EXTRA"]
    d = nat.HrnetDesc()
    name = str(model["NAME"]) if "NAME" in model else "pose_hrnet"
    if name not in HEAD_CODES:
        raise nat.NativeError("MODEL.NAME=%s: supported models are %s" % (name, ", ".join(HEAD_CODES)))
    d.head = HEAD_CODES[name]
    d.num_joints = int(model["NUM_JOINTS"])
    d.final_conv_kernel = int(extra["FINAL_CONV_KERNEL"])
    d.num_stages = 3
    for si, name in enumerate(("STAGE2", "STAGE3", "STAGE4")):
        s = extra[name]
        if str(s["BLOCK"]) not in ("BASIC", "BOTTLENECK"):
            raise nat.NativeError("%s.BLOCK=%s: stage blocks are BASIC or BOTTLENECK" % (name, s["BLOCK"]))
        d.block[si] = 1 if str(s["BLOCK"]) == "BOTTLENECK" else 0
        d.num_modules[si] = int(s["NUM_MODULES"])
        d.num_branches[si] = int(s["NUM_BRANCHES"])
        for b in range(int(s["NUM_BRANCHES"])):
            d.num_blocks[si][b] = int(s["NUM_BLOCKS"][b])
            d.num_channels[si][b] = int(s["NUM_CHANNELS"][b])
    d.dtype = dtype_code(dtype)
    for i in range(3):
        d.mean[i] = mean[i]
        d.std[i] = std[i]
    return d


class HrnetEngine:
    """Owns a native scpose_hrnet handle built from a state_dict (host f32 copies are
    only needed during create) and a torch-allocated workspace arena."""

    def __init__(self, cfg, state_dict, dtype="bf16", allow_missing=False, device=None):
        if not torch.cuda.is_available():
            raise nat.NativeError("no ROCm device visible: the HRNet engine only runs on the GPU")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.desc = desc_from_cfg(cfg, dtype)
        self.num_joints = self.desc.num_joints
        names, ptrs, numels, keep = [], [], [], []
        for k, v in state_dict.items():
            if not torch.is_floating_point(v):
                continue
            t = v.detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            names.append(k.encode())
            ptrs.append(t.data_ptr())
            numels.append(t.numel())
        n = len(names)
        c_names = (c_char_p * n)(*names)
        c_ptrs = (c_void_p * n)(*ptrs)
        c_numels = (c_int64 * n)(*numels)
        h = c_void_p()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().scpose_hrnet_create(ctypes.byref(self.desc), c_names, c_ptrs, c_numels, n,
                                                    int(allow_missing), ctypes.byref(h)), "hrnet_create")
        self._h = h
        self._ws = None

    def workspace_bytes(self, n, h, w):
        b = c_size_t()
        nat.check(nat.lib().scpose_hrnet_workspace_bytes(self._h, n, h, w, ctypes.byref(b)), "hrnet_workspace_bytes")
        return b.value

    def heatmap_size(self, h, w):
        oh = c_int32(); ow = c_int32()
        nat.check(nat.lib().scpose_hrnet_heatmap_size(self._h, h, w, ctypes.byref(oh), ctypes.byref(ow)), "hrnet_heatmap_size")
        return oh.value, ow.value

    def stats(self, h, w):
        l = c_int32(); f = c_double(); by = c_double()
        nat.check(nat.lib().scpose_hrnet_stats(self._h, h, w, ctypes.byref(l), ctypes.byref(f), ctypes.byref(by)), "hrnet_stats")
        return {"launches": l.value, "flops_per_frame": f.value, "act_bytes_per_frame": by.value}

    def forward(self, x, out=None, profile=False):
        """x: float32 (N,3,H,W) normalised, or uint8 (N,H,W,3) raw RGB.  Returns f32 (N,J,H/4,W/4).
        profile=True records a HIP event around every launch (read with profile_read())."""
        _need_cuda(x)
        x = x.contiguous()
        if x.dtype == torch.uint8:
            n, h, w, c = x.shape
            fmt = nat.IN_U8_NHWC
        else:
            x = x.float()
            n, c, h, w = x.shape
            fmt = nat.IN_F32_NCHW
        if c != 3:
            raise nat.NativeError("HRNet input must have 3 channels, got %d" % c)
        need = self.workspace_bytes(n, h, w)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        oh, ow = self.heatmap_size(h, w)
        if out is None:
            out = torch.empty((n, self.num_joints, oh, ow), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (n, self.num_joints, oh, ow) or out.dtype != torch.float32 or not out.is_contiguous():
            raise nat.NativeError("hrnet_forward: out must be contiguous float32 %s" % ((n, self.num_joints, oh, ow),))
        fn = nat.lib().scpose_hrnet_forward_profiled if profile else nat.lib().scpose_hrnet_forward
        nat.check(fn(self._h, _ptr(x), fmt, n, h, w, _ptr(out), _ptr(self._ws), self._ws.numel(), _stream()), "hrnet_forward")
        self._last_hw = (h, w)
        return out

    def tail_fused(self, n, h, w):
        """True when a forward of this shape runs the last fuse sum + final_layer (+ decode) as one kernel (head_fused.hip)."""
        k = c_int32()
        nat.check(nat.lib().scpose_hrnet_tail_fused(self._h, n, h, w, ctypes.byref(k)), "hrnet_tail_fused")
        return bool(k.value)

    def forward_decode(self, x, center, scale, post_process=True, heatmaps=False, profile=False):
        """Key points straight from the forward (scpose_hrnet_forward_decode): (N, J, 3) [x_img, y_img, maxval], bit-identical
        to decode(forward(x), center, scale, post_process).  For pose_hrnet with a 1x1 final layer no heat-map is written
        unless heatmaps=True (then (preds, heatmaps) is returned); other heads always go through a heat-map buffer.
        profile=True records a HIP event around every launch (read with profile_read())."""
        _need_cuda(x, center, scale)
        x = x.contiguous()
        if x.dtype == torch.uint8:
            n, h, w, c = x.shape
            fmt = nat.IN_U8_NHWC
        else:
            x = x.float()
            n, c, h, w = x.shape
            fmt = nat.IN_F32_NCHW
        if c != 3:
            raise nat.NativeError("HRNet input must have 3 channels, got %d" % c)
        need = self.workspace_bytes(n, h, w)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        oh, ow = self.heatmap_size(h, w)
        hm = None
        if heatmaps or not self.tail_fused(n, h, w):
            hm = torch.empty((n, self.num_joints, oh, ow), dtype=torch.float32, device=x.device)
        cc = center.contiguous().float(); ss = scale.contiguous().float()
        preds = torch.empty((n, self.num_joints, 3), dtype=torch.float32, device=x.device)
        fn = nat.lib().scpose_hrnet_forward_decode_profiled if profile else nat.lib().scpose_hrnet_forward_decode
        nat.check(fn(self._h, _ptr(x), fmt, n, h, w, _ptr(cc), _ptr(ss), int(bool(post_process)),
                     _ptr(preds), _ptr(hm) if hm is not None else None, _ptr(self._ws),
                     self._ws.numel(), _stream()), "hrnet_forward_decode")
        self._last_hw = (h, w)
        return (preds, hm) if heatmaps else preds

    def capture_decode(self, x, center, scale, post_process=True, concurrent=True, heatmaps=False):
        """capture() for forward_decode: x, center and scale are bound by address (refill them in place); .replay() returns
        the bound (N, J, 3) key-point buffer; .heatmaps is the bound heat-map buffer when one exists."""
        _need_cuda(x, center, scale)
        for t, name in ((x, "input"), (center, "center"), (scale, "scale")):
            if not t.is_contiguous():
                raise nat.NativeError("capture_decode: the %s buffer must be contiguous (it is bound by address)" % name)
        if center.dtype != torch.float32 or scale.dtype != torch.float32:
            raise nat.NativeError("capture_decode: center / scale must be float32")
        if x.dtype == torch.uint8:
            n, h, w, c = x.shape
            fmt = nat.IN_U8_NHWC
        elif x.dtype == torch.float32:
            n, c, h, w = x.shape
            fmt = nat.IN_F32_NCHW
        else:
            raise nat.NativeError("capture_decode: input must be uint8 NHWC or float32 NCHW")
        oh, ow = self.heatmap_size(h, w)
        hm = None
        if heatmaps or not self.tail_fused(n, h, w):
            hm = torch.empty((n, self.num_joints, oh, ow), dtype=torch.float32, device=x.device)
        preds = torch.empty((n, self.num_joints, 3), dtype=torch.float32, device=x.device)
        return HrnetGraph(self, x, fmt, (n, h, w), preds, concurrent, decode=(center, scale, bool(post_process), hm))

    def capture(self, x, out=None, concurrent=True):
        """Record the forward of the input BUFFER x (uint8 NHWC or float32 NCHW, fixed shape) into a hipGraph and return
        an HrnetGraph: refill x in place (x.copy_(...)), call .replay(), read .out.  concurrent=True (1) records the
        independent ops (module branches, fuse rows, transition convs) on parallel graph branches; concurrent=2 only the
        fuse rows and transition convs (large batches, whose branch kernels each fill the chip); 0 / False none."""
        _need_cuda(x)
        if not x.is_contiguous():
            raise nat.NativeError("capture: the input buffer must be contiguous (it is bound by address)")
        if x.dtype == torch.uint8:
            n, h, w, c = x.shape
            fmt = nat.IN_U8_NHWC
        elif x.dtype == torch.float32:
            n, c, h, w = x.shape
            fmt = nat.IN_F32_NCHW
        else:
            raise nat.NativeError("capture: input must be uint8 NHWC or float32 NCHW")
        if c != 3:
            raise nat.NativeError("HRNet input must have 3 channels, got %d" % c)
        oh, ow = self.heatmap_size(h, w)
        if out is None:
            out = torch.empty((n, self.num_joints, oh, ow), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (n, self.num_joints, oh, ow) or out.dtype != torch.float32 or not out.is_contiguous():
            raise nat.NativeError("capture: out must be contiguous float32 %s" % ((n, self.num_joints, oh, ow),))
        return HrnetGraph(self, x, fmt, (n, h, w), out, concurrent)

    def tap_names(self):
        buf = ctypes.create_string_buffer(4096)
        nat.check(nat.lib().scpose_hrnet_tap_names(self._h, buf, 4096), "hrnet_tap_names")
        return buf.value.decode().split(",")

    def forward_tap(self, x, tap):
        """Intermediate tensor `tap` ("stem1", "stem2", "layer1", "stage3.1.out0", ...) of the forward of x as float32
        (N, C, h, w): the forward is run up to the op that produces it (unit-level parity against the oracle's taps)."""
        _need_cuda(x)
        x = x.contiguous()
        if x.dtype == torch.uint8:
            n, h, w, _ = x.shape
            fmt = nat.IN_U8_NHWC
        else:
            x = x.float()
            n, _, h, w = x.shape
            fmt = nat.IN_F32_NCHW
        need = self.workspace_bytes(n, h, w)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        c = c_int32(); oh = c_int32(); ow = c_int32()
        fn = nat.lib().scpose_hrnet_forward_tap
        nat.check(fn(self._h, None, fmt, n, h, w, tap.encode(), None, ctypes.byref(c), ctypes.byref(oh), ctypes.byref(ow), None, 0, None),
                  "hrnet_forward_tap")
        out = torch.empty((n, c.value, oh.value, ow.value), dtype=torch.float32, device=x.device)
        nat.check(fn(self._h, _ptr(x), fmt, n, h, w, tap.encode(), _ptr(out), None, None, None, _ptr(self._ws), self._ws.numel(), _stream()),
                  "hrnet_forward_tap")
        return out

    def profile_read(self):
        """Per-launch records of the last profiled forward: list of dicts
        {ms, flops_per_frame, bytes_per_frame, kind, a, cin, cout}."""
        cnt = c_int32()
        nat.check(nat.lib().scpose_hrnet_profile_read(self._h, 0, 0, 0, None, None, None, None, ctypes.byref(cnt)), "profile_read")
        k = cnt.value
        ms = np.zeros(k, dtype=np.float32); fl = np.zeros(k); by = np.zeros(k); sig = np.zeros((k, 4), dtype=np.int32)
        h, w = self._last_hw
        nat.check(nat.lib().scpose_hrnet_profile_read(self._h, h, w, k, c_void_p(ms.ctypes.data), c_void_p(fl.ctypes.data),
                                                      c_void_p(by.ctypes.data), c_void_p(sig.ctypes.data), ctypes.byref(cnt)),
                  "profile_read")
        return [{"ms": float(ms[i]), "flops_per_frame": float(fl[i]), "bytes_per_frame": float(by[i]),
                 "kind": int(sig[i, 0]), "a": int(sig[i, 1]), "cin": int(sig[i, 2]), "cout": int(sig[i, 3])} for i in range(k)]

    __call__ = forward

    @staticmethod
    def kernel_classes(recs):
        """Class name of every record of profile_read(): 'kind:a:cin:cout', with '@<output pixels per frame>' appended where one
        forward runs the same layer shape at two map sizes (the 48->48 stride-2 convs of the 96x96 -> 12x12 fuse chain), so
        that per-class statistics never mix launches with different work."""
        base = ["%d:%d:%d:%d" % (r["kind"], r["a"], r["cin"], r["cout"]) for r in recs]
        work = {}
        for b, r in zip(base, recs):
            work.setdefault(b, set()).add(r["flops_per_frame"])      # same layer shape, different map size (bytes alone also differ with a residual)
        out = []
        for b, r in zip(base, recs):
            if len(work[b]) > 1 and r["kind"] == 1:
                k = r["a"] // 10
                out.append("%s@%d" % (b, round(r["flops_per_frame"] / (2.0 * r["cin"] * r["cout"] * k * k))))
            else:
                out.append(b)
        return out

    def close(self):
        # captured graphs bake in this engine's packed-weight pointers: destroy them first, so that no graph can replay
        # kernels that read freed weights (their replay() raises afterwards)
        for ref in list(getattr(self, "_graphs", ())):
            g = ref()
            if g is not None:
                g.close()
        self._graphs = []
        if getattr(self, "_h", None):
            nat.lib().scpose_hrnet_destroy(self._h)
            self._h = None
        self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HrnetGraph:
    """A captured forward (scpose_hrnet_graph_*): fixed input / output / workspace buffers, one launch per replay."""

    def __init__(self, engine, x, fmt, nhw, out, concurrent, decode=None):
        n, h, w = nhw
        b = c_size_t()
        nat.check(nat.lib().scpose_hrnet_graph_workspace_bytes(engine._h, n, h, w, ctypes.byref(b)), "hrnet_graph_workspace_bytes")
        self.engine, self.x, self.out = engine, x, out
        self._ws = torch.empty(b.value, dtype=torch.uint8, device=x.device)       # owned by the graph: addresses are baked in
        torch.cuda.synchronize(x.device)     # create runs one eager forward on an internal stream: every pending write to x / out,
                                             # on any stream of x's device (not only the current one), must have landed
        g = c_void_p()
        self.heatmaps = None
        with torch.cuda.device(x.device):
            if decode is None:
                nat.check(nat.lib().scpose_hrnet_graph_create(engine._h, _ptr(x), fmt, n, h, w, _ptr(out), _ptr(self._ws), self._ws.numel(),
                                                              int(concurrent), ctypes.byref(g)), "hrnet_graph_create")
            else:   # out = the key-point buffer; the decode's inputs are bound like x
                self.center, self.scale, post, self.heatmaps = decode
                nat.check(nat.lib().scpose_hrnet_graph_create_decode(
                    engine._h, _ptr(x), fmt, n, h, w, _ptr(self.center), _ptr(self.scale), int(post), _ptr(out),
                    _ptr(self.heatmaps) if self.heatmaps is not None else None, _ptr(self._ws), self._ws.numel(), int(concurrent),
                    ctypes.byref(g)), "hrnet_graph_create_decode")
        self._g = g
        k = c_int32()
        nat.check(nat.lib().scpose_hrnet_graph_nodes(self._g, ctypes.byref(k)), "hrnet_graph_nodes")
        self.nodes = k.value
        import weakref
        if not hasattr(engine, "_graphs"):
            engine._graphs = []
        engine._graphs.append(weakref.ref(self))     # HrnetEngine.close() destroys its live graphs before its weights

    def replay(self):
        """Enqueue the captured forward on the current stream; returns the (bound) heat-map buffer."""
        if not getattr(self, "_g", None):
            raise RuntimeError("HrnetGraph.replay(): the graph (or its engine) has been closed")
        nat.check(nat.lib().scpose_hrnet_graph_launch(self._g, _stream()), "hrnet_graph_launch")
        return self.out

    def close(self):
        if getattr(self, "_g", None):
            nat.lib().scpose_hrnet_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
