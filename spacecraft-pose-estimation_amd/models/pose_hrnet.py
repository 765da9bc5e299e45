"""Host-side mirror of the reference pose_hrnet module API, executing on the HIP engine.

Interface kept from landmark_regression/lib/models/pose_hrnet.py:
  get_pose_net(cfg, is_train, **kwargs) -> nn.Module              (:495-501)
  module(x: float32 N x 3 x H x W, normalised) -> float32 N x J x H/4 x W/4   (:425-460)
  (hrnet_cms.py / hrnet_cms_384.py subclass this tree with their four transposed-conv heads)
  .state_dict() / .load_state_dict(sd, strict=False) with the reference's key names and shapes
  (conv weights OIHW, BatchNorm weight/bias/running_mean/running_var/num_batches_tracked).

The module tree below only HOLDS parameters (so checkpoints interchange and .cuda()/.eval()
behave); it contains no torch compute.  forward() hands the tensors to libscpose_hip.so
(csrc/hrnet.cpp), which folds BatchNorm, packs bf16/f16 MFMA weights and runs the hand-written
kernels.  There is no eager fallback: without the HIP library or a GPU, forward() raises.
Inference only (is_train=True is rejected: training is out of scope).
"""
import logging

import torch
import torch.nn as nn

from .. import ops

logger = logging.getLogger(__name__)
BN_MOMENTUM = 0.1


def _conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=bias)


def _cbn(cin, cout, k, stride=1, relu=False, momentum=BN_MOMENTUM):
    layers = [_conv(cin, cout, k, stride), nn.BatchNorm2d(cout, momentum=momentum)]
    if relu:
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class _Params(nn.Module):
    """A bag of named sub-modules that is never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: compute runs in the HIP engine")


def _basic_block(c):
    m = _Params()
    m.conv1 = _conv(c, c, 3); m.bn1 = nn.BatchNorm2d(c, momentum=BN_MOMENTUM)
    m.conv2 = _conv(c, c, 3); m.bn2 = nn.BatchNorm2d(c, momentum=BN_MOMENTUM)
    return m


def _bottleneck(cin, planes, downsample):
    m = _Params()
    m.conv1 = _conv(cin, planes, 1); m.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
    m.conv2 = _conv(planes, planes, 3); m.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
    m.conv3 = _conv(planes, planes * 4, 1); m.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM)
    if downsample:
        m.downsample = nn.Sequential(_conv(cin, planes * 4, 1), nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM))
    return m


def _hr_module(channels, num_blocks, multi_scale_output, bottleneck=False):
    nb = len(channels)
    m = _Params()
    # STAGEk.BLOCK = BOTTLENECK (blocks_dict, pose_hrnet.py:266-269): channels[b] = 4 * planes in and out, identity residual (:142-154)
    block = (lambda c: _bottleneck(c, c // 4, False)) if bottleneck else _basic_block
    m.branches = nn.ModuleList([nn.Sequential(*[block(channels[b]) for _ in range(num_blocks[b])]) for b in range(nb)])
    rows = []
    for i in range(nb if multi_scale_output else 1):
        row = []
        for j in range(nb):
            if j > i:      # 1x1 + BN (+ nearest upsample, parameter-free; index 2 kept for key parity)
                row.append(nn.Sequential(_conv(channels[j], channels[i], 1), nn.BatchNorm2d(channels[i]),
                                         nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
            elif j == i:
                row.append(None)
            else:          # (i-j) stride-2 3x3 convs, ReLU on all but the last (fuse-layer BNs use default momentum)
                chain = []
                for k in range(i - j):
                    last = k == i - j - 1
                    chain.append(_cbn(channels[j], channels[i] if last else channels[j], 3, 2, relu=not last, momentum=0.1))
                row.append(nn.Sequential(*chain))
        rows.append(nn.ModuleList(row))
    m.fuse_layers = nn.ModuleList(rows)
    return m


class PoseHighResolutionNet(nn.Module):
    MODEL_NAME = "pose_hrnet"     # what the engine is told (ops.HEAD_CODES); overridden by hrnet_cms / hrnet_cms_384
    HEAD = None                   # (state_dict key suffix, kernel, stride) of the transposed-conv heads, if any

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg["MODEL"]["EXTRA"]
        self._cfg_model = {"NAME": self.MODEL_NAME, "NUM_JOINTS": int(cfg["MODEL"]["NUM_JOINTS"]),
                           "EXTRA": {k: (dict(extra[k]) if k.startswith("STAGE") else extra[k]) for k in extra}}
        self.dtype_name = kwargs.get("dtype", "bf16")
        self.conv1 = _conv(3, 64, 3, 2); self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = _conv(64, 64, 3, 2); self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = nn.Sequential(*[_bottleneck(64 if b == 0 else 256, 64, b == 0) for b in range(4)])
        pre = [256]
        for si, name in enumerate(("STAGE2", "STAGE3", "STAGE4")):
            scfg = extra[name]
            if str(scfg["BLOCK"]) not in ("BASIC", "BOTTLENECK"):
                raise ValueError("%s.BLOCK=%s: stage blocks are BASIC or BOTTLENECK (blocks_dict, pose_hrnet.py:266-269)" % (name, scfg["BLOCK"]))
            bneck = str(scfg["BLOCK"]) == "BOTTLENECK"
            cur = [int(c) * (4 if bneck else 1) for c in scfg["NUM_CHANNELS"]]     # num_channels * block.expansion (:393-400)
            trans = []
            for i in range(len(cur)):
                if i < len(pre):
                    trans.append(_cbn(pre[i], cur[i], 3, 1, relu=True) if cur[i] != pre[i] else None)
                else:
                    chain = []
                    for j in range(i + 1 - len(pre)):
                        cout = cur[i] if j == i - len(pre) else pre[-1]
                        chain.append(_cbn(pre[-1], cout, 3, 2, relu=True))
                    trans.append(nn.Sequential(*chain))
            setattr(self, "transition%d" % (si + 1), nn.ModuleList(trans))
            nm = int(scfg["NUM_MODULES"])
            mods = [_hr_module(cur, [int(b) for b in scfg["NUM_BLOCKS"]],
                               self.HEAD is not None or not (name == "STAGE4" and m == nm - 1), bneck) for m in range(nm)]
            setattr(self, "stage%d" % (si + 2), nn.Sequential(*mods))
            pre = cur
        self._make_head(pre, int(cfg["MODEL"]["NUM_JOINTS"]), int(extra["FINAL_CONV_KERNEL"]))
        self.pretrained_layers = extra["PRETRAINED_LAYERS"] if "PRETRAINED_LAYERS" in extra else ["*"]
        self._engine = None
        self._engine_version = None
        self._param_tensors = None

    def _make_head(self, channels, num_joints, fk):
        if self.HEAD is None:
            self.final_layer = _conv(channels[0], num_joints, fk, 1, bias=True)
            return
        suffix, k, stride = self.HEAD        # hrnet_cms.py:353-419: one ConvTranspose2d + Conv2d pair per branch
        for b, c in enumerate(channels):
            name = "final_layer%s_%s" % ("" if b == 0 else str(b + 1), suffix)
            setattr(self, name, nn.Sequential(
                nn.ConvTranspose2d(c, 32, kernel_size=k, stride=stride, padding=1, output_padding=1),
                _conv(32, num_joints, fk, 1, bias=True)))

    # ---- engine lifetime: rebuilt whenever parameters may have changed ----
    # The engine holds folded, packed copies of the parameters, so every call has to know whether they changed since.  Walking
    # state_dict() for that (1 754 tensors: 5.8 ms per call, a cap of ~2 800 frames/s at the shipped BATCH_SIZE_PER_GPU of 16 -- VERDICT r5)
    # is replaced by (a) invalidation where torch replaces or rewrites parameters wholesale -- _apply (.cuda() / .to() / .half()) and
    # load_state_dict -- and (b) the sum of the in-place version counters of a CACHED tensor list (0.1 ms), which catches p.copy_() /
    # p.add_() / optimiser-style updates.  Re-binding a parameter object by hand (module.weight = nn.Parameter(...)) is the one case
    # neither sees: call invalidate_engine() after it.
    def invalidate_engine(self):
        self._param_tensors = None
        self._engine_version = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_engine()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_engine()
        return out

    def _param_version(self):
        ts = getattr(self, "_param_tensors", None)
        if ts is None:
            ts = self._param_tensors = list(self.state_dict().values())
            self._param_ptrs = sum(t.data_ptr() for t in ts)
        v = 0
        for t in ts:
            v += t._version
        return (len(ts), self._param_ptrs, v)

    def _get_engine(self, device):
        ver = self._param_version()
        if self._engine is None or self._engine_version != ver or self._engine.device != device:
            if self._engine is not None:
                self._engine.close()             # (closes the graphs captured from it)
            self._fast_graph = self._fast_seen = None      # forward_decode's captured forward belonged to the old engine
            cfg = {"MODEL": self._cfg_model}
            self._engine = ops.HrnetEngine(cfg, self.state_dict(), dtype=self.dtype_name, device=device)
            self._engine_version = ver
        return self._engine

    def forward(self, x):
        if self.training:
            raise RuntimeError("the MI355X pose_hrnet is inference-only: call .eval() (training is out of scope)")
        if not x.is_cuda:
            raise ops.nat.NativeError("pose_hrnet.forward needs a ROCm device tensor: there is no CPU fallback")
        return self._get_engine(x.device)(x)

    def forward_decode(self, x, center, scale, post_process=True):
        """Key points (N, J, 3) [x_img, y_img, maxval] straight from the network (scpose_hrnet_forward_decode): for pose_hrnet with a
        1x1 final layer the last fuse row, final_layer and get_final_preds (lib/core/inference.py:49-79) are one kernel and no
        heat-map is written; bit-identical to get_final_preds_device(cfg, self(x), center, scale).  Batches of one shape replay a
        captured hipGraph of the forward (concurrent branches), bound to buffers owned by the module; a batch of another shape
        (the last, ragged one of a data set) runs the same launches eagerly."""
        if self.training:
            raise RuntimeError("the MI355X pose_hrnet is inference-only: call .eval() (training is out of scope)")
        if not x.is_cuda:
            raise ops.nat.NativeError("pose_hrnet.forward_decode needs a ROCm device tensor: there is no CPU fallback")
        eng = self._get_engine(x.device)
        x = x.contiguous() if x.dtype == torch.uint8 else x.float().contiguous()
        center = center.to(x.device, torch.float32).contiguous(); scale = scale.to(x.device, torch.float32).contiguous()
        key = (tuple(x.shape), x.dtype, bool(post_process), id(eng))
        g = getattr(self, "_fast_graph", None)
        if g is None or g[0] != key:
            seen = getattr(self, "_fast_seen", None)
            self._fast_seen = key
            if seen != key:          # first batch of this shape: eager (a shape seen once is not worth a capture)
                return eng.forward_decode(x, center, scale, post_process)
            if g is not None:
                g[1].close()
            bx, bc, bs = x.clone(), center.clone(), scale.clone()
            g = self._fast_graph = (key, eng.capture_decode(bx, bc, bs, post_process, concurrent=True), bx, bc, bs)
        _, graph, bx, bc, bs = g
        bx.copy_(x); bc.copy_(center); bs.copy_(scale)
        return graph.replay().clone()

    def init_weights(self, pretrained=""):
        raise RuntimeError("init_weights: training-time initialisation is out of scope; load a checkpoint instead")


def _get_pose_net(cls, cfg, is_train, **kwargs):
    if is_train:
        raise ValueError("get_pose_net(is_train=True): the MI355X build covers the inference path only")
    return cls(cfg, **kwargs)


def get_pose_net(cfg, is_train, **kwargs):
    return _get_pose_net(PoseHighResolutionNet, cfg, is_train, **kwargs)
