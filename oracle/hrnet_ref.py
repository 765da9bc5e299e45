"""fp32 CPU restatement of the reference pose_hrnet forward (TEST ORACLE, not product).

Follows /root/reference/landmark_regression/lib/models/pose_hrnet.py:
  * stem + layer1 ................. :425-432, Bottleneck :78-98, _make_layer :374-391
  * transition layers ............. :333-372, use in forward :434-455
  * HighResolutionModule .......... branches :139-185, fuse layers :187-242, forward :247-265
  * stages ........................ _make_stage :393-423 (last stage-4 module fuses to branch 0 only)
  * final 1x1 (or 3x3) conv + bias  :323-329, :458
and, for MODEL.NAME hrnet_cms / hrnet_cms_384 (lib/models/hrnet_cms.py, hrnet_cms_384.py):
  * stage 4 keeps all four outputs  hrnet_cms.py:321-322 (multi_scale_output=True)
  * four heads ConvTranspose2d(C_b -> 32, k5 s4 p1 op1 | k3 s2 p1 op1) + Conv2d(32 -> J)   :353-419
  * top-down pyramid: x_b = head_b(y_b) + bilinear_x2(x_{b+1}), eval returns x_0             :551-562

The network is evaluated functionally from a plain ``state_dict`` whose keys and
shapes are those of the reference module (``state_dict_spec``), so checkpoints
interchange.  No nn.Module tree of the reference is reproduced here.

``emulate`` selects the arithmetic model:
  None   -- the reference's own arithmetic: fp32 conv, eval BatchNorm, ReLU, adds.
  'bf16' / 'f16' -- the storage model of the HIP path: BatchNorm folded into the
           conv weight/bias, folded weights rounded to the 16-bit type, every
           tensor that the HIP path writes to HBM rounded to it, fp32 accumulation.
           Used to separate "logic differs" from "precision differs" in the tests.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
EXPANSION = {"BASIC": 1, "BOTTLENECK": 4}


# ----------------------------------------------------------------------------- cfg
def extra_of(cfg):
    return cfg["MODEL"]["EXTRA"]


def w32_cfg(num_joints=11, image=256):
    return _wN_cfg(32, num_joints, image)


def w48_cfg(num_joints=11, image=384):
    return _wN_cfg(48, num_joints, image)


HEADS = {"pose_hrnet": None, "hrnet_cms": ("equal_to_image", 5, 4), "hrnet_cms_384": ("4x", 3, 2)}


def head_of(cfg):
    """None for pose_hrnet, else (key suffix, kernel, stride) of the four transposed-conv heads."""
    return HEADS[cfg["MODEL"].get("NAME", "pose_hrnet")]


def head_names(cfg):
    suffix = head_of(cfg)[0]
    return ["final_layer%s_%s" % ("" if b == 0 else str(b + 1), suffix) for b in range(4)]


def with_model(cfg, name):
    """Copy of cfg for another member of the family (pose_hrnet / hrnet_cms / hrnet_cms_384)."""
    out = {"MODEL": dict(cfg["MODEL"])}
    out["MODEL"]["NAME"] = name
    up = {None: 1}.get(head_of(out), None) or head_of(out)[2]
    img = out["MODEL"]["IMAGE_SIZE"][0]
    out["MODEL"]["HEATMAP_SIZE"] = [img // 4 * up, img // 4 * up]
    return out


def _wN_cfg(c, num_joints, image, modules=(1, 4, 3), block="BASIC", blocks=4):
    def stage(nb, nm):
        return {"NUM_MODULES": nm, "NUM_BRANCHES": nb, "BLOCK": block,
                "NUM_BLOCKS": [blocks] * nb, "NUM_CHANNELS": [c * (2 ** i) for i in range(nb)],
                "FUSE_METHOD": "SUM"}
    return {"MODEL": {"NAME": "pose_hrnet", "NUM_JOINTS": num_joints, "INIT_WEIGHTS": False,
                      "PRETRAINED": "", "IMAGE_SIZE": [image, image],
                      "HEATMAP_SIZE": [image // 4, image // 4],
                      "EXTRA": {"PRETRAINED_LAYERS": ["*"], "FINAL_CONV_KERNEL": 1,
                                "STAGE2": stage(2, modules[0]), "STAGE3": stage(3, modules[1]),
                                "STAGE4": stage(4, modules[2])}}}


def tiny_cfg(num_joints=11, image=64, c=16, modules=(1, 1, 1)):
    """Small HRNet of the same topology (for fast CPU tests)."""
    return _wN_cfg(c, num_joints, image, modules)


def bneck_cfg(num_joints=11, image=64, c=16, modules=(1, 1, 1), blocks=2):
    """STAGEk.BLOCK = BOTTLENECK (blocks_dict, pose_hrnet.py:266-269): branch b carries 4 * c * 2^b channels.  No shipped YAML
    uses it; the reference module builds and runs it (tests/golden/hrnet_bneck_reference_outputs.npz)."""
    return _wN_cfg(c, num_joints, image, modules, block="BOTTLENECK", blocks=blocks)


# ----------------------------------------------------------------------------- key/shape census
def _conv(sd, name, cout, cin, k, bias=False):
    sd[name + ".weight"] = (cout, cin, k, k)
    if bias:
        sd[name + ".bias"] = (cout,)


def _bn(sd, name, c):
    sd[name + ".weight"] = (c,)
    sd[name + ".bias"] = (c,)
    sd[name + ".running_mean"] = (c,)
    sd[name + ".running_var"] = (c,)
    sd[name + ".num_batches_tracked"] = ()


def _stage_channels(scfg):
    e = EXPANSION[scfg["BLOCK"]]
    return [c * e for c in scfg["NUM_CHANNELS"]]


def state_dict_spec(cfg):
    """name -> shape, in the reference module's registration order."""
    ex = extra_of(cfg)
    sd = OrderedDict()
    _conv(sd, "conv1", 64, 3, 3); _bn(sd, "bn1", 64)
    _conv(sd, "conv2", 64, 64, 3); _bn(sd, "bn2", 64)
    inpl = 64
    for b in range(4):                               # layer1: Bottleneck(planes 64) x4
        p = "layer1.%d" % b
        _conv(sd, p + ".conv1", 64, inpl, 1); _bn(sd, p + ".bn1", 64)
        _conv(sd, p + ".conv2", 64, 64, 3); _bn(sd, p + ".bn2", 64)
        _conv(sd, p + ".conv3", 256, 64, 1); _bn(sd, p + ".bn3", 256)
        if b == 0:
            _conv(sd, p + ".downsample.0", 256, inpl, 1); _bn(sd, p + ".downsample.1", 256)
        inpl = 256
    pre = [256]
    for si, sname in enumerate(("STAGE2", "STAGE3", "STAGE4")):
        scfg = ex[sname]
        cur = _stage_channels(scfg)
        tname = "transition%d" % (si + 1)
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    _conv(sd, "%s.%d.0" % (tname, i), cur[i], pre[i], 3); _bn(sd, "%s.%d.1" % (tname, i), cur[i])
            else:
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = cur[i] if j == i - len(pre) else cin
                    _conv(sd, "%s.%d.%d.0" % (tname, i, j), cout, cin, 3); _bn(sd, "%s.%d.%d.1" % (tname, i, j), cout)
        nb = scfg["NUM_BRANCHES"]
        last_stage = sname == "STAGE4"
        stname = "stage%d" % (si + 2)
        for m in range(scfg["NUM_MODULES"]):
            multi = head_of(cfg) is not None or not (last_stage and m == scfg["NUM_MODULES"] - 1)
            mp = "%s.%d" % (stname, m)
            for b in range(nb):
                for k in range(scfg["NUM_BLOCKS"][b]):
                    p = "%s.branches.%d.%d" % (mp, b, k)
                    if scfg["BLOCK"] == "BOTTLENECK":      # pose_hrnet.py:60-98, :142-154: in = out = 4 * planes, no downsample
                        pl = scfg["NUM_CHANNELS"][b]
                        _conv(sd, p + ".conv1", pl, cur[b], 1); _bn(sd, p + ".bn1", pl)
                        _conv(sd, p + ".conv2", pl, pl, 3); _bn(sd, p + ".bn2", pl)
                        _conv(sd, p + ".conv3", cur[b], pl, 1); _bn(sd, p + ".bn3", cur[b])
                        continue
                    _conv(sd, p + ".conv1", cur[b], cur[b], 3); _bn(sd, p + ".bn1", cur[b])
                    _conv(sd, p + ".conv2", cur[b], cur[b], 3); _bn(sd, p + ".bn2", cur[b])
            for i in range(nb if multi else 1):
                for j in range(nb):
                    fp = "%s.fuse_layers.%d.%d" % (mp, i, j)
                    if j > i:
                        _conv(sd, fp + ".0", cur[i], cur[j], 1); _bn(sd, fp + ".1", cur[i])
                    elif j < i:
                        for k in range(i - j):
                            cout = cur[i] if k == i - j - 1 else cur[j]
                            _conv(sd, "%s.%d.0" % (fp, k), cout, cur[j], 3); _bn(sd, "%s.%d.1" % (fp, k), cout)
        pre = cur
    fk = ex["FINAL_CONV_KERNEL"]
    if head_of(cfg) is None:
        _conv(sd, "final_layer", cfg["MODEL"]["NUM_JOINTS"], pre[0], fk, bias=True)
    else:
        _, k, _ = head_of(cfg)
        for b, name in enumerate(head_names(cfg)):
            sd[name + ".0.weight"] = (pre[b], 32, k, k)          # ConvTranspose2d: (in, out, k, k)
            sd[name + ".0.bias"] = (32,)
            _conv(sd, name + ".1", cfg["MODEL"]["NUM_JOINTS"], 32, fk, bias=True)
    return sd


def make_state_dict(cfg, seed=0):
    """Seeded synthetic checkpoint (SURVEY.md 8d recipe): kaiming-uniform(a=sqrt 5) convs,
    BN gamma~U[.75,1.25], beta,mean~N(0,.1^2), var~U[.75,1.25].  Deterministic for a torch build."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in state_dict_spec(cfg).items():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = torch.tensor(0, dtype=torch.long)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = (1.0 / fan_in) ** 0.5                    # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.startswith("final_layer") and leaf == "bias":
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        elif leaf in ("weight", "running_var"):
            sd[name] = 0.75 + 0.5 * torch.rand(shape, generator=g)
        else:                                                # bn bias / running_mean
            sd[name] = 0.1 * torch.randn(shape, generator=g)
    return sd


# ----------------------------------------------------------------------------- arithmetic models
class _Arith:
    """conv(+bn)(+residual)(+relu) under one of the two arithmetic models."""

    def __init__(self, sd, emulate=None):
        self.sd = sd
        self.emulate = emulate
        self.dt = {None: None, "bf16": torch.bfloat16, "f16": torch.float16}[emulate]

    def rnd(self, t):
        return t if self.dt is None else t.to(self.dt).to(torch.float32)

    def conv_bn(self, x, conv, bn, stride=1, relu=False, residual=None, bias=None, store=True, round_w=True):
        sd = self.sd
        w = sd[conv + ".weight"].float()
        pad = (w.shape[-1] - 1) // 2
        if self.dt is None:
            y = F.conv2d(x, w, sd.get(conv + ".bias") if bias is None else bias, stride, pad)
            if bn is not None:
                y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"],
                                 sd[bn + ".weight"], sd[bn + ".bias"], False, 0.0, BN_EPS)
            if residual is not None:
                y = y + residual
            return F.relu(y) if relu else y
        # HIP storage model: fold, round weights, fp32 accumulate, epilogue in fp32, round on store
        if bn is not None:
            s = (sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS))
            b = (sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * s).float()
            w = (w.double() * s.view(-1, 1, 1, 1)).float()
        else:
            b = sd[conv + ".bias"].float() if (conv + ".bias") in sd else None
        y = F.conv2d(x, self.rnd(w) if round_w else w, b, stride, pad)
        if residual is not None:
            y = y + residual
        if relu:
            y = F.relu(y)
        return self.rnd(y) if store else y


    def head(self, x, name, k, s):
        """ConvTranspose2d(k, stride s, padding 1, output_padding 1) + Conv2d (hrnet_cms.py:353-368)."""
        sd = self.sd
        wt, bt = sd[name + ".0.weight"].float(), sd[name + ".0.bias"].float()
        wc, bc = sd[name + ".1.weight"].float(), sd[name + ".1.bias"].float()
        if self.dt is None:
            y = F.conv_transpose2d(x, wt, bt, stride=s, padding=1, output_padding=1)
            return F.conv2d(y, wc, bc, 1, (wc.shape[-1] - 1) // 2)
        # HIP storage model (csrc/head.hip): the two linear layers are folded into one transposed convolution
        # C -> J; every kernel tap is a 1x1 convolution whose result ("tap map") is stored in the 16-bit type;
        # the taps that reach an output pixel are summed in fp32 together with the folded bias.
        if wc.shape[-1] != 1:
            raise NotImplementedError("folded heads need FINAL_CONV_KERNEL == 1")
        w = torch.einsum("cmyx,jm->cjyx", wt.double(), wc[:, :, 0, 0].double()).float()      # (C, J, k, k)
        b = (wc[:, :, 0, 0].double() @ bt.double() + bc.double()).float()
        n, _, h, wd = x.shape
        out = b.view(1, -1, 1, 1).repeat(n, 1, s * h, s * wd).contiguous()
        for ky in range(k):
            for kx in range(k):
                t = self.rnd(F.conv2d(x, self.rnd(w[:, :, ky, kx]).t().reshape(w.shape[1], w.shape[0], 1, 1)))
                # output row = s*iy - 1 + ky, for the input rows whose target lies inside the map
                iy0 = 1 if ky == 0 else 0
                ix0 = 1 if kx == 0 else 0
                iy1 = min(h, (s * h - ky) // s + 1)
                ix1 = min(wd, (s * wd - kx) // s + 1)
                out[:, :, s * iy0 - 1 + ky: s * (iy1 - 1) + ky: s, s * ix0 - 1 + kx: s * (ix1 - 1) + kx: s] += t[:, :, iy0:iy1, ix0:ix1]
        return out


def forward(sd, cfg, x, emulate=None, taps=None):
    """x: (N,3,H,W) float32, ImageNet-normalised.  Returns (N,J,H/4,W/4) float32 heatmaps
    (hrnet_cms: (N,J,H,W); hrnet_cms_384: (N,J,H/2,W/2)).
    ``taps``: optional dict that receives named intermediate tensors."""
    ex = extra_of(cfg)
    A = _Arith(sd, emulate)

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = x.float()
    # the HIP stem (csrc/stem_fused.hip) runs conv1 on MFMA like every other layer: the normalised input and the folded
    # weights are rounded to the 16-bit operand type, the sum is fp32, the stored result is rounded
    x = tap("stem1", A.conv_bn(A.rnd(x), "conv1", "bn1", 2, relu=True))
    x = tap("stem2", A.conv_bn(x, "conv2", "bn2", 2, relu=True))
    for b in range(4):
        p = "layer1.%d" % b
        # storage model: the HIP plan runs the first Bottleneck's downsample 1x1 conv and its conv3 as ONE K-concatenated
        # convolution (csrc/hrnet.cpp: conv_cat), so that residual is summed in fp32 and never rounded to 16 bits
        res = A.conv_bn(x, p + ".downsample.0", p + ".downsample.1", store=False) if b == 0 else x
        y = A.conv_bn(x, p + ".conv1", p + ".bn1", relu=True)
        y = A.conv_bn(y, p + ".conv2", p + ".bn2", relu=True)
        x = A.conv_bn(y, p + ".conv3", p + ".bn3", relu=True, residual=res)
    tap("layer1", x)

    ylist = [x]
    pre = [256]
    for si, sname in enumerate(("STAGE2", "STAGE3", "STAGE4")):
        scfg = ex[sname]
        cur = _stage_channels(scfg)
        tname = "transition%d" % (si + 1)
        xs = []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    src = ylist[-1]                            # reference :437,:445,:453 -- always the LAST branch
                    xs.append(A.conv_bn(src, "%s.%d.0" % (tname, i), "%s.%d.1" % (tname, i), relu=True))
                else:
                    xs.append(ylist[i])
            else:
                t = ylist[-1]                                  # reference :445,:453 -- from the LAST branch
                for j in range(i + 1 - len(pre)):
                    t = A.conv_bn(t, "%s.%d.%d.0" % (tname, i, j), "%s.%d.%d.1" % (tname, i, j), 2, relu=True)
                xs.append(t)
        nb = scfg["NUM_BRANCHES"]
        stname = "stage%d" % (si + 2)
        for m in range(scfg["NUM_MODULES"]):
            multi = head_of(cfg) is not None or not (sname == "STAGE4" and m == scfg["NUM_MODULES"] - 1)
            mp = "%s.%d" % (stname, m)
            for b in range(nb):
                t = xs[b]
                for k in range(scfg["NUM_BLOCKS"][b]):
                    p = "%s.branches.%d.%d" % (mp, b, k)
                    if scfg["BLOCK"] == "BOTTLENECK":      # pose_hrnet.py:78-98 with identity residual
                        u = A.conv_bn(t, p + ".conv1", p + ".bn1", relu=True)
                        u = A.conv_bn(u, p + ".conv2", p + ".bn2", relu=True)
                        t = A.conv_bn(u, p + ".conv3", p + ".bn3", relu=True, residual=t)
                        continue
                    u = A.conv_bn(t, p + ".conv1", p + ".bn1", relu=True)
                    t = A.conv_bn(u, p + ".conv2", p + ".bn2", relu=True, residual=t)
                xs[b] = t
            outs = []
            for i in range(nb if multi else 1):
                acc = None
                for j in range(nb):
                    fp = "%s.fuse_layers.%d.%d" % (mp, i, j)
                    if j == i:
                        term = xs[j]
                    elif j > i:
                        z = A.conv_bn(xs[j], fp + ".0", fp + ".1")
                        term = F.interpolate(z, scale_factor=2 ** (j - i), mode="nearest")
                    else:
                        term = xs[j]
                        for k in range(i - j):
                            last = k == i - j - 1
                            term = A.conv_bn(term, "%s.%d.0" % (fp, k), "%s.%d.1" % (fp, k), 2, relu=not last)
                    acc = term if acc is None else acc + term
                outs.append(A.rnd(F.relu(acc)))
            xs = outs
            tap("%s.out0" % mp, xs[0])
        ylist = xs
        pre = cur
    if head_of(cfg) is None:
        out = A.conv_bn(ylist[0], "final_layer", None, store=False)
        return tap("heatmaps", out)
    # hrnet_cms.py:551-557: coarse-to-fine sum of the four heads
    _, k, s = head_of(cfg)
    names = head_names(cfg)
    out = None
    for b in (3, 2, 1, 0):
        y = A.head(ylist[b], names[b], k, s)
        if out is not None:
            y = y + F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=False)
        out = tap("head%d" % b, y)
    return tap("heatmaps", out)
