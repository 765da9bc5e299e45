"""Seeded synthetic workloads for the benchmark and smoke run: model
configs, a random-init checkpoint of the pose_hrnet architecture, and PnP keypoints made by
projecting the 11 Tango landmarks through random poses.  Product-side data generation: no
file under oracle/ is imported here.
"""
from collections import OrderedDict

import numpy as np
import torch

# Reference fixtures (data): object_detection/speed_plus_utils/landmarks.csv:2-12 and
# calibration.json:1-24 of mohsij/spacecraft-pose-estimation.
TANGO_LANDMARKS = np.array([
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
SPEEDPLUS_K = np.array([[2988.5795163815555, 0, 960], [0, 2988.3401159176124, 600], [0, 0, 1]], dtype=np.float64)
SPEEDPLUS_DIST = np.array([-0.22383016606510672, 0.51409797089106379, -0.00066499611998340662,
                           -0.00021404771667484594, -0.13124227429077406], dtype=np.float64)


def hrnet_cfg(width=48, num_joints=11, image=384, modules=(1, 4, 3)):
    """Plain-dict cfg in the shape of the reference YAMLs (experiments/events/events-config.yaml
    MODEL subtree): NUM_CHANNELS (w, 2w, 4w, 8w), 4 BASIC blocks per branch, NUM_MODULES 1/4/3."""
    def stage(nb, nm):
        return {"NUM_MODULES": nm, "NUM_BRANCHES": nb, "BLOCK": "BASIC", "NUM_BLOCKS": [4] * nb,
                "NUM_CHANNELS": [width * (2 ** i) for i in range(nb)], "FUSE_METHOD": "SUM"}
    return {"MODEL": {"NAME": "pose_hrnet", "NUM_JOINTS": num_joints, "INIT_WEIGHTS": False, "PRETRAINED": "",
                      "IMAGE_SIZE": [image, image], "HEATMAP_SIZE": [image // 4, image // 4],
                      "EXTRA": {"PRETRAINED_LAYERS": ["*"], "FINAL_CONV_KERNEL": 1, "STAGE2": stage(2, modules[0]),
                                "STAGE3": stage(3, modules[1]), "STAGE4": stage(4, modules[2])}}}


def random_checkpoint(cfg, seed=0):
    """Random-init state_dict of the architecture: Conv2d default init (kaiming-uniform, a=sqrt 5 ==
    U(+-1/sqrt(fan_in))), BN gamma,var ~ U[.75,1.25], beta,mean ~ N(0,.1^2) so activations stay O(1)
    through ~300 layers.  Keys/shapes come from the module tree (== the reference's state_dict)."""
    from importlib import import_module
    net_cls = import_module(".models." + str(cfg["MODEL"].get("NAME", "pose_hrnet")), __package__).PoseHighResolutionNet
    with torch.device("meta"):
        spec = net_cls(cfg).state_dict()
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, meta in spec.items():
        shape = tuple(meta.shape)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = torch.tensor(0, dtype=torch.long)
        elif len(shape) == 4:
            bound = (1.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.startswith("final_layer") and leaf == "bias":
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        elif leaf in ("weight", "running_var"):
            sd[name] = 0.75 + 0.5 * torch.rand(shape, generator=g)
        else:
            sd[name] = 0.1 * torch.randn(shape, generator=g)
    return sd


def project(R, t, X, K=SPEEDPLUS_K, dist=SPEEDPLUS_DIST):
    """Pinhole + (k1,k2,p1,p2,k3) projection, the model of the reference's project()
    (pose_estimation/export_predicted_poses_real.py:104-121)."""
    pc = X @ R.T + t
    x0, y0 = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    r2 = x0 * x0 + y0 * y0
    cd = 1 + dist[0] * r2 + dist[1] * r2 * r2 + dist[4] * r2 * r2 * r2
    x1 = x0 * cd + dist[2] * 2 * x0 * y0 + dist[3] * (r2 + 2 * x0 * x0)
    y1 = y0 * cd + dist[2] * (r2 + 2 * y0 * y0) + dist[3] * 2 * x0 * y0
    return np.stack([K[0, 0] * x1 + K[0, 2], K[1, 1] * y1 + K[1, 2]], 1)


def random_rotation(rng):
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-np.pi, np.pi)
    kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * kx @ kx


def keypoints(n, rng, noise_px=1.0, outlier_frac=0.0, landmarks=TANGO_LANDMARKS, K=SPEEDPLUS_K, dist=SPEEDPLUS_DIST,
              width=1920, height=1200):
    """(n, J, 3) float32 [u, v, conf=1] + ground-truth (R, t): |t| in [3,10] m, target in frame,
    N(0, noise) px jitter, a fraction of landmarks replaced by uniform image points."""
    j = len(landmarks)
    kp = np.zeros((n, j, 3), dtype=np.float32)
    rs = np.zeros((n, 3, 3)); ts = np.zeros((n, 3))
    for i in range(n):
        while True:
            r = random_rotation(rng)
            z = rng.uniform(3.0, 10.0)
            t = np.array([rng.uniform(-0.25, 0.25) * z, rng.uniform(-0.15, 0.15) * z, z])
            uv = project(r, t, landmarks, K, dist)
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
        rs[i], ts[i] = r, t
    return kp, rs, ts


def rgb_crops(n, image, generator):
    """uint8 (n, H, W, 3) uniform noise crops (SURVEY.md section 8d 'RGB crops')."""
    return torch.randint(0, 256, (n, image, image, 3), generator=generator, dtype=torch.uint8)


def event_frames(n, image, generator, p_one=0.02, p_two=0.01):
    """uint8 (n, H, W, 3) synthetic v2e event frames (BASELINE.json configs[4]).

    The reference renders accumulated event counts c (all polarities folded to +1, v2e/e2v.py:128-130) as
    (c + full_scale) / (2 * full_scale) with full_scale = 2 (v2e/v2ecore/renderer.py:247-249), scales by 255, truncates
    to uint8 and replicates the gray value to three channels (cv2.COLOR_GRAY2BGR, renderer.py:343): a frame holds only
    the values 127 (no event), 191 (one event) and 255 (two or more).  Here: Bernoulli pixels, p_one at 191, p_two at 255."""
    u = torch.rand((n, image, image), generator=generator)
    gray = torch.full((n, image, image), 127, dtype=torch.uint8)
    gray[u < p_one + p_two] = 191
    gray[u < p_two] = 255
    return gray.unsqueeze(-1).expand(n, image, image, 3).contiguous()


def mixed_batch(n, image, generator):
    """First half RGB noise crops, second half event frames: the mixed-modality batch of configs[4]."""
    n_rgb = n - n // 2
    return torch.cat([rgb_crops(n_rgb, image, generator), event_frames(n // 2, image, generator)], 0)


# ------------------------------------------------------------------------------------------------------------------
# Frames whose CONTENT determines the key points (tests/golden/fit_chain_checkpoint.py fits a small HRNet to them, so that the
# chain image -> heat-maps -> key points -> pose can be compared end to end on peaked maps).
# ------------------------------------------------------------------------------------------------------------------
# one colour per landmark (RGB in [0, 1]): the network tells the landmarks apart by colour, the blob gives the position
LANDMARK_COLOURS = np.array([
    [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 1.0],
    [1.0, 0.5, 0.0], [0.5, 0.0, 1.0], [0.0, 1.0, 0.5], [1.0, 1.0, 1.0], [1.0, 0.0, 0.5]], dtype=np.float64)   # distinct channel RATIOS
CHAIN_HEATMAP_SIGMA = 1.5     # heat-map pixels: sigma of the gaussian targets the chain checkpoint was fitted to


def landmark_frames(n, rng, image=128, blob_sigma=2.5, landmarks=TANGO_LANDMARKS, K=SPEEDPLUS_K, dist=SPEEDPLUS_DIST,
                    width=1920, height=1200):
    """n synthetic crops of a 1920 x 1200 frame showing the projected landmarks as coloured gaussian blobs.

    A random pose (|t| in [3, 10] m, target in frame) is projected with the SPEED+ camera; the crop box is the landmarks'
    bounding box * 1.3 (centre c, scale s = side / 200 as in lib/dataset/events.py's _box2cs convention, square), the crop
    coordinate of a frame point is (x - c) * image / side + image / 2 (lib/utils/transforms.py:57-95 with rot = 0) and its
    heat-map coordinate a quarter of that.  Every landmark is DRAWN at the position whose heat-map coordinate has the fractional
    part .25 or .75 nearest to the projected one (a shift of at most a quarter heat-map pixel = one crop pixel): for a peak
    there, get_final_preds' arg-max + quarter-pixel rule (lib/core/inference.py:49-79) returns exactly the drawn position, with
    margin on both decisions, so two 16-bit pipelines that differ by rounding noise decode identical key points.  Poses whose
    landmarks come closer than two heat-map pixels are redrawn.

    Returns dict: crops uint8 (n, image, image, 3); center, scale float32 (n, 2); kp float64 (n, J, 2) the drawn positions in
    FRAME pixels (what a perfect network + decode returns); hm float64 (n, J, 2) heat-map coordinates; R (n, 3, 3), t (n, 3)."""
    j = len(landmarks)
    hs = image // 4
    crops = np.zeros((n, image, image, 3), dtype=np.uint8)
    center = np.zeros((n, 2), dtype=np.float32); scale = np.zeros((n, 2), dtype=np.float32)
    kp = np.zeros((n, j, 2)); hm = np.zeros((n, j, 2)); rs = np.zeros((n, 3, 3)); ts = np.zeros((n, 3))
    yy, xx = np.mgrid[0:image, 0:image].astype(np.float64)
    for i in range(n):
        while True:
            r = random_rotation(rng)
            z = rng.uniform(3.0, 10.0)
            t = np.array([rng.uniform(-0.25, 0.25) * z, rng.uniform(-0.15, 0.15) * z, z])
            uv = project(r, t, landmarks, K, dist)
            if not ((uv[:, 0] > 0).all() and (uv[:, 0] < width).all() and (uv[:, 1] > 0).all() and (uv[:, 1] < height).all()):
                continue
            lo, hi = uv.min(0), uv.max(0)
            c32 = ((lo + hi) / 2).astype(np.float32)
            s32 = np.float32(max(1.3 * (hi - lo).max(), 48.0) / 200.0)
            side = float(s32) * 200.0
            h = ((uv - c32.astype(np.float64)) * (image / side) + image / 2) / 4.0
            fl = np.floor(h)
            h = fl + np.where(h - fl < 0.5, 0.25, 0.75)
            if h.min() < 2.0 or h.max() > hs - 3.0:
                continue
            d = np.linalg.norm(h[:, None, :] - h[None, :, :], axis=2) + 1e9 * np.eye(j)
            if d.min() < 2.0:
                continue
            break
        img = rng.uniform(0.0, 40.0, (image, image, 3))
        for k in range(j):
            g = np.exp(-((xx - 4.0 * h[k, 0]) ** 2 + (yy - 4.0 * h[k, 1]) ** 2) / (2.0 * blob_sigma ** 2))
            img += 200.0 * g[:, :, None] * LANDMARK_COLOURS[k % len(LANDMARK_COLOURS)]
        crops[i] = np.clip(img, 0.0, 255.0).astype(np.uint8)
        center[i] = c32; scale[i] = s32
        hm[i] = h
        kp[i] = (4.0 * h - image / 2) * (side / image) + c32.astype(np.float64)
        rs[i], ts[i] = r, t
    return {"crops": crops, "center": center, "scale": scale, "kp": kp, "hm": hm, "R": rs, "t": ts}


def gaussian_targets(hm_xy, size, sigma=CHAIN_HEATMAP_SIGMA):
    """(n, J, size, size) float32 unit-peak gaussians at the (unquantised) heat-map coordinates -- generate_target of
    lib/dataset/JointsDataset.py:264-332 without its rounding of the centre to whole pixels (the fitted network has to reproduce
    the sub-pixel phase for the quarter-pixel rule to have something to read)."""
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    d2 = (xx[None, None] - hm_xy[:, :, 0, None, None]) ** 2 + (yy[None, None] - hm_xy[:, :, 1, None, None]) ** 2
    return np.exp(-d2 / (2.0 * sigma ** 2)).astype(np.float32)


def chain_cfg(image=128):
    """The small HRNet the chain checkpoint (tests/golden/chain_checkpoint.npz) was fitted for: 16 / 32 / 64 / 128 channels,
    one module per stage, two BASIC blocks per branch -- every layer class of pose_hrnet, 1.6 M parameters."""
    cfg = hrnet_cfg(16, 11, image, modules=(1, 1, 1))
    for st in ("STAGE2", "STAGE3", "STAGE4"):
        cfg["MODEL"]["EXTRA"][st]["NUM_BLOCKS"] = [2] * cfg["MODEL"]["EXTRA"][st]["NUM_BRANCHES"]
    return cfg


def load_chain_checkpoint(path):
    """state_dict (float32 tensors, exactly the float16-representable values stored in the fixture) of chain_cfg()."""
    z = np.load(path)
    sd = OrderedDict()
    for k in z.files:
        if not k.startswith("sd/"):
            continue
        a = z[k]
        sd[k[3:]] = torch.from_numpy(a.astype(np.int64)) if a.dtype.kind in "iu" else torch.from_numpy(a.astype(np.float32))
    return sd


# ---------------------------------------------------------------------------------------------------------------------------
# A CONSTRUCTED HRNet-W48 384 x 384 checkpoint whose heat-maps are peaked (the headline geometry of BASELINE.json: fitting all
# 63.6 M parameters of that network on host cores is out of reach, so the weights are written down instead of trained).
# ---------------------------------------------------------------------------------------------------------------------------
W48_CHAIN_BLOB_SIGMA = 7.5      # crop pixels at 384 x 384 (1.9 heat-map pixels), the blobs landmark_frames draws for this checkpoint
W48_CHAIN_EPS = 0.02


def w48_chain_cfg(image=384):
    return hrnet_cfg(48, 11, image)


def w48_chain_checkpoint(seed=0, eps=W48_CHAIN_EPS):
    """state_dict of w48_chain_cfg(): random_checkpoint(seed) with the branch-0 path turned into "detect the landmark colours,
    then carry the eleven maps to final_layer", every residual / cross-branch path kept RANDOM but scaled by `eps`:

      * conv1 + bn1 + ReLU (3 -> 64, stride 2): 18 units P = relu(v_c - a_l), N = relu(a_l - v_c) of the [1 2 1] x [1 2 1] / 16 blurred
        colour v (c = r, g, b; a_l = the three levels a landmark colour's component takes at a blob centre), the other 46 random * eps;
      * conv2 + bn2 + ReLU (64 -> 64, stride 2): units 0..10 = relu(1 - sum_c |v_c - a_jc| / r_j) = relu(1 - sum_c (P + N) / r_j), an L1
        bump around landmark j's colour (r_j below the distance to the nearest other colour), again through the blur kernel; the
        rest random * eps.  A map peaks where its blob is brightest, and falls monotonically with the blob's gaussian;
      * layer1.0.downsample, transition1.0: identity on the leading channels (+ eps * random); the last BatchNorm of every Bottleneck
        and of every BasicBlock of branch 0, and of every fuse up-path into branch 0: gamma, beta * eps -- so a block is
        relu(x + eps * f(x)) with f the random-init function of random_checkpoint; branches 1-3 stay O(1) random;
      * final_layer: identity on channels 0..10 (+ eps * random), zero bias.

    So every kernel of the forward computes on real data, the eleven heat-maps are narrow peaks (~0.9) on a near-zero floor with
    margins far above 16-bit rounding noise, and the key points of lib/core/inference.get_final_preds are the drawn landmark
    positions.  Deterministic in `seed` (the fixture tests/golden/chain_w48_reference.npz stores the seed, not 127 MB of weights)."""
    cfg = w48_chain_cfg()
    sd = random_checkpoint(cfg, seed)
    mean = np.array([0.485, 0.456, 0.406]); std = np.array([0.229, 0.224, 0.225])
    blur = torch.tensor([[1.0, 2.0, 1.0], [2.0, 4.0, 2.0], [1.0, 2.0, 1.0]]) / 16.0
    att = W48_CHAIN_BLOB_SIGMA ** 2 / (W48_CHAIN_BLOB_SIGMA ** 2 + 0.5 + 2.0)      # peak attenuation by the two blurs (variances 0.5, 2 crop px^2)
    levels = (20.0 + att * 200.0 * np.array([0.0, 0.5, 1.0])) / 255.0            # landmark_frames: background U(0, 40) + 200 * colour

    def identity_bn(prefix, n, beta=None):
        sd[prefix + ".weight"][:n] = 1.0
        sd[prefix + ".bias"][:n] = 0.0 if beta is None else torch.as_tensor(beta, dtype=torch.float32)
        sd[prefix + ".running_mean"][:n] = 0.0
        sd[prefix + ".running_var"][:n] = 1.0 - 1e-5

    def scale_bn(prefix, s):
        sd[prefix + ".weight"] *= s
        sd[prefix + ".bias"] *= s

    # ---- conv1: colour features of the blurred, de-normalised image ----
    w = sd["conv1.weight"]; w *= eps
    beta = np.zeros(18)
    for c in range(3):
        for l in range(3):
            k = 6 * c + 2 * l
            w[k] = 0.0; w[k + 1] = 0.0
            w[k, c] = float(std[c]) * blur; beta[k] = mean[c] - levels[l]              # P = relu(v_c - a_l)
            w[k + 1, c] = -float(std[c]) * blur; beta[k + 1] = levels[l] - mean[c]     # N = relu(a_l - v_c)
    scale_bn("bn1", eps)
    identity_bn("bn1", 18, beta)
    # ---- conv2: one L1 bump per landmark colour ----
    cols = LANDMARK_COLOURS[:11]
    a = (20.0 + att * 200.0 * cols) / 255.0
    dist = np.abs(a[:, None, :] - a[None, :, :]).sum(2) + 1e9 * np.eye(11)
    r = np.minimum(0.9 * dist.min(1), 0.35 * (att * 200.0 / 255.0) * cols.sum(1))
    w = sd["conv2.weight"]; w *= eps
    for j in range(11):
        w[j] = 0.0
        for c in range(3):
            l = int(round(cols[j, c] * 2))
            k = 6 * c + 2 * l
            w[j, k] = -blur / float(r[j]); w[j, k + 1] = -blur / float(r[j])
    scale_bn("bn2", eps)
    identity_bn("bn2", 11, np.ones(11))
    # ---- carry channels 0..10 along branch 0 ----
    w = sd["layer1.0.downsample.0.weight"]; w[:64] *= eps
    for k in range(64):
        w[k, k, 0, 0] += 1.0
    identity_bn("layer1.0.downsample.1", 64)
    for b in range(4):
        scale_bn("layer1.%d.bn3" % b, eps)
    w = sd["transition1.0.0.weight"]; w *= eps
    for k in range(48):
        w[k, k, 1, 1] += 1.0
    identity_bn("transition1.0.1", 48)
    for name in list(sd):
        p = name.split(".")
        if p[0] in ("stage2", "stage3", "stage4") and p[2] == "branches" and p[3] == "0" and p[5] == "bn2" and p[6] == "weight":
            scale_bn(name[:-len(".weight")], eps)                     # BasicBlocks of branch 0
        if p[0] in ("stage2", "stage3", "stage4") and p[2] == "fuse_layers" and p[3] == "0" and p[5] == "1" and p[6] == "weight":
            scale_bn(name[:-len(".weight")], eps)                     # up paths into branch 0 (fuse_layers.0.j = [conv1x1, bn, upsample])
    w = sd["final_layer.weight"]; w *= eps
    for j in range(11):
        w[j, j, 0, 0] += 1.0
    sd["final_layer.bias"].zero_()
    return sd


def landmark_scene(n, rng, image=384, blob_sigma=W48_CHAIN_BLOB_SIGMA, landmarks=TANGO_LANDMARKS, K=SPEEDPLUS_K, dist=SPEEDPLUS_DIST,
                   width=1920, height=1200):
    """landmark_frames at the FILE boundary: n whole 1920 x 1200 frames showing the projected landmarks as coloured blobs, with the COCO
    bounding box a detector would hand to `EventsDataset` -- so that the crop JointsDataset.py:134-198 cuts (centre / scale by
    events.py:94-113's _xywh2cs, float32) is the crop landmark_frames draws directly: every landmark sits at the frame position whose
    heat-map coordinate has the fractional part .25 / .75 nearest to its projection, with sigma = blob_sigma crop pixels after the warp.
    Returns dict: frames [n] uint8 (height, width, 3) RGB; bbox (n, 4) [x, y, w, h]; kp (n, J, 2) drawn positions in frame pixels; R, t."""
    j = len(landmarks)
    hs = image // 4
    frames, bbox = [], np.zeros((n, 4))
    kp = np.zeros((n, j, 2)); rs = np.zeros((n, 3, 3)); ts = np.zeros((n, 3))
    for i in range(n):
        while True:
            r = random_rotation(rng)
            z = rng.uniform(3.0, 10.0)
            t = np.array([rng.uniform(-0.25, 0.25) * z, rng.uniform(-0.15, 0.15) * z, z])
            uv = project(r, t, landmarks, K, dist)
            if not ((uv[:, 0] > 0).all() and (uv[:, 0] < width).all() and (uv[:, 1] > 0).all() and (uv[:, 1] < height).all()):
                continue
            lo, hi = uv.min(0), uv.max(0)
            w = max(1.3 * (hi - lo).max(), 48.0) / 1.5                # the detector's box: EventsDataset enlarges it 1.5 x
            x, y = (lo[0] + hi[0]) / 2 - w / 2, (lo[1] + hi[1]) / 2 - w / 2
            c32 = np.array([x + 0.5 * w, y + 0.5 * w], dtype=np.float32)                 # _xywh2cs, events.py:94-113
            s32 = (np.array([w, w], dtype=np.float64) / 200).astype(np.float32) * 1.5
            side = float(s32[0]) * 200.0
            h = ((uv - c32.astype(np.float64)) * (image / side) + image / 2) / 4.0
            fl = np.floor(h)
            h = fl + np.where(h - fl < 0.5, 0.25, 0.75)
            if h.min() < 2.0 or h.max() > hs - 3.0:
                continue
            d = np.linalg.norm(h[:, None, :] - h[None, :, :], axis=2) + 1e9 * np.eye(j)
            if d.min() < 4.0:          # blobs of sigma 1.9 heat-map pixels must not merge
                continue
            break
        pos = (4.0 * h - image / 2) * (side / image) + c32.astype(np.float64)
        sig = blob_sigma * side / image
        img = 20.0 + rng.uniform(-4.0, 4.0, (height, width, 1)) * np.ones((1, 1, 3))
        for k in range(j):
            x0, x1 = int(max(pos[k, 0] - 4 * sig, 0)), int(min(pos[k, 0] + 4 * sig + 1, width))
            y0, y1 = int(max(pos[k, 1] - 4 * sig, 0)), int(min(pos[k, 1] + 4 * sig + 1, height))
            yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float64)
            g = np.exp(-((xx - pos[k, 0]) ** 2 + (yy - pos[k, 1]) ** 2) / (2.0 * sig ** 2))
            img[y0:y1, x0:x1] += 200.0 * g[:, :, None] * LANDMARK_COLOURS[k % len(LANDMARK_COLOURS)]
        frames.append(np.clip(img, 0.0, 255.0).astype(np.uint8))
        bbox[i] = (x, y, w, w); kp[i] = pos; rs[i], ts[i] = r, t
    return {"frames": frames, "bbox": bbox, "kp": kp, "R": rs, "t": ts}
