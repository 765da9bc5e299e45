"""core.function.validate against the REFERENCE's own evaluation loop.

tests/golden/validate_reference_outputs.npz was produced by running landmark_regression/lib/core/function.py:318-459 `validate()` ITSELF in
the build container (tests/golden/make_validate_golden.py: the reference's module, loss, accuracy, get_final_preds, all_preds / all_boxes
assembly and log lines; fp32 on CPU) on the 64 decisive frames of the fitted chain checkpoint, in batches of 12.  Here the same loader
content goes through this repository's validate() on the HIP path, twice: with metric logging (heat-map path: loss / PCK as the
reference logs them) and without (the fused forward -> key points path the CLI takes by default).

  all_boxes   bit-identical (centre, scale, prod(scale * 200), score: host arithmetic)
  all_preds   every key point within 0.5 px of the reference's (measured 0.000), maxvals within 0.05 (16-bit pipeline vs fp32)
  image_path, pred_file_name, returned perf indicator   identical
  loss / accuracy averages and the log lines (heat-map path)   accuracy identical, loss within 5 %, same line format"""
import logging
import os
import re
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def _loader(syn, batch, as_u8):
    z = np.load(os.path.join(GOLD, "chain_checkpoint.npz"))
    image, n_cand, seed, _ = [int(v) for v in z["meta"]]
    cand = syn.landmark_frames(n_cand, np.random.default_rng(seed), image)
    fr = {k: v[z["test_index"]] for k, v in cand.items()}
    n = fr["crops"].shape[0]
    x = torch.from_numpy(fr["crops"]) if as_u8 else (torch.from_numpy(fr["crops"]).permute(0, 3, 1, 2).float() / 255.0 - MEAN) / STD
    target = torch.from_numpy(syn.gaussian_targets(fr["hm"], image // 4))
    weight = torch.ones(n, 11, 1)
    score = torch.linspace(0.5, 1.0, n, dtype=torch.float64)
    batches = []
    for i in range(0, n, batch):
        meta = {"center": torch.from_numpy(fr["center"][i:i + batch]), "scale": torch.from_numpy(fr["scale"][i:i + batch]),
                "score": score[i:i + batch], "image": ["frame_%03d.png" % k for k in range(i, min(i + batch, n))]}
        batches.append((x[i:i + batch], target[i:i + batch], weight[i:i + batch], meta))
    return image, n, batches


@pytest.mark.parametrize("mode", ["heatmaps_and_metrics", "fused_keypoints", "fused_keypoints_u8"])
def test_validate_equals_the_reference_validate(scpose, gpu_ops, mode):
    from importlib import import_module
    P = "spacecraft-pose-estimation_amd"
    syn = import_module(P + ".synthetic"); fn = import_module(P + ".core.function"); models = import_module(P + ".models")
    loss_mod = import_module(P + ".core.loss")
    g = np.load(os.path.join(GOLD, "validate_reference_outputs.npz"))
    image, n, batches = _loader(syn, int(g["batch"]), as_u8=mode.endswith("u8"))
    net = models.pose_hrnet.get_pose_net(syn.chain_cfg(image), is_train=False)
    net.load_state_dict(syn.load_chain_checkpoint(os.path.join(GOLD, "chain_checkpoint.npz")), strict=True)
    net = net.cuda().eval()
    N = types.SimpleNamespace
    config = N(MODEL=N(NUM_JOINTS=11, NAME="pose_hrnet", IMAGE_SIZE=[image, image], HEATMAP_SIZE=[image // 4, image // 4]),
               TEST=N(FLIP_TEST=False, SHIFT_HEATMAP=True, POST_PROCESS=True), PRINT_FREQ=2)
    got = {}

    class DS:
        flip_pairs = []

        def __len__(self):
            return n

        def evaluate(self, c, preds, output_dir, pred_file_name, all_boxes, image_path, filenames, imgnums):
            got.update(preds=preds.copy(), boxes=all_boxes.copy(), image_path=list(image_path), pred_file_name=pred_file_name)
            return {"Null": 0}, 0
    lines = []

    class H(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    h = H(); fn.logger.addHandler(h); fn.logger.setLevel(logging.INFO)
    try:
        metrics = mode == "heatmaps_and_metrics"
        perf = fn.validate(config, batches, DS(), net, loss_mod.JointsMSELoss(use_target_weight=True).cuda() if metrics else None,
                           "", "", pred_file_name="pred_test", log_metrics=metrics)
    finally:
        fn.logger.removeHandler(h)
    assert perf == int(g["perf"]) and got["pred_file_name"] == str(g["pred_file_name"]) and got["image_path"] == [str(v) for v in g["image_path"]]
    assert np.array_equal(got["boxes"], g["boxes"])
    err = np.linalg.norm(got["preds"][:, :, :2] - g["preds"][:, :, :2], axis=2)
    dv = np.abs(got["preds"][:, :, 2] - g["preds"][:, :, 2]).max()
    print("%s: max |key point - reference validate()| = %.3e px over %d joints, maxval diff %.3e" % (mode, err.max(), err.size, dv))
    assert got["preds"].dtype == np.float32 and err.max() <= 0.5 and dv <= 0.05
    lines = [re.sub(r"Time \S+ \(\S+\)", "Time T (T)", l) for l in lines if not l.startswith("validate: fused")]
    ref_lines = [str(v) for v in g["log"]]
    if metrics:
        assert len(lines) == len(ref_lines) and lines[-3:] == ref_lines[-3:]            # the markdown table of name_values
        num = re.compile(r"Test: \[(\d+)/(\d+)\]\tTime T \(T\)\tLoss (\S+) \((\S+)\)\tAccuracy (\S+) \((\S+)\)")
        for a, b in zip(lines[:-3], ref_lines[:-3]):
            ma, mb = num.fullmatch(a), num.fullmatch(b)
            assert ma and mb and ma.group(1, 2, 5, 6) == mb.group(1, 2, 5, 6), (a, b)       # batch index / count and both accuracies
            assert abs(float(ma.group(3)) - float(mb.group(3))) <= 1e-4 and abs(float(ma.group(4)) - float(mb.group(4))) <= 1e-4   # losses as printed (.4f)
    else:
        assert lines[-3:] == ref_lines[-3:]


def test_validate_cv_equals_the_reference_validate_cv(scpose, gpu_ops):
    """The ensemble loop (lib/core/function.py:500-592, tools/test_cv_ensemble.py) against the reference's own validate_cv() on three members
    (the chain checkpoint with its final layer scaled by 0.8 / 1.0 / 1.3): heat-maps summed in member order and divided once, same key
    points, boxes, log cadence (every fifth batch, whatever PRINT_FREQ says) and no name_values table."""
    from importlib import import_module
    P = "spacecraft-pose-estimation_amd"
    syn = import_module(P + ".synthetic"); fn = import_module(P + ".core.function"); models = import_module(P + ".models")
    loss_mod = import_module(P + ".core.loss")
    g = np.load(os.path.join(GOLD, "validate_reference_outputs.npz"))
    image, n, batches = _loader(syn, int(g["batch"]), as_u8=False)
    nets = []
    for f in g["cv_scales"]:
        m = models.pose_hrnet.get_pose_net(syn.chain_cfg(image), is_train=False)
        sd = syn.load_chain_checkpoint(os.path.join(GOLD, "chain_checkpoint.npz"))
        sd["final_layer.weight"] = sd["final_layer.weight"] * float(f); sd["final_layer.bias"] = sd["final_layer.bias"] * float(f)
        m.load_state_dict(sd, strict=True)
        nets.append(m.cuda().eval())
    N = types.SimpleNamespace
    config = N(MODEL=N(NUM_JOINTS=11, NAME="pose_hrnet", IMAGE_SIZE=[image, image], HEATMAP_SIZE=[image // 4, image // 4]),
               TEST=N(FLIP_TEST=False, SHIFT_HEATMAP=True, POST_PROCESS=True), PRINT_FREQ=2)
    got = {}

    class DS:
        flip_pairs = []

        def __len__(self):
            return n

        def evaluate(self, c, preds, output_dir, pred_file_name, all_boxes, image_path, filenames, imgnums):
            got.update(preds=preds.copy(), boxes=all_boxes.copy(), pred_file_name=pred_file_name)
            return {"Null": 0}, 0
    lines = []

    class H(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    h = H(); fn.logger.addHandler(h); fn.logger.setLevel(logging.INFO)
    try:
        perf = fn.validate_cv(config, batches, DS(), nets, loss_mod.JointsMSELoss(use_target_weight=True).cuda(), "", "", "pred_real")
    finally:
        fn.logger.removeHandler(h)
    assert perf == int(g["cv_perf"]) and got["pred_file_name"] == str(g["cv_pred_file_name"]) and np.array_equal(got["boxes"], g["cv_boxes"])
    err = np.linalg.norm(got["preds"][:, :, :2] - g["cv_preds"][:, :, :2], axis=2)
    dv = np.abs(got["preds"][:, :, 2] - g["cv_preds"][:, :, 2]).max()
    print("validate_cv: max |key point - reference validate_cv()| = %.3e px over %d joints, maxval diff %.3e" % (err.max(), err.size, dv))
    assert err.max() <= 0.5 and dv <= 0.05
    lines = [re.sub(r"Time \S+ \(\S+\)", "Time T (T)", l) for l in lines]
    ref_lines = [str(v) for v in g["cv_log"]]
    num = re.compile(r"Test: \[(\d+)/(\d+)\]\tTime T \(T\)\tLoss (\S+) \((\S+)\)\tAccuracy (\S+) \((\S+)\)")
    assert len(lines) == len(ref_lines) == 2
    for a, b in zip(lines, ref_lines):
        ma, mb = num.fullmatch(a), num.fullmatch(b)
        assert ma and mb and ma.group(1, 2, 5, 6) == mb.group(1, 2, 5, 6), (a, b)
        assert abs(float(ma.group(3)) - float(mb.group(3))) <= 1e-4 and abs(float(ma.group(4)) - float(mb.group(4))) <= 1e-4


@pytest.mark.parametrize("mode", ["fused_keypoints_u8", "fused_keypoints", "flip_test", "ensemble"])
def test_engine_batch_coalescing_gives_the_same_rows(scpose, gpu_ops, mode):
    """core.function._Coalescer: the loader's batches (5 frames here, 16 in the reference's shipped YAML; the last
    one ragged) are queued and the engine runs on `engine_batch` frames at a time -- 8 (loader batches straddle engine batches, the
    second and third engine batch replay the captured forward, the tail is ragged), 256 (everything in one ragged launch) -- against one
    launch per loader batch (engine_batch = 0).  all_preds, all_boxes and image_path must be identical, bit for bit, in every mode that
    coalesces: fused key points (u8 crops / normalised tensors), the flip test (function.py:347-366) and the ensemble (:500-592)."""
    from importlib import import_module
    P = "spacecraft-pose-estimation_amd"
    syn = import_module(P + ".synthetic"); fn = import_module(P + ".core.function"); models = import_module(P + ".models")
    image, n, batches = _loader(syn, 5, as_u8=mode.endswith("u8"))
    assert n % 5 != 0 and n > 3 * 8
    nets = []
    for f in ((1.0, 0.8) if mode == "ensemble" else (1.0,)):
        m = models.pose_hrnet.get_pose_net(syn.chain_cfg(image), is_train=False)
        sd = syn.load_chain_checkpoint(os.path.join(GOLD, "chain_checkpoint.npz"))
        sd["final_layer.weight"] = sd["final_layer.weight"] * f; sd["final_layer.bias"] = sd["final_layer.bias"] * f
        m.load_state_dict(sd, strict=True)
        nets.append(m.cuda().eval())
    N = types.SimpleNamespace
    config = N(MODEL=N(NUM_JOINTS=11, NAME="pose_hrnet", IMAGE_SIZE=[image, image], HEATMAP_SIZE=[image // 4, image // 4]),
               TEST=N(FLIP_TEST=mode == "flip_test", SHIFT_HEATMAP=True, POST_PROCESS=True), PRINT_FREQ=100)
    got = {}

    class DS:
        flip_pairs = [[1, 2], [3, 4], [5, 6]]

        def __len__(self):
            return n

        def evaluate(self, c, preds, output_dir, pred_file_name, all_boxes, image_path, filenames, imgnums):
            got.update(preds=preds.copy(), boxes=all_boxes.copy(), image_path=list(image_path))
            return {"Null": 0}, 0
    lines = []

    class H(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    h = H(); fn.logger.addHandler(h); fn.logger.setLevel(logging.INFO)
    res = {}
    try:
        for eb in (0, 8, 256, None):
            del lines[:]
            if mode == "ensemble":
                fn.validate_cv(config, batches, DS(), nets, None, "", "", "pred_real", log_metrics=False, engine_batch=eb)
            else:
                fn.validate(config, batches, DS(), nets[0], None, "", "", pred_file_name="pred_test", log_metrics=False, engine_batch=eb)
            res[eb] = dict(got)
            assert any("coalesced into engine batches" in l for l in lines) == (eb != 0)
            assert any("Loss n/a" in l for l in lines) and not any("Loss 0.0000" in l for l in lines)     # nothing computed them: not printed as zeros
    finally:
        fn.logger.removeHandler(h)
    for eb in (8, 256, None):
        assert np.array_equal(res[eb]["preds"].view(np.int32), res[0]["preds"].view(np.int32)), (mode, eb)
        assert np.array_equal(res[eb]["boxes"], res[0]["boxes"]) and res[eb]["image_path"] == res[0]["image_path"]
    assert res[0]["preds"].shape == (n, 11, 3) and np.abs(res[0]["preds"][:, :, 2]).max() > 0.1
    if mode.startswith("fused"):      # the second engine batch of 8 is captured, the third replays it (models/pose_hrnet.py: forward_decode)
        assert getattr(nets[0], "_fast_graph", None) is not None
