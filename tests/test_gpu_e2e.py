"""End-to-end on the GPU through the reference-shaped interfaces: module API, core.inference
signatures, validate(), and the two CLIs on a synthetic scene -- compared with the CPU oracle.

Keypoint parity on random-init networks is MARGIN-GATED: random heatmaps are
not peaked, so bf16 noise may legitimately move an argmax whose runner-up is within the noise.
For every joint whose argmax margin and quarter-pixel sign tests exceed twice the measured heatmap deviation the
decoded keypoint must agree within 0.5 px (BASELINE.json north_star)."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from oracle import decode_ref as D
from oracle import hrnet_ref as R
from oracle import pnp_ref as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pk(gpu_ops):
    from importlib import import_module
    ns = types.SimpleNamespace(ops=gpu_ops)
    for m in ("config", "models.pose_hrnet", "core.inference", "core.evaluate", "core.function", "core.loss", "dataset", "utils.transforms", "pose_export"):
        setattr(ns, m.split(".")[-1], import_module("spacecraft-pose-estimation_amd." + m))
    return ns


def test_module_api_matches_oracle(pk):
    """get_pose_net -> load_state_dict(strict=False) -> .cuda().eval() -> model(x), tools/test.py:84-98."""
    cfg = R.tiny_cfg()
    sd = R.make_state_dict(cfg, seed=11)
    model = pk.pose_hrnet.get_pose_net(cfg, is_train=False)
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        y = model(x.cuda())
        ref = R.forward(sd, cfg, x, emulate="bf16")
    assert y.shape == (3, 11, 16, 16) and y.dtype == torch.float32 and y.is_cuda
    assert ((y.cpu() - ref).norm() / ref.norm()).item() < 1.3e-2
    # parameters changed -> engine re-packed, result follows
    sd2 = R.make_state_dict(cfg, seed=13)
    model.load_state_dict(sd2)
    with torch.no_grad():
        y2 = model(x.cuda())
        ref2 = R.forward(sd2, cfg, x, emulate="bf16")
    assert ((y2.cpu() - ref2).norm() / ref2.norm()).item() < 1.3e-2
    model.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        model(x.cuda())


def test_inference_numpy_signatures(pk):
    rng = np.random.default_rng(5)
    hm = rng.standard_normal((4, 11, 32, 32)).astype(np.float32)
    c = (rng.random((4, 2)) * 900).astype(np.float32); s = (rng.random((4, 2)) * 2 + 0.5).astype(np.float32)
    cfg = types.SimpleNamespace(TEST=types.SimpleNamespace(POST_PROCESS=True))
    preds, maxvals = pk.inference.get_final_preds(cfg, hm, c, s)
    rp, rm = D.get_final_preds(True, hm.copy(), c, s)
    assert preds.shape == (4, 11, 2) and maxvals.shape == (4, 11, 1) and preds.dtype == np.float32
    assert np.abs(preds - rp).max() <= 2e-4 and np.array_equal(maxvals, rm)
    coords, mv = pk.inference.get_max_preds(hm)
    rc, rmv = D.get_max_preds(hm)
    assert np.array_equal(coords, rc) and np.array_equal(mv, rmv)


@pytest.mark.parametrize("model,size,n", [("w32", 128, 8), ("w48", 384, 4)])     # w48 384x384: BASELINE configs[2] geometry
def test_margin_gated_keypoint_parity(pk, model, size, n):
    cfg = R.w32_cfg(11, size) if model == "w32" else R.w48_cfg(11, size)
    sd = R.make_state_dict(cfg, seed=21)
    x = torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(22))
    eng = pk.ops.HrnetEngine(cfg, sd)
    hm_gpu = eng(x.cuda())
    with torch.no_grad():
        hm_ref = R.forward(sd, cfg, x)
    c = torch.full((n, 2), 700.0); s = torch.full((n, 2), 1.3)
    kp_gpu = pk.ops.decode(hm_gpu, c.cuda(), s.cuda(), True).cpu().numpy()
    kp_ref = D.decode_xyc(True, hm_ref.numpy(), c.numpy(), s.numpy())
    dev = (hm_gpu.cpu() - hm_ref).abs().flatten(2).amax(2).numpy()               # (N,J) max |bf16 deviation| per map
    top2 = hm_ref.flatten(2).topk(2, dim=2).values.numpy()
    # A decoded keypoint is PROVABLY unaffected by a perturbation bounded by dev when the argmax
    # margin and the two quarter-pixel sign tests (inference.py:62-69) each exceed 2*dev.
    hr = hm_ref.numpy()
    stable = (top2[:, :, 0] - top2[:, :, 1]) > 2 * dev
    for n in range(hr.shape[0]):
        for j in range(hr.shape[1]):
            py, px = np.unravel_index(hr[n, j].argmax(), hr[n, j].shape)
            if 1 < px < hr.shape[3] - 1 and 1 < py < hr.shape[2] - 1:
                dx = hr[n, j, py, px + 1] - hr[n, j, py, px - 1]
                dy = hr[n, j, py + 1, px] - hr[n, j, py - 1, px]
                stable[n, j] &= abs(dx) > 2 * dev[n, j] and abs(dy) > 2 * dev[n, j]
    err = np.linalg.norm(kp_gpu[:, :, :2] - kp_ref[:, :, :2], axis=2)
    step = 1.3 * 200 / (size // 4)
    print("stable joints %d/%d, max err on stable %.3f px (one heatmap px = %.2f image px)" % (stable.sum(), stable.size, err[stable].max() if stable.any() else 0, step))
    # random-init heat-maps are flat: on 96x96 maps (w48, 384 px) only ~1 joint in 7 has a provably stable argmax
    assert stable.sum() >= (stable.size // 4 if model == "w32" else 4)
    assert (err[stable] <= 0.5).all()
    eng.close()


def test_every_keypoint_agrees_or_is_explained_w48_384(pk):
    """Complete (not margin-gated) keypoint parity at the headline geometry (BASELINE configs[2]: W48, 384x384), 64 frames:
    EVERY joint is either within 0.5 px of the oracle keypoint (north_star) or its disagreement is explained by the 16-bit
    noise of the heat-map -- the oracle map's value at the GPU's argmax lies within 2*dev of the oracle maximum (a near-tie
    the noise may flip, lib/core/inference.py:30-40), or the argmax agrees and the quarter-pixel sign test that differs
    (:62-69) has |difference| <= 2*dev in the oracle.  dev is the measured max |HIP - oracle| of that map, and is itself
    bounded against the map's dynamic range (measured: median 2.5 %, max 4.1 % on these nearly flat random-init maps; bound 6 %),
    so that 'explained' cannot be satisfied by an inaccurate heat-map.  Measured split: 515 of 704 within 0.5 px, 181 near-tie
    argmax, 8 quarter-pixel signs, 0 unexplained.
    No joint may be neither; the split is printed."""
    n, size = 64, 384
    cfg = R.w48_cfg(11, size)
    sd = R.make_state_dict(cfg, seed=21)
    x = torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(23))
    eng = pk.ops.HrnetEngine(cfg, sd)
    hm_gpu = eng(x.cuda())
    with torch.no_grad():
        hm_ref = torch.cat([R.forward(sd, cfg, x[i:i + 16]) for i in range(0, n, 16)], 0)
    c = torch.full((n, 2), 700.0); s = torch.full((n, 2), 1.3)
    kp_gpu = pk.ops.decode(hm_gpu, c.cuda(), s.cuda(), True).cpu().numpy()
    kp_ref = D.decode_xyc(True, hm_ref.numpy(), c.numpy(), s.numpy())
    hg, hr = hm_gpu.cpu().numpy(), hm_ref.numpy()
    H, W = hr.shape[2:]
    dev = np.abs(hg - hr).reshape(n, 11, -1).max(2)
    span = hr.reshape(n, 11, -1).max(2) - hr.reshape(n, 11, -1).min(2)
    err = np.linalg.norm(kp_gpu[:, :, :2] - kp_ref[:, :, :2], axis=2)
    agree = near_tie = sign_flip = 0
    bad = []
    for i in range(n):
        for j in range(11):
            if err[i, j] <= 0.5:
                agree += 1
                continue
            gy, gx = np.unravel_index(hg[i, j].argmax(), (H, W))
            ry, rx = np.unravel_index(hr[i, j].argmax(), (H, W))
            if (gy, gx) != (ry, rx):
                if hr[i, j, gy, gx] >= hr[i, j, ry, rx] - 2 * dev[i, j]:
                    near_tie += 1
                else:
                    bad.append((i, j, "argmax", float(err[i, j])))
                continue
            ok = 1 < rx < W - 1 and 1 < ry < H - 1          # same argmax: only a quarter-pixel sign can differ
            if ok:
                dx = hr[i, j, ry, rx + 1] - hr[i, j, ry, rx - 1]; dy = hr[i, j, ry + 1, rx] - hr[i, j, ry - 1, rx]
                gdx = hg[i, j, ry, rx + 1] - hg[i, j, ry, rx - 1]; gdy = hg[i, j, ry + 1, rx] - hg[i, j, ry - 1, rx]
                ok = all(np.sign(a) == np.sign(b) or abs(a) <= 2 * dev[i, j] for a, b in ((dx, gdx), (dy, gdy)))
            if ok:
                sign_flip += 1
            else:
                bad.append((i, j, "quarter-pixel", float(err[i, j])))
    rel = dev / span
    print("W48 384x384, %d joints: %d within 0.5 px, %d near-tie argmax, %d quarter-pixel sign within noise, %d unexplained; "
          "dev / dynamic range of the map: median %.4f max %.4f" % (n * 11, agree, near_tie, sign_flip, len(bad), np.median(rel), rel.max()))
    assert not bad, bad[:8]
    assert agree + near_tie + sign_flip == n * 11
    assert rel.max() <= 0.06, "heat-map noise is not small against the map's dynamic range: %.4f" % rel.max()
    eng.close()


def _scene(tmp_path, n=6, j=11, size=(160, 120)):
    from PIL import Image
    rng = np.random.default_rng(3)
    (tmp_path / "frames").mkdir()
    images, anns = [], []
    for i in range(n):
        name = "frame_%03d.png" % i
        Image.fromarray(rng.integers(0, 255, (size[1], size[0], 3), dtype=np.uint8)).save(tmp_path / "frames" / name)
        images.append({"id": i + 1, "file_name": name, "width": size[0], "height": size[1]})
        anns.append({"image_id": i + 1, "bbox": [10 + 3 * i, 8, 90, 70], "keypoints": [2.0] * (3 * j), "id": i, "category_id": 1})
    (tmp_path / "data").mkdir()
    (tmp_path / "data" / "real_test.json").write_text(json.dumps({"images": images, "annotations": anns}))
    return images, anns


def test_cli_tools_test_and_pose_export(pk, tmp_path):
    """evaluate_pipeline.py:67-91 in miniature: tools/test.py writes pred_test.mat/pred.mat, then
    export_predicted_poses_real.py turns a pred .mat into opencv_poses.json (+ overlays)."""
    images, anns = _scene(tmp_path)
    cfg_node = R.w32_cfg(11, 64)
    sd = R.make_state_dict(cfg_node, seed=31)
    torch.save(sd, tmp_path / "model.pth")
    yaml_path = os.path.join(ROOT, "landmark_regression", "experiments", "bench", "w32_256.yaml")
    out = tmp_path / "out"
    cmd = [sys.executable, "tools/test.py", "--cfg", yaml_path, "OUTPUT_DIR", str(out), "LOG_DIR", str(tmp_path / "log"),
           "DATA_DIR", str(tmp_path / "frames"), "DATASET.ROOT", str(tmp_path / "data"), "DATASET.TEST_SET", "test",
           "MODEL.NUM_JOINTS", "11", "MODEL.IMAGE_SIZE", "[64, 64]", "MODEL.HEATMAP_SIZE", "[16, 16]",
           "TEST.MODEL_FILE", str(tmp_path / "model.pth"), "TEST.BATCH_SIZE_PER_GPU", "4"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "landmark_regression"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    final = out / "EventsDataset" / "pose_hrnet" / "w32_256"
    from scipy.io import loadmat
    preds = loadmat(final / "pred_test.mat")["preds"]
    assert preds.shape == (6, 11, 3) and preds.dtype == np.float32
    assert np.array_equal(preds, loadmat(final / "pred.mat")["preds"])
    # the same crops through the oracle (dataset code is host-side and shared; the network + decode are the HIP path)
    c = pk.config._defaults()
    pk.config.update_config(c, types.SimpleNamespace(cfg=yaml_path, opts=cmd[cmd.index(yaml_path) + 1:], modelDir="", logDir="", dataDir=""))
    T = pk.transforms
    ds = pk.dataset.EventsDataset(c, c.DATASET.ROOT, c.DATA_DIR, "test", False,
                                  T.Compose([T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])]))
    xs = torch.stack([ds[i][0] for i in range(6)])
    with torch.no_grad():
        hm = R.forward(sd, cfg_node, xs)
    cs = np.stack([ds.db[i]["center"] for i in range(6)]); ss = np.stack([ds.db[i]["scale"] for i in range(6)])
    ref = D.decode_xyc(True, hm.numpy(), cs, ss)
    rel = np.abs(preds[:, :, 2] - ref[:, :, 2]).max() / np.abs(ref[:, :, 2]).max()
    assert rel < 5e-2, "maxvals of the CLI run deviate %.3g from the fp32 oracle" % rel

    # What the CLI does by default (round 5): crops warped on the GPU (scpose_crop_warp) and key points straight out of the network's
    # last kernel (no heat-map).  The reference's data flow -- crops warped and normalised in the loader, heat-maps written, loss /
    # PCK logged, get_final_preds on the heat-maps -- is --host_crop --log_metrics.  Every combination must give the same pred .mat,
    # bit for bit; so must a batch size of 2, whose second and third batch replay the captured forward (models/pose_hrnet.py).
    def run(tag, flags, batch="4", workers="0"):
        o = tmp_path / ("out_" + tag)
        c2 = cmd[:2] + flags + cmd[2:] + ["WORKERS", workers]
        c2[c2.index("OUTPUT_DIR") + 1] = str(o)
        c2[c2.index("TEST.BATCH_SIZE_PER_GPU") + 1] = batch
        r2 = subprocess.run(c2, cwd=os.path.join(ROOT, "landmark_regression"), capture_output=True, text=True, timeout=600)
        assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
        return loadmat(o / "EventsDataset" / "pose_hrnet" / "w32_256" / "pred_test.mat")["preds"], r2.stdout + r2.stderr
    assert "fused forward -> key-point path" in r.stdout + r.stderr                  # the default run above took the fused path
    for tag, flags, batch in (("ref_flow", ["--host_crop", "--log_metrics"], "4"), ("dev_crop_heatmaps", ["--log_metrics"], "4"),
                              ("host_crop_fused", ["--host_crop"], "4"), ("graph_replay", ["--engine_batch", "0"], "2"), ("legacy_flag", ["--device_crop"], "4")):
        preds_v, log = run(tag, flags, batch)
        assert np.array_equal(preds_v, preds), tag
        assert ("fused forward -> key-point path" in log) == ("--log_metrics" not in flags), tag
        if "--log_metrics" in flags:
            assert "Accuracy" in log and "Loss" in log
    # loader workers (cfg.WORKERS > 0) come from a pre-loaded fork server, never forked from the process that holds the HIP context
    preds_w, _ = run("workers2", [], "4", workers="2")
    assert np.array_equal(preds_w, preds)
    preds_w, _ = run("workers2_host_crop", ["--host_crop"], "4", workers="2")
    assert np.array_equal(preds_w, preds)
    # engine-batch coalescing (core/function.py: _Coalescer): loader batches of 1 queued into engine batches of 2 (the second is captured, the
    # third replays), one launch per loader batch (--engine_batch 0), and the reference's data flow, which keeps the loader's batches
    for tag, flags, batch in (("coalesce_1_into_2", ["--engine_batch", "2"], "1"), ("no_coalescing", ["--engine_batch", "0"], "2"),
                              ("coalesce_ignored_with_metrics", ["--engine_batch", "4", "--host_crop", "--log_metrics"], "1")):
        preds_w, log = run(tag, flags, batch)
        assert np.array_equal(preds_w, preds), tag
        assert ("coalesced into engine batches" in log) == (tag == "coalesce_1_into_2"), tag

    # a data set of 512+ frames is decoded by worker processes even though the YAML says WORKERS: 0 (as the reference's events-config.yaml does);
    # rows follow annotations[] order whatever the workers' completion order
    big = {"images": images, "annotations": [dict(anns[k % 6], id=k) for k in range(520)]}
    (tmp_path / "data_big").mkdir()
    (tmp_path / "data_big" / "real_test.json").write_text(json.dumps(big))
    cb = list(cmd); cb[cb.index("DATASET.ROOT") + 1] = str(tmp_path / "data_big"); cb[cb.index("OUTPUT_DIR") + 1] = str(tmp_path / "out_big")
    cb[cb.index("TEST.BATCH_SIZE_PER_GPU") + 1] = "64"
    rb = subprocess.run(cb, cwd=os.path.join(ROOT, "landmark_regression"), capture_output=True, text=True, timeout=900)
    assert rb.returncode == 0, rb.stdout[-2000:] + rb.stderr[-2000:]
    assert "decoding with" in rb.stdout + rb.stderr
    pb_ = loadmat(tmp_path / "out_big" / "EventsDataset" / "pose_hrnet" / "w32_256" / "pred_test.mat")["preds"]
    assert pb_.shape == (520, 11, 3) and all(np.array_equal(pb_[k], preds[k % 6]) for k in range(520))

    # stage 3 on known-answer keypoints written in the same .mat format
    kp, Rs, ts = P.synth_keypoints(6, np.random.default_rng(4), 0.5, 0.0)
    from scipy.io import savemat
    savemat(tmp_path / "kp.mat", {"preds": kp})
    (tmp_path / "landmarks.csv").write_text("x,y,z\n" + "\n".join(",".join(repr(float(v)) for v in r) for r in P.LANDMARKS))
    (tmp_path / "calib.json").write_text(json.dumps({"intrinsics": {"camera_matrix": P.CAMERA_K.tolist(),
                                                                    "distortion_coefficients": P.CAMERA_DIST.tolist()}}))
    cmd = [sys.executable, "export_predicted_poses_real.py", "--frames_dir", str(tmp_path / "frames"),
           "--detection_annotations", str(tmp_path / "data" / "real_test.json"), "--pose_annotations", str(tmp_path / "kp.mat"),
           "--landmarks_file", str(tmp_path / "landmarks.csv"), "--calibration_file_path", str(tmp_path / "calib.json"),
           "--output_dir", str(tmp_path / "poses")]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "pose_estimation"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    poses = json.load(open(tmp_path / "poses" / "opencv_poses.json"))
    assert [p["image_name"] for p in poses] == [im["file_name"] for im in images]
    o = P.solve_batch(kp)
    for i, p in enumerate(poses):
        assert np.array(p["T"]).shape == (3, 1) and np.array(p["rotation_matrix"]).shape == (3, 3)
        assert P.rot_angle(np.array(p["rotation_matrix"]), o["R"][i]) <= 1e-4
        assert np.linalg.norm(np.array(p["T"]).ravel() - o["t"][i]) / np.linalg.norm(o["t"][i]) <= 1e-4
        assert os.path.exists(tmp_path / "poses" / ("frame_%03d.jpg" % i))


def test_accuracy_matches_loop_restatement(pk):
    """core.evaluate.accuracy (vectorised) against a plain double loop over samples and joints following
    landmark_regression/lib/core/evaluate.py:16-71 (PCK@0.5 on argmax positions normalised by (H, W) / 10)."""
    rng = np.random.default_rng(5)
    n, j, h, w = 5, 7, 16, 12
    out = rng.standard_normal((n, j, h, w)).astype(np.float32)
    tgt = np.zeros((n, j, h, w), dtype=np.float32)
    for a in range(n):
        for b in range(j):
            if (a + b) % 4 == 0:
                tgt[a, b, 0, 1] = 1.0        # argmax at x = 1: joint excluded (needs both coordinates > 1)
            else:
                y, x = int(rng.integers(2, h)), int(rng.integers(2, w))
                tgt[a, b, y, x] = 1.0
                if (a + b) % 3 == 0:
                    out[a, b, y, x] = 50.0   # exact hit
    acc, avg, cnt, pred = pk.evaluate.accuracy(out, tgt)
    p_idx = out.reshape(n, j, -1).argmax(2); t_idx = tgt.reshape(n, j, -1).argmax(2)
    per = []
    for b in range(j):
        hits, used = 0, 0
        for a in range(n):
            tx, ty = t_idx[a, b] % w, t_idx[a, b] // w
            if tx > 1 and ty > 1:
                used += 1
                px, py = p_idx[a, b] % w, p_idx[a, b] // w
                d = np.hypot((px - tx) / (h / 10.0), (py - ty) / (w / 10.0))
                hits += d < 0.5
        per.append(hits / used if used else -1)
    per = np.array(per, dtype=np.float64)
    assert np.allclose(acc[1:], per)
    good = per >= 0
    assert cnt == int(good.sum()) and np.isclose(avg, per[good].mean()) and np.isclose(acc[0], avg)


def test_evaluate_pipeline_two_scenes(pk, tmp_path):
    """evaluate_pipeline.py (reference :62-91) over two scene directories: per scene tools/test.py -> pred.mat ->
    export_predicted_poses_real.py -> opencv_poses.json, with the reference driver's command line."""
    from PIL import Image
    from scipy.io import loadmat
    rng = np.random.default_rng(11)
    cfg_node = R.w32_cfg(11, 64)
    torch.save(R.make_state_dict(cfg_node, seed=33), tmp_path / "model.pth")
    names = {}
    for scene, n in (("scene_a", 3), ("scene_b", 5)):
        frames = tmp_path / "data" / scene / "event-frames"
        frames.mkdir(parents=True)
        images, anns = [], []
        for i in range(n):
            name = "%s_%02d.png" % (scene, i)
            Image.fromarray(rng.integers(0, 255, (120, 160, 3), dtype=np.uint8)).save(frames / name)
            images.append({"id": i + 1, "file_name": name, "width": 160, "height": 120})
            anns.append({"image_id": i + 1, "bbox": [12 + 2 * i, 9, 88, 72], "keypoints": [2.0] * 33, "id": i, "category_id": 1})
        det = tmp_path / "det" / scene
        det.mkdir(parents=True)
        (det / "real_test.json").write_text(json.dumps({"images": images, "annotations": anns}))     # what stage 1 writes
        names[scene] = [im["file_name"] for im in images]
    (tmp_path / "landmarks.csv").write_text("x,y,z\n" + "\n".join(",".join(repr(float(v)) for v in r) for r in P.LANDMARKS))
    (tmp_path / "calib.json").write_text(json.dumps({"intrinsics": {"camera_matrix": P.CAMERA_K.tolist(),
                                                                    "distortion_coefficients": P.CAMERA_DIST.tolist()}}))
    yaml_path = os.path.join(ROOT, "landmark_regression", "experiments", "bench", "w32_256.yaml")
    cmd = [sys.executable, "evaluate_pipeline.py", "--data_dir", str(tmp_path / "data"), "--detection_model_file", "unused.pth",
           "--regression_model_file", str(tmp_path / "model.pth"), "--detection_annotations_base", str(tmp_path / "det"),
           "--regression_annotations_base", str(tmp_path / "reg"), "--pose_estimation_base", str(tmp_path / "poses"),
           "--validation_annotations", "unused.json", "--landmarks_file", str(tmp_path / "landmarks.csv"),
           "--calibration_file_path", str(tmp_path / "calib.json"), "--image_width", "160", "--image_height", "120",
           "--joints_count", "11", "--cfg", yaml_path, "--no_overlay",
           "--regression_opts", "MODEL.IMAGE_SIZE", "[64, 64]", "MODEL.HEATMAP_SIZE", "[16, 16]", "LOG_DIR", str(tmp_path / "log")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for scene, files in names.items():
        preds = loadmat(tmp_path / "reg" / scene / "EventsDataset" / "pose_hrnet" / "w32_256" / "pred.mat")["preds"]
        assert preds.shape == (len(files), 11, 3)
        poses = json.load(open(tmp_path / "poses" / scene / "opencv_poses.json"))
        assert [p["image_name"] for p in poses] == files
        assert all(np.array(p["rotation_matrix"]).shape == (3, 3) and np.array(p["T"]).shape == (3, 1) for p in poses)
    # a scene without detection annotations is reported, not silently skipped
    (tmp_path / "data" / "scene_c" / "event-frames").mkdir(parents=True)
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "stage 1" in (r.stdout + r.stderr)


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """`bench.py --gpus 2` stand-alone: parallel.spawn_local_ranks starts two fresh ranks (file-store rendezvous), each takes its
    frame shard, the (R, t, status) blocks are all-gathered over RCCL and rank 0 prints the JSON line with n_gpus = 2.
    Needs two visible devices (the 1-GPU test boxes skip it; the driver's 8-GPU scaling run goes through torch.distributed.run)."""
    from importlib import import_module
    par = import_module("spacecraft-pose-estimation_amd.parallel")
    if (par.visible_gpu_count() or torch.cuda.device_count()) < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-frames", "0"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_step"] == 512 and line["poses_ok"] > 0


def test_cli_files_to_poses_recovers_the_rendered_poses_w48(pk, tmp_path):
    """The whole path at the FILE boundary on frames with content, at the headline geometry: 1920 x 1200 PNG frames showing the landmarks of
    seeded poses (synthetic.landmark_scene) + the COCO boxes a detector would write -> tools/test.py (HRNet-W48 384 x 384, the constructed
    peaked-heat-map checkpoint saved as a .pth; default flags: loader ships frame windows, GPU crop warp, fused forward -> key points)
    -> pred.mat -> export_predicted_poses_real.py -> opencv_poses.json.  Known answer: the key points are the drawn landmark
    positions, the poses are the ones the frames were rendered from (to the <= 1 crop pixel between a landmark's projection and
    the lattice point it is drawn at).  The reference's data flow (--host_crop --log_metrics) must give the same pred.mat."""
    from PIL import Image
    from scipy.io import loadmat
    from importlib import import_module
    syn = import_module("spacecraft-pose-estimation_amd.synthetic")
    n = 6
    sc = syn.landmark_scene(n, np.random.default_rng(77))
    (tmp_path / "frames").mkdir(); (tmp_path / "data").mkdir()
    images, anns = [], []
    for i in range(n):
        name = "f%02d.png" % i
        Image.fromarray(sc["frames"][i]).save(tmp_path / "frames" / name)
        images.append({"id": i + 1, "file_name": name, "width": 1920, "height": 1200})
        anns.append({"image_id": i + 1, "bbox": [float(v) for v in sc["bbox"][i]], "keypoints": [2.0] * 33, "id": i, "category_id": 1})
    (tmp_path / "data" / "real_test.json").write_text(json.dumps({"images": images, "annotations": anns}))
    torch.save(syn.w48_chain_checkpoint(0), tmp_path / "w48_chain.pth")
    yaml_path = os.path.join(ROOT, "landmark_regression", "experiments", "bench", "w48_384.yaml")
    preds = {}
    for tag, flags in (("default", []), ("ref_flow", ["--host_crop", "--log_metrics"])):
        out = tmp_path / ("out_" + tag)
        cmd = [sys.executable, "tools/test.py"] + flags + ["--cfg", yaml_path, "OUTPUT_DIR", str(out), "LOG_DIR", str(tmp_path / "log"),
               "DATA_DIR", str(tmp_path / "frames"), "DATASET.ROOT", str(tmp_path / "data"), "DATASET.TEST_SET", "test", "MODEL.NUM_JOINTS", "11",
               "TEST.MODEL_FILE", str(tmp_path / "w48_chain.pth"), "TEST.BATCH_SIZE_PER_GPU", "4", "WORKERS", "2"]
        r = subprocess.run(cmd, cwd=os.path.join(ROOT, "landmark_regression"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        preds[tag] = loadmat(out / "EventsDataset" / "pose_hrnet" / "w48_384" / "pred.mat")["preds"]
    assert np.array_equal(preds["default"], preds["ref_flow"])
    p = preds["default"]
    err = np.linalg.norm(p[:, :, :2] - sc["kp"], axis=2)
    print("files -> key points: max |decoded - drawn| = %.3f frame px; peaks %.2f .. %.2f" % (err.max(), p[:, :, 2].min(), p[:, :, 2].max()))
    # (frames are not pre-selected for margin here, unlike tests/golden/chain_w48_reference.npz: a quarter-pixel decision within 16-bit
    # noise may fall the other way -- 0.25 or 0.5 heat-map pixels = one or two crop pixels of side / 384 frame pixels)
    crop_px = 1.5 * sc["bbox"][:, 2] / 384.0
    assert (err <= 0.5).mean() >= 0.95 and (err <= 2.2 * crop_px[:, None]).all() and p[:, :, 2].min() > 0.4
    (tmp_path / "landmarks.csv").write_text("x,y,z\n" + "\n".join(",".join(repr(float(v)) for v in r) for r in P.LANDMARKS))
    (tmp_path / "calib.json").write_text(json.dumps({"intrinsics": {"camera_matrix": P.CAMERA_K.tolist(),
                                                                    "distortion_coefficients": P.CAMERA_DIST.tolist()}}))
    final = tmp_path / "out_default" / "EventsDataset" / "pose_hrnet" / "w48_384"
    cmd = [sys.executable, "export_predicted_poses_real.py", "--frames_dir", str(tmp_path / "frames"),
           "--detection_annotations", str(tmp_path / "data" / "real_test.json"), "--pose_annotations", str(final / "pred.mat"),
           "--landmarks_file", str(tmp_path / "landmarks.csv"), "--calibration_file_path", str(tmp_path / "calib.json"),
           "--output_dir", str(tmp_path / "poses"), "--no_overlay"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "pose_estimation"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    poses = json.load(open(tmp_path / "poses" / "opencv_poses.json"))
    assert [q["image_name"] for q in poses] == [im["file_name"] for im in images]
    Rg = np.array([q["rotation_matrix"] for q in poses]); tg = np.array([q["T"] for q in poses]).reshape(n, 3)
    ang = P.rot_angle(Rg, sc["R"]); terr = np.linalg.norm(tg - sc["t"], axis=1) / np.linalg.norm(sc["t"], axis=1)
    print("files -> poses vs the rendered poses: rotation max %.2e rad, translation max %.2e" % (ang.max(), terr.max()))
    assert ang.max() < 5e-2 and terr.max() < 2e-2


def test_pose_export_on_the_gpu_equals_the_reference_main_output(pk, tmp_path):
    """The same scene as tests/test_host.py::test_pose_export_writes_what_the_reference_main_writes, with the real kernel: the poses in
    opencv_poses.json are those the reference's main() wrote (its cv2 answered by the C oracle) to 1e-4 rad / 1e-4, record for record."""
    from test_host import _export_scene
    g = np.load(os.path.join(ROOT, "tests", "golden", "export_reference_outputs.npz"))
    _export_scene(tmp_path, g, with_frames=False)
    cmd = [sys.executable, "export_predicted_poses_real.py", "--frames_dir", str(tmp_path / "frames"), "--detection_annotations", str(tmp_path / "det.json"),
           "--pose_annotations", str(tmp_path / "pred.mat"), "--landmarks_file", str(tmp_path / "landmarks.csv"),
           "--calibration_file_path", str(tmp_path / "calib.json"), "--output_dir", str(tmp_path / "out")]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "pose_estimation"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = json.load(open(tmp_path / "out" / "opencv_poses.json")); ref = json.loads(str(g["json_text"]))
    assert [q["image_name"] for q in got] == [q["image_name"] for q in ref] and all(list(a) == list(b) for a, b in zip(got, ref))
    Rg = np.array([q["rotation_matrix"] for q in got]); Rr = np.array([q["rotation_matrix"] for q in ref])
    tg = np.array([q["T"] for q in got]).reshape(-1, 3); tr_ = np.array([q["T"] for q in ref]).reshape(-1, 3)
    assert P.rot_angle(Rg, Rr).max() <= 1e-4 and (np.linalg.norm(tg - tr_, axis=1) / np.linalg.norm(tr_, axis=1)).max() <= 1e-4
