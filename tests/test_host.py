"""Host-side mirror of the reference interfaces (config, logger paths, dataset, module tree,
sharding): everything that runs without a GPU."""
import json
import os
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import decode_ref as D
from oracle import hrnet_ref as R


@pytest.fixture(scope="module")
def P(scpose):
    from importlib import import_module
    ns = types.SimpleNamespace()
    for m in ("config", "parallel", "synthetic", "utils.utils", "utils.transforms", "dataset", "models.pose_hrnet", "pose_export"):
        setattr(ns, m.split(".")[-1], import_module("spacecraft-pose-estimation_amd." + m))
    return ns


def _args(cfg_path, opts=(), **kw):
    a = types.SimpleNamespace(cfg=cfg_path, opts=list(opts), modelDir="", logDir="", dataDir="")
    a.__dict__.update(kw)
    return a


YAML = """
GPUS: (0,1)
OUTPUT_DIR: 'out'
WORKERS: 0
DATASET: {DATASET: EventsDataset, ROOT: 'data/', TEST_SET: test, COLOR_RGB: true, DATA_FORMAT: png}
MODEL:
  NAME: pose_hrnet
  NUM_JOINTS: 17
  IMAGE_SIZE: [64, 64]
  HEATMAP_SIZE: [16, 16]
  SIGMA: 2
  EXTRA:
    FINAL_CONV_KERNEL: 1
    PRETRAINED_LAYERS: ['*']
    STAGE2: {NUM_MODULES: 1, NUM_BRANCHES: 2, BLOCK: BASIC, NUM_BLOCKS: [4, 4], NUM_CHANNELS: [16, 32], FUSE_METHOD: SUM}
    STAGE3: {NUM_MODULES: 1, NUM_BRANCHES: 3, BLOCK: BASIC, NUM_BLOCKS: [4, 4, 4], NUM_CHANNELS: [16, 32, 64], FUSE_METHOD: SUM}
    STAGE4: {NUM_MODULES: 1, NUM_BRANCHES: 4, BLOCK: BASIC, NUM_BLOCKS: [4, 4, 4, 4], NUM_CHANNELS: [16, 32, 64, 128], FUSE_METHOD: SUM}
TEST: {BATCH_SIZE_PER_GPU: 4, POST_PROCESS: true}
"""


@pytest.fixture()
def cfg(P, tmp_path):
    p = tmp_path / "tiny-config.yaml"
    p.write_text(YAML)
    c = P.config._defaults()
    P.config.update_config(c, _args(str(p), ["MODEL.NUM_JOINTS", "11", "OUTPUT_DIR", str(tmp_path / "o"), "LOG_DIR", str(tmp_path / "l"),
                                             "DATASET.ROOT", str(tmp_path), "DATA_DIR", str(tmp_path / "img"), "TEST.MODEL_FILE", "x.pth"]))
    return c


def test_config_yacs_semantics(P, cfg, tmp_path):
    assert cfg.GPUS == (0, 1) and cfg.MODEL.NUM_JOINTS == 11 and cfg.TEST.POST_PROCESS is True
    assert cfg.MODEL.EXTRA.STAGE3.NUM_CHANNELS == [16, 32, 64] and cfg.MODEL.SIGMA2 == 4      # default kept
    assert cfg["MODEL"]["EXTRA"]["STAGE2"]["BLOCK"] == "BASIC"                                  # dict-style access (pose_hrnet.py:278)
    with pytest.raises(AttributeError):
        cfg.OUTPUT_DIR = "frozen"
    c = P.config._defaults()
    with pytest.raises(KeyError):
        c.merge_from_list(["MODEL.NO_SUCH_KEY", "1"])
    with pytest.raises(ValueError):
        c.merge_from_list(["WORKERS", "abc"])
    c.merge_from_list(["MODEL.EXTRA.ANYTHING", "[1, 2]", "TRAIN.LR", "1"])                   # new_allowed subtree; int -> float
    assert c.MODEL.EXTRA.ANYTHING == [1, 2] and c.TRAIN.LR == 1.0
    ref_yaml = "/root/reference/landmark_regression/experiments/events/events-config.yaml"
    if os.path.exists(ref_yaml):       # build container only: the reference's own YAML parses unchanged
        c2 = P.config._defaults()
        P.config.update_config(c2, _args(ref_yaml))
        assert c2.MODEL.IMAGE_SIZE == [512, 512] and c2.GPUS == (0,)


def test_shipped_experiment_yamls_load(P):
    root = os.path.join(os.path.dirname(os.path.dirname(__file__)), "landmark_regression", "experiments")
    for rel, ch, img in (("events/events-config.yaml", 32, 512), ("bench/w32_256.yaml", 32, 256), ("bench/w48_384.yaml", 48, 384)):
        c = P.config._defaults()
        P.config.update_config(c, _args(os.path.join(root, rel)))
        assert c.MODEL.EXTRA.STAGE2.NUM_CHANNELS[0] == ch and c.MODEL.IMAGE_SIZE == [img, img]
        assert len(P.pose_hrnet.get_pose_net(c, False).state_dict()) == 1754


def test_logger_output_dir_layout(P, cfg, tmp_path):
    _, out, tb = P.utils.create_logger(cfg, "experiments/events/tiny-config.yaml", "valid")
    assert out == str(tmp_path / "o" / "EventsDataset" / "pose_hrnet" / "tiny-config")           # evaluate_pipeline.py:88
    assert os.path.isdir(out) and os.path.isdir(tb)


def _make_dataset(tmp_path, n=5, j=11):
    from PIL import Image
    (tmp_path / "img").mkdir(exist_ok=True)
    rng = np.random.default_rng(0)
    images, anns = [], []
    for i in range(n):
        name = "f%03d.png" % i
        Image.fromarray(rng.integers(0, 255, (120, 160, 3), dtype=np.uint8)).save(tmp_path / "img" / name)
        images.append({"id": 100 + i, "file_name": name, "width": 160, "height": 120})
        anns.append({"image_id": 100 + i, "bbox": [20 + i, 10, 80, 60 + i], "keypoints": [2.0] * (3 * j), "id": i, "category_id": 1})
    (tmp_path / "test.json").write_text(json.dumps({"images": images, "annotations": anns[::-1]}))   # annotations[] order != images[] order
    return images, anns[::-1]


def test_events_dataset_center_scale_and_mat(P, cfg, tmp_path):
    images, anns = _make_dataset(tmp_path)
    T = P.transforms
    ds = P.dataset.EventsDataset(cfg, cfg.DATASET.ROOT, cfg.DATA_DIR, "test", False,
                                 T.Compose([T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])]))
    assert len(ds) == 5
    for rec, a in zip(ds.db, anns):                                    # rows follow annotations[] order (events.py:58)
        x, y, w, h = a["bbox"]
        assert np.array_equal(rec["center"], np.array([x + w * 0.5, y + h * 0.5], np.float32))
        assert np.array_equal(rec["scale"], np.array([w / 200.0, h / 200.0], np.float32) * 1.5)
    inp, target, tw, meta = ds[0]
    assert inp.shape == (3, 64, 64) and inp.dtype == torch.float32 and target.shape == (11, 16, 16)
    assert meta["score"] == 1 and meta["image"].endswith(anns[0] and "f004.png")
    preds = np.arange(5 * 11 * 3, dtype=np.float32).reshape(5, 11, 3)
    ds.evaluate(cfg, preds, str(tmp_path), "pred_test")
    from scipy.io import loadmat
    for f in ("pred_test.mat", "pred.mat"):                            # both spellings (SURVEY 3.1 quirk ii)
        assert np.array_equal(loadmat(tmp_path / f)["preds"], preds)
    os.rename(tmp_path / "test.json", tmp_path / "real_test.json")     # quirk (i): stage 1 writes real_test.json
    assert len(P.dataset.EventsDataset(cfg, cfg.DATASET.ROOT, cfg.DATA_DIR, "test", False)) == 5
    with pytest.raises(ValueError, match="Fail to read"):
        ds.db[0]["image"] = str(tmp_path / "missing.png")
        ds[0]


def test_host_affine_matches_oracle(P):
    rng = np.random.default_rng(1)
    for _ in range(10):
        c = (rng.random(2) * 800).astype(np.float32); s = (rng.random(2) * 2 + 0.3).astype(np.float32)
        for inv in (0, 1):
            assert np.allclose(P.transforms.get_affine_transform(c, s, 0, [64, 48], inv=inv),
                               D.get_affine_transform(c, s, 0, [64, 48], inv=inv), atol=1e-9)
    img = rng.integers(0, 255, (50, 60, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(P.transforms.warp_affine_bilinear(img, ident, (60, 50)), img)
    shifted = P.transforms.warp_affine_bilinear(img, np.array([[1.0, 0, 2], [0, 1.0, 3]]), (60, 50))
    assert np.array_equal(shifted[3:, 2:], img[:-3, :-2]) and (shifted[:3] == 0).all()


def test_module_tree_state_dict_and_cpu_refusal(P):
    cfg = R.tiny_cfg()
    net = P.pose_hrnet.get_pose_net(cfg, False).eval()
    sd = R.make_state_dict(cfg, seed=1)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    missing = dict(sd); del missing["final_layer.bias"]
    res = net.load_state_dict(missing, strict=False)                    # tools/test.py:90
    assert res.missing_keys == ["final_layer.bias"]
    with pytest.raises(Exception, match="no CPU fallback|ROCm"):
        net(torch.zeros(1, 3, 64, 64))                                  # fails loudly, never an eager fallback
    with pytest.raises(ValueError):
        P.pose_hrnet.get_pose_net(cfg, True)
    # STAGEk.BLOCK = BOTTLENECK (blocks_dict, pose_hrnet.py:266-269): same keys, in the same order, as the reference module registers
    bcfg = R.bneck_cfg(c=16)
    bsd = R.make_state_dict(bcfg, seed=2)
    bnet = P.pose_hrnet.get_pose_net(bcfg, False).eval()
    assert list(bnet.state_dict().keys()) == list(bsd.keys())
    bnet.load_state_dict(bsd, strict=True)


def test_synthetic_checkpoint_equals_oracle_recipe(P):
    cfg = P.synthetic.hrnet_cfg(16, 11, 64, modules=(1, 1, 1))
    assert cfg == R.tiny_cfg()
    a, b = P.synthetic.random_checkpoint(cfg, 5), R.make_state_dict(cfg, 5)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    kp, rs, ts = P.synthetic.keypoints(4, np.random.default_rng(0), 0.0, 0.0)
    assert np.abs(P.synthetic.project(rs[0], ts[0], P.synthetic.TANGO_LANDMARKS) - kp[0, :, :2]).max() < 1e-3


def test_shard_range_is_a_contiguous_partition(P):
    for n in (0, 1, 7, 256, 2048, 2051):
        for ws in (1, 2, 3, 8):
            spans = [P.parallel.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_landmark_csv_reader(P, tmp_path):
    (tmp_path / "lm.csv").write_text("x,y,z\n0.1,0.2,0.3\n-1,2,3.5\n")
    assert np.array_equal(P.pose_export.read_landmarks(tmp_path / "lm.csv"), np.array([[0.1, 0.2, 0.3], [-1, 2, 3.5]]))


def test_evaluate_pipeline_path_conventions(tmp_path):
    """evaluate_pipeline.py (reference :49-91): stage-relative paths, scene discovery, detection-file spellings."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("evaluate_pipeline", os.path.join(ROOT, "evaluate_pipeline.py"))
    ep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ep)
    assert ep.under("regression", "out/x") == os.path.join(ROOT, "landmark_regression", "out", "x")
    assert ep.under("detection", "../shared/det") == os.path.join(ROOT, "shared", "det")
    assert ep.under("pose", "/abs/lm.csv") == "/abs/lm.csv"
    (tmp_path / "b_scene").mkdir(); (tmp_path / "a_scene").mkdir(); (tmp_path / "notes.txt").write_text("x")
    assert ep.scenes_of(str(tmp_path)) == ["a_scene", "b_scene"]
    with pytest.raises(SystemExit, match="stage 1"):
        ep.detection_file(str(tmp_path / "a_scene"))
    (tmp_path / "a_scene" / "real_test.json").write_text("{}")
    assert ep.detection_file(str(tmp_path / "a_scene")).endswith("real_test.json")
    (tmp_path / "a_scene" / "test.json").write_text("{}")
    assert ep.detection_file(str(tmp_path / "a_scene")).endswith(os.sep + "test.json")
    cfg = os.path.join(ROOT, "landmark_regression", "experiments", "events", "events-config.yaml")
    assert ep.output_names(cfg, []) == ("EventsDataset", "pose_hrnet")
    assert ep.output_names(cfg, ["MODEL.NAME", "hrnet_cms", "DATASET.DATASET", "PEdataset"]) == ("PEdataset", "hrnet_cms")
    a = ep.parse_args(["--data_dir", "d", "--regression_model_file", "m.pth", "--detection_annotations_base", "det",
                       "--regression_annotations_base", "reg", "--pose_estimation_base", "pose", "--landmarks_file", "lm.csv",
                       "--calibration_file_path", "calib.json"])
    assert (a.image_width, a.image_height, a.joints_count) == (640, 480, 24)      # reference defaults (:38-43)


def test_warp_affine_is_opencvs_fixed_point_algorithm():
    """utils.transforms.warp_affine_bilinear (the data loader's cv2.warpAffine stand-in, JointsDataset.py:191-195) against
    the scalar restatement of OpenCV 3.4's uint8 INTER_LINEAR path (oracle/warp_ref.py), and the algorithm's known
    answers: identity and integer shifts copy pixels, a half-pixel shift averages neighbours rounding .5 UP,
    coordinates are quantised to 1/32 px."""
    from importlib import import_module
    from oracle import warp_ref as W
    T = import_module("spacecraft-pose-estimation_amd.utils.transforms")
    tab = W.bilinear_tab_i()
    assert tab[0] == [32767, 0, 0, 1] and tab[1] == [31744, 1024, 0, 0] and all(sum(t) == 32768 for t in tab)
    rng = np.random.default_rng(0)
    for (h, w, c, s, o) in [(30, 40, (20.0, 15.0), (0.1, 0.1), (24, 20)), (17, 23, (2.0, 3.0), (0.2, 0.05), (16, 12)),
                            (40, 40, (50.0, -5.0), (0.3, 0.3), (20, 20)), (9, 7, (3.0, 4.0), (0.02, 0.02), (12, 10))]:
        f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        t = T.get_affine_transform(np.array(c, np.float32), np.array(s, np.float32), 0, np.array(o))
        assert np.array_equal(T.warp_affine_bilinear(f, t, o), W.warp_affine_linear_u8(f, t, o))
    g = rng.integers(0, 256, (12, 14), dtype=np.uint8)
    assert np.array_equal(T.warp_affine_bilinear(g, np.array([[1, 0, 0], [0, 1, 0]], float), (14, 12)), g)
    shifted = T.warp_affine_bilinear(g, np.array([[1, 0, 3], [0, 1, -2]], float), (14, 12))
    want = np.zeros_like(g); want[:10, 3:] = g[2:, :11]
    assert np.array_equal(shifted, want)
    half = T.warp_affine_bilinear(g, np.array([[1, 0, 0.5], [0, 1, 0]], float), (14, 12))
    left = np.concatenate([np.zeros((12, 1), int), g.astype(int)], 1)
    assert np.array_equal(half, ((left[:, :-1] + left[:, 1:] + 1) >> 1).astype(np.uint8))
    # 1/32-px quantisation: a shift of 1/100 px is a shift of 0 (0.01 * 32 rounds to 0), 1/50 px is 1/32 px
    tiny = T.warp_affine_bilinear(g, np.array([[1, 0, 0.01], [0, 1, 0]], float), (14, 12))
    assert np.array_equal(tiny, g)
    q = T.warp_affine_bilinear(g, np.array([[1, 0, -0.02], [0, 1, 0]], float), (14, 12)).astype(int)
    right = np.concatenate([g.astype(int), np.zeros((12, 1), int)], 1)
    assert np.array_equal(q, (right[:, :-1] * 31 * 1024 + right[:, 1:] * 1024 + 16384) >> 15)


def test_mat_v5_writer_and_reader_interoperate_with_scipy(tmp_path):
    """utils/matio.py (own Level-5 MAT writer / reader for pred*.mat; events.py:121-125, export_predicted_poses_real.py:172-173):
    SciPy reads what it writes, it reads what SciPy writes (plain and compressed), and for the pipeline's file the bytes
    after the 128-byte text header are identical to scipy.io.savemat's."""
    import scipy.io
    from importlib import import_module
    M = import_module("spacecraft-pose-estimation_amd.utils.matio")
    rng = np.random.default_rng(0)
    preds = rng.standard_normal((7, 11, 3)).astype(np.float32)
    extra = {"k": np.arange(5, dtype=np.int32), "s": np.float64(3.5), "u": rng.integers(0, 255, (2, 3, 4, 2)).astype(np.uint8),
             "d": rng.standard_normal((4, 1)), "e": np.zeros((0, 3), dtype=np.float32)}
    M.savemat(tmp_path / "a.mat", dict(preds=preds, **extra))
    r = scipy.io.loadmat(tmp_path / "a.mat")
    assert np.array_equal(r["preds"], preds) and r["preds"].dtype == np.float32
    assert np.array_equal(r["k"], extra["k"].reshape(1, -1)) and r["s"][0, 0] == 3.5 and np.array_equal(r["u"], extra["u"])
    assert np.array_equal(r["d"], extra["d"]) and r["e"].shape == (0, 3)
    mine = M.loadmat(tmp_path / "a.mat")
    assert np.array_equal(mine["preds"], preds) and mine["u"].dtype == np.uint8 and np.array_equal(mine["u"], extra["u"])
    for comp in (False, True):
        scipy.io.savemat(tmp_path / "b.mat", {"preds": preds, "x": np.arange(3.0), "b": np.uint8(7)}, do_compression=comp)
        m = M.loadmat(tmp_path / "b.mat")
        assert np.array_equal(m["preds"], preds) and m["preds"].dtype == np.float32
        assert np.array_equal(m["x"], np.arange(3.0).reshape(1, 3)) and int(m["b"][0, 0]) == 7
    scipy.io.savemat(tmp_path / "c.mat", {"preds": preds})
    M.savemat(tmp_path / "d.mat", {"preds": preds})
    assert (tmp_path / "c.mat").read_bytes()[128:] == (tmp_path / "d.mat").read_bytes()[128:]
    with pytest.raises(TypeError):
        M.savemat(tmp_path / "e.mat", {"s": np.array(["text"])})


def test_warp_window_contains_every_tap_of_the_fixed_point_warp():
    """ops.warp_window (what the loader ships to scpose_crop_warp_roi): every in-frame tap of cv2.warpAffine's fixed-point coordinate
    computation (csrc/crop.hip: X = (round((M1 y + M2) 1024) + 16 + round(M0 x 1024)) >> 5, taps (X >> 5, X >> 5 + 1)) lies inside
    the window, for boxes inside, across and outside the frame and for scales from 0.1 to 4."""
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    T = import_module("spacecraft-pose-estimation_amd.utils.transforms")
    ops = import_module("spacecraft-pose-estimation_amd.ops")
    rng = np.random.default_rng(3)
    for k in range(200):
        fh, fw = int(rng.integers(40, 1300)), int(rng.integers(40, 2000))
        c = np.array([rng.uniform(-0.2, 1.2) * fw, rng.uniform(-0.2, 1.2) * fh], np.float32)
        s = np.float32(rng.uniform(0.1, 4.0)) * np.ones(2, np.float32)
        ow, oh = (96, 128) if k % 2 else (384, 384)
        t = T.get_affine_transform(c, s, 0, np.array([ow, oh]))
        m = np.asarray(T.invert_affine_cv(t), dtype=np.float64).reshape(6)
        xs = np.arange(ow, dtype=np.float64)[None, :]; ys = np.arange(oh, dtype=np.float64)[:, None]
        X = (np.rint((m[1] * ys + m[2]) * 1024.0).astype(np.int64) + 16 + np.rint(m[0] * xs * 1024.0).astype(np.int64)) >> 5
        Y = (np.rint((m[4] * ys + m[5]) * 1024.0).astype(np.int64) + 16 + np.rint(m[3] * xs * 1024.0).astype(np.int64)) >> 5
        x0, y0 = X >> 5, Y >> 5
        rx, ry, rw, rh = ops.warp_window(t, (ow, oh), (fh, fw))
        for dx in (0, 1):
            for dy in (0, 1):
                xx, yy = x0 + dx, y0 + dy
                inside = (xx >= 0) & (xx < fw) & (yy >= 0) & (yy < fh)
                assert ((xx[inside] >= rx) & (xx[inside] < rx + rw) & (yy[inside] >= ry) & (yy[inside] < ry + rh)).all(), (k, fh, fw, c, s)
        assert 0 <= rx <= fw and 0 <= ry <= fh and rx + rw <= fw and ry + rh <= fh


def test_crop_affine_equals_the_reference_get_affine_transform():
    """utils.transforms.get_affine_transform / affine_transform (and the oracle's copy) against vectors made by the reference's own
    lib/utils/transforms.py:57-95 (tests/golden/affine_reference_outputs.npz, make_golden.py: affine_vectors): the affine the crop is cut
    with, its inverse, and joints mapped through it -- the geometry scpose_crop_warp and ops.warp_window start from."""
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    T = import_module("spacecraft-pose-estimation_amd.utils.transforms")
    from oracle import decode_ref as D
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "affine_reference_outputs.npz"))
    for i in range(len(g["center"])):
        c, s, size = g["center"][i], g["scale"][i], g["sizes"][i]
        for mod in (T, D):
            assert np.abs(mod.get_affine_transform(c, s, 0, size) - g["forward"][i]).max() <= 1e-9 * max(1.0, np.abs(g["forward"][i]).max())
            assert np.abs(mod.get_affine_transform(c, s, 0, size, inv=1) - g["inverse"][i]).max() <= 1e-9 * max(1.0, np.abs(g["inverse"][i]).max())
        got = np.stack([T.affine_transform(p, g["forward"][i]) for p in g["points"][i]])
        assert np.abs(got - g["mapped"][i]).max() <= 1e-9 * max(1.0, np.abs(g["mapped"][i]).max())


def test_dataset_box2cs_and_generate_target_equal_the_reference_classes():
    """EventsDataset._xywh2cs and JointsDataset.generate_target against vectors produced by the reference's own classes
    (tests/golden/dataset_reference_outputs.npz, make_golden.py: dataset_vectors -- lib/dataset/events.py:94-113, lib/dataset/JointsDataset.py:264-332
    called unbound): bit-identical centres / scales; bit-identical gaussian targets and weights for joints inside, on the corners, just
    outside the map and invisible, at 96 x 96, 48 x 64 and the 768 x 768 maps of the hrnet_cms configurations."""
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    ds = import_module("spacecraft-pose-estimation_amd.dataset")
    jd = import_module("spacecraft-pose-estimation_amd.dataset.JointsDataset")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_reference_outputs.npz"))
    ns = types.SimpleNamespace(pixel_std=200)
    for b, c, s in zip(g["boxes"], g["center"], g["scale"]):
        cc, ss = ds.EventsDataset._xywh2cs(ns, *b)
        assert cc.dtype == np.float32 and ss.dtype == np.float32 and np.array_equal(cc, c) and np.array_equal(ss, s)
    for name in ("w48", "rect", "cms768"):
        iw, ih, hw, hh, sigma = [int(v) for v in g[name + "/meta"]]
        o = types.SimpleNamespace(num_joints=11, target_type="gaussian", image_size=np.array([iw, ih]), heatmap_size=np.array([hw, hh]), sigma=sigma,
                                  use_different_joints_weight=False, joints_weight=1)
        pos = 0
        for i in range(len(g[name + "/joints"])):
            t, w = jd.JointsDataset.generate_target(o, g[name + "/joints"][i].copy(), g[name + "/vis"][i].copy())
            assert t.dtype == np.float32 and np.array_equal(w, g[name + "/weight"][i])
            if name != "cms768":
                assert np.array_equal(t, g[name + "/target"][i])
            else:
                for k in range(11):
                    cnt = int(g[name + "/nz_count"][i * 11 + k])
                    nz = np.nonzero(t[k])
                    assert len(nz[0]) == cnt and np.array_equal((nz[0] * hw + nz[1]).astype(np.int32), g[name + "/nz_index"][pos:pos + cnt])
                    assert np.array_equal(t[k][nz], g[name + "/nz_value"][pos:pos + cnt])
                    pos += cnt


def test_output_tree_naming_equals_the_reference_create_logger(tmp_path):
    """utils.utils.create_logger against the directories the reference's own create_logger (lib/utils/utils.py:22-57) made
    (tests/golden/naming_reference_outputs.npz): evaluate_pipeline.py:88 finds pred.mat through this naming."""
    import logging
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    U = import_module("spacecraft-pose-estimation_amd.utils.utils")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "naming_reference_outputs.npz"))
    before = list(logging.getLogger().handlers)
    for k, (ds, hy, model, cfg_name, phase) in enumerate(g["cases"]):
        cfg = types.SimpleNamespace(OUTPUT_DIR=str(tmp_path / ("o%d" % k)), LOG_DIR=str(tmp_path / ("l%d" % k)),
                                    DATASET=types.SimpleNamespace(DATASET=str(ds), HYBRID_JOINTS_TYPE=str(hy)), MODEL=types.SimpleNamespace(NAME=str(model)))
        _, fo, tb = U.create_logger(cfg, str(cfg_name), str(phase))
        assert os.path.relpath(fo, cfg.OUTPUT_DIR) == str(g["rel_out"][k]) and os.path.isdir(fo)
        assert os.path.relpath(os.path.dirname(tb), cfg.LOG_DIR) + "|" + os.path.basename(tb)[:-17] == str(g["rel_log"][k])
    for h in list(logging.getLogger().handlers):
        if h not in before:
            logging.getLogger().removeHandler(h); h.close()


def test_driver_command_lines_follow_the_reference_driver(tmp_path, monkeypatch):
    """evaluate_pipeline.py against the command lines the REFERENCE's driver issues (tests/golden/driver_reference_commands.npz: the
    reference's evaluate_pipeline.py:62-91 run with subprocess.run replaced by a recorder): per scene the same script, the same
    flags / yacs keys in the same order, equal literal values, and every path value naming the same file (the reference's value,
    stripped of the leading '..' components its working-directory convention needs, is a suffix of the absolute path used here)."""
    import importlib.util
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "driver_reference_commands.npz"))
    ref_calls = [c for c in json.loads(str(g["calls"])) if "export_object_detection" not in c[1][1]]      # stage 1 is out of scope
    spec = importlib.util.spec_from_file_location("our_evaluate_pipeline", os.path.join(ROOT, "evaluate_pipeline.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    for sub in ("data/scene_a/event-frames", "data/scene_b/event-frames", "object_detection/det_out/scene_a", "object_detection/det_out/scene_b"):
        os.makedirs(tmp_path / sub)
    for sc in ("scene_a", "scene_b"):
        (tmp_path / "object_detection" / "det_out" / sc / "test.json").write_text("{}")
    calls = []
    monkeypatch.setattr(drv, "run", lambda cmd, cwd: calls.append((os.path.basename(cwd), list(cmd))))
    drv.main(["--data_dir", str(tmp_path / "data"), "--detection_model_file", "det.pth", "--regression_model_file", "models/reg.pth",
              "--detection_annotations_base", str(tmp_path / "object_detection" / "det_out"),
              "--regression_annotations_base", str(tmp_path / "landmark_regression" / "reg_out"),
              "--pose_estimation_base", str(tmp_path / "pose_estimation" / "pose_out"), "--validation_annotations", "val.json",
              "--landmarks_file", "landmarks.csv", "--calibration_file_path", "calib/calibration.json",
              "--image_width", "1920", "--image_height", "1200", "--joints_count", "11"])
    assert len(calls) == len(ref_calls) == 4
    literal = {"DATASET.TEST_SET", "DATASET.TRAIN_SET", "DATASET.IMAGE_WIDTH", "DATASET.IMAGE_HEIGHT", "MODEL.NUM_JOINTS"}

    def strip(p):
        parts = [q for q in p.split("/") if q not in ("..", ".")]
        return "/".join(parts)
    for (cwd, cmd), (rcwd, rcmd) in zip(calls, ref_calls):
        assert cwd == os.path.basename(rcwd)                                  # landmark_regression / pose_estimation
        ours, ref = cmd[1:], rcmd[1:]                                         # [script, flag, value, ...] behind the interpreter
        assert ours[0] == ref[0] and len(ours) == len(ref)
        assert ours[1::2] == ref[1::2], (ours[1::2], ref[1::2])              # flags and yacs keys, in order
        for key, a, b in zip(ref[1::2], ours[2::2], ref[2::2]):
            if key in literal:
                assert a == b, (key, a, b)
            else:
                assert os.path.isabs(a) and a.endswith("/" + strip(b)), (key, a, b)


def test_cli_argparse_surfaces_cover_the_reference_clis(monkeypatch):
    """landmark_regression/tools/test.py and pose_estimation/export_predicted_poses_real.py against the argparse surfaces of the
    reference's scripts (tests/golden/cli_reference_surfaces.npz: tools/test.py:35-66, export_predicted_poses_real.py:127-148): every
    reference option exists here with the same flags, required-ness, type, nargs and default (extensions are extra, optional flags),
    and the command lines the reference's driver issues parse to the same values."""
    import argparse
    import importlib.util
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_reference_surfaces.npz"))
    surfaces, parsed, argv = (json.loads(str(g[k])) for k in ("surfaces", "parsed", "argv"))

    class Done(Exception):
        pass
    got = {}
    real_parse = argparse.ArgumentParser.parse_args

    def surface(parser):
        return [[list(a.option_strings) or [a.dest], bool(a.required), getattr(a.type, "__name__", None), a.nargs if a.nargs is None else str(a.nargs),
                 a.default if isinstance(a.default, (str, int, float, type(None))) else None] for a in parser._actions if a.dest != "help"]
    for name, path in (("tools/test.py", "landmark_regression/tools/test.py"), ("export_predicted_poses_real.py", "pose_estimation/export_predicted_poses_real.py")):
        def parse(self, args=None, namespace=None, _n=name):
            got[_n] = (surface(self), vars(real_parse(self, argv[_n])))
            raise Done()
        monkeypatch.setattr(argparse.ArgumentParser, "parse_args", parse)
        spec = importlib.util.spec_from_file_location("cli_" + name.replace("/", "_").replace(".", "_"), os.path.join(ROOT, path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        with pytest.raises(Done):
            (mod.parse_args if hasattr(mod, "parse_args") else mod.main)()
        monkeypatch.setattr(argparse.ArgumentParser, "parse_args", real_parse)
        ours, vals = got[name]
        for opt in surfaces[name]:
            assert opt in ours, "%s: reference option %s is missing or differs (have %s)" % (name, opt, ours)
        for extra in (o for o in ours if o not in surfaces[name]):
            assert extra[1] is False, "%s: an extension option must not be required: %s" % (name, extra)
        for k, v in parsed[name].items():
            assert vals[k] == v, (name, k, vals[k], v)


def test_config_defaults_equal_the_reference_default_py():
    """config._defaults() against every key and value of the reference's lib/config/default.py:17-142
    (tests/golden/config_reference_defaults.npz, dumped by importing the module under a stand-in for yacs): same tree, same
    values, same tuple-vs-list kinds -- so every YAML and KEY VAL override the reference accepts is accepted here."""
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    C = import_module("spacecraft-pose-estimation_amd.config")
    want = json.loads(str(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_reference_defaults.npz"))["defaults"]))
    cfg = C._defaults()
    missing, differs = [], []

    def walk(w, node, path):
        for k, v in w.items():
            p = path + [k]
            try:
                have = node[k] if isinstance(node, dict) else getattr(node, k)
            except (KeyError, AttributeError):
                missing.append(".".join(p)); continue
            if isinstance(v, dict) and "__tuple__" not in v:
                walk(v, have, p)
            else:
                exp = tuple(v["__tuple__"]) if isinstance(v, dict) else v
                got = have
                if isinstance(exp, tuple):
                    if not (isinstance(got, tuple) and tuple(got) == exp):
                        differs.append((".".join(p), got, exp))
                elif isinstance(exp, list):
                    if list(got) != exp:
                        differs.append((".".join(p), got, exp))
                elif got != exp or type(got) is not type(exp):
                    differs.append((".".join(p), got, exp))
    walk(want, cfg, [])
    assert not missing, "default keys of the reference that config._defaults() lacks: %s" % missing
    assert not differs, "default values that differ: %s" % differs[:10]
    # the experiment file evaluate_pipeline.py:70 names.  (a) the REFERENCE's own file content (stored as parsed data) merges into this config
    # unchanged -- training, debug and cuDNN keys included -- and every value arrives; (b) the shipped copy is that file restricted to
    # the inference path: wherever both have a key the values agree, except the three the copy deliberately neutralises.
    import yaml
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_reference_defaults.npz"))
    ref_yaml = json.loads(str(g["events_yaml"]))
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as fh:
        yaml.safe_dump(ref_yaml, fh)
    try:
        C.update_config(cfg, types.SimpleNamespace(cfg=fh.name, opts=["MODEL.NUM_JOINTS", "11", "GPUS", "(0,)"], modelDir="", logDir="", dataDir=""))
    finally:
        os.unlink(fh.name)
    assert cfg.MODEL.NUM_JOINTS == 11 and tuple(cfg.GPUS) == (0,) and cfg.WORKERS == ref_yaml["WORKERS"] == 0

    def arrived(w, node, path):
        for k, v in w.items():
            have = node[k] if isinstance(node, dict) else getattr(node, k)
            if isinstance(v, dict):
                arrived(v, have, path + [k])
            elif path + [k] not in (["MODEL", "NUM_JOINTS"], ["GPUS"]):
                assert (list(have) == list(v)) if isinstance(v, list) else (have == v or str(have) == str(v)), (".".join(path + [k]), have, v)
    arrived(ref_yaml, cfg, [])
    with open(os.path.join(ROOT, "landmark_regression", "experiments", "events", "events-config.yaml")) as fh2:
        ours = yaml.safe_load(fh2)
    neutralised = {"DEBUG.DEBUG", "TEST.MODEL_FILE", "DATA_DIR", "MODEL.EXTRA.PRETRAINED_LAYERS"}     # debug dumps off, no personal paths, load every layer

    def common(a, b, path):
        for k in a:
            if k in b:
                if isinstance(a[k], dict) and isinstance(b[k], dict):
                    common(a[k], b[k], path + [k])
                elif ".".join(path + [k]) not in neutralised:
                    assert a[k] == b[k], (".".join(path + [k]), a[k], b[k])
    common(ours, ref_yaml, [])


def _export_scene(tmp_path, g, with_frames):
    from PIL import Image
    from scipy.io import savemat
    from oracle import pnp_ref as P
    det = json.loads(str(g["det"]))
    (tmp_path / "det.json").write_text(json.dumps(det))
    savemat(tmp_path / "pred.mat", {"preds": g["preds"]})
    (tmp_path / "landmarks.csv").write_text("x,y,z\n" + "\n".join(",".join(repr(float(v)) for v in r) for r in P.LANDMARKS))
    (tmp_path / "calib.json").write_text(json.dumps({"intrinsics": {"camera_matrix": P.CAMERA_K.tolist(), "distortion_coefficients": P.CAMERA_DIST.tolist()}}))
    if with_frames:
        for im in det["images"]:
            f = tmp_path / "frames" / im["file_name"]
            f.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(np.zeros((24, 32, 3), np.uint8)).save(f)
    return det


def test_pose_export_writes_what_the_reference_main_writes(tmp_path, monkeypatch):
    """pose_export.export against the reference's own export_predicted_poses_real.py main() (:126-236), run on a scratch scene with cv2
    replaced by a recorder (tests/golden/export_reference_outputs.npz): given the poses, opencv_poses.json is the reference's file
    BYTE FOR BYTE (record order = images[] order through the id -> file_name map, "T" 3 x 1, "rotation_matrix" 3 x 3, indent 2), the
    overlay files carry the reference's names, and what the reference hands to solvePnPRansac per frame -- float32 image points and
    float64 landmarks of exactly the landmarks its threshold loop admits, SOLVEPNP_EPNP, 10 000 iterations, 15 px -- is what the
    oracle's filter selects and what the kernel's defaults are."""
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    from oracle import pnp_ref as P
    pe = import_module("spacecraft-pose-estimation_amd.pose_export")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "export_reference_outputs.npz"))
    det = _export_scene(tmp_path, g, with_frames=True)
    ref_text = str(g["json_text"])
    ref = json.loads(ref_text)
    R = np.array([r["rotation_matrix"] for r in ref]); T = np.array([r["T"] for r in ref]).reshape(-1, 3)
    monkeypatch.setattr(pe, "solve_poses", lambda preds, lm, K, dist, **kw: (R, T, np.full(len(R), 11, np.int32)))
    pe.export(str(tmp_path / "frames"), str(tmp_path / "det.json"), str(tmp_path / "pred.mat"), str(tmp_path / "landmarks.csv"),
              str(tmp_path / "calib.json"), str(tmp_path / "out"), overlay=True)
    assert (tmp_path / "out" / "opencv_poses.json").read_text() == ref_text
    assert sorted(f for f in os.listdir(tmp_path / "out") if f.endswith(".jpg")) == sorted(str(v) for v in g["overlays"])
    # the call contract
    n = int(g["ncalls"])
    assert n == len(det["images"]) and set(g["flags"].tolist()) == {1} and set(g["iters"].tolist()) == {10000} and set(g["err"].tolist()) == {15.0}
    assert set(g["img_dtype"].tolist()) == {"float32"} and set(g["obj_dtype"].tolist()) == {"float64"} and json.loads(str(g["extra"])) == [[]] * n
    assert np.array_equal(g["K"], P.CAMERA_K) and np.array_equal(g["dist"], P.CAMERA_DIST)
    pos = 0
    for i in range(n):
        m = P.conf_mask(g["preds"][i, :, 2])
        k = int(g["npts"][i])
        assert m.sum() == k and np.array_equal(g["img_points"][pos:pos + k], g["preds"][i, m, :2])
        # (landmarks as pd.read_csv parses them: one float64 ulp off the literals in places, identical as the float32 values OpenCV computes with)
        assert np.array_equal(g["obj_points"][pos:pos + k].astype(np.float32), P.LANDMARKS[m].astype(np.float32))
        pos += k
    import inspect
    sig = inspect.signature(import_module("spacecraft-pose-estimation_amd.ops").pnp_epnp_ransac).parameters
    assert sig["max_iters"].default == 10000 and sig["reproj_err"].default == 15.0 and sig["conf_thr0"].default == 0.95 and sig["min_pts"].default == 15


@pytest.mark.parametrize("rgb", [True, False])
def test_events_dataset_equals_the_reference_classes_item_for_item(tmp_path, rgb):
    """EventsDataset (db construction, __getitem__ in eval mode) against the reference's own classes run on the same scratch scene
    (tests/golden/dataset_item_reference_outputs.npz, make_golden.py: dataset_item_vectors -- lib/dataset/events.py:24-92,
    lib/dataset/JointsDataset.py:120-229; cv2.warpAffine answered by utils.transforms.warp_affine_bilinear, the one unpinned piece):
    db records in annotations[] order (non-monotonic ids, an image used twice, file names with a directory part), centre / scale,
    joints and visibility, the crop, the joints mapped into it (only the visible ones), targets, weights and the meta dict -- bit for bit,
    for COLOR_RGB on and off; and the device-crop path returns the window of the same frame with the same meta."""
    import importlib.util
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    spec = importlib.util.spec_from_file_location("make_golden_for_scene", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)            # only its scene builder is used (this repository's code; nothing of /root/reference is touched on import)
    n = mg.scratch_coco_scene(str(tmp_path), np.random.default_rng(17))
    g = np.load(os.path.join(ROOT, "tests", "golden", "dataset_item_reference_outputs.npz"))
    C = import_module("spacecraft-pose-estimation_amd.config"); D = import_module("spacecraft-pose-estimation_amd.dataset")
    cfg = C._defaults()
    cfg.defrost() if hasattr(cfg, "defrost") else None
    cfg.DATASET.DATA_FORMAT = "png"; cfg.DATASET.COLOR_RGB = rgb; cfg.DATASET.IMAGE_WIDTH = 160; cfg.DATASET.IMAGE_HEIGHT = 120
    cfg.MODEL.IMAGE_SIZE = [64, 48]; cfg.MODEL.HEATMAP_SIZE = [16, 12]; cfg.MODEL.SIGMA = 2; cfg.MODEL.NUM_JOINTS = 11; cfg.MODEL.TARGET_TYPE = "gaussian"
    ds = D.EventsDataset(cfg, str(tmp_path / "ann"), str(tmp_path / "frames"), "test", False, None)
    tag = "rgb%d/" % rgb
    assert len(ds) == int(g[tag + "len"]) == n
    for i in range(n):
        rec = ds.db[i]
        assert os.path.relpath(rec["image"], tmp_path) == str(g[tag + "%d/db_image" % i])
        for k in ("center", "scale", "joints_3d", "joints_3d_vis"):
            want = g[tag + "%d/db_%s" % (i, k)]
            assert np.asarray(rec[k]).dtype == want.dtype and np.array_equal(rec[k], want), (i, k)
        assert np.array_equal(np.array([rec["box_w"], rec["box_h"]]), g[tag + "%d/db_box" % i])
        inp, target, weight, meta = ds[i]
        assert np.array_equal(np.asarray(inp), g[tag + "%d/input" % i])
        assert np.array_equal(target.numpy(), g[tag + "%d/target" % i]) and np.array_equal(weight.numpy(), g[tag + "%d/weight" % i])
        for k in ("joints", "joints_vis", "center", "scale"):
            assert np.array_equal(np.asarray(meta[k]), g[tag + "%d/meta_%s" % (i, k)]), (i, k)
        misc = json.loads(str(g[tag + "%d/meta_misc" % i]))
        assert [os.path.relpath(meta["image"], tmp_path), meta["filename"], meta["imgnum"], meta["rotation"], meta["score"]] == misc
    # device-crop path: the window of the frame, same meta, and the affine the crop above was cut with
    ds.device_crop = True
    T = import_module("spacecraft-pose-estimation_amd.utils.transforms")
    for i in range(n):
        win, target, weight, meta = ds[i]
        rx, ry, rw, rh = [int(v) for v in meta["roi"]]
        assert tuple(win.shape) == (rh, rw, 3) and np.array_equal(np.asarray(meta["center"]), g[tag + "%d/meta_center" % i])
        full = np.zeros((int(meta["frame_hw"][0]), int(meta["frame_hw"][1]), 3), np.uint8)
        full[ry:ry + rh, rx:rx + rw] = win.numpy()
        assert np.array_equal(T.warp_affine_bilinear(full, meta["trans"], (64, 48)), g[tag + "%d/input" % i])      # every tap of the crop lies in the window


def test_dumps_poses_is_json_dumps_indent_2_byte_for_byte():
    """pose_export.dumps_poses (the hand-assembled writer of opencv_poses.json) against json.dumps(poses, indent=2), what
    export_predicted_poses_real.py:235-236 calls: names that need escaping, signed zeros, denormals, values over 16 decades, the
    optional status field, the empty list."""
    import numpy as np
    import scpose  # noqa: F401
    from importlib import import_module
    pe = import_module("spacecraft-pose-estimation_amd.pose_export")
    rng = np.random.default_rng(0)
    poses = [{"image_name": "scene/a \"q\" é %d.png" % i, "T": rng.standard_normal((3, 1)).tolist(),
              "rotation_matrix": (rng.standard_normal((3, 3)) * 10.0 ** float(rng.integers(-8, 8))).tolist()} for i in range(200)]
    poses[3]["T"] = [[0.0], [-0.0], [1e-320]]
    poses[4]["rotation_matrix"] = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    # a degenerate frame (non-finite pose): json.dumps writes NaN / Infinity / -Infinity, and the file must stay loadable
    poses[5]["T"] = [[float("nan")], [float("inf")], [float("-inf")]]
    poses[6]["rotation_matrix"] = [[float("nan"), 0.0, float("-inf")], [0.0, float("inf"), 0.0], [0.0, 0.0, 1.0]]
    assert pe.dumps_poses(poses) == json.dumps(poses, indent=2)
    back = json.loads(pe.dumps_poses(poses))
    assert back[5]["T"][1] == [float("inf")] and back[5]["T"][0][0] != back[5]["T"][0][0] and back[7] == poses[7]
    with_status = [dict(p, status=i - 3) for i, p in enumerate(poses)]
    assert pe.dumps_poses(with_status) == json.dumps(with_status, indent=2)
    assert pe.dumps_poses([]) == json.dumps([], indent=2)


def test_loader_workers_start_with_the_package_imported(tmp_path):
    """parallel.loader_worker_context: DataLoader workers are forked from a fork server that has torch AND this package
    imported already.  CPython 3.10's fork server imports its preload list without the parent's sys.path, so the repository root has
    to be on PYTHONPATH for it.  The probe's target is the builtin exec (unpickling it imports nothing), so what it reports is what the
    worker had at start."""
    import scpose  # noqa: F401
    from importlib import import_module
    par = import_module("spacecraft-pose-estimation_amd.parallel")
    ctx = par.loader_worker_context(2)
    assert ROOT in os.environ["PYTHONPATH"].split(os.pathsep)
    out = tmp_path / "mods.json"
    code = ("import sys, json; json.dump({k: k in sys.modules for k in ('torch', 'numpy', 'PIL.Image', 'scpose', "
            "'spacecraft-pose-estimation_amd')}, open(%r, 'w'))" % str(out))
    p = ctx.Process(target=exec, args=(code,))
    p.start(); p.join(120)
    assert p.exitcode == 0
    mods = json.loads(out.read_text())
    assert all(mods.values()), mods
    assert par.loader_worker_context(0) is None
    assert par.auto_workers(100, 0) == 0 and par.auto_workers(5000, 3) == 3 and par.auto_workers(5000, 0, keep=True) == 0
    assert par.auto_workers(5000, 0) == max(1, min(32, (os.cpu_count() or 1) // 4))


@pytest.mark.parametrize("loader,engine,total", [(5, 2, 23), (16, 256, 1000), (7, 7, 21), (3, 10, 31), (4, 6, 4), (256, 16, 600)])
def test_engine_batch_coalescer_keeps_order_and_sizes(loader, engine, total):
    """core.function._Coalescer on CPU tensors (no engine: the step echoes its frames): every engine call but the last gets exactly `engine`
    frames, loader batches that straddle engine batches are sliced, nothing is lost or reordered, centres / scales travel with their
    frames -- whatever the ratio of the two batch sizes (the GPU test compares real rows, tests/test_gpu_validate_golden.py)."""
    import scpose  # noqa: F401
    from importlib import import_module
    fn = import_module("spacecraft-pose-estimation_amd.core.function")
    calls = []

    def step(x, c, s):
        assert x.shape[0] == c.shape[0] == s.shape[0] and torch.equal(c[:, 0], x[:, 0]) and torch.equal(s[:, 1], x[:, 0] * 2)
        calls.append(int(x.shape[0]))
        return x.clone()
    co = fn._Coalescer(engine, step)
    out = []
    for i in range(0, total, loader):
        ids = torch.arange(i, min(i + loader, total), dtype=torch.float32)
        x = ids.view(-1, 1).repeat(1, 3)
        out.extend(co.push(x, torch.stack([ids, ids], 1), torch.stack([ids, ids * 2], 1)))
    out.extend(co.drain(True))
    got = torch.cat(out)[:, 0]
    assert torch.equal(got, torch.arange(total, dtype=torch.float32))
    assert all(n == engine for n in calls[:-1]) and 0 < calls[-1] <= engine and sum(calls) == total
    assert co.n == 0 and co.q == [] and co.drain(True) == []


def test_packed_loader_batches_equal_the_plain_ones(tmp_path):
    """dataset.collate_device_crop_packed / unpack_device_crop_batch (round 6): a device-crop batch crosses the worker -> main-process
    boundary as TWO tensors (the frame windows and one blob of everything small) instead of fifteen, because every tensor costs the
    feeding thread an authenticated descriptor hand-over.  The 4-tuple rebuilt from the blob must equal collate_device_crop's, field
    for field, dtype for dtype; and the loader the CLIs build (parallel.valid_loader: worker processes from the pre-loaded fork server)
    must yield the same batches as the dataset iterated in this process."""
    import scpose  # noqa: F401
    from importlib import import_module
    from PIL import Image
    P = "spacecraft-pose-estimation_amd"
    dataset = import_module(P + ".dataset"); config_mod = import_module(P + ".config"); par = import_module(P + ".parallel")
    T = import_module(P + ".utils.transforms")
    rng = np.random.default_rng(2)
    (tmp_path / "frames").mkdir(); (tmp_path / "data").mkdir()
    images, anns = [], []
    for i in range(11):
        Image.fromarray(rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)).save(tmp_path / "frames" / ("f%02d.png" % i))
        images.append({"id": i + 1, "file_name": "f%02d.png" % i, "width": 160, "height": 120})
        anns.append({"image_id": i + 1, "bbox": [10.0 + i, 8.0, 90.0, 70.0 + i], "keypoints": [2.0] * 33, "id": i, "category_id": 1})
    (tmp_path / "data" / "real_test.json").write_text(json.dumps({"images": images, "annotations": anns}))
    cfg = config_mod._defaults()
    config_mod.update_config(cfg, types.SimpleNamespace(cfg=os.path.join(ROOT, "landmark_regression", "experiments", "events", "events-config.yaml"),
                                                        opts=["DATA_DIR", str(tmp_path / "frames"), "DATASET.ROOT", str(tmp_path / "data"), "DATASET.TEST_SET", "test",
                                                              "MODEL.NUM_JOINTS", "11", "OUTPUT_DIR", str(tmp_path / "o"), "LOG_DIR", str(tmp_path / "l")], modelDir="", logDir="", dataDir=""))
    ds = dataset.EventsDataset(cfg, cfg.DATASET.ROOT, cfg.DATA_DIR, "test", False, T.Compose([T.ToTensor()]))
    ds.device_crop = True; ds.want_target = False

    def same(x, y):
        if torch.is_tensor(x):
            return torch.is_tensor(y) and x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)
        if isinstance(x, dict):
            return x.keys() == y.keys() and all(same(x[k], y[k]) for k in x)
        if isinstance(x, (list, tuple)):
            return len(x) == len(y) and all(same(p, q) for p, q in zip(x, y))
        return x == y
    items = [ds[i] for i in range(11)]
    plain = ds.collate_device_crop(items[:4])
    packed = ds.collate_device_crop_packed(items[:4])
    assert sum(torch.is_tensor(v) for v in packed.values()) == 2            # the windows and the blob
    assert same(plain, ds.unpack_device_crop_batch(packed))
    ld = par.valid_loader(ds, 0, len(ds), 1, 4, 2, True)                    # two worker processes, batches of 4 + a ragged one of 3
    got = list(ld)
    assert len(ld) == 3 and len(got) == 3
    for k, b in enumerate(got):
        assert same(ds.collate_device_crop(items[4 * k:4 * k + 4]), b), k
    assert same(list(par.valid_loader(ds, 0, len(ds), 1, 4, 0, True)), got)   # no workers: the plain collate, same batches
    # the data set pickles its record list once and hands every further worker the cached bytes (a DataLoader pickles it per worker)
    import pickle
    one = pickle.dumps(ds)
    assert ds._db_blob[1] is pickle.loads(pickle.dumps(ds.__getstate__()))["_db_pickled"] or len(ds._db_blob[1]) > 0
    back = pickle.loads(one)
    assert len(back) == len(ds) and back.db[3]["image"] == ds.db[3]["image"] and np.array_equal(back.db[9]["center"], ds.db[9]["center"]) and back.device_crop
    assert same(back.collate_device_crop([back[i] for i in range(4)]), plain)
