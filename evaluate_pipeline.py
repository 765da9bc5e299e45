#!/usr/bin/env python3
"""Stages 2 and 3 of the reference's evaluate_pipeline.py (:62-91) on the MI355X HIP path.

Same command line as the reference driver.  For every scene directory <data_dir>/<scene>/ it runs
    landmark_regression/tools/test.py              (HRNet + decode  -> .../EventsDataset/<MODEL.NAME>/<cfg>/pred.mat)
    pose_estimation/export_predicted_poses_real.py (EPnP + RANSAC   -> <pose_estimation_base>/<scene>/opencv_poses.json)
as child processes with the working directories, relative-path conventions and KEY VAL overrides of :69-91.
Stage 1 (object detection, :49-60) is outside this repository's scope: its result, the COCO file
<object_detection>/<detection_annotations_base>/<scene>/test.json (or real_test.json, which is what the reference's
stage 1 actually writes), must already exist; --detection_model_file and --validation_annotations are accepted and
ignored.  Path quirks of the reference that are absorbed instead of reproduced: stage 2 looks for the scenes two
levels up (:66) although it is one level below the root, and stage 3 reads pred.mat although tools/test.py writes
pred_test.mat (both are written here).

Extensions (all optional): --cfg, --regression_opts KEY VAL ..., --nproc_per_node N (frames of a scene sharded over N
GPUs, one process each), --no_overlay.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
STAGE_DIRS = {"detection": os.path.join(ROOT, "object_detection"), "regression": os.path.join(ROOT, "landmark_regression"),
              "pose": os.path.join(ROOT, "pose_estimation")}


def under(stage, path):
    """Absolute form of a path the reference interprets relative to a stage's working directory."""
    return path if os.path.isabs(path) else os.path.normpath(os.path.join(STAGE_DIRS[stage], path))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Evaluate the pose estimation pipeline (landmark regression + pose recovery stages).")
    p.add_argument("--data_dir", required=True, help="directory with one sub-directory per scene, each holding event-frames/")
    p.add_argument("--detection_model_file", default="", help="(stage 1, ignored)")
    p.add_argument("--regression_model_file", required=True, help="landmark regression state_dict, relative to landmark_regression/")
    p.add_argument("--detection_annotations_base", required=True, help="directory of the detection annotations, relative to object_detection/")
    p.add_argument("--regression_annotations_base", required=True, help="directory for the landmark regression output, relative to landmark_regression/")
    p.add_argument("--pose_estimation_base", required=True, help="directory for the pose results, relative to pose_estimation/")
    p.add_argument("--validation_annotations", default="", help="(stage 1, ignored)")
    p.add_argument("--landmarks_file", required=True, help="landmarks CSV, relative to pose_estimation/")
    p.add_argument("--calibration_file_path", required=True, help="camera calibration JSON, relative to the repository root")
    p.add_argument("--image_width", type=int, default=640)
    p.add_argument("--image_height", type=int, default=480)
    p.add_argument("--joints_count", type=int, default=24)
    p.add_argument("--cfg", default="experiments/events/events-config.yaml", help="experiment YAML, relative to landmark_regression/")
    p.add_argument("--regression_opts", nargs="*", default=[], help="extra yacs KEY VAL overrides for tools/test.py")
    p.add_argument("--nproc_per_node", type=int, default=1)
    p.add_argument("--no_overlay", action="store_true")
    return p.parse_args(argv)


def scenes_of(data_dir):
    return sorted(d for d in os.listdir(data_dir) if os.path.isdir(os.path.join(data_dir, d)))


def detection_file(det_dir):
    for name in ("test.json", "real_test.json"):
        if os.path.exists(os.path.join(det_dir, name)):
            return os.path.join(det_dir, name)
    raise SystemExit("evaluate_pipeline: no detection annotations in %s (test.json / real_test.json); stage 1 "
                     "(object detection) is not part of this build -- run the reference's detector first" % det_dir)


def output_names(cfg_path, opts):
    """(DATASET.DATASET, MODEL.NAME) as tools/test.py will see them: they name its output directory (:88)."""
    import yaml
    with open(cfg_path) as fh:
        y = yaml.safe_load(fh) or {}
    names = {"DATASET.DATASET": (y.get("DATASET") or {}).get("DATASET", "EventsDataset"),
             "MODEL.NAME": (y.get("MODEL") or {}).get("NAME", "pose_hrnet")}
    for k, v in zip(opts[0::2], opts[1::2]):
        if k in names:
            names[k] = v
    return names["DATASET.DATASET"], names["MODEL.NAME"]


def run(cmd, cwd):
    print("[evaluate_pipeline] (%s) %s" % (os.path.relpath(cwd, ROOT), " ".join(cmd)), flush=True)
    rc = subprocess.run(cmd, cwd=cwd).returncode
    if rc != 0:      # the reference ignores the exit status and fails later on the missing file
        raise SystemExit("evaluate_pipeline: stage failed with exit status %d" % rc)


def main(argv=None):
    a = parse_args(argv)
    data_dir = a.data_dir if os.path.isabs(a.data_dir) else os.path.join(ROOT, a.data_dir)
    cfg = under("regression", a.cfg)
    cfg_name = os.path.splitext(os.path.basename(cfg))[0]
    launcher = [sys.executable]
    if a.nproc_per_node > 1:
        launcher += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.nproc_per_node),
                     "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29571")]
    scenes = scenes_of(data_dir)
    if not scenes:
        raise SystemExit("evaluate_pipeline: no scene directories in %s" % data_dir)
    jobs = []
    for scene in scenes:
        frames = os.path.join(data_dir, scene, "event-frames")
        det_dir = os.path.join(under("detection", a.detection_annotations_base), scene)
        jobs.append({"scene": scene, "frames": frames, "det_dir": det_dir, "det_file": detection_file(det_dir),
                     "reg_out": os.path.join(under("regression", a.regression_annotations_base), scene),
                     "pose_out": os.path.join(under("pose", a.pose_estimation_base), scene)})
    for j in jobs:       # :62-79
        os.makedirs(j["reg_out"], exist_ok=True)
        run(launcher + ["tools/test.py", "--cfg", cfg, "DATA_DIR", j["frames"], "OUTPUT_DIR", j["reg_out"],
                        "DATASET.ROOT", j["det_dir"], "DATASET.TEST_SET", "test", "DATASET.TRAIN_SET", "synthetic_train",
                        "DATASET.IMAGE_WIDTH", str(a.image_width), "DATASET.IMAGE_HEIGHT", str(a.image_height),
                        "MODEL.NUM_JOINTS", str(a.joints_count), "TEST.MODEL_FILE", under("regression", a.regression_model_file)]
            + list(a.regression_opts), STAGE_DIRS["regression"])
    names = output_names(cfg, a.regression_opts)
    for j in jobs:       # :81-91
        pred = os.path.join(j["reg_out"], names[0], names[1], cfg_name, "pred.mat")
        run([sys.executable, "export_predicted_poses_real.py", "--frames_dir", j["frames"], "--detection_annotations", j["det_file"],
             "--pose_annotations", pred, "--landmarks_file", under("pose", a.landmarks_file),
             "--calibration_file_path", a.calibration_file_path if os.path.isabs(a.calibration_file_path)
             else os.path.join(ROOT, a.calibration_file_path), "--output_dir", j["pose_out"]]
            + (["--no_overlay"] if a.no_overlay else []), STAGE_DIRS["pose"])
    return [j["pose_out"] for j in jobs]


if __name__ == "__main__":
    main()
