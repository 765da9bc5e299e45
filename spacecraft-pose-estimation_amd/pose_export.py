"""Host side of pose_estimation/export_predicted_poses_real.py:126-236 around the HIP PnP kernel.

Inputs/outputs are the reference's files: landmarks CSV (columns x,y,z), calibration JSON
({"intrinsics": {"camera_matrix", "distortion_coefficients"}}), COCO detection JSON, pred .mat
('preds' N x J x 3); writes <output_dir>/opencv_poses.json = [{"image_name", "T" (3x1),
"rotation_matrix" (3x3)}] in images[] order (:224-226, :235-236) and, unless disabled, the
per-frame overlay JPEG (:206-233).  The per-frame cv2.solvePnPRansac loop (:177-203) becomes ONE
batched launch of csrc/pnp.hip.  Deviation (documented): frames whose solve fails get identity
R / zero T and a "status" < 0 instead of the reference's exception / garbage.
"""
import csv
import json
import os
from pathlib import Path

import numpy as np
import torch
from .utils.matio import loadmat

from . import ops, parallel


def read_landmarks(path):
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    return np.array([[float(r["x"]), float(r["y"]), float(r["z"])] for r in rows], dtype=np.float64)


def solve_poses(preds, landmarks, K, dist, device=None, **kw):
    """preds (N,J,3) float32 -> R (N,3,3), T (N,3), status (N,) as NumPy arrays (GPU batched EPnP+RANSAC).
    With WORLD_SIZE>1 the frames are sharded over ranks and the rows all-gathered."""
    if not torch.cuda.is_available():
        raise ops.nat.NativeError("PnP runs on the GPU only (no CPU fallback)")
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    ws, rank, _ = parallel.world()
    dist_pg = parallel.init() if ws > 1 else None
    n = preds.shape[0]
    lo, hi = parallel.shard_range(n, rank, ws)
    kp = torch.from_numpy(np.ascontiguousarray(preds[lo:hi], dtype=np.float32)).to(dev)
    block = torch.empty((hi - lo, 13), dtype=torch.float64, device=dev)     # [R (9), t (3), status] rows, written by the kernel itself
    ops.pnp_epnp_ransac(kp, torch.from_numpy(np.asarray(landmarks, dtype=np.float64)).to(dev),
                        torch.from_numpy(np.asarray(K, dtype=np.float64)).to(dev),
                        torch.from_numpy(np.asarray(dist, dtype=np.float64)).to(dev), rows=block, **kw)
    block = parallel.gather_rows(block, n, dist_pg).cpu().numpy()
    return block[:, :9].reshape(-1, 3, 3), block[:, 9:12], block[:, 12].astype(np.int32)


def draw_overlay(frames_dir, file_name, out_path, bbox, K, R, T, landmarks):
    """Reprojection overlay of :206-233 (pinhole K[R|T]X without distortion, green bbox, r=5 discs)."""
    from PIL import Image, ImageDraw
    src = os.path.join(frames_dir, file_name)
    if not os.path.exists(src):
        return False
    img = Image.open(src).convert("RGB")
    d = ImageDraw.Draw(img)
    pts = (K @ np.column_stack((R, T.reshape(3, 1)))) @ np.column_stack((landmarks, np.ones(len(landmarks)))).T
    pts = (pts / pts[2]).T
    x, y, w, h = [int(v) for v in bbox]
    d.rectangle([x, y, x + w, y + h], outline=(0, 255, 0), width=2)
    for px, py in pts[:, :2]:
        if np.isfinite(px) and np.isfinite(py):
            d.ellipse([int(px) - 5, int(py) - 5, int(px) + 5, int(py) + 5], fill=(0, 0, 255))   # BGR (255,0,0) = blue
    img.save(out_path, quality=95)
    return True


def export(frames_dir, detection_annotations, pose_annotations, landmarks_file, calibration_file_path, output_dir,
           overlay=True, include_status=False):
    Path(output_dir).mkdir(parents=True, exist_ok=True)
    landmarks = read_landmarks(landmarks_file)
    with open(calibration_file_path, "r") as f:
        calib = json.load(f)
    K = np.array(calib["intrinsics"]["camera_matrix"], dtype=np.float64)
    dist = np.array(calib["intrinsics"]["distortion_coefficients"], dtype=np.float64)
    with open(detection_annotations, "r") as f:
        ann = json.load(f)
    image_ids = [im["id"] for im in ann["images"]]
    names = {im["id"]: im["file_name"] for im in ann["images"]}
    preds = np.array(loadmat(pose_annotations)["preds"], dtype=np.float32)
    n = min(len(image_ids), preds.shape[0])           # zip() semantics of :174-175
    min_pts = 15                                       # :192
    R, T, status = solve_poses(preds[:n], landmarks, K, dist, min_pts=min_pts)
    poses = []
    if parallel.world()[1] != 0:
        return poses
    for i in range(n):
        name = names[image_ids[i]]
        rec = {"image_name": name, "T": T[i].reshape(3, 1).tolist(), "rotation_matrix": R[i].tolist()}
        if include_status:
            rec["status"] = int(status[i])
        poses.append(rec)
        if overlay:
            out = os.path.join(output_dir, os.path.basename(name).split(".")[0] + ".jpg")
            draw_overlay(frames_dir, name, out, ann["annotations"][i]["bbox"], K, R[i], T[i], landmarks)
    with open(os.path.join(output_dir, "opencv_poses.json"), "w") as f:
        f.write(dumps_poses(poses))
    return poses


_INF = float("inf")


def dumps_poses(poses):
    """json.dumps(poses, indent=2) (:235-236) for a list of pose records, written out directly: the standard library formats indented
    output with its pure-Python encoder (0.11 ms per record); the records have a fixed shape, so the same text -- float repr, key order,
    two-space indentation, one number per line -- is assembled by hand (tests/test_host.py compares with the reference's own file and
    with json.dumps byte for byte)."""
    if not poses:
        return "[]"

    def fr(v, _repr=float.__repr__):
        # json.dumps writes non-finite floats as NaN / Infinity / -Infinity (allow_nan=True), float.__repr__ as nan / inf / -inf,
        # which json.loads rejects: one degenerate frame must not make the whole file unreadable
        if v != v:
            return "NaN"
        if v in (_INF, -_INF):
            return "Infinity" if v > 0 else "-Infinity"
        return _repr(v)
    recs = []
    for p in poses:
        t = ",\n".join("      [\n        %s\n      ]" % fr(float(row[0])) for row in p["T"])
        r = ",\n".join("      [\n%s\n      ]" % ",\n".join("        " + fr(float(v)) for v in row) for row in p["rotation_matrix"])
        body = '    "image_name": %s,\n    "T": [\n%s\n    ],\n    "rotation_matrix": [\n%s\n    ]' % (json.dumps(p["image_name"]), t, r)
        if "status" in p:
            body += ',\n    "status": %d' % p["status"]
        recs.append("  {\n" + body + "\n  }")
    return "[\n" + ",\n".join(recs) + "\n]"
