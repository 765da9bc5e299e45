"""EventsDataset: COCO-dict in, pred .mat out (landmark_regression/lib/dataset/events.py:25-125).

  _get_db      :47-91   annotations[] order defines the row order of preds
  _box2cs      :94-113  center = (x + w/2, y + h/2) f32; scale = (w/200, h/200) * 1.5 f32
  evaluate     :116-125 savemat(<output_dir>/<pred_file_name>.mat, {'preds': preds}) -- written by utils/matio.py (own Level-5 writer)
Driver quirk absorbed: stage 1 writes real_test.json while stage 2 is told
TEST_SET=test -> fall back to real_<set>.json when <set>.json is absent; and stage 3 is pointed
at pred.mat while tools/test.py writes pred_test.mat -> both are written.
"""
import json
import logging
import os

import numpy as np
from ..utils.matio import savemat

from .JointsDataset import JointsDataset

logger = logging.getLogger(__name__)


class EventsDataset(JointsDataset):
    def __init__(self, cfg, root, image_dir, image_set, is_train, transform=None, numpy_transform=None):
        super().__init__(cfg, root, image_set, is_train, transform, numpy_transform, multi_scale_target=False)
        self.DATA_DIR = image_dir
        self.num_joints = cfg.MODEL.NUM_JOINTS
        self.flip_pairs = []
        self.parent_ids = []
        self.upper_body_ids = None
        self.lower_body_ids = None
        self.image_width = cfg.DATASET.IMAGE_WIDTH
        self.image_height = cfg.DATASET.IMAGE_HEIGHT
        self.aspect_ratio = self.image_width * 1.0 / self.image_height
        self.pixel_std = 200
        self.db = self._get_db()
        logger.info("=> load {} samples".format(len(self.db)))

    def annotation_file(self):
        primary = os.path.join(self.root, self.image_set + ".json")
        if os.path.exists(primary):
            return primary
        alt = os.path.join(self.root, "real_" + self.image_set + ".json")
        if os.path.exists(alt):
            logger.info("=> {} not found, using {}".format(primary, alt))
            return alt
        return primary

    def _get_db(self):
        """One record per entry of annotations[] (that order is the row order of preds)."""
        with open(self.annotation_file()) as fh:
            coco = json.load(fh)
        file_of = {im["id"]: im["file_name"] for im in coco["images"]}
        records = []
        for ann in coco["annotations"]:
            bbox = np.asarray(ann["bbox"], dtype=np.float64).reshape(-1)
            center, scale = self._box2cs(bbox)
            kp = np.asarray(ann["keypoints"], dtype=np.float64).reshape(-1, 3)
            xy = np.zeros((self.num_joints, 3), dtype=np.float64)
            vis = np.zeros((self.num_joints, 3), dtype=np.float64)
            xy[:, :2] = kp[:, :2]
            vis[:, :2] = (kp[:, 2] - 1)[:, None]          # COCO visibility {1, 2} -> {0, 1} on both axes
            records.append({"image": os.path.join(self.DATA_DIR, file_of[ann["image_id"]]), "center": center,
                            "scale": scale, "box_w": bbox[2], "box_h": bbox[3], "joints_3d": xy,
                            "joints_3d_vis": vis, "filename": "", "imgnum": 0})
        return records

    def _box2cs(self, box):
        return self._xywh2cs(*box[:4])

    def _xywh2cs(self, x, y, w, h):
        """bbox -> (center, scale): centre of the box in float32; scale = box size / 200, enlarged 1.5x
        (the aspect-ratio fix of the COCO loader is disabled in the reference, events.py:103-106)."""
        center = np.array([x + 0.5 * w, y + 0.5 * h], dtype=np.float32)
        scale = np.array([w, h], dtype=np.float64) / self.pixel_std
        scale = scale.astype(np.float32)
        if center[0] != -1:
            scale = scale * 1.5
        return center, scale

    def evaluate(self, cfg, preds, output_dir, pred_file_name, *args, **kwargs):
        if output_dir:
            savemat(os.path.join(output_dir, "{}.mat".format(pred_file_name)), mdict={"preds": preds})
            if pred_file_name == "pred_test":    # evaluate_pipeline.py:88 reads pred.mat
                savemat(os.path.join(output_dir, "pred.mat"), mdict={"preds": preds})
        return {"Null": 0}, 0
