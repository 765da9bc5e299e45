"""EventsDataset: COCO-dict in, pred .mat out (landmark_regression/lib/dataset/events.py:25-125).

  _get_db      :47-91   annotations[] order defines the row order of preds
  _box2cs      :94-113  center = (x + w/2, y + h/2) f32; scale = (w/200, h/200) * 1.5 f32
  evaluate     :116-125 scipy.io.savemat(<output_dir>/<pred_file_name>.mat, {'preds': preds})
Driver quirk absorbed: stage 1 writes real_test.json while stage 2 is told
TEST_SET=test -> fall back to real_<set>.json when <set>.json is absent; and stage 3 is pointed
at pred.mat while tools/test.py writes pred_test.mat -> both are written.
"""
import json
import logging
import os

import numpy as np
from scipy.io import savemat

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
        with open(self.annotation_file()) as anno_file:
            anno = json.load(anno_file)
        gt_db = []
        image_annots = {im["id"]: im for im in anno["images"]}
        for a in anno["annotations"]:
            image_name = image_annots[a["image_id"]]["file_name"]
            box = np.array(a["bbox"]).flatten()
            c, s = self._box2cs(box)
            joints_3d = np.zeros((self.num_joints, 3), dtype=np.float64)
            joints_3d_vis = np.zeros((self.num_joints, 3), dtype=np.float64)
            jr = np.array(a["keypoints"]).reshape((-1, 3))
            joints_3d[:, 0:2] = jr[:, 0:2]
            joints_3d_vis[:, 0] = jr[:, -1] - 1      # detectron visibility -> mpii
            joints_3d_vis[:, 1] = jr[:, -1] - 1
            x, y, w, h = box[:4]
            gt_db.append({"image": os.path.join(self.DATA_DIR, image_name), "center": c, "scale": s, "box_w": w,
                          "box_h": h, "joints_3d": joints_3d, "joints_3d_vis": joints_3d_vis, "filename": "", "imgnum": 0})
        return gt_db

    def _box2cs(self, box):
        x, y, w, h = box[:4]
        return self._xywh2cs(x, y, w, h)

    def _xywh2cs(self, x, y, w, h):
        center = np.zeros((2), dtype=np.float32)
        center[0] = x + w * 0.5
        center[1] = y + h * 0.5
        scale = np.array([w * 1.0 / self.pixel_std, h * 1.0 / self.pixel_std], dtype=np.float32)
        if center[0] != -1:
            scale = scale * 1.5
        return center, scale

    def evaluate(self, cfg, preds, output_dir, pred_file_name, *args, **kwargs):
        if output_dir:
            savemat(os.path.join(output_dir, "{}.mat".format(pred_file_name)), mdict={"preds": preds})
            if pred_file_name == "pred_test":    # evaluate_pipeline.py:88 reads pred.mat
                savemat(os.path.join(output_dir, "pred.mat"), mdict={"preds": preds})
        return {"Null": 0}, 0
