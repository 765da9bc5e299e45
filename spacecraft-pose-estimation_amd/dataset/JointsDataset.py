"""Inference-side JointsDataset (landmark_regression/lib/dataset/JointsDataset.py:27-229, :264-332).

__getitem__ returns the reference's tuple (input, target, target_weight, meta) for is_train=False:
image read -> optional BGR/RGB handling -> get_affine_transform(c, s, 0, IMAGE_SIZE) -> bilinear
warp to the crop -> transform (ToTensor + Normalize) -> gaussian target.  Training-time
augmentation (flip / scale / rotation / half-body, :158-177) is out of scope and rejected.
Image decoding uses PIL (cv2 is not available on this image).  The crop warp is either the NumPy
restatement of cv2.warpAffine(INTER_LINEAR) (utils/transforms.py; default, the reference's data flow)
or, with ``device_crop = True``, the HIP kernel scpose_crop_warp: then
__getitem__ returns the whole frame and its affine, ``collate_device_crop`` keeps the frames of a batch
as a list, and ``validate`` warps + normalises the batch on the GPU (bit-identical crops).
"""
import copy
import logging

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.transforms import affine_transform, get_affine_transform, warp_affine_bilinear

logger = logging.getLogger(__name__)


def _imread(path, rgb):
    """cv2.imread(path, IMREAD_COLOR | IMREAD_IGNORE_ORIENTATION) followed, when `rgb`, by the BGR -> RGB conversion of
    JointsDataset.py:149-150: HxWx3 uint8 in BGR (rgb = False) or RGB order, or None.  With COLOR_RGB the decoder's own RGB array is
    returned as it is: reversing a 1920 x 1200 frame to BGR and back cost more than decoding it (8.9 ms decode, +12 ms for the
    reversed copy, +5 ms for slicing the re-reversed view: round 5, loader throughput x 2.8 per core)."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            arr = np.asarray(im.convert("RGB"))
        return arr if rgb else arr[:, :, ::-1].copy()
    except Exception:
        return None



class JointsDataset(Dataset):
    def __init__(self, cfg, root, image_set, is_train, transform=None, numpy_transform=None, multi_scale_target=False):
        if is_train:
            raise ValueError("JointsDataset(is_train=True): training augmentation is out of scope of this build")
        self.num_joints = 0
        self.pixel_std = 200
        self.flip_pairs = []
        self.parent_ids = []
        self.is_train = is_train
        self.root = root
        self.image_set = image_set
        self.output_path = cfg.OUTPUT_DIR
        self.data_format = cfg.DATASET.DATA_FORMAT
        self.color_rgb = cfg.DATASET.COLOR_RGB
        self.target_type = cfg.MODEL.TARGET_TYPE
        self.image_size = np.array(cfg.MODEL.IMAGE_SIZE)
        self.heatmap_size = np.array(cfg.MODEL.HEATMAP_SIZE)
        self.sigma = cfg.MODEL.SIGMA
        self.use_different_joints_weight = cfg.LOSS.USE_DIFFERENT_JOINTS_WEIGHT
        self.joints_weight = 1
        self.transform = transform
        self.numpy_transform = numpy_transform
        self.device_crop = False   # True: hand whole frames to the GPU crop kernel instead of warping here
        self.want_target = True    # False: skip generate_target (the maps feed only the loss / PCK that validate() logs)
        self.db = []

    def _get_db(self):
        raise NotImplementedError

    # Pickling (a DataLoader pickles its data set once PER WORKER PROCESS, in the main process, one worker after the other): the list of
    # records -- 0.7 KB and 0.03 ms each, 1 s for a 32 768-frame scene, 19 s of start-up for 32 workers -- is serialised once and the
    # bytes are cached; every further worker costs a copy of them, and the workers unpickle in parallel.
    def __getstate__(self):
        import pickle
        st = self.__dict__.copy()
        key = (id(self.db), len(self.db))
        cache = st.pop("_db_blob", None)
        if cache is None or cache[0] != key:
            cache = (key, pickle.dumps(self.db, protocol=pickle.HIGHEST_PROTOCOL))
            self._db_blob = cache
        st["db"] = None
        st["_db_pickled"] = cache[1]
        return st

    def __setstate__(self, st):
        import pickle
        blob = st.pop("_db_pickled", None)
        self.__dict__.update(st)
        if blob is not None:
            self.db = pickle.loads(blob)

    def evaluate(self, cfg, preds, output_dir, pred_file_name, *args, **kwargs):
        raise NotImplementedError

    def __len__(self):
        return len(self.db)

    def __getitem__(self, idx):
        db_rec = copy.deepcopy(self.db[idx])
        image_file = db_rec["image"]
        if self.data_format == "zip":
            raise ValueError("DATA_FORMAT 'zip' is not supported (unused by the shipped configs)")
        data_numpy = _imread(image_file, bool(self.color_rgb))      # BGR as cv2.imread gives it, or RGB when COLOR_RGB (:149-150)
        if data_numpy is None:
            logger.error("=> fail to read {}".format(image_file))
            raise ValueError("Fail to read {}".format(image_file))
        joints = db_rec["joints_3d"]
        joints_vis = db_rec["joints_3d_vis"]
        c, s = db_rec["center"], db_rec["scale"]
        score = db_rec["score"] if "score" in db_rec else 1
        r = 0
        if self.numpy_transform:
            data_numpy = self.numpy_transform(data_numpy)
        trans = get_affine_transform(c, s, r, self.image_size)
        if self.device_crop:
            # only the window of the frame the warp can read travels (worker -> main process -> device): [x0, y0, w, h] in meta["roi"]
            from ..ops import warp_window
            fh, fw = int(data_numpy.shape[0]), int(data_numpy.shape[1])
            roi = warp_window(trans, (int(self.image_size[0]), int(self.image_size[1])), (fh, fw))
            win = np.ascontiguousarray(data_numpy[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]])
            input = torch.from_numpy(win if win.flags.writeable else win.copy())   # uint8, final channel order (the decoder's array is read-only)
        else:
            input = warp_affine_bilinear(np.ascontiguousarray(data_numpy), trans, (int(self.image_size[0]), int(self.image_size[1])))
            if self.transform:
                input = self.transform(input)
        for i in range(self.num_joints):
            if joints_vis[i, 0] > 0.0:
                joints[i, 0:2] = affine_transform(joints[i, 0:2], trans)
        if self.want_target:
            target, target_weight = self.generate_target(joints, joints_vis)
        else:
            target = np.zeros((self.num_joints, 1, 1), dtype=np.float32)
            target_weight = np.asarray(joints_vis[:, 0:1], dtype=np.float32).copy()
        meta = {"image": image_file, "filename": db_rec.get("filename", ""), "imgnum": db_rec.get("imgnum", ""),
                "joints": joints, "joints_vis": joints_vis, "center": c, "scale": s, "rotation": r, "score": score}
        if self.device_crop:
            meta["trans"] = np.asarray(trans, dtype=np.float64)
            meta["roi"] = np.asarray(roi, dtype=np.int32)
            meta["frame_hw"] = np.asarray([fh, fw], dtype=np.int32)
        return input, torch.from_numpy(target), torch.from_numpy(target_weight), meta

    @staticmethod
    def collate_device_crop(batch):
        """DataLoader collate_fn for device_crop: the frame windows (sizes differ) are packed into one flat tensor + offsets + sizes
        (ops.pack_frames: one object through the worker -> main-process queue instead of one per frame), the rest is default-collated."""
        from torch.utils.data import default_collate
        from ..ops import pack_frames
        frames = pack_frames([b[0] for b in batch])
        rest = default_collate([(b[1], b[2], b[3]) for b in batch])
        return frames, rest[0], rest[1], rest[2]

    @staticmethod
    def collate_device_crop_packed(batch):
        """collate_device_crop for the trip worker -> main process: TWO tensors per batch instead of fifteen.  Every tensor of a batch is
        its own shared-memory segment whose descriptor reaches the main process over an authenticated connection of its own
        (multiprocessing.resource_sharer): 0.7 ms each, 11 ms per batch of the reference's 16 frames, all in the one thread that feeds the
        GPU -- at 32 workers that thread, not JPEG decoding, set the pace (1 265 against 1 536 frames/s at loader batch 256, round 6).  Here
        the small tensors (offsets, sizes, targets, every numeric meta field) are copied into ONE byte blob with a plain-Python layout;
        unpack_device_crop_batch() rebuilds the same 4-tuple from views of it, bit for bit."""
        frames, target, weight, meta = JointsDataset.collate_device_crop(batch)
        layout, chunks, off = [], [], 0

        def put(path, t):
            nonlocal off
            t = t.contiguous()
            raw = t.reshape(-1).view(torch.uint8) if t.numel() else torch.zeros(0, dtype=torch.uint8)
            pad = (-off) % 8
            if pad:
                chunks.append(torch.zeros(pad, dtype=torch.uint8)); off += pad
            layout.append((path, str(t.dtype).replace("torch.", ""), tuple(t.shape), off, int(raw.numel())))
            chunks.append(raw); off += int(raw.numel())
        put(("frames", "offsets"), frames["offsets"]); put(("frames", "hw"), frames["hw"])
        put(("target",), target); put(("weight",), weight)
        plain = {}
        for k, v in meta.items():
            if torch.is_tensor(v):
                put(("meta", k), v)
            else:
                plain[k] = v          # lists of strings / numbers: pickled inline
        return {"flat": frames["flat"], "blob": torch.cat(chunks) if chunks else torch.zeros(0, dtype=torch.uint8), "layout": layout, "plain": plain}

    @staticmethod
    def unpack_device_crop_batch(packed):
        """The (frames, target, target_weight, meta) tuple collate_device_crop builds, from collate_device_crop_packed's dict: the tensors
        are views of the blob (no copy; pinned when the loader pinned the blob)."""
        blob = packed["blob"]
        frames, meta, out = {"flat": packed["flat"]}, dict(packed["plain"]), {}
        for path, dt, shape, off, nbytes in packed["layout"]:
            t = blob[off:off + nbytes].view(getattr(torch, dt)).reshape(shape) if nbytes else torch.zeros(shape, dtype=getattr(torch, dt))
            if path[0] == "frames":
                frames[path[1]] = t
            elif path[0] == "meta":
                meta[path[1]] = t
            else:
                out[path[0]] = t
        return frames, out["target"], out["weight"], meta

    def generate_target(self, joints, joints_vis):
        """Gaussian heat-maps (JointsDataset.py:264-332): value exp(-d^2 / 2 sigma^2) inside the (6 sigma + 1)^2
        window centred on the rounded joint position, 0 elsewhere; a joint whose window misses the map
        entirely gets weight 0 and an empty map.  Evaluated on the whole grid at once in float32."""
        assert self.target_type == "gaussian", "Only support gaussian map now!"
        wmap, hmap = int(self.heatmap_size[0]), int(self.heatmap_size[1])
        weight = np.ones((self.num_joints, 1), dtype=np.float32)
        weight[:, 0] = joints_vis[:, 0]
        target = np.zeros((self.num_joints, hmap, wmap), dtype=np.float32)
        reach = self.sigma * 3
        stride = self.image_size / self.heatmap_size
        gx = np.arange(wmap, dtype=np.float32)[None, :]
        gy = np.arange(hmap, dtype=np.float32)[:, None]
        denom = np.float32(2 * self.sigma ** 2)
        for k in range(self.num_joints):
            cx = int(joints[k][0] / stride[0] + 0.5)
            cy = int(joints[k][1] / stride[1] + 0.5)
            if cx - reach >= wmap or cy - reach >= hmap or cx + reach + 1 < 0 or cy + reach + 1 < 0:
                weight[k] = 0
                continue
            if weight[k] > 0.5:
                dx, dy = gx - np.float32(cx), gy - np.float32(cy)
                inside = (np.abs(dx) <= reach) & (np.abs(dy) <= reach)
                target[k] = np.where(inside, np.exp(-(dx ** 2 + dy ** 2) / denom), np.float32(0))
        if self.use_different_joints_weight:
            weight = np.multiply(weight, self.joints_weight)
        return target, weight
