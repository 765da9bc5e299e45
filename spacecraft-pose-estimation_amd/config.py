"""yacs-compatible experiment configuration (yacs itself is not a dependency here).

Mirrors the interface of landmark_regression/lib/config/default.py: a module-level ``cfg``
CfgNode holding the reference's default tree (:17-142) and ``update_config(cfg, args)`` =
merge_from_file(args.cfg) + merge_from_list(args.opts) + --modelDir/--logDir/--dataDir
overrides + freeze (:145-172).  Semantics kept from yacs: values (from YAML and from the
KEY VAL command line) are literal_eval'ed when they parse ("(0,)" -> tuple, "11" -> int),
unknown keys are errors except under nodes created with new_allowed=True (MODEL.EXTRA),
list/tuple are interchangeable, a frozen node rejects mutation.
"""
import copy
from ast import literal_eval

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"
    NEW_ALLOWED = "__new_allowed__"

    def __init__(self, init_dict=None, new_allowed=False):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        self.__dict__[CfgNode.NEW_ALLOWED] = new_allowed
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v, new_allowed=new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def is_new_allowed(self):
        return self.__dict__[CfgNode.NEW_ALLOWED]

    def _set_frozen(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode(new_allowed=self.is_new_allowed())
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return out

    def __str__(self):
        return yaml.safe_dump(_to_plain(self), default_flow_style=None, sort_keys=True)

    __repr__ = __str__

    # merging
    def merge_from_file(self, path):
        with open(path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        _merge(loaded, self, [])

    def merge_from_other_cfg(self, other):
        _merge(other, self, [])

    def merge_from_list(self, cfg_list):
        cfg_list = list(cfg_list or [])
        if len(cfg_list) % 2 != 0:
            raise AssertionError("Override list has odd length: {}; it must be a list of pairs".format(cfg_list))
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            keys = full_key.split(".")
            d = self
            for sub in keys[:-1]:
                if sub not in d:
                    raise KeyError("Non-existent key: {}".format(full_key))
                d = d[sub]
            leaf = keys[-1]
            if leaf not in d and not d.is_new_allowed():
                raise KeyError("Non-existent key: {}".format(full_key))
            value = _decode(v)
            if leaf in d:
                value = _coerce(value, d[leaf], full_key)
            dict.__setitem__(d, leaf, value)


def _to_plain(x):
    if isinstance(x, dict):
        return {k: _to_plain(v) for k, v in x.items()}
    if isinstance(x, tuple):
        return [_to_plain(v) for v in x]
    if isinstance(x, list):
        return [_to_plain(v) for v in x]
    return x


def _decode(v):
    if isinstance(v, dict):
        return CfgNode(v, new_allowed=True)
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(replacement, original, full_key):
    ot, rt = type(original), type(replacement)
    if rt == ot or original is None or replacement is None:
        return replacement
    for a, b in ((list, tuple), (tuple, list)):
        if rt == a and ot == b:
            return b(replacement)
    if ot is float and rt is int:
        return float(replacement)
    if isinstance(original, CfgNode) and isinstance(replacement, dict):
        return CfgNode(replacement, new_allowed=original.is_new_allowed())
    raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
        ot, rt, original, replacement, full_key))


def _merge(a, b, key_list):
    for k, v_ in a.items():
        full_key = ".".join(key_list + [k])
        v = _decode(copy.deepcopy(v_))
        if k in b:
            if isinstance(b[k], CfgNode) and isinstance(v, dict):
                _merge(v, b[k], key_list + [k])
            else:
                dict.__setitem__(b, k, _coerce(v, b[k], full_key))
        elif b.is_new_allowed():
            dict.__setitem__(b, k, CfgNode(v, new_allowed=True) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)
        else:
            raise KeyError("Non-existent config key: {}".format(full_key))


CN = CfgNode


def _defaults():
    """The default tree of landmark_regression/lib/config/default.py:17-142 (keys and values)."""
    c = CN()
    c.OUTPUT_DIR = ""; c.LOG_DIR = ""; c.DATA_DIR = ""; c.DATA_DIR_ADVERSARIAL = ""
    c.GPUS = (0,); c.WORKERS = 4; c.PRINT_FREQ = 20; c.AUTO_RESUME = False; c.PIN_MEMORY = True
    c.RANK = 0; c.D_LOSS = 1; c.BETA = 0.0002
    c.CUDNN = CN({"BENCHMARK": True, "DETERMINISTIC": False, "ENABLED": True})
    c.MODEL = CN({"NAME": "pose_hrnet", "INIT_WEIGHTS": True, "PRETRAINED": "", "NUM_JOINTS": 17,
                  "TAG_PER_JOINT": True, "TARGET_TYPE": "gaussian", "MULTI_SCALE_TARGET": False,
                  "IMAGE_SIZE": [256, 256], "HEATMAP_SIZE": [64, 64], "HEATMAP_SIZE_ADVERSARIAL": [16, 16],
                  "SIGMA": 5, "SIGMA2": 4, "SIGMA3": 3, "SIGMA4": 2})
    c.MODEL.EXTRA = CN(new_allowed=True)
    c.LOSS = CN({"USE_OHKM": False, "TOPK": 8, "USE_TARGET_WEIGHT": True, "USE_DIFFERENT_JOINTS_WEIGHT": False})
    c.DATASET = CN({"ROOT": "", "ROOT_ADVERSARIAL": "", "DATASET": "mpii", "DATASET_ADVERSARIAL": "",
                    "TRAIN_SET": "train", "TRAIN_SET_ADVERSARIAL": "", "TEST_SET": "valid", "DATA_FORMAT": "jpg",
                    "IMAGE_WIDTH": 1280, "IMAGE_HEIGHT": 720, "HYBRID_JOINTS_TYPE": "", "SELECT_DATA": False,
                    "FLIP": True, "SCALE_FACTOR": 0.25, "ROT_FACTOR": 30, "PROB_HALF_BODY": 0.0,
                    "NUM_JOINTS_HALF_BODY": 8, "COLOR_RGB": False})
    c.TRAIN = CN({"LR_FACTOR": 0.1, "LR_STEP": [90, 110], "LR": 0.001, "OPTIMIZER": "adam", "MOMENTUM": 0.9,
                  "WD": 0.0001, "NESTEROV": False, "GAMMA1": 0.99, "GAMMA2": 0.0, "BEGIN_EPOCH": 0,
                  "END_EPOCH": 140, "RESUME": False, "CHECKPOINT": "", "BATCH_SIZE_PER_GPU": 32,
                  "BATCH_SIZE_PER_GPU_ADVERSARIAL_SET": 3, "SHUFFLE": True})
    c.TEST = CN({"BATCH_SIZE_PER_GPU": 32, "FLIP_TEST": False, "POST_PROCESS": False, "SHIFT_HEATMAP": False,
                 "USE_GT_BBOX": False, "IMAGE_THRE": 0.1, "NMS_THRE": 0.6, "SOFT_NMS": False, "OKS_THRE": 0.5,
                 "IN_VIS_THRE": 0.0, "COCO_BBOX_FILE": "", "BBOX_THRE": 1.0, "MODEL_FILE": "", "MODEL_FILE2": "",
                 "MODEL_FILE3": "", "MODEL_FILE4": "", "MODEL_FILE5": "", "MODEL_FILE6": ""})
    c.DEBUG = CN({"DEBUG": False, "SAVE_BATCH_IMAGES_GT": False, "SAVE_BATCH_IMAGES_PRED": False,
                  "SAVE_HEATMAPS_GT": False, "SAVE_HEATMAPS_PRED": False})
    return c


_C = _defaults()
cfg = _C


def update_config(cfg, args):
    cfg.defrost()
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    if getattr(args, "modelDir", ""):
        cfg.OUTPUT_DIR = args.modelDir
    if getattr(args, "logDir", ""):
        cfg.LOG_DIR = args.logDir
    if getattr(args, "dataDir", ""):
        cfg.DATA_DIR = args.dataDir
    cfg.freeze()
