"""create_logger with the reference's output-directory naming
(landmark_regression/lib/utils/utils.py:22-57): <OUTPUT_DIR>/<DATASET>[_<HYBRID_JOINTS_TYPE>]/<MODEL.NAME>/<cfg basename>.
evaluate_pipeline.py:88 depends on that layout; tests/golden/naming_reference_outputs.npz holds what the reference's own function creates."""
import logging
import os
import time
from pathlib import Path


def _tree_names(cfg, cfg_name):
    """(dataset directory, model directory, experiment name): ':' in the data-set name becomes '_', the experiment name is the cfg file's
    base name up to its first dot."""
    dataset = cfg.DATASET.DATASET
    if cfg.DATASET.HYBRID_JOINTS_TYPE:
        dataset += "_" + cfg.DATASET.HYBRID_JOINTS_TYPE
    return dataset.replace(":", "_"), cfg.MODEL.NAME, os.path.basename(cfg_name).split(".")[0]


def create_logger(cfg, cfg_name, phase="train"):
    dataset, model, exp = _tree_names(cfg, cfg_name)
    stamp = time.strftime("%Y-%m-%d-%H-%M")
    out_dir = Path(cfg.OUTPUT_DIR) / dataset / model / exp
    tb_dir = Path(cfg.LOG_DIR) / dataset / model / ("%s_%s" % (exp, stamp))
    for d in (out_dir, tb_dir):
        print("=> creating {}".format(d))
        d.mkdir(parents=True, exist_ok=True)
    logging.basicConfig(filename=str(out_dir / ("%s_%s_%s.log" % (exp, stamp, phase))), format="%(asctime)-15s %(message)s")
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    if not any(type(h) is logging.StreamHandler for h in root.handlers):      # one console echo, however often this is called
        root.addHandler(logging.StreamHandler())
    return root, str(out_dir), str(tb_dir)
