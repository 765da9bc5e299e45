"""create_logger with the reference's output-directory naming
(landmark_regression/lib/utils/utils.py:22-57): <OUTPUT_DIR>/<DATASET>/<MODEL.NAME>/<cfg basename>.
evaluate_pipeline.py:88 depends on that layout."""
import logging
import os
import time
from pathlib import Path


def create_logger(cfg, cfg_name, phase="train"):
    root_output_dir = Path(cfg.OUTPUT_DIR)
    if not root_output_dir.exists():
        print("=> creating {}".format(root_output_dir))
        root_output_dir.mkdir(parents=True, exist_ok=True)
    dataset = cfg.DATASET.DATASET + "_" + cfg.DATASET.HYBRID_JOINTS_TYPE if cfg.DATASET.HYBRID_JOINTS_TYPE else cfg.DATASET.DATASET
    dataset = dataset.replace(":", "_")
    model = cfg.MODEL.NAME
    cfg_name = os.path.basename(cfg_name).split(".")[0]
    final_output_dir = root_output_dir / dataset / model / cfg_name
    print("=> creating {}".format(final_output_dir))
    final_output_dir.mkdir(parents=True, exist_ok=True)
    time_str = time.strftime("%Y-%m-%d-%H-%M")
    log_file = "{}_{}_{}.log".format(cfg_name, time_str, phase)
    logging.basicConfig(filename=str(final_output_dir / log_file), format="%(asctime)-15s %(message)s")
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    if not any(isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler) for h in logger.handlers):
        logging.getLogger("").addHandler(logging.StreamHandler())
    tensorboard_log_dir = Path(cfg.LOG_DIR) / dataset / model / (cfg_name + "_" + time_str)
    print("=> creating {}".format(tensorboard_log_dir))
    tensorboard_log_dir.mkdir(parents=True, exist_ok=True)
    return logger, str(final_output_dir), str(tensorboard_log_dir)
