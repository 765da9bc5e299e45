"""validate(): the evaluation loop of landmark_regression/lib/core/function.py:318-459 on the HIP path.

Same signature and outputs (all_preds (N,J,3) f32 [x_img,y_img,maxval], all_boxes (N,6),
dataset.evaluate -> <pred_file_name>.mat).  Differences in HOW, not WHAT:
  * the forward is the HIP engine; heatmaps never leave the device -- decode runs there
    (the reference copies the full heatmap tensor to the host twice per batch, :376/:390);
  * with WORLD_SIZE > 1 every rank evaluates its contiguous shard of the dataset and the
    (N,J,3)/(N,6) rows are all-gathered (parallel.py) -- replaces DataParallel;
  * loss / PCK are logged like the reference (they do not affect any output).

validate_cv(): the ensemble loop of lib/core/function.py:500-592 (tools/test_cv_ensemble.py): the heat-maps of
up to six models are summed and divided by their number on the device (scpose_heatmap_accumulate), then decoded
exactly like validate() (no flip test there, as in the reference).
"""
import logging
import os
import time

import numpy as np
import torch

from .. import ops, parallel
from .evaluate import accuracy
from .inference import get_final_preds_device

logger = logging.getLogger(__name__)


class AverageMeter:
    """Running mean weighted by sample count; `.val` is the last value, `.avg` the mean so far (what the log lines print)."""

    def __init__(self):
        self.val = self.sum = self.count = self.avg = 0

    reset = __init__

    def update(self, val, n=1):
        self.val, self.sum, self.count = val, self.sum + val * n, self.count + n
        self.avg = self.sum / self.count if self.count else 0


def _print_name_value(name_value, full_arch_name):
    """The three-line markdown table lib/core/function.py:462-479 logs for a dict of metrics (arch names over 15 characters are cut to 8 + '...')."""
    arch = full_arch_name if len(full_arch_name) <= 15 else full_arch_name[:8] + "..."
    logger.info("| Arch " + " ".join("| %s" % k for k in name_value) + " |")
    logger.info("|---" * (len(name_value) + 1) + "|")
    logger.info("| " + arch + " " + " ".join("| %.3f" % v for v in name_value.values()) + " |")


ENGINE_BATCH = 256     # frames per engine launch when the loop coalesces loader batches (the batch BASELINE's metric is quoted on)


def validate(config, val_loader, val_dataset, model, criterion, output_dir, tb_log_dir, pred_file_name="pred",
             writer_dict=None, log_metrics=True, engine_batch=None):
    return _run(config, val_loader, val_dataset, [model], criterion, output_dir, pred_file_name, log_metrics,
                flip_test=bool(config.TEST.FLIP_TEST), engine_batch=engine_batch)


def validate_cv(config, val_loader, val_dataset, models, criterion, output_dir, tb_log_dir, pred_file_name,
                writer_dict=None, log_metrics=True, engine_batch=None):
    if not models:
        raise ValueError("validate_cv: no model given (none of TEST.MODEL_FILE .. MODEL_FILE6 exists?)")
    # (:500-592 logs every fifth batch whatever PRINT_FREQ says, and does not print the name_values table)
    return _run(config, val_loader, val_dataset, list(models), criterion, output_dir, pred_file_name, log_metrics,
                flip_test=False, print_freq=5, print_table=False, engine_batch=engine_batch)


def _last(outputs):
    return outputs[-1] if isinstance(outputs, (list, tuple)) else outputs


class _Coalescer:
    """Loader batches -> engine batches.  The reference's shipped YAML says TEST.BATCH_SIZE_PER_GPU: 16 (events-config.yaml:74) and
    evaluate_pipeline.py:69-79 passes it on unchanged; sixteen 384 x 384 frames fill a tenth of an MI355X.  A frame's result
    does not depend on the batch it is computed in, bit for bit (tests/test_gpu_hrnet.py: test_batch_2048_equals_eight_batches_of_256),
    so the loop queues the (device) crops of the loader's batches and runs the engine on `size` frames at a time -- always exactly
    `size`, so that the captured forward of that shape is replayed -- and on whatever is left at the end.  Rows come out in loader order."""

    def __init__(self, size, step):
        self.size, self.step, self.q, self.n = int(size), step, [], 0

    def push(self, x, c, s):
        self.q.append((x, c, s)); self.n += int(x.shape[0])
        return self.drain(False)

    def drain(self, final):
        out = []
        while self.n >= self.size or (final and self.n > 0):
            need = take = min(self.size, self.n)
            xs, cs, ss = [], [], []
            while need > 0:
                x, c, s = self.q[0]
                n = int(x.shape[0])
                if n <= need:
                    self.q.pop(0)
                else:       # a loader batch that straddles two engine batches
                    self.q[0] = (x[need:], c[need:], s[need:])
                    x, c, s, n = x[:need], c[:need], s[:need], need
                xs.append(x); cs.append(c); ss.append(s); need -= n
            self.n -= take
            one = len(xs) == 1
            out.append(self.step(xs[0] if one else torch.cat(xs, 0), cs[0] if one else torch.cat(cs, 0), ss[0] if one else torch.cat(ss, 0)))
        return out


def _run(config, val_loader, val_dataset, models, criterion, output_dir, pred_file_name, log_metrics, flip_test, print_freq=None, print_table=True,
         engine_batch=None):
    batch_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter()
    for m in models:
        m.eval()
    model = models[0]
    warned = False
    metrics = bool(log_metrics and criterion is not None)
    # The fused path (models/pose_hrnet.py: forward_decode) whenever nothing downstream needs a heat-map: one model, no flip test,
    # no loss / accuracy logging.  Same key points, bit for bit (tests/test_gpu_e2e.py).
    fast = (len(models) == 1 and not flip_test and not metrics and hasattr(model, "forward_decode"))
    if fast:
        logger.info("validate: fused forward -> key-point path (no heat-maps); pass log_metrics / a flip test to get the heat-map path")
    dev = torch.device("cuda", torch.cuda.current_device())
    dist = parallel.init() if parallel.world()[0] > 1 else None
    num_samples = len(val_dataset)
    local_preds, local_boxes, image_path = [], [], []
    post = bool(config.TEST.POST_PROCESS)

    def heatmaps(input):
        """model (or ensemble mean, :530-536; or flip-test average, :347-366) -> heat-maps on the device"""
        output = _last(model(input))
        if len(models) > 1:       # sum in model order, one division by len(models)
            output = output.clone()
            for k, other in enumerate(models[1:], start=2):
                ops.heatmap_accumulate(output, _last(other(input)), float(len(models)) if k == len(models) else 1.0)
        if flip_test:
            out_f = _last(model(input.flip(2) if input.dtype == torch.uint8 else input.flip(3)))   # x axis: NHWC crops / NCHW tensors
            # flip_back + SHIFT_HEATMAP + average (:354-366) in one device kernel, no D2H round trip
            output = ops.flip_merge(output, out_f, val_dataset.flip_pairs, config.TEST.SHIFT_HEATMAP)
        return output

    def engine_step(input, c_d, s_d):
        if fast:   # key points straight from the network's last kernel: no heat-map is written, copied or re-read
            return model.forward_decode(input, c_d, s_d, post)
        return get_final_preds_device(config, heatmaps(input), c_d, s_d)

    # Loss / PCK are per LOADER batch (the reference logs them per batch), so the metric-logging mode keeps the loader's batches;
    # every other mode coalesces them into engine batches (engine_batch = 0 turns that off)
    size = ENGINE_BATCH if engine_batch is None else int(engine_batch)
    co = _Coalescer(size, engine_step) if (size > 0 and not metrics) else None
    if co is not None:
        logger.info("validate: loader batches are coalesced into engine batches of %d frames (same rows, bit for bit)" % size)
    with torch.no_grad():
        end = time.time()
        for i, (input, target, target_weight, meta) in enumerate(val_loader):
            if isinstance(input, (list, tuple, dict)):   # dataset.device_crop: frame windows -> uint8 NHWC crops on the GPU
                size_wh = config.MODEL.IMAGE_SIZE
                input = ops.crop_warp(input, meta["trans"].numpy(), (int(size_wh[0]), int(size_wh[1])), device=dev,
                                      roi=meta["roi"].numpy() if "roi" in meta else None,
                                      frame_hw=meta["frame_hw"].numpy() if "frame_hw" in meta else None)
            else:
                input = input.to(dev, non_blocking=True)
            c = meta["center"].float()
            s = meta["scale"].float()
            num_images = int(input.shape[0])
            c_d, s_d = c.to(dev, non_blocking=True), s.to(dev, non_blocking=True)
            if co is not None:
                local_preds.extend(co.push(input, c_d, s_d))
            elif not metrics:
                local_preds.append(engine_step(input, c_d, s_d))
            else:
                output = heatmaps(input)
                if tuple(target.shape) != tuple(output.shape):
                    if not warned:    # the reference would raise inside the loss here
                        logger.warning("MODEL.HEATMAP_SIZE targets %s do not match the model's heat-maps %s: loss / accuracy "
                                       "logging is skipped", tuple(target.shape[2:]), tuple(output.shape[2:]))
                        warned = True
                else:
                    target_d = target.to(dev, non_blocking=True)
                    loss = criterion(output, target_d, target_weight.to(dev, non_blocking=True))
                    losses.update(loss.item(), num_images)
                    _, avg_acc, cnt, _ = accuracy(output, target_d)      # both arg-max passes on the device tensors
                    acc.update(avg_acc, cnt)
                local_preds.append(get_final_preds_device(config, output, c_d, s_d))
            score = meta["score"].double() if torch.is_tensor(meta["score"]) else torch.tensor(meta["score"]).double()
            boxes = torch.zeros((num_images, 6), dtype=torch.float64)
            boxes[:, 0:2] = c[:, 0:2].double(); boxes[:, 2:4] = s[:, 0:2].double()
            boxes[:, 4] = torch.prod(s * 200, 1).double(); boxes[:, 5] = score     # np.prod(s * 200, 1) on the float32 scales (:397): float32 arithmetic
            local_boxes.append(boxes)
            image_path.extend(meta["image"])
            batch_time.update(time.time() - end)
            end = time.time()
            if i % (print_freq or config.PRINT_FREQ) == 0:
                if metrics:
                    logger.info("Test: [{0}/{1}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tLoss {loss.val:.4f} ({loss.avg:.4f})\t"
                                "Accuracy {acc.val:.3f} ({acc.avg:.3f})".format(i, len(val_loader), bt=batch_time, loss=losses, acc=acc))
                else:   # nothing computed them: say so instead of printing zeros
                    logger.info("Test: [{0}/{1}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tLoss n/a\tAccuracy n/a".format(i, len(val_loader), bt=batch_time))
        if co is not None:
            local_preds.extend(co.drain(True))
        preds_d = torch.cat(local_preds, 0) if local_preds else torch.zeros((0, config.MODEL.NUM_JOINTS, 3), device=dev)
        boxes_d = (torch.cat(local_boxes, 0) if local_boxes else torch.zeros((0, 6), dtype=torch.float64)).to(dev)
        if dist is not None:
            preds_d = parallel.gather_rows(preds_d, num_samples, dist)
            boxes_d = parallel.gather_rows(boxes_d, num_samples, dist)
        all_preds = np.zeros((num_samples, config.MODEL.NUM_JOINTS, 3), dtype=np.float32)
        all_boxes = np.zeros((num_samples, 6))
        all_preds[: preds_d.shape[0]] = preds_d.cpu().numpy()
        all_boxes[: boxes_d.shape[0]] = boxes_d.cpu().numpy()
        name_values, perf_indicator = {"Null": 0}, 0
        if parallel.world()[1] == 0:
            name_values, perf_indicator = val_dataset.evaluate(config, all_preds, output_dir, pred_file_name, all_boxes,
                                                               image_path, [], [])
            model_name = config.MODEL.NAME
            for nv in (name_values if isinstance(name_values, list) else [name_values]) if print_table else []:
                _print_name_value(nv, model_name)
    return perf_indicator
