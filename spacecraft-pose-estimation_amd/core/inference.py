"""Heatmap decode with the reference's Python signatures, executed by the HIP decode kernel.

  get_max_preds(batch_heatmaps)                         landmark_regression/lib/core/inference.py:18-46
  get_final_preds(config, batch_heatmaps, center, scale)                                        :49-79

NumPy arrays in / NumPy arrays out like the reference (uploaded, decoded by
csrc/decode.hip, downloaded); torch device tensors are accepted too and then stay on the
device (`get_final_preds_device`, used by validate() to avoid the reference's two D2H copies
of the whole heatmap tensor per batch, lib/core/function.py:376,:390).
"""
import numpy as np
import torch

from .. import ops


def _dev():
    if not torch.cuda.is_available():
        raise ops.nat.NativeError("heatmap decode runs on the GPU only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def get_max_preds(batch_heatmaps):
    """heatmaps: numpy.ndarray([batch_size, num_joints, height, width]) -> (preds (N,J,2) f32, maxvals (N,J,1))"""
    assert isinstance(batch_heatmaps, np.ndarray), "batch_heatmaps should be numpy.ndarray"
    assert batch_heatmaps.ndim == 4, "batch_images should be 4-ndim"
    hm = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).to(_dev())
    coords, maxvals = ops.max_preds(hm)
    return coords.cpu().numpy(), maxvals.cpu().numpy()


def get_final_preds_device(config, heatmaps, center, scale):
    """Device tensors in, (N,J,3) [x_img, y_img, maxval] device tensor out."""
    return ops.decode(heatmaps, center, scale, bool(config.TEST.POST_PROCESS))


def get_final_preds(config, batch_heatmaps, center, scale):
    dev = _dev()
    hm = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).to(dev)
    c = torch.from_numpy(np.ascontiguousarray(center, dtype=np.float32)).to(dev)
    s = torch.from_numpy(np.ascontiguousarray(scale, dtype=np.float32)).to(dev)
    xyc = get_final_preds_device(config, hm, c, s).cpu().numpy()
    return xyc[:, :, 0:2].copy(), xyc[:, :, 2:3].copy()
