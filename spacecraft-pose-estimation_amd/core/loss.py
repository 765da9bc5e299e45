"""JointsMSELoss (landmark_regression/lib/core/loss.py:15-39): only logged by validate()."""
import torch.nn as nn


class JointsMSELoss(nn.Module):
    def __init__(self, use_target_weight):
        super().__init__()
        self.criterion = nn.MSELoss(reduction="mean")
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight):
        b, j = output.size(0), output.size(1)
        pred = output.reshape((b, j, -1)).split(1, 1)
        gt = target.reshape((b, j, -1)).split(1, 1)
        loss = 0
        for idx in range(j):
            p, g = pred[idx].squeeze(), gt[idx].squeeze()
            if self.use_target_weight:
                loss += 0.5 * self.criterion(p.mul(target_weight[:, idx]), g.mul(target_weight[:, idx]))
            else:
                loss += 0.5 * self.criterion(p, g)
        return loss / j
