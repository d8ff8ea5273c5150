"""Loss modules of the YOLOX head (reference: models/detection/yolox/models/losses.py:11-85).

The training path does not call them: the IoU / BCE / focal terms and their gradients are evaluated inside ``leod_yolox_loss``
(``functions.HeadTailFn``, one launch for all six losses), and the head only reads their configuration.  ``forward`` is provided for
callers that use the modules on their own (analysis scripts, CPU checks): plain tensor arithmetic with the reference's conventions,
checked against ``oracle/head.py`` in ``tests/test_host_cpu.py``."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reduce(loss: torch.Tensor, reduction: str) -> torch.Tensor:
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    return loss


class IOUloss(nn.Module):
    """IoU losses of boxes given as (cx, cy, w, h) rows (:11-66): ``'iou'`` = 1 - IoU^2, ``'giou'`` = 1 - GIoU; optional per-box weights."""

    def __init__(self, reduction="none", loss_type="iou"):
        super().__init__()
        if loss_type not in ('iou', 'giou'):
            raise NotImplementedError(f'IOUloss: loss_type {loss_type!r}')
        self.reduction = reduction
        self.loss_type = loss_type

    def forward(self, pred: torch.Tensor, target: torch.Tensor, weights=None):
        if pred.shape[0] != target.shape[0]:
            raise ValueError('IOUloss: one target box per predicted box')
        if pred.shape[0] == 0:
            return 0.                                           # the reference's convention for an empty match set (:21-22)
        pred, target = pred.reshape(-1, 4), target.reshape(-1, 4)
        half_p, half_t = pred[:, 2:] * 0.5, target[:, 2:] * 0.5
        lo = torch.maximum(pred[:, :2] - half_p, target[:, :2] - half_t)
        hi = torch.minimum(pred[:, :2] + half_p, target[:, :2] + half_t)
        overlap = ((hi - lo).prod(dim=1)) * (lo < hi).all(dim=1).to(pred.dtype)
        union = pred[:, 2:].prod(dim=1) + target[:, 2:].prod(dim=1) - overlap
        iou = overlap / (union + 1e-16)
        if self.loss_type == 'iou':
            loss = 1.0 - iou * iou
        else:                                                   # enclosing-box penalty (:39-48)
            c_lo = torch.minimum(pred[:, :2] - half_p, target[:, :2] - half_t)
            c_hi = torch.maximum(pred[:, :2] + half_p, target[:, :2] + half_t)
            area_c = (c_hi - c_lo).prod(dim=1)
            giou = iou - (area_c - union) / area_c.clamp(1e-16)
            loss = 1.0 - giou.clamp(min=-1.0, max=1.0)
        if weights is not None and isinstance(weights, torch.Tensor) and bool((weights != 1.).any()):
            if weights.shape[0] != loss.shape[0]:
                raise ValueError('IOUloss: one weight per box')
            loss = loss * weights
        return _reduce(loss, self.reduction)


class FocalLoss(nn.Module):
    """Sigmoid focal loss on logits (:69-85; torchvision.ops.sigmoid_focal_loss semantics): alpha_t (1 - p_t)^gamma BCE."""

    def __init__(self, alpha=0.25, gamma=2, reduction='none'):
        super().__init__()
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction

    def forward(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        prob = torch.sigmoid(inputs)
        bce = F.binary_cross_entropy_with_logits(inputs, targets, reduction='none')
        p_t = torch.where(targets > 0.5, prob, 1.0 - prob) if targets.dtype == torch.bool else prob * targets + (1.0 - prob) * (1.0 - targets)
        loss = bce * (1.0 - p_t) ** self.gamma
        if self.alpha >= 0:
            loss = (self.alpha * targets + (1.0 - self.alpha) * (1.0 - targets)) * loss
        return _reduce(loss, self.reduction)
