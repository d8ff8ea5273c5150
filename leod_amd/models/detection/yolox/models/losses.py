"""Loss modules kept for API compatibility (reference: models/detection/yolox/models/losses.py).
On the HIP path the IoU / BCE / focal terms and their gradients are computed inside leod_yolox_loss
(see functions.HeadTailFn); these classes only carry the configuration."""
import torch.nn as nn


class IOUloss(nn.Module):
    def __init__(self, reduction="none", loss_type="iou"):
        super().__init__()
        if loss_type != 'iou':
            raise NotImplementedError('only the plain IoU loss (1 - iou^2) is used by the reference configs')
        self.reduction = reduction
        self.loss_type = loss_type


class FocalLoss(nn.Module):
    def __init__(self, alpha=0.25, gamma=2, reduction='none'):
        super().__init__()
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction
