"""make_golden-only stand-in for torchvision.ops.

torchvision is not installed here, so when the *reference* is imported to record golden
vectors its NMS calls are served by the oracle's restatement (oracle/nms.py, CPU
semantics: per-class loop above 4000 box elements).  Consequently the golden vectors pin
everything around NMS (score products, class argmax, thresholds, output packing, order)
but NOT torchvision's kernel itself -- "parity unpinned" at that boundary, as stated in
oracle/nms.py and DESIGN.md.
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..', '..')))
from oracle import nms as _onms  # noqa: E402


def nms(boxes, scores, iou_threshold):
    keep = _onms.nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), iou_threshold)
    return torch.from_numpy(keep).to(boxes.device)


def batched_nms(boxes, scores, idxs, iou_threshold):
    keep = _onms.batched_nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(),
                             idxs.detach().cpu().numpy(), iou_threshold, device_semantics='cpu')
    return torch.from_numpy(keep).to(boxes.device)


def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2, reduction='none'):
    # torchvision 0.15 ops/focal_loss.py semantics
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction='none')
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == 'mean':
        loss = loss.mean()
    elif reduction == 'sum':
        loss = loss.sum()
    return loss
