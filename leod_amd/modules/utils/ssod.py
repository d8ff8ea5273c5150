"""Pseudo-label helpers with the reference's interface (modules/utils/ssod.py): label sub-sampling
indices and ``pred2label`` / ``filter_pred_boxes`` running on the HIP filter kernel."""
from typing import Callable, List, Optional, Tuple, Union

import torch
import torch as th

from leod_amd import ops
from leod_amd.data.genx_utils.labels import ObjectLabels

DATASET2HEIGHT = {'gen1': 240, 'gen4': 720}
DATASET2WIDTH = {'gen1': 304, 'gen4': 1280}


def get_subsample_label_idx(L: int, use_every: int = -1, remove_every: int = -1) -> Tuple[int, ...]:
    assert use_every == -1 or remove_every == -1
    idx = list(range(L))
    if use_every == 1:
        return tuple(idx)
    if use_every > 0:
        use = idx[1::use_every]
    elif remove_every > 0:
        use = sorted(set(idx) - set(idx[::remove_every]))
    else:
        raise ValueError('Either use_every or remove_every must be > 0')
    if L - 1 not in use:
        use.append(L - 1)
    return tuple(use)


def frame_hw(dataset_name: str = 'gen1', downsampled_by_2: bool = False) -> Tuple[int, int]:
    h, w = DATASET2HEIGHT[dataset_name], DATASET2WIDTH[dataset_name]
    return (h // 2, w // 2) if downsampled_by_2 else (h, w)


def pred2label_padded(det: th.Tensor, cnt: th.Tensor, obj_thresh, cls_thresh, dataset_name: Optional[str] = 'gen1',
                      downsampled_by_2: bool = False):
    """Sync-free form: det [B,max_det,7] + cnt [B] (from postprocess_padded) -> labels [B,max_det,8] + counts [B]."""
    filt = dataset_name is not None
    return ops.pseudo_filter(det, cnt, obj_thresh, cls_thresh, filt, frame_hw(dataset_name, downsampled_by_2) if filt else (1, 1))


def pred2label(pred: List[th.Tensor], obj_thresh: Union[float, List[float]] = 0.9, cls_thresh: Union[float, List[float]] = 0.9,
               filter_bbox_fn: Callable = None, hw: Tuple[int, int] = (-1, -1), dataset_name: str = 'gen1',
               downsampled_by_2: bool = False) -> List[ObjectLabels]:
    """pred: list of [n_i,7] (xyxy, obj, cls_conf, cls_id) -> list of ObjectLabels (t=0, corner xywh).
    ``filter_bbox_fn`` only switches the box filters on/off (they run inside the kernel)."""
    if len(pred) == 0:
        return []
    dev = pred[0].device
    nmax = max(1, max(len(p) for p in pred))
    det = torch.zeros((len(pred), nmax, 7), dtype=torch.float32, device=dev)
    for i, p in enumerate(pred):
        det[i, :len(p)] = p
    cnt = torch.tensor([len(p) for p in pred], dtype=torch.int32, device=dev)
    lab, lcnt = ops.pseudo_filter(det, cnt, obj_thresh, cls_thresh, filter_bbox_fn is not None,
                                  frame_hw(dataset_name, downsampled_by_2))
    return [ObjectLabels(lab[i, :n], hw) for i, n in enumerate(ops.host_counts(lcnt, 'pred2label'))]


def filter_pred_boxes(boxes: th.Tensor, dataset_name: str = 'gen1', downsampled_by_2: bool = False):
    """Marker used as ``filter_bbox_fn``; kept for API compatibility (returns clamped boxes + keep mask)."""
    h, w = frame_hw(dataset_name, downsampled_by_2)
    x1, y1 = boxes[..., 0].clamp(0., w - 1.), boxes[..., 1].clamp(0., h - 1.)
    x2, y2 = boxes[..., 2].clamp(0., w - 1.), boxes[..., 3].clamp(0., h - 1.)
    bw, bh = x2 - x1, y2 - y1
    keep = (bw > 0) & (bh > 0) & (bw >= 5) & (bh >= 5) & (bw <= (9 * w) // 10)
    return th.stack([x1, y1, x2, y2], dim=-1), keep
