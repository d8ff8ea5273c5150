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


# ---- quality of the pseudo labels on frames whose GT was withheld (modules/utils/ssod.py:192-350 of the reference) -----------------
def merge_label(gt_label: List[Optional[th.Tensor]], pseudo_label: List[Optional[th.Tensor]]):
    """GT where a frame has it, the pseudo label elsewhere (in place) -> (labels, [frame had GT])  (:192-206)."""
    assert len(gt_label) == len(pseudo_label)
    gt_mask = [lbl is not None for lbl in gt_label]
    for i, lbl in enumerate(gt_label):
        if lbl is None:
            gt_label[i] = pseudo_label[i]
    return gt_label, gt_mask


def _flat(*lists):
    """The reference's ``temporal_wrapper`` (utils/helpers.py:55-110) lets these functions take [L][B] lists as well as flat ones."""
    out = []
    for x in lists:
        if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], (list, tuple)):
            x = [e for row in x for e in row]
        out.append(x)
    return out


def _center_iou(a: th.Tensor, b: th.Tensor) -> th.Tensor:
    """IoU of (cx, cy, w, h) boxes, [M,4] x [N,4] -> [M,N] (models/detection/yolox/utils/boxes.py:89-113, xyxy=False)."""
    tl = th.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = th.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    inter = th.prod(br - tl, 2) * (tl < br).type(tl.type()).prod(dim=2)
    return inter / (th.prod(a[:, 2:], 1)[:, None] + th.prod(b[:, 2:], 1) - inter)


def evaluate_label(gt_label, pseudo_label, pred_mask, num_cls: int, prefix: str = '', all_thresh: Tuple[float, ...] = (0.25, 0.50, 0.75)):
    """Per class: the share of GT boxes matched by a pseudo box (``teacher_AR@t``) and of pseudo boxes matching a GT box (``teacher_AP@t``) at
    IoU > t, averaged over the frames that have GT boxes of the class, plus the mean box counts (:209-281).  Frames with
    ``pred_mask`` False were not predicted and do not count as misses."""
    from leod_amd.utils.evaluation.prophesee.evaluator import get_labelmap
    import numpy as np
    gt_label, pseudo_label = _flat(gt_label, pseudo_label)
    pred_mask = np.asarray(pred_mask).reshape(-1)
    assert len(gt_label) == len(pseudo_label) == len(pred_mask)
    per_cls = [[] for _ in range(num_cls)]
    n_gt = [[] for _ in range(num_cls)]
    n_pred = [[] for _ in range(num_cls)]
    nt = len(all_thresh)
    for gt, pse, is_pred in zip(gt_label, pseudo_label, pred_mask):
        if gt is None or len(gt) == 0 or not is_pred:
            continue
        g5, p5 = gt.get_labels_as_tensors()[:, :5], pse.get_labels_as_tensors()[:, :5]
        for c in range(num_cls):
            gb, pb = g5[g5[:, 0] == c, 1:], p5[p5[:, 0] == c, 1:]
            if len(gb) == 0:
                continue
            m = [0.] * (2 * nt)
            if len(pb):
                ious = _center_iou(gb, pb)
                for ti, t in enumerate(all_thresh):
                    hit = ious > t
                    m[ti], m[ti + nt] = hit.any(dim=1).float().mean(), hit.any(dim=0).float().mean()
            per_cls[c].append(m)
            n_gt[c].append(len(gb))
            n_pred[c].append(len(pb))
    log, names = {}, get_labelmap(num_cls=num_cls)
    for c in range(num_cls):
        if not per_cls[c]:
            continue
        name = names[c]
        log[f'num_{name}'] = len(per_cls[c])
        avg = th.tensor(per_cls[c]).mean(dim=0).cpu().numpy()
        for ti, t in enumerate(all_thresh):
            log[f'{prefix}teacher_AR@{int(t * 100)}_{name}'] = avg[ti]
            log[f'{prefix}teacher_AP@{int(t * 100)}_{name}'] = avg[ti + nt]
        log[f'{prefix}gt_num_{name}'] = np.array(n_gt[c]).mean()
        log[f'{prefix}pred_num_{name}'] = np.array(n_pred[c]).mean()
    return log


def get_scores_ious(gt_label, pseudo_label, pred_mask, num_cls: int, prefix: str = ''):
    """For every pseudo box on a frame with GT: its best IoU with a GT box (of its class; and over all classes under ``_all``) next to its class
    / objectness confidence -- the raw material of the score-vs-quality plots (:284-350).  -> {prefix}{true_ious|cls_scores|obj_scores}_{class}: lists."""
    from leod_amd.utils.evaluation.prophesee.evaluator import get_labelmap
    import numpy as np
    gt_label, pseudo_label = _flat(gt_label, pseudo_label)
    pred_mask = np.asarray(pred_mask).reshape(-1)
    assert len(gt_label) == len(pseudo_label) == len(pred_mask)
    ious_l = [[] for _ in range(num_cls + 1)]
    cls_l = [[] for _ in range(num_cls + 1)]
    obj_l = [[] for _ in range(num_cls + 1)]
    for gt, pse, is_pred in zip(gt_label, pseudo_label, pred_mask):
        if gt is None or len(gt) == 0 or not is_pred:
            continue
        g5, p5 = gt.get_labels_as_tensors()[:, :5], pse.get_labels_as_tensors()[:, :5]
        cc, oc = th.as_tensor(pse.class_confidence), th.as_tensor(pse.objectness)
        for i, c in enumerate(list(range(num_cls)) + [None]):
            gb, sel = (g5, th.ones(len(p5), dtype=th.bool)) if c is None else (g5[g5[:, 0] == c], p5[:, 0] == c)
            if len(gb) == 0:
                continue
            best = _center_iou(gb[:, 1:], p5[sel][:, 1:]).max(dim=0)[0]
            ious_l[i].append(best)
            cls_l[i].append(cc[sel])
            obj_l[i].append(oc[sel])
    cat = lambda xs: (th.cat(xs) if xs else th.tensor([])).cpu().numpy().tolist()  # noqa: E731
    log, names = {}, get_labelmap(num_cls=num_cls)
    for i in range(num_cls + 1):
        name = 'all' if i == num_cls else names[i]
        log[f'{prefix}true_ious_{name}'] = cat(ious_l[i])
        log[f'{prefix}cls_scores_{name}'] = cat(cls_l[i])
        log[f'{prefix}obj_scores_{name}'] = cat(obj_l[i])
    return log


def filter_w_thresh(scores: th.Tensor, class_ids: th.Tensor, thresh) -> th.Tensor:
    """scores above one threshold, or above the threshold of their class (:136-145)."""
    if isinstance(thresh, float):
        return scores > thresh
    mask = th.zeros_like(scores, dtype=th.bool)
    for i, t in enumerate(thresh):
        mask |= (class_ids == i) & (scores > t)
    return mask


class AverageMeter:
    """Running weighted mean (the reference keeps its label-quality metrics in nerv.utils.AverageMeter, pseudo_labeler.py:604-607)."""

    def __init__(self):
        self.sum, self.count = 0., 0

    def update(self, val, n: int = 1):
        self.sum += float(val) * n
        self.count += n

    @property
    def avg(self) -> float:
        return self.sum / max(self.count, 1)
