"""Oracle: detection post-processing, pseudo-label filtering, label packing, voxelisation.
TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations relative to /root/reference.
"""
from typing import List, Optional, Sequence, Tuple, Union

import math
import numpy as np
import torch

from . import nms as _nms


# ------------------------------------------------------------------------------------------------
# postprocess (models/detection/yolox/utils/boxes.py:32-86)
# ------------------------------------------------------------------------------------------------
def postprocess(prediction: torch.Tensor, num_classes: int, conf_thre=0.7, nms_thre=0.45,
                class_agnostic=False, pad=None, device_semantics='gpu'):
    """prediction [B,A,5+nc] = (cx,cy,w,h,obj,cls...) -- MUTATED in place to xyxy like the
    reference (boxes.py:41-46).  Returns list of [n_i,7] = (x1,y1,x2,y2,obj,cls_conf,cls_id) in
    NMS (score-descending) order, or ``pad`` where nothing survives."""
    box = prediction.new_empty(prediction.shape)
    box[:, :, 0] = prediction[:, :, 0] - prediction[:, :, 2] / 2
    box[:, :, 1] = prediction[:, :, 1] - prediction[:, :, 3] / 2
    box[:, :, 2] = prediction[:, :, 0] + prediction[:, :, 2] / 2
    box[:, :, 3] = prediction[:, :, 1] + prediction[:, :, 3] / 2
    prediction[:, :, :4] = box[:, :, :4]
    out = [pad] * len(prediction)
    for i, ip in enumerate(prediction):
        if not ip.size(0):
            continue
        class_conf, class_pred = torch.max(ip[:, 5:5 + num_classes], 1, keepdim=True)
        mask = (ip[:, 4] * class_conf.squeeze(1) >= conf_thre)
        det = torch.cat((ip[:, :5], class_conf, class_pred.float()), 1)[mask]
        if not det.size(0):
            continue
        d = det.detach().cpu().numpy()
        scores = (d[:, 4] * d[:, 5]).astype(np.float32)
        if class_agnostic:
            keep = _nms.nms(d[:, :4], scores, nms_thre)
        else:
            keep = _nms.batched_nms(d[:, :4], scores, d[:, 6], nms_thre, device_semantics)
        out[i] = det[torch.from_numpy(keep)]
    return out


def tta_postprocess(preds: List[torch.Tensor], conf_thre=0.7, nms_thre=0.45, class_agnostic=False,
                    pad=None, device_semantics='gpu'):
    """Tensor flavour, modules/utils/tta.py:18-61: merge the boxes of several TTA views of one frame.
    preds: list of [n,7] = (xyxy, obj, cls_conf, cls_id)."""
    out = [pad] * len(preds)
    for i, p in enumerate(preds):
        if not p.size(0):
            continue
        det = p[(p[:, 4] * p[:, 5]) >= conf_thre]
        if not det.size(0):
            continue
        d = det.detach().cpu().numpy()
        scores = (d[:, 4] * d[:, 5]).astype(np.float32)
        keep = _nms.nms(d[:, :4], scores, nms_thre) if class_agnostic else \
            _nms.batched_nms(d[:, :4], scores, d[:, 6], nms_thre, device_semantics)
        out[i] = det[torch.from_numpy(keep)]
    return out


# ------------------------------------------------------------------------------------------------
# pseudo-label filters (modules/utils/ssod.py:40-188)
# ------------------------------------------------------------------------------------------------
DATASET2HEIGHT = {'gen1': 240, 'gen4': 720}
DATASET2WIDTH = {'gen1': 304, 'gen4': 1280}


def filter_pred_boxes(boxes, dataset_name='gen1', downsampled_by_2=False):
    """ssod.py:113-133: clamp to the field of view, drop w/h <= 0 after the clamp, drop sides < 5,
    drop width > 0.9 * frame width.  boxes [N,4] xyxy -> (clamped boxes, keep mask)."""
    fh, fw = DATASET2HEIGHT[dataset_name], DATASET2WIDTH[dataset_name]
    if downsampled_by_2:
        fh //= 2
        fw //= 2
    x1 = torch.clamp(boxes[..., 0], min=0., max=fw - 1.)
    y1 = torch.clamp(boxes[..., 1], min=0., max=fh - 1.)
    x2 = torch.clamp(boxes[..., 2], min=0., max=fw - 1.)
    y2 = torch.clamp(boxes[..., 3], min=0., max=fh - 1.)
    w, h = x2 - x1, y2 - y1
    keep = (w > 0) & (h > 0)
    keep &= (w >= 5) & (h >= 5)
    keep &= (w <= (9 * fw) // 10)
    return torch.stack([x1, y1, x2, y2], dim=-1), keep


def filter_w_thresh(scores, class_ids, thresh: Union[float, Sequence[float]]):
    """ssod.py:136-144."""
    if isinstance(thresh, float):
        return scores > thresh
    m = torch.zeros_like(scores, dtype=torch.bool)
    for i, t in enumerate(thresh):
        m |= (class_ids == i) & (scores > t)
    return m


def pred2label(pred: List[torch.Tensor], obj_thresh=0.9, cls_thresh=0.9, dataset_name='gen1',
               downsampled_by_2=False, filter_boxes=True):
    """ssod.py:147-188 for a flat list.  pred: list of [n_i,7] (xyxy,obj,cls_conf,cls_id) ->
    list of [m_i,8] = (t=0, x, y, w, h, cls_id, cls_conf, obj) with corner xy (ObjectLabels layout,
    data/genx_utils/labels.py:27-36)."""
    lens = [len(p) for p in pred]
    allp = torch.cat(pred, dim=0).clone()
    obj, clsc, cid = allp[:, 4], allp[:, 5], allp[:, 6]
    sel = filter_w_thresh(obj, cid, obj_thresh) & filter_w_thresh(clsc, cid, cls_thresh)
    if filter_boxes:
        nb, keep = filter_pred_boxes(allp[:, :4], dataset_name, downsampled_by_2)
        sel &= keep
        allp[:, :4] = nb
    out, s = [], 0
    for n in lens:
        lab = allp[s:s + n][sel[s:s + n]]
        s += n
        xywh = torch.stack([lab[:, 0], lab[:, 1], lab[:, 2] - lab[:, 0], lab[:, 3] - lab[:, 1]], -1)
        out.append(torch.cat([torch.zeros_like(xywh[:, :1]), xywh, lab[:, 6:7], lab[:, 5:6], lab[:, 4:5]], 1))
    return out


# ------------------------------------------------------------------------------------------------
# labels (data/genx_utils/labels.py)
# ------------------------------------------------------------------------------------------------
def labels_to_yolox(lab8: torch.Tensor) -> torch.Tensor:
    """get_labels_as_tensors('yolox'), labels.py:543-560: [n,8] (t,x,y,w,h,cls,cls_conf,obj) ->
    [n,7] (cls, cx, cy, w, h, obj, cls_conf)."""
    out = torch.zeros((lab8.shape[0], 7), dtype=torch.float32)
    if lab8.shape[0] == 0:
        return out
    out[:, 0] = lab8[:, 5]
    out[:, 1] = lab8[:, 1] + 0.5 * lab8[:, 3]
    out[:, 2] = lab8[:, 2] + 0.5 * lab8[:, 4]
    out[:, 3] = lab8[:, 3]
    out[:, 4] = lab8[:, 4]
    out[:, 5] = lab8[:, 7]
    out[:, 6] = lab8[:, 6]
    return out


def batched_yolox_labels(lab_list: List[torch.Tensor]) -> torch.Tensor:
    """get_labels_as_batched_tensor, labels.py:573-603: zero-pad to the max box count."""
    N = max(len(x) for x in lab_list)
    assert N > 0
    out = torch.zeros((len(lab_list), N, 7), dtype=torch.float32)
    for i, l in enumerate(lab_list):
        out[i, :len(l)] = labels_to_yolox(l)
    return out


def flip_lr_labels(lab8: torch.Tensor, width: int) -> torch.Tensor:
    """ObjectLabels.flip_lr_, labels.py:506-509: x <- W - 1 - x - w."""
    out = lab8.clone()
    if len(out):
        out[:, 1] = width - 1 - out[:, 1] - out[:, 3]
    return out


def get_subsample_label_idx(L, use_every=-1, remove_every=-1):
    """modules/utils/ssod.py:19-37."""
    assert use_every == -1 or remove_every == -1
    all_idx = list(range(L))
    if use_every == 1:
        return tuple(all_idx)
    if use_every > 0:
        use_idx = all_idx[1::use_every]
    elif remove_every > 0:
        use_idx = list(set(all_idx) - set(all_idx[::remove_every]))
    else:
        raise ValueError('Either use_every or remove_every must be > 0')
    if L - 1 not in use_idx:
        use_idx.append(L - 1)
    return tuple(use_idx)


# ------------------------------------------------------------------------------------------------
# voxelisation (data/utils/representations.py:78-123)
# ------------------------------------------------------------------------------------------------
def stacked_histogram(x, y, pol, time, bins, height, width, count_cutoff=None, fastmode=True):
    """StackedHistogram.construct: events -> uint8 [2*bins, H, W].  numpy int arrays in.
    fastmode accumulates in uint8 (wraps mod 256) then clamps to count_cutoff; otherwise int16
    accumulate, clamp, cast."""
    cutoff = 255 if count_cutoff is None else min(count_cutoff, 255)
    n = 2 * bins * height * width
    if len(x) == 0:
        return np.zeros((2 * bins, height, width), dtype=np.uint8)
    t0, t1 = int(time[0]), int(time[-1])
    # reference: (time - t0) is int64; "/ max(t1 - t0, 1)" is a true division that ATen performs in
    # fp32 (default dtype), then "* bins" in fp32, floor, clamp
    t_norm = (np.asarray(time, dtype=np.int64) - t0).astype(np.float32) / np.float32(max(t1 - t0, 1))
    t_idx = np.floor(t_norm * np.float32(bins))
    t_idx = np.minimum(t_idx, bins - 1).astype(np.int64)
    idx = np.asarray(x, np.int64) + width * np.asarray(y, np.int64) + height * width * t_idx + \
        bins * height * width * np.asarray(pol, np.int64)
    counts = np.bincount(idx, minlength=n).astype(np.int64)
    if fastmode:
        rep = (counts % 256).astype(np.uint8)
    else:
        rep = ((counts + 32768) % 65536 - 32768).astype(np.int16)   # int16 wrap
    rep = np.clip(rep, 0, cutoff).astype(np.uint8)
    return rep.reshape(2 * bins, height, width)


def mixed_density_stack(x, y, pol, time, bins, height, width, count_cutoff=None):
    """MixedDensityEventStack.construct (data/utils/representations.py:168-221): events sorted in time -> int8 [bins, H, W].
    Polarity sums (+-1) in the logarithmic time bin  floor(max(bins - log(t_norm)/log(1/2), 0)),  t_norm clamped to [1e-6, 1-1e-6];
    put_(accumulate) adds in int8 (wraps), cumsum_channel (representations.py:125-128) assigns the int64 prefix sums over the bins back
    into int8 (wraps again), then the clamp to +-count_cutoff."""
    if len(x) == 0:
        return np.zeros((bins, height, width), dtype=np.int8)
    t0, t1 = int(time[0]), int(time[-1])
    t_norm = (np.asarray(time, dtype=np.int64) - t0).astype(np.float32) / np.float32(max(t1 - t0, 1))
    t_norm = np.clip(t_norm, np.float32(1e-6), np.float32(1 - 1e-6))
    # fp32 log, fp32 division by the python scalar log(1/2), fp32 subtraction from the integer bin count
    bin_float = np.float32(bins) - np.log(t_norm, dtype=np.float32) / np.float32(math.log(1 / 2))
    t_idx = np.floor(np.maximum(bin_float, np.float32(0))).astype(np.int64)
    idx = np.asarray(x, np.int64) + width * np.asarray(y, np.int64) + height * width * t_idx
    counts = np.zeros(bins * height * width, dtype=np.int64)
    np.add.at(counts, idx, 2 * np.asarray(pol, np.int64) - 1)
    run = np.cumsum(counts.reshape(bins, height, width), axis=0)
    rep = ((run + 128) % 256 - 128).astype(np.int8)              # int8 wrap of every stage = wrap of the exact prefix sum
    if count_cutoff is not None:
        rep = np.clip(rep, -count_cutoff, count_cutoff).astype(np.int8)
    return rep
