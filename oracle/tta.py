"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the evaluation-time TTA bookkeeping of the reference, modules/utils/tta.py:64-195 (EventSeqResult):
detections of the plain / h-flipped / time-reversed / doubly flipped views of a recording are mapped back to the plain
frame (x1 <- W - 1 - x1 - w; reversed frame i -> plain frame i + offset), concatenated per labelled frame, merged by
`tta_postprocess` (oracle.postproc) when any augmented view took part, and converted to Prophesee records
(utils/evaluation/prophesee/io/box_loading.py:57-107).  Pinned by tests/golden/g16_tta_result.npz (recorded from the
reference; its torchvision NMS call is served by the same restatement as everywhere else, see oracle/nms.py)."""
from typing import Dict, List

import numpy as np
import torch

from . import postproc as op
from .synth import EVAL_BBOX_DTYPE


def to_records(labels: torch.Tensor, dets) -> (np.ndarray, np.ndarray):
    """labels [n,8] (t,x,y,w,h,cls,cls_conf,obj), dets [m,7] (xyxy,obj,cls_conf,cls) or None -> Prophesee records."""
    l = labels.numpy()
    rec = np.zeros((len(l),), dtype=EVAL_BBOX_DTYPE)
    for name, col in (('t', 0), ('x', 1), ('y', 2), ('w', 3), ('h', 4), ('class_id', 5), ('class_confidence', 6)):
        rec[name] = np.asarray(l[:, col], dtype=EVAL_BBOX_DTYPE[name])
    m = 0 if dets is None else len(dets)
    prd = np.zeros((m,), dtype=EVAL_BBOX_DTYPE)
    if m:
        d = dets.numpy()
        prd['t'] = np.ones((m,), dtype=np.int64) * np.unique(l[:, 0]).item()
        prd['x'], prd['y'], prd['w'], prd['h'] = d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]
        prd['class_id'] = np.asarray(d[:, 6], dtype=np.uint32)
        prd['class_confidence'] = d[:, 5]
    return rec, prd


def aggregate_views(views: List[dict], img_hw, conf_thre: float, nms_thre: float, tflip_offset: int = -1):
    """views as produced by oracle.synth.synth_tta_views -> (label records, detection records) per labelled frame."""
    W = img_hw[1]
    preds: Dict[int, torch.Tensor] = {}
    gts: Dict[int, torch.Tensor] = {}
    aug = False
    for v in views:
        for gt, p, idx in zip(v['gts'], v['preds'], v['ev_idx']):
            if not torch.is_tensor(gt) or len(gt) == 0:
                continue
            p = p.clone()
            if v['hflip']:
                w = p[:, 2] - p[:, 0]
                p[:, 0] = W - 1 - p[:, 0] - w
                p[:, 2] = p[:, 0] + w
            f = idx + tflip_offset if v['tflip'] else idx
            if v['hflip'] or v['tflip']:
                aug = True
            else:
                assert f not in gts
                gts[f] = gt
            preds[f] = p if f not in preds else torch.cat([preds[f], p], 0)
    frames = sorted(preds)
    assert frames == sorted(gts)
    merged = [preds[f] for f in frames]
    if aug:
        merged = op.tta_postprocess(merged, conf_thre, nms_thre)
    return [to_records(gts[f], m) for f, m in zip(frames, merged)]
