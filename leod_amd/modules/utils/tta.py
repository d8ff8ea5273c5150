"""TTA merge helpers with the reference's interface (modules/utils/tta.py:18-61): the boxes that several views of one
frame produced are merged by confidence filtering + batched NMS -- here ONE kernel launch for all frames."""
from typing import List

import torch as th

from leod_amd import ops
from leod_amd.models.detection.yolox.utils.boxes import GPU_VANILLA_LIMIT


def tta_postprocess_padded(rows: th.Tensor, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """rows [F, N, 7] = (xyxy, obj, cls_conf, cls_id), padding rows must carry obj < 0.  -> (det [F,N,7], cnt [F])."""
    return ops.postprocess_nms(rows, 0, conf_thre, nms_thre, class_agnostic, None, GPU_VANILLA_LIMIT)


def tta_postprocess(preds: List[th.Tensor], conf_thre: float = 0.7, nms_thre: float = 0.45, class_agnostic: bool = False,
                    pad=None) -> List[th.Tensor]:
    """preds: list of [n_i,7] (xyxy, obj_conf, cls_conf, cls_idx) -> same with NMS applied (score order)."""
    out = [pad] * len(preds)
    live = [i for i, p in enumerate(preds) if p is not None and p.shape[0] > 0]
    if not live:
        return out
    nmax = max(preds[i].shape[0] for i in live)
    dev = preds[live[0]].device
    rows = th.zeros((len(live), nmax, 7), dtype=th.float32, device=dev)
    rows[:, :, 4] = -1.0
    for j, i in enumerate(live):
        rows[j, :preds[i].shape[0]] = preds[i]
    det, cnt = tta_postprocess_padded(rows, conf_thre, nms_thre, class_agnostic)
    for j, (i, n) in enumerate(zip(live, ops.host_counts(cnt, 'tta_postprocess'))):
        if n > 0:
            out[i] = det[j, :n]
    return out
