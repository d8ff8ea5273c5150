"""Detection post-processing, HIP-backed (mirror of the reference's
models/detection/yolox/utils/boxes.py:32-113).

``postprocess`` keeps the reference signature and side effect (the input's boxes are rewritten in place
to xyxy) but runs confidence filtering, the class argmax, the stable score sort and torchvision-semantics
batched NMS for the WHOLE batch in one kernel launch (one workgroup per image), instead of a Python loop
with one torchvision call per image.  ``postprocess_padded`` is the sync-free form used by the
pseudo-label pipeline."""
from typing import List, Optional, Tuple

import torch

from leod_amd import ops

GPU_VANILLA_LIMIT = 20000      # torchvision.ops.batched_nms: per-class loop above this many box ELEMENTS on a GPU


def postprocess_padded(prediction: torch.Tensor, num_classes: int, conf_thre=0.7, nms_thre=0.45, class_agnostic=False,
                       max_det: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (det [B,max_det,7], count [B]) on the device, no host synchronisation."""
    if prediction.dim() != 3:
        raise ValueError('prediction must be [B, A, 5+num_classes]')
    if not prediction.is_contiguous():
        raise ValueError('postprocess mutates its input in place and needs a contiguous tensor')
    return ops.postprocess_nms(prediction, num_classes, conf_thre, nms_thre, class_agnostic, max_det, GPU_VANILLA_LIMIT)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False, pad=None,
                host: bool = False) -> List:
    """prediction [B,N,5+nc] (cx,cy,w,h,obj,cls..) -> list of [n_i,7] (x1,y1,x2,y2,obj,cls_conf,cls_id) in
    score-descending order, ``pad`` where nothing survives.  ``host=True`` hands the rows over as CPU tensors copied
    in ONE transfer (for callers that convert every frame to numpy records, e.g. the validation step)."""
    if len(prediction) == 0:
        return []
    det, cnt = postprocess_padded(prediction, num_classes, conf_thre, nms_thre, class_agnostic)
    counts = ops.host_counts(cnt)              # the only host sync of the call (the API returns ragged lists)
    if host:
        det = det[:, :max(max(counts), 1)].cpu()
    return [det[i, :n] if n > 0 else pad for i, n in enumerate(counts)]


def bboxes_iou(bboxes_a, bboxes_b, xyxy=True):
    """Pairwise IoU [M,N]; used by metric code outside the hot path (modules/utils/ssod.py:219,298)."""
    if bboxes_a.shape[1] != 4 or bboxes_b.shape[1] != 4:
        raise IndexError
    if xyxy:
        tl = torch.max(bboxes_a[:, None, :2], bboxes_b[:, :2])
        br = torch.min(bboxes_a[:, None, 2:], bboxes_b[:, 2:])
        area_a = torch.prod(bboxes_a[:, 2:] - bboxes_a[:, :2], 1)
        area_b = torch.prod(bboxes_b[:, 2:] - bboxes_b[:, :2], 1)
    else:
        tl = torch.max(bboxes_a[:, None, :2] - bboxes_a[:, None, 2:] / 2, bboxes_b[:, :2] - bboxes_b[:, 2:] / 2)
        br = torch.min(bboxes_a[:, None, :2] + bboxes_a[:, None, 2:] / 2, bboxes_b[:, :2] + bboxes_b[:, 2:] / 2)
        area_a = torch.prod(bboxes_a[:, 2:], 1)
        area_b = torch.prod(bboxes_b[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)
