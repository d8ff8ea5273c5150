"""Oracle: greedy NMS with torchvision-0.15 semantics (numpy fp32).  TEST INFRASTRUCTURE.

PARITY UNPINNED at this boundary: the reference calls ``torchvision.ops.batched_nms`` /
``ops.nms`` (models/detection/yolox/utils/boxes.py:67,73; modules/pseudo_labeler.py:70,76;
modules/utils/tta.py:45,51).  torchvision (pinned 0.15.x: docs/install.md:13,
environment.yml:94) is neither installed in the build container nor vendored by the
reference and no reference test holds NMS vectors, so this file restates the published
algorithm of torchvision 0.15:

* ``nms``: boxes (x1,y1,x2,y2) fp32; order = stable sort of scores, descending; walk the
  order, keep a box unless an earlier *kept* box suppresses it; box j is suppressed by kept
  box i when ``inter / (area_i + area_j - inter) > thr`` (strict), with
  ``area = (x2-x1)*(y2-y1)`` and ``inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1))``,
  every operation rounded to fp32 (no fused multiply-add).  Returns kept indices in score
  order.
* ``batched_nms``: if ``boxes.numel() > limit`` (4000 on CPU, 20000 on a GPU device) run
  ``nms`` per class and return the kept indices sorted by score (descending); otherwise the
  "coordinate trick": ``nms(boxes + idxs.to(fp32) * (boxes.max() + 1), scores)`` -- the IoUs
  are evaluated on the *offset* fp32 coordinates and that rounding is part of the behaviour.
"""
import numpy as np

F32 = np.float32


def nms(boxes, scores, iou_threshold):
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F32).reshape(-1)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1).astype(F32) * (y2 - y1).astype(F32)).astype(F32)
    # stable, descending: stable argsort of the negated key keeps equal scores in index order
    order = np.argsort(-scores, kind='stable')
    thr = F32(iou_threshold)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for pos in range(n):
        i = order[pos]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        if rest.size == 0:
            break
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(F32(0), (xx2 - xx1).astype(F32))
        h = np.maximum(F32(0), (yy2 - yy1).astype(F32))
        inter = (w * h).astype(F32)
        with np.errstate(divide='ignore', invalid='ignore'):
            union = ((areas[i] + areas[rest]).astype(F32) - inter).astype(F32)
            ovr = (inter / union).astype(F32)
        suppressed[rest[ovr > thr]] = True   # NaN > thr is False, as in C
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, iou_threshold, device_semantics='gpu'):
    """device_semantics: 'gpu' -> per-class loop above 20000 box elements (what the reference's
    CUDA runs use), 'cpu' -> above 4000."""
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F32).reshape(-1)
    idxs = np.asarray(idxs)
    if boxes.size == 0:
        return np.zeros((0,), dtype=np.int64)
    limit = 20000 if device_semantics == 'gpu' else 4000
    if boxes.size > limit:
        keep_mask = np.zeros(scores.shape[0], dtype=bool)
        for cid in np.unique(idxs):
            curr = np.nonzero(idxs == cid)[0]
            k = nms(boxes[curr], scores[curr], iou_threshold)
            keep_mask[curr[k]] = True
        keep = np.nonzero(keep_mask)[0]
        # torchvision: scores[keep].sort(descending=True) (not declared stable; we use the
        # stable order = ascending index among equal scores, which is what both ATen sort
        # back-ends produce for this size)
        return keep[np.argsort(-scores[keep], kind='stable')]
    max_coordinate = boxes.max()
    offsets = (idxs.astype(F32) * (max_coordinate + F32(1)).astype(F32)).astype(F32)
    return nms((boxes + offsets[:, None]).astype(F32), scores, iou_threshold)
