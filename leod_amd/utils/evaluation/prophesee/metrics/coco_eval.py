"""COCO-protocol KPIs over Prophesee box records (reference: utils/evaluation/prophesee/metrics/coco_eval.py:32-194).

The reference windows the boxes with a Python two-pointer sweep, converts every box to a COCO dictionary and runs
pycocotools' ``COCOeval`` (or detectron2's C++ ``COCOeval_opt``).  Here the windows are index ranges from
``np.searchsorted`` and the matching / accumulation runs in host C++ behind ``leod_coco_eval``
(leod_amd/csrc/coco_eval.cpp); the final averages are taken with numpy exactly as ``COCOeval.summarize`` takes them."""
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from leod_amd._lib import lib, check

OUT_KEYS = ('AP', 'AP_50', 'AP_75', 'AP_S', 'AP_M', 'AP_L')
STAT_KEYS = OUT_KEYS + ('AR_1', 'AR_10', 'AR_100', 'AR_S', 'AR_M', 'AR_L')
IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)


def _ranges_to_index(lo: np.ndarray, hi: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenation of arange(lo[i], hi[i]) for every i, and the offsets [n+1] of the pieces."""
    n = hi - lo
    off = np.zeros(len(n) + 1, dtype=np.int64)
    np.cumsum(n, out=off[1:])
    idx = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1] - lo, n)
    return idx, off


def match_times(gt_boxes: np.ndarray, dt_boxes: np.ndarray, time_tol: int):
    """One image per distinct label timestamp: the labels at that time and the detections within +-time_tol of it
    (:49-97).  Returns (gt rows, gt offsets, dt rows, dt offsets) -- windows of the detections may overlap."""
    gt_t, dt_t = gt_boxes['t'], dt_boxes['t']
    ts = np.unique(gt_t)
    g_idx, g_off = _ranges_to_index(np.searchsorted(gt_t, ts, 'left'), np.searchsorted(gt_t, ts, 'right'))
    d_idx, d_off = _ranges_to_index(np.searchsorted(dt_t, ts - time_tol, 'left'), np.searchsorted(dt_t, ts + time_tol, 'right'))
    return g_idx, g_off, d_idx, d_off


def _xywh(boxes: np.ndarray) -> np.ndarray:
    return np.stack([boxes['x'], boxes['y'], boxes['w'], boxes['h']], axis=1).astype(np.float32)


def coco_tables(gt: np.ndarray, gt_off: np.ndarray, dt: np.ndarray, dt_off: np.ndarray, n_cat: int):
    """precision [T,R,K,4,3], recall [T,K,4,3] of COCOeval.accumulate for image windows given as offset arrays."""
    n_img = len(gt_off) - 1
    assert len(dt_off) - 1 == n_img
    gb, db = np.ascontiguousarray(_xywh(gt)), np.ascontiguousarray(_xywh(dt))
    gc, dc = np.ascontiguousarray(gt['class_id'].astype(np.int32)), np.ascontiguousarray(dt['class_id'].astype(np.int32))
    ds = np.ascontiguousarray(dt['class_confidence'].astype(np.float32))
    go, do = np.ascontiguousarray(gt_off.astype(np.int32)), np.ascontiguousarray(dt_off.astype(np.int32))
    T, R = len(IOU_THRS), len(REC_THRS)
    precision = np.empty((T, R, n_cat, 4, 3), dtype=np.float64)
    recall = np.empty((T, n_cat, 4, 3), dtype=np.float64)
    p = lambda a: a.ctypes.data  # noqa: E731
    check(lib().leod_coco_eval(p(gb), p(gc), p(go), p(db), p(dc), p(ds), p(do), n_img, n_cat, p(IOU_THRS), T, p(REC_THRS), R,
                               p(precision), p(recall)), 'coco_eval')
    return precision, recall


def summarize(precision: np.ndarray, recall: np.ndarray) -> np.ndarray:
    """The 12 numbers of COCOeval.summarize: mean over the defined (> -1) entries, -1 if there are none."""
    def mean_valid(x):
        x = x[x > -1]
        return float(np.mean(x)) if x.size else -1.0

    def ap(iou=None, area=0, m=2):
        s = precision if iou is None else precision[np.where(iou == IOU_THRS)[0]]
        return mean_valid(s[:, :, :, area, m])

    def ar(area=0, m=2):
        return mean_valid(recall[:, :, area, m])

    return np.array([ap(), ap(.5), ap(.75), ap(area=1), ap(area=2), ap(area=3),
                     ar(m=0), ar(m=1), ar(m=2), ar(area=1), ar(area=2), ar(area=3)])


def evaluate_detection(gt_boxes_list, dt_boxes_list, classes: Sequence[str] = ('car', 'pedestrian'), height: int = 240,
                       width: int = 304, time_tol: int = 50000, return_aps: bool = True,
                       return_all_stats: bool = False) -> Optional[Dict[str, float]]:
    """KPIs over lists of box records (one array per recording or per labelled frame); only timestamps that carry at
    least one label are scored (:32-47).  ``height`` / ``width`` only label the COCO images in the reference and do
    not enter any number."""
    gts, g_offs, dts, d_offs = [], [np.zeros(1, np.int64)], [], [np.zeros(1, np.int64)]
    g_base = d_base = 0
    for gt_boxes, dt_boxes in zip(gt_boxes_list, dt_boxes_list):
        assert np.all(gt_boxes['t'][1:] >= gt_boxes['t'][:-1])
        assert np.all(dt_boxes['t'][1:] >= dt_boxes['t'][:-1])
        g_idx, g_off, d_idx, d_off = match_times(gt_boxes, dt_boxes, time_tol)
        gts.append(gt_boxes[g_idx])
        dts.append(dt_boxes[d_idx])
        g_offs.append(g_off[1:] + g_base)
        d_offs.append(d_off[1:] + d_base)
        g_base += g_off[-1]
        d_base += d_off[-1]
    out = {k: 0.0 for k in (STAT_KEYS if return_all_stats else OUT_KEYS)}
    if d_base == 0:                      # nothing detected yet (start of training): all zeros (:113-116)
        print('no detections for evaluation found.')
        return out if return_aps else None
    precision, recall = coco_tables(np.concatenate(gts), np.concatenate(g_offs), np.concatenate(dts), np.concatenate(d_offs),
                                    len(classes))
    stats = summarize(precision, recall)
    if not return_aps:
        return None
    return {k: float(stats[i]) for i, k in enumerate(STAT_KEYS if return_all_stats else OUT_KEYS)}
