"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the Prophesee / COCO detection evaluation the reference runs in validation and test:

  * paper box filters                          utils/evaluation/prophesee/io/box_filtering.py:18-36
  * filter thresholds per camera               utils/evaluation/prophesee/evaluation.py:5-42
  * +-50 ms time matching, image windows       utils/evaluation/prophesee/metrics/coco_eval.py:32-97
  * conversion to COCO records                 utils/evaluation/prophesee/metrics/coco_eval.py:143-194
  * overall + per-class metric dictionary      utils/evaluation/prophesee/evaluator.py:70-110

Those are pinned by tests/golden/g15_evaluator.npz (recorded from the reference itself by tests/golden/make_golden.py).

PARITY UNPINNED for the AP computation proper: the reference hands the COCO records to `pycocotools.cocoeval.COCOeval`
(coco_eval.py:121-139; pycocotools is pinned to 2.0.6 in the reference's environment.yml, detectron2's `COCOeval_opt` is an
optional drop-in for it).  pycocotools is not installed in this image and its source is not under /root/reference, so
`coco_stats` below restates the published algorithm of COCOeval (evaluate -> accumulate -> summarize, bbox mode, no crowd
regions, useCats=1, maxDets (1,10,100), area ranges all/small/medium/large, 10 IoU and 101 recall thresholds) as plain loops,
and is pinned only by hand-checkable cases (tests/test_evaluator_cpu.py).  First thing to do on a machine that has pycocotools:
diff `coco_stats` against it.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

OUT_KEYS = ('AP', 'AP_50', 'AP_75', 'AP_S', 'AP_M', 'AP_L')
AREA_RANGES = ((0.0, 1e5 ** 2), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e5 ** 2))
MAX_DETS = (1, 10, 100)


def iou_thresholds():
    return np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)


def recall_thresholds():
    return np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)


# ---- the reference's own (pinned) part -----------------------------------------------------------------------------------
def filter_boxes(boxes: np.ndarray, skip_ts=int(5e5), min_box_diag=60, min_box_side=20) -> np.ndarray:
    """box_filtering.py:18-36."""
    w, h = boxes['w'], boxes['h']
    keep = (boxes['t'] > skip_ts) * (w ** 2 + h ** 2 >= min_box_diag ** 2) * (w >= min_box_side) * (h >= min_box_side)
    return boxes[keep]


def filter_thresholds(camera: str, downsampled_by_2: bool) -> Tuple[int, int]:
    """evaluation.py:24-33: (min_box_diag, min_box_side)."""
    diag, side = (60, 20) if camera == 'gen4' else (30, 10)
    if downsampled_by_2:
        diag, side = diag // 2, side // 2
    return diag, side


def match_times(gt: np.ndarray, dt: np.ndarray, time_tol=50000) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """coco_eval.py:49-97 as the same two-pointer sweep: one 'image' per distinct GT timestamp."""
    gts, dts = [], []
    lo_g = hi_g = lo_d = hi_d = 0
    for ts in np.unique(gt['t']):
        while lo_g < len(gt) and gt[lo_g]['t'] < ts:
            lo_g += 1
        hi_g = max(lo_g, hi_g)
        while hi_g < len(gt) and gt[hi_g]['t'] <= ts:
            hi_g += 1
        while lo_d < len(dt) and dt[lo_d]['t'] < ts - time_tol:
            lo_d += 1
        hi_d = max(lo_d, hi_d)
        while hi_d < len(dt) and dt[hi_d]['t'] <= ts + time_tol:
            hi_d += 1
        gts.append(gt[lo_g:hi_g])
        dts.append(dt[lo_d:hi_d])
    return gts, dts


def to_coco_records(gts: Sequence[np.ndarray], dts: Sequence[np.ndarray]):
    """coco_eval.py:143-194 + what COCO.loadRes adds to a bbox result (area = w*h, id, iscrowd 0).  Image ids start at 1,
    category ids are class_id + 1; areas are the float32 products the reference computes."""
    anns, res = [], []
    for k, (g, d) in enumerate(zip(gts, dts)):
        for b in g:
            anns.append(dict(image_id=k + 1, category_id=int(b['class_id']) + 1, area=float(b['w'] * b['h']),
                             bbox=[float(b['x']), float(b['y']), float(b['w']), float(b['h'])], id=len(anns) + 1))
        for b in d:
            res.append(dict(image_id=k + 1, category_id=int(b['class_id']) + 1, score=float(b['class_confidence']),
                            area=float(b['w'] * b['h']), bbox=[float(b['x']), float(b['y']), float(b['w']), float(b['h'])],
                            id=len(res) + 1))
    return anns, res


# ---- COCOeval restated (parity unpinned, see the module docstring) -----------------------------------------------------------
def _iou(d, g):
    """maskApi bbIou for non-crowd boxes (x, y, w, h), double precision."""
    w = min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0])
    h = min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1])
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    return inter / (d[2] * d[3] + g[2] * g[3] - inter)


def _evaluate_image(gt, dt, area_rng, max_det, thrs):
    """COCOeval.evaluateImg for one (image, category, area range)."""
    if not gt and not dt:
        return None
    g_ig = [1 if (g['area'] < area_rng[0] or g['area'] > area_rng[1]) else 0 for g in gt]
    g_order = np.argsort(g_ig, kind='mergesort')
    gt = [gt[i] for i in g_order]
    g_ig = [g_ig[i] for i in g_order]
    d_order = np.argsort([-d['score'] for d in dt], kind='mergesort')
    dt = [dt[i] for i in d_order[:max_det]]
    ious = [[_iou(d['bbox'], g['bbox']) for g in gt] for d in dt]
    T, G, D = len(thrs), len(gt), len(dt)
    gtm = np.zeros((T, G), dtype=np.int64)
    dtm = np.zeros((T, D), dtype=np.int64)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(thrs):
        for di in range(D):
            best = min(t, 1 - 1e-10)
            m = -1
            for gi in range(G):
                if gtm[ti, gi] > 0:
                    continue
                if m > -1 and g_ig[m] == 0 and g_ig[gi] == 1:
                    break
                if ious[di][gi] < best:
                    continue
                best = ious[di][gi]
                m = gi
            if m == -1:
                continue
            dt_ig[ti, di] = bool(g_ig[m])
            dtm[ti, di] = gt[m]['id']
            gtm[ti, m] = dt[di]['id']
    outside = np.array([d['area'] < area_rng[0] or d['area'] > area_rng[1] for d in dt], dtype=bool).reshape(1, D)
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(outside, T, 0)))
    return dict(dtm=dtm, dt_ig=dt_ig, scores=[d['score'] for d in dt], g_ig=np.array(g_ig, dtype=np.int64))


def coco_tables(anns, res, n_img: int, n_cat: int):
    """precision [T,R,K,A,M] and recall [T,K,A,M] (COCOeval.evaluate + accumulate), -1 where undefined."""
    thrs, recs = iou_thresholds(), recall_thresholds()
    T, R, K, A, M = len(thrs), len(recs), n_cat, len(AREA_RANGES), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    by_key_g: Dict[Tuple[int, int], list] = {}
    by_key_d: Dict[Tuple[int, int], list] = {}
    for a in anns:
        by_key_g.setdefault((a['image_id'], a['category_id']), []).append(a)
    for r in res:
        by_key_d.setdefault((r['image_id'], r['category_id']), []).append(r)
    # only categories that occur in the ground truth are evaluated (COCO.getCatIds of the GT data set lists the declared
    # categories; the reference declares all of them, coco_eval.py:107-108)
    for k in range(K):
        for ai, rng in enumerate(AREA_RANGES):
            per_img = [_evaluate_image(by_key_g.get((i + 1, k + 1), []), by_key_d.get((i + 1, k + 1), []), rng, MAX_DETS[-1], thrs)
                       for i in range(n_img)]
            per_img = [e for e in per_img if e is not None]
            if not per_img:
                continue
            for mi, max_det in enumerate(MAX_DETS):
                scores = np.concatenate([np.asarray(e['scores'][:max_det], dtype=np.float64) for e in per_img])
                order = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e['dtm'][:, :max_det] for e in per_img], axis=1)[:, order]
                dt_ig = np.concatenate([e['dt_ig'][:, :max_det] for e in per_img], axis=1)[:, order]
                g_ig = np.concatenate([e['g_ig'] for e in per_img])
                npig = int(np.count_nonzero(g_ig == 0))
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                for ti in range(T):
                    tp, fp = tp_sum[ti], fp_sum[ti]
                    nd = len(tp)
                    rc = tp / npig
                    pr = (tp / (fp + tp + np.spacing(1))).tolist()
                    recall[ti, k, ai, mi] = rc[-1] if nd else 0
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = [0.0] * R
                    for ri, pi in enumerate(np.searchsorted(rc, recs, side='left')):
                        if pi >= nd:
                            break
                        q[ri] = pr[pi]
                    precision[ti, :, k, ai, mi] = q
    return precision, recall


def coco_stats(anns, res, n_img: int, n_cat: int) -> np.ndarray:
    """The 12 numbers of COCOeval.summarize (AP, AP50, AP75, APs, APm, APl, AR1, AR10, AR100, ARs, ARm, ARl)."""
    precision, recall = coco_tables(anns, res, n_img, n_cat)
    thrs = iou_thresholds()

    def mean_valid(x):
        x = x[x > -1]
        return float(np.mean(x)) if x.size else -1.0

    def ap(iou=None, area=0, m=2):
        p = precision if iou is None else precision[np.where(iou == thrs)[0]]
        return mean_valid(p[:, :, :, area, m])

    def ar(area=0, m=2):
        return mean_valid(recall[:, :, area, m])

    return np.array([ap(), ap(.5), ap(.75), ap(area=1), ap(area=2), ap(area=3),
                     ar(m=0), ar(m=1), ar(m=2), ar(area=1), ar(area=2), ar(area=3)])


def evaluate_detection(gt_list, dt_list, n_cat: int, time_tol=50000) -> Dict[str, float]:
    """coco_eval.py:32-139."""
    gts, dts = [], []
    for g, d in zip(gt_list, dt_list):
        gw, dw = match_times(g, d, time_tol)
        gts += gw
        dts += dw
    if sum(d.size for d in dts) == 0:
        return {k: 0.0 for k in OUT_KEYS}
    anns, res = to_coco_records(gts, dts)
    stats = coco_stats(anns, res, len(gts), n_cat)
    return {k: float(stats[i]) for i, k in enumerate(OUT_KEYS)}


def evaluate_list(dt_list, gt_list, camera='gen1', downsampled_by_2=False, apply_bbox_filters=True) -> Dict[str, float]:
    """evaluation.py:5-42."""
    n_cat = 3 if camera == 'gen4' else 2
    if apply_bbox_filters:
        diag, side = filter_thresholds(camera, downsampled_by_2)
        gt_list = [filter_boxes(b, int(5e5), diag, side) for b in gt_list]
        dt_list = [filter_boxes(b, int(5e5), diag, side) for b in dt_list]
    return evaluate_detection(gt_list, dt_list, n_cat)


def evaluate_buffer(labels, predictions, camera='gen1', downsampled_by_2=False) -> Dict[str, float]:
    """evaluator.py:70-110: overall metrics plus `<key>_<class name>` per class."""
    names = ('car', 'ped') if camera == 'gen1' else ('ped', 'cyc', 'car')
    out = evaluate_list(predictions, labels, camera, downsampled_by_2)
    for c, name in enumerate(names):
        m = evaluate_list([p[p['class_id'] == c] for p in predictions], [l[l['class_id'] == c] for l in labels],
                          camera, downsampled_by_2)
        out.update({f'{k}_{name}': v for k, v in m.items()})
    return out
