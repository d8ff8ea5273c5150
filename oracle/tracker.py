"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the pseudo-label tracking post-filter of the reference:

  * linear-velocity box tracklets            modules/tracking/linear.py:10-151 (LinearBoxTracker)
  * confidence-ordered greedy association    modules/tracking/linear.py:154-193, utils.py:7-51
  * the online tracker loop                  modules/tracking/linear.py:196-292, tracker.py:9-47
  * short-tracklet filter + in-painting      modules/pseudo_labeler.py:201-258 (EventSeqData._track)
  * forward / "forward or backward" merge    modules/pseudo_labeler.py:260-333 (EventSeqData._track_filter)

Pinned by tests/golden/g13_tracker.npz (recorded from the reference itself by tests/golden/make_golden.py).

Arithmetic follows what the reference computes under NumPy >= 2 (NEP 50): boxes are float32 and stay float32 through
`x - w / 2.`, np.clip with Python-float bounds and the IoU; tracklet confidences are Python floats (float64); the
association threshold 0.45 is compared against float32 IoUs *as float32*.  `np.argsort` of the (negated) confidences
is restated as a stable sort: NumPy's introsort is an insertion sort below 17 elements, beyond that the order of exactly
tied confidences is implementation defined (AVX-512 builds use a different network) -- parity unpinned in that corner.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

F32 = np.float32


def _xyxy(b):
    """[cx,cy,w,h] float32 -> x1,y1,x2,y2 float32 (utils.py:65-72)."""
    half_w, half_h = F32(b[2] / F32(2.)), F32(b[3] / F32(2.))
    return F32(b[0] - half_w), F32(b[1] - half_h), F32(b[0] + half_w), F32(b[1] + half_h)


def clamp_state(b, img_hw):
    """utils.py:75-96 with format_='xywh': clamp the corners to [0, W-1] x [0, H-1], back to centre format."""
    H, W = img_hw
    x1_, y1_, x2_, y2_ = _xyxy(b)
    x1 = F32(min(max(x1_, F32(0.)), F32(W - 1.)))
    x2 = F32(min(max(x2_, F32(0.)), F32(W - 1.)))
    y1 = F32(min(max(y1_, F32(0.)), F32(H - 1.)))
    y2 = F32(min(max(y2_, F32(0.)), F32(H - 1.)))
    out = np.array([F32(F32(x1 + x2) / F32(2.)), F32(F32(y1 + y2) / F32(2.)), F32(x2 - x1), F32(y2 - y1)], dtype=F32)
    return out, bool(y1 != y1_), bool(y2 != y2_), bool(x1 != x1_), bool(x2 != x2_)


class _Track:
    """linear.py:10-151."""

    def __init__(self, tid, box, bbox_idx, is_gt, img_hw, q):
        self.img_hw, self.q, self.id = img_hw, q, tid
        self.bbox = np.array(box[:4], dtype=F32)
        self.class_id = F32(box[4])
        self.last_bbox = None
        self.vxvy = np.zeros(2, dtype=F32)              # float64 zeros in the reference: adding 0.0 is exact
        self.clamp = (False, False, False, False)       # top, down, left, right
        self.bbox_idx = [bbox_idx]
        self.missed, self.missed_cache = [], []         # (frame, box5) in insertion order
        self.is_gt = bool(is_gt)
        self.conf = q
        self.age, self.hits, self.done = 0, 1, False
        self.pred = None

    @property
    def area(self):
        return F32(self.bbox[2] * self.bbox[3])

    def predict(self):
        self.age += 1
        self.last_bbox = self.bbox.copy()
        self.bbox[:2] = (self.bbox[:2] + self.vxvy).astype(F32)
        st, t, d, l, r = clamp_state(self.bbox, self.img_hw)
        self.clamp = (t, d, l, r)
        self.pred = np.concatenate([st, [self.class_id]]).astype(F32)
        return self.pred.copy()

    def update(self, new_box, bbox_idx, is_gt):
        self.hits = self.age + 1
        v = (new_box[:2] - self.last_bbox[:2]).astype(F32)
        t, d, l, r = self.clamp
        if t or d or l or r:                             # linear.py:100-122: use the un-clamped edge
            ox1, oy1, ox2, oy2 = _xyxy(self.last_bbox)
            nx1, ny1, nx2, ny2 = _xyxy(new_box)
            if t:
                v[1] = F32(ny2 - oy2)
            if d:
                v[1] = F32(ny1 - oy1)
            if l:
                v[0] = F32(nx2 - ox2)
            if r:
                v[0] = F32(nx1 - ox1)
        self.vxvy = v
        self.bbox = np.array(new_box[:4], dtype=F32)
        self.bbox_idx.append(bbox_idx)
        self.is_gt = self.is_gt or bool(is_gt)
        w = self.q * (1. - self.q ** self.age) / (1. - self.q)
        self.conf = (w * self.conf + 1.) / (w + 1.)
        self.missed.extend(self.missed_cache)
        self.missed_cache = []

    def miss(self, frame_idx, has_gt):
        self.conf *= self.q
        if not has_gt:
            self.missed_cache.append((frame_idx, self.pred.copy()))


def iou_xywh(trks: np.ndarray, dets: np.ndarray) -> np.ndarray:
    """utils.py:21-51, float32 throughout; different classes -> 0."""
    a, b = trks[:, None, :].astype(F32), dets[None, :, :].astype(F32)
    two = F32(2.)
    xx1 = np.maximum(a[..., 0] - a[..., 2] / two, b[..., 0] - b[..., 2] / two)
    yy1 = np.maximum(a[..., 1] - a[..., 3] / two, b[..., 1] - b[..., 3] / two)
    xx2 = np.minimum(a[..., 0] + a[..., 2] / two, b[..., 0] + b[..., 2] / two)
    yy2 = np.minimum(a[..., 1] + a[..., 3] / two, b[..., 1] + b[..., 3] / two)
    w, h = np.maximum(F32(0.), xx2 - xx1), np.maximum(F32(0.), yy2 - yy1)
    wh = w * h
    with np.errstate(divide='ignore', invalid='ignore'):
        o = wh / (a[..., 2] * a[..., 3] + b[..., 2] * b[..., 3] - wh)
    o = np.where(np.broadcast_to(a[..., 4] != b[..., 4], o.shape), F32(0.), o)
    return o.astype(F32)


def associate(trks, order, dets, thr):
    """linear.py:154-193 + utils.py:7-18."""
    T, D = len(trks), len(dets)
    if T == 0:
        return [], [], list(range(D))
    if D == 0:
        return [], list(range(T)), []
    iou = iou_xywh(trks, dets)
    matched = []
    if iou.max() > 0:
        cost = iou.copy()
        thr32 = F32(thr)
        for i in order:
            if not (cost[i].max() >= thr32):             # `max < thresh -> continue`
                continue
            j = int(np.argmax(cost[i]))
            cost[:, j] = -np.inf
            matched.append((int(i), j))
    mt, md = {m[0] for m in matched}, {m[1] for m in matched}
    return matched, [t for t in range(T) if t not in mt], [d for d in range(D) if d not in md]


class LinearTracker:
    """linear.py:196-292 + tracker.py:9-47."""

    def __init__(self, img_hw, min_conf=0.55, iou_threshold=0.45, q=0.9):
        self.img_hw, self.min_conf, self.iou_threshold, self.q = img_hw, min_conf, iou_threshold, q
        self.trackers: List[_Track] = []
        self.prev: List[_Track] = []
        self.box2trk: Dict[int, _Track] = {}
        self.track_count = self.bbox_count = 0

    def _delete(self, idx, done=True):
        t = self.trackers.pop(idx)
        t.done = done
        self.prev.append(t)
        for b in t.bbox_idx:
            self.box2trk[b] = t

    def update(self, frame_idx, dets=None, is_gt=None):
        dets = np.zeros((0, 5), F32) if dets is None else np.asarray(dets, dtype=F32)
        if len(dets) == 0 and len(self.trackers) == 0:
            return
        is_gt = np.zeros((len(dets),), bool) if is_gt is None or len(is_gt) == 0 else np.asarray(is_gt, dtype=bool)
        to_del, trks, neg_conf = [], [], []
        for t, trk in enumerate(self.trackers):
            if trk.area <= 0.:
                to_del.append(t)
                continue
            trks.append(trk.predict())
            neg_conf.append(-trk.conf)
        for t in reversed(to_del):
            self._delete(t)
        order = np.argsort(np.asarray(neg_conf, dtype=np.float64), kind='stable')
        trks = np.stack(trks, 0) if trks else np.zeros((0, 5), F32)
        matched, un_t, un_d = associate(trks, order, dets, self.iou_threshold)
        for ti, di in matched:
            self.trackers[ti].update(dets[di], self.bbox_count + di, is_gt[di])
        for t in un_t:
            self.trackers[t].miss(frame_idx, bool(is_gt.any()))
        for d in un_d:
            self.trackers.append(_Track(self.track_count, dets[d], self.bbox_count + d, is_gt[d], self.img_hw, self.q))
            self.track_count += 1
        for i in reversed(range(len(self.trackers))):
            if self.trackers[i].conf < self.min_conf:
                self._delete(i)
        self.bbox_count += len(dets)

    def finish(self):
        for i in reversed(range(len(self.trackers))):
            self._delete(i, done=False)


def track(boxes: Sequence[np.ndarray], is_gt: Sequence[np.ndarray], frame_idx: Sequence[int], img_hw,
          min_track_len: int = 6, inpaint: bool = False) -> Tuple[List[int], Dict[int, np.ndarray]]:
    """EventSeqData._track (pseudo_labeler.py:201-258).  boxes[k]: [n_k,5] float32 (cx,cy,w,h,class) of labelled frame
    frame_idx[k].  Returns (indices of boxes on finished, non-GT tracklets shorter than min_track_len;
    {frame: [m,8] in-painted labels (t,x,y,w,h,class,0,0), corner format})."""
    if len(boxes) == 0:
        return [], {}
    model = LinearTracker(img_hw)
    frame_idx = list(frame_idx)
    for f in range(max(frame_idx) + 1):
        if f not in frame_idx:
            model.update(f)
            continue
        k = frame_idx.index(f)
        model.update(f, boxes[k], is_gt[k])
    model.finish()
    n = sum(len(b) for b in boxes)
    remove = []
    for b in range(n):
        t = model.box2trk[b]
        if t.done and not t.is_gt and t.hits < min_track_len:
            remove.append(b)
    if not inpaint:
        return remove, {}
    per_frame: Dict[int, list] = {}
    for t in model.prev:
        if t.done and not t.is_gt and t.hits < min_track_len:
            continue
        for f, box in t.missed:
            per_frame.setdefault(f, []).append(box)
    out = {}
    for f, lst in per_frame.items():
        b = np.stack(lst).astype(F32)
        lab = np.zeros((len(b), 8), F32)
        lab[:, 1] = b[:, 0] - b[:, 2] / F32(2.)
        lab[:, 2] = b[:, 1] - b[:, 3] / F32(2.)
        lab[:, 3:6] = b[:, 2:5]
        out[f] = lab
    return remove, out


def track_filter(boxes, is_gt, frame_idx, img_hw, min_track_len=6, track_method='forward or backward', inpaint=False):
    """EventSeqData._track_filter (pseudo_labeler.py:260-333), reduced to its decisions: the global indices of the
    boxes whose class id becomes `ignore_label`, and the in-painted boxes per frame (forward pass only)."""
    if len(boxes) == 0 or min_track_len <= 0:
        return [], {}
    remove, inp = track(boxes, is_gt, frame_idx, img_hw, min_track_len, inpaint)
    if 'backward' in track_method:
        rb = [b[::-1].copy() for b in boxes[::-1]]       # ObjectLabels.get_reverse: rows flipped, frames reversed
        rg = [g[::-1].copy() for g in is_gt[::-1]]
        rf = [max(frame_idx) - f for f in list(frame_idx)[::-1]]
        bremove, _ = track(rb, rg, rf, img_hw, min_track_len, False)
        n = sum(len(b) for b in boxes)
        bremove = [n - i - 1 for i in bremove[::-1]]
        remove = sorted(set(remove) & set(bremove))
    return remove, inp


def apply_track_filter(rows: List[np.ndarray], frame_idx: Sequence[int], img_hw, min_track_len=6,
                       track_method='forward or backward', inpaint=False, ignore_label=1024):
    """Full effect of EventSeqData._track_filter on the label rows ([n,8]: t,x,y,w,h,class,cls_conf,obj; corner xywh):
    short-tracklet boxes get class `ignore_label`, in-painted boxes (class `ignore_label`, t = conf = 0) are appended to
    their frame or inserted as a new frame; frames come back sorted (pseudo_labeler.py:291-333)."""
    rows = [np.array(r, dtype=F32, copy=True) for r in rows]
    frame_idx = [int(f) for f in frame_idx]
    if len(rows) == 0 or min_track_len <= 0:
        return frame_idx, rows
    boxes = [np.stack([r[:, 1] + F32(0.5) * r[:, 3], r[:, 2] + F32(0.5) * r[:, 4], r[:, 3], r[:, 4], r[:, 5]], -1).astype(F32)
             for r in rows]                                      # labels.py:521-531 get_xywh('center', add_class_id)
    is_gt = [r[:, 0] != 0 for r in rows]
    remove, inp = track_filter(boxes, is_gt, frame_idx, img_hw, min_track_len, track_method, inpaint)
    remove = set(remove)
    b = 0
    for r in rows:
        for i in range(len(r)):
            if b in remove:
                r[i, 5] = ignore_label
            b += 1
    if inp:
        for f in range(max(frame_idx) + 1):
            if f not in inp:
                continue
            lab = inp[f].copy()
            lab[:, 5] = ignore_label
            if f in frame_idx:
                k = frame_idx.index(f)
                rows[k] = np.concatenate([rows[k], lab], 0)
            else:
                frame_idx.append(f)
                rows.append(lab)
        order = sorted(range(len(frame_idx)), key=lambda k: frame_idx[k])
        frame_idx, rows = [frame_idx[k] for k in order], [rows[k] for k in order]
    return frame_idx, rows
