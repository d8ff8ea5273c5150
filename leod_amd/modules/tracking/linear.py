"""``track`` = EventSeqData._track (modules/pseudo_labeler.py:201-258) on plain arrays; ``track_filter`` adds the
forward / "forward or backward" combination of EventSeqData._track_filter (:260-290).  The tracker itself
(modules/tracking/linear.py:10-292) runs in C++ behind ``leod_track_filter``."""
import ctypes
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

from leod_amd._lib import lib, check


@dataclass
class LinearTrackerConfig:
    """Constructor defaults of the reference's LinearTracker (modules/tracking/linear.py:199-204)."""
    min_conf: float = 0.55
    iou_threshold: float = 0.45
    q: float = 0.9


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def track(boxes: Sequence[np.ndarray], is_gt: Sequence[np.ndarray], frame_idx: Sequence[int], img_hw: Tuple[int, int],
          min_track_len: int = 6, inpaint: bool = False, cfg: LinearTrackerConfig = LinearTrackerConfig()) \
        -> Tuple[List[int], Dict[int, np.ndarray]]:
    """boxes[k]: [n_k,5] float32 (cx,cy,w,h,class) of labelled frame frame_idx[k] (ascending); is_gt[k]: [n_k] bool.
    Returns (global indices of the boxes to ignore, {frame: [m,8] in-painted label rows (t,x,y,w,h,class,0,0)})."""
    assert min_track_len > 0, f'{min_track_len=} <= 0'
    assert len(boxes) == len(frame_idx) == len(is_gt)
    if len(boxes) == 0:
        return [], {}
    counts = np.array([len(b) for b in boxes], dtype=np.int32)
    allb = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float32).reshape(-1, 5) for b in boxes], 0))
    gt = np.ascontiguousarray(np.concatenate([np.asarray(g).reshape(-1) for g in is_gt]).astype(np.uint8))
    fidx = np.ascontiguousarray(np.asarray(list(frame_idx), dtype=np.int32))
    n = int(counts.sum())
    remove = np.zeros((max(n, 1),), dtype=np.uint8)
    # a tracklet can miss at most every frame of the recording once: (frames x live tracklets) bounds the in-paint list
    cap = int((fidx[-1] + 1) * max(n, 1)) if inpaint else 0
    cap = min(cap, 1 << 22)
    inp_frame = np.zeros((max(cap, 1),), dtype=np.int32)
    inp_box = np.zeros((max(cap, 1), 5), dtype=np.float32)
    n_inp = ctypes.c_int(0)
    rc = lib().leod_track_filter(_ptr(allb), _ptr(gt), _ptr(fidx), _ptr(counts), len(fidx), int(img_hw[0]), int(img_hw[1]),
                                 int(min_track_len), float(cfg.min_conf), float(cfg.iou_threshold), float(cfg.q),
                                 _ptr(remove), 1 if inpaint else 0, _ptr(inp_frame), _ptr(inp_box), cap, ctypes.byref(n_inp))
    check(rc, 'track_filter')
    remove_idx = np.nonzero(remove[:n])[0].tolist()
    inpainted: Dict[int, np.ndarray] = {}
    if inpaint and n_inp.value:
        m = n_inp.value
        fr, bx = inp_frame[:m], inp_box[:m]
        for f in dict.fromkeys(fr.tolist()):                          # frames in order of first appearance
            b = bx[fr == f]
            lab = np.zeros((len(b), 8), dtype=np.float32)             # pseudo_labeler.py:251-259: corner xywh, t = conf = 0
            lab[:, 1] = b[:, 0] - b[:, 2] / np.float32(2.)
            lab[:, 2] = b[:, 1] - b[:, 3] / np.float32(2.)
            lab[:, 3:6] = b[:, 2:5]
            inpainted[int(f)] = lab
    return remove_idx, inpainted


def track_filter(boxes: Sequence[np.ndarray], is_gt: Sequence[np.ndarray], frame_idx: Sequence[int], img_hw,
                 min_track_len: int = 6, track_method: str = 'forward or backward', inpaint: bool = False):
    """Which boxes get the ignore label and which boxes are in-painted (pseudo_labeler.py:260-290): forward tracking,
    optionally AND-ed with tracking the time-reversed recording."""
    assert track_method in ('forward', 'forward or backward'), f'Unknown tracking post-processing {track_method}'
    if len(boxes) == 0 or min_track_len <= 0:
        return [], {}
    remove, inp = track(boxes, is_gt, frame_idx, img_hw, min_track_len, inpaint)
    if 'backward' in track_method:
        fmax = max(frame_idx)
        rb = [np.ascontiguousarray(np.asarray(b)[::-1]) for b in list(boxes)[::-1]]
        rg = [np.ascontiguousarray(np.asarray(g)[::-1]) for g in list(is_gt)[::-1]]
        rf = [fmax - f for f in list(frame_idx)[::-1]]
        bremove, _ = track(rb, rg, rf, img_hw, min_track_len, False)
        n = sum(len(b) for b in boxes)
        remove = sorted(set(remove) & {n - i - 1 for i in bremove})   # both directions must agree to drop a box
    return remove, inp
