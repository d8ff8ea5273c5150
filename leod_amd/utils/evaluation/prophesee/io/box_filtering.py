"""Box filters of the Prophesee evaluation protocol (reference: utils/evaluation/prophesee/io/box_filtering.py:18-36):
drop everything in the first half second and every box whose diagonal or shorter side is below the camera's minimum."""
import numpy as np


def filter_boxes(boxes: np.ndarray, skip_ts: int = int(5e5), min_box_diag: int = 60, min_box_side: int = 20) -> np.ndarray:
    """boxes: structured array with at least the fields t, w, h.  Defaults are the 1 Mpx thresholds."""
    w, h = boxes['w'], boxes['h']
    keep = boxes['t'] > skip_ts
    keep &= w ** 2 + h ** 2 >= min_box_diag ** 2
    keep &= (w >= min_box_side) & (h >= min_box_side)
    return boxes[keep]
