"""Box-format helpers with the interface of the reference's utils/bbox.py:11-92 (numpy arrays or torch tensors; coordinates either in the
last axis, ``last4=True``, or as the four leading rows, ``last4=False``; ``format_`` 'center' = (cx, cy, w, h), 'corner' = (x0, y0, w, h)).
The hot path does these conversions inside its kernels (pseudo_filter_kernel, postprocess_nms_kernel); the module exists for callers of
the reference's helpers (analysis scripts, the tracker's host code)."""
from typing import List, Tuple, Union

import numpy as np
import torch as th

Boxes = Union[np.ndarray, th.Tensor]


def _lib(values):
    first = values[0]
    if isinstance(first, np.ndarray):
        return 'np'
    if isinstance(first, th.Tensor):
        return 'th'
    raise ValueError(f'Unknown type {type(first)}')


def np_th_stack(values: List[Boxes], axis: int = 0):
    return np.stack(values, axis=axis) if _lib(values) == 'np' else th.stack(values, dim=axis)


def np_th_concat(values: List[Boxes], axis: int = 0):
    return np.concatenate(values, axis=axis) if _lib(values) == 'np' else th.cat(values, dim=axis)


def get_bbox_coords(bbox: Boxes, last4: bool = None) -> Tuple[Tuple[Boxes, Boxes, Boxes, Boxes], bool]:
    """The four coordinate arrays of ``bbox`` and where they were found.  ``last4=None`` guesses from the shape -- four leading rows win, as in
    the reference (a [4, 4] array is therefore read row-wise)."""
    if isinstance(bbox, list):
        bbox = np_th_stack(bbox, axis=0)
    if last4 is None:
        if bbox.shape[0] == 4:
            last4 = False
        elif bbox.shape[-1] == 4:
            last4 = True
        else:
            raise ValueError(f'Unknown shape {bbox.shape}')
    if last4:
        return (bbox[..., 0], bbox[..., 1], bbox[..., 2], bbox[..., 3]), True
    a, b, c, d = bbox
    return (a, b, c, d), False


def construct_bbox(abcd: Tuple[Boxes, Boxes, Boxes, Boxes], last4: bool):
    return np_th_stack(list(abcd), axis=-1 if last4 else 0)


# (origin of the box in x / y as a multiple of its width / height: centre boxes start half a size before their anchor point)


# per anchor convention: (x, w) -> (x1, x2) and (x1, x2, w) -> x, written with the reference's own arithmetic (utils/bbox.py:63-68,83-86) so that
# rounding and integer dtypes come out the same: centre boxes halve the extent, corner boxes never leave the input dtype
_TO_CORNERS = {'center': lambda a, e: (a - e / 2., a + e / 2.), 'corner': lambda a, e: (a, a + e)}
_TO_ANCHOR = {'center': lambda lo, hi: (lo + hi) / 2., 'corner': lambda lo, hi: lo}


def _convention(table, format_: str):
    if format_ not in table:
        raise NotImplementedError(f'Unknown format {format_}')
    return table[format_]


def xywh2xyxy(xywh, format_: str = 'center', last4: bool = None):
    """(x, y, w, h) -> (x1, y1, x2, y2); ``ObjectLabels`` answer with their own corner boxes."""
    from leod_amd.data.genx_utils.labels import ObjectLabels
    if isinstance(xywh, ObjectLabels):
        return xywh.get_xyxy()
    span = _convention(_TO_CORNERS, format_)
    (ax, ay, bw, bh), last4 = get_bbox_coords(xywh, last4=last4)
    (left, right), (top, bottom) = span(ax, bw), span(ay, bh)
    return construct_bbox((left, top, right, bottom), last4=last4)


def xyxy2xywh(xyxy, format_: str = 'center', last4: bool = None):
    """(x1, y1, x2, y2) -> (x, y, w, h) with the anchor point at the centre or at the top-left corner."""
    from leod_amd.data.genx_utils.labels import ObjectLabels
    if isinstance(xyxy, ObjectLabels):
        return xyxy.get_xywh(format_=format_)
    anchor = _convention(_TO_ANCHOR, format_)
    (left, top, right, bottom), last4 = get_bbox_coords(xyxy, last4=last4)
    return construct_bbox((anchor(left, right), anchor(top, bottom), right - left, bottom - top), last4=last4)
