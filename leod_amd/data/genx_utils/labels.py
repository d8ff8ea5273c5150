"""Box-label containers on the boundary of the hot path (the subset of the reference's
data/genx_utils/labels.py that the detection / pseudo-label modules touch).

Layout contract kept from the reference: ``object_labels`` is float [N, 8] =
(t, x, y, w, h, class_id, class_confidence, objectness) with (x, y) the TOP-LEFT corner; pseudo labels
carry t == 0; the loss target layout is [B, Nmax, 7] = (cls, cx, cy, w, h, obj, cls_conf) zero padded
(labels.py:543-603).  BBOX_DTYPE is the 40-byte on-disk record (labels.py:12-16)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch as th

BBOX_DTYPE = np.dtype({
    'names': ['t', 'x', 'y', 'w', 'h', 'class_id', 'class_confidence', 'objectness'],
    'formats': ['<i8', '<f4', '<f4', '<f4', '<f4', '<u4', '<f4', '<f4'],
    'offsets': [0, 8, 12, 16, 20, 24, 28, 32], 'itemsize': 40})

FIELDS = ('t', 'x', 'y', 'w', 'h', 'class_id', 'class_confidence', 'objectness')
_IDX = {k: i for i, k in enumerate(FIELDS)}


class ObjectLabels:
    _str2idx = dict(_IDX)

    def __init__(self, object_labels: th.Tensor, input_size_hw: Tuple[int, int]):
        if isinstance(object_labels, np.ndarray):
            object_labels = th.from_numpy(object_labels)
        assert object_labels.is_floating_point() and object_labels.dim() == 2 and object_labels.shape[1] == len(FIELDS)
        assert isinstance(input_size_hw, tuple) and len(input_size_hw) == 2
        self.object_labels = object_labels
        self._input_size_hw = input_size_hw

    # ---- field access -------------------------------------------------------------------------------
    def __getattr__(self, name):
        idx = _IDX.get(name)
        if idx is None:
            raise AttributeError(name)
        return self.object_labels[:, idx]

    def _set(self, name, value):
        self.object_labels[:, _IDX[name]] = value

    def get(self, request: str):
        return self.object_labels[:, _IDX[request]]

    def numpy_(self) -> None:
        """In-place conversion to numpy (labels.py:82-87); used by the evaluation records."""
        if th.is_tensor(self.object_labels):
            self.object_labels = self.object_labels.detach().cpu().numpy()

    def torch_(self) -> None:
        if not th.is_tensor(self.object_labels):
            self.object_labels = th.from_numpy(self.object_labels)

    @property
    def input_size_hw(self):
        return self._input_size_hw

    @input_size_hw.setter
    def input_size_hw(self, hw):
        assert isinstance(hw, tuple) and len(hw) == 2 and hw[0] > 0 and hw[1] > 0
        self._input_size_hw = hw

    @classmethod
    def keys(cls) -> List[str]:
        return list(FIELDS)

    @property
    def dtype(self):
        return self.object_labels.dtype

    @property
    def device(self):
        return self.object_labels.device

    def __len__(self):
        return self.object_labels.shape[0]

    def __add__(self, other: 'ObjectLabels') -> 'ObjectLabels':
        assert isinstance(other, ObjectLabels) and self.input_size_hw == other.input_size_hw
        return ObjectLabels(th.cat([self.object_labels, other.object_labels], dim=0), self.input_size_hw)

    def to(self, *args, **kwargs):
        self.object_labels = self.object_labels.to(*args, **kwargs)
        return self

    def new_zeros(self) -> 'ObjectLabels':
        return ObjectLabels(self.object_labels.new_zeros((0, len(FIELDS))), self.input_size_hw)

    def __eq__(self, other) -> bool:
        """Same frame size and the same boxes in any order, to 1e-3 per field (labels.py:271-286)."""
        if not isinstance(other, ObjectLabels) or self.input_size_hw != other.input_size_hw or \
                tuple(self.object_labels.shape) != tuple(other.object_labels.shape):
            return False
        if len(self) == 0:
            return True
        a, b = th.as_tensor(self.object_labels).float(), th.as_tensor(other.object_labels).float()
        near = (a[None, :, :] - b[:, None, :]).abs().amax(dim=2) < 1e-3        # [other row, own row]
        return bool(near.any(dim=1).all())

    __hash__ = None

    def get_reverse(self) -> 'ObjectLabels':
        """Rows in reverse order (labels.py:515-519): the time-reversed view used by backward tracking."""
        return ObjectLabels(self.object_labels.flip(0), self.input_size_hw)

    # ---- predicates ---------------------------------------------------------------------------------
    def is_pseudo_label(self):
        return self.object_labels[:, 0] == 0          # pseudo labels are written with t == 0

    def is_gt_label(self):
        return ~self.is_pseudo_label()

    def is_ignore(self, ignore_label):
        return self.object_labels[:, 5] == ignore_label

    # ---- transforms the hot path uses -----------------------------------------------------------------
    def flip_lr_(self) -> None:
        if len(self):
            self._set('x', self.input_size_hw[1] - 1 - self.object_labels[:, 1] - self.object_labels[:, 3])

    def reverse_flip_lr_(self) -> None:
        self.flip_lr_()

    # ---- spatial augmentation of the labels (labels.py:67-69, 372-408, 436-457, 486-504) ---------------------------------
    def remove_flat_labels_(self) -> None:
        keep = (self.object_labels[:, 3] > 0) & (self.object_labels[:, 4] > 0)
        self.object_labels = self.object_labels[keep]

    def scale_(self, scaling_multiplier: float) -> None:
        if len(self) == 0:
            return
        assert scaling_multiplier > 0
        if scaling_multiplier == 1:
            return
        img_ht, img_wd = self.input_size_hw
        new_ht, new_wd = scaling_multiplier * img_ht, scaling_multiplier * img_wd
        self.input_size_hw = (new_ht, new_wd)
        o = self.object_labels
        x1 = th.clamp((o[:, 1] + o[:, 3]) * scaling_multiplier, max=new_wd - 1)
        y1 = th.clamp((o[:, 2] + o[:, 4]) * scaling_multiplier, max=new_ht - 1)
        self._set('x', o[:, 1] * scaling_multiplier)
        self._set('y', o[:, 2] * scaling_multiplier)
        self._set('w', x1 - self.object_labels[:, 1])
        self._set('h', y1 - self.object_labels[:, 2])
        self.remove_flat_labels_()

    def zoom_in_and_rescale_(self, zoom_coordinates_x0y0, zoom_in_factor: float) -> None:
        """Crop to the zoom window at ``zoom_coordinates_x0y0`` (size = frame / factor), drop boxes that fall outside,
        rescale back to the frame resolution."""
        if len(self) == 0:
            return
        assert len(zoom_coordinates_x0y0) == 2 and zoom_in_factor >= 1
        if zoom_in_factor == 1:
            return
        z_x0, z_y0 = zoom_coordinates_x0y0
        h_orig, w_orig = self.input_size_hw
        assert 0 <= z_x0 <= w_orig - 1 and 0 <= z_y0 <= h_orig - 1
        win_h, win_w = tuple(v / zoom_in_factor for v in self.input_size_hw)
        z_x1 = min(z_x0 + win_w, w_orig - 1)
        z_y1 = min(z_y0 + win_h, h_orig - 1)
        o = self.object_labels
        x0 = th.clamp(o[:, 1], min=z_x0, max=z_x1 - 1)
        y0 = th.clamp(o[:, 2], min=z_y0, max=z_y1 - 1)
        x1 = th.clamp(o[:, 1] + o[:, 3], min=z_x0, max=z_x1 - 1)
        y1 = th.clamp(o[:, 2] + o[:, 4], min=z_y0, max=z_y1 - 1)
        self._set('x', x0 - z_x0)
        self._set('y', y0 - z_y0)
        self._set('w', x1 - x0)
        self._set('h', y1 - y0)
        self.input_size_hw = (win_h, win_w)
        self.remove_flat_labels_()
        self.scale_(scaling_multiplier=zoom_in_factor)

    def zoom_out_and_rescale_(self, zoom_coordinates_x0y0, zoom_out_factor: float) -> None:
        """Shrink by 1 / factor and shift to the paste position of the shrunk frame."""
        if len(self) == 0:
            return
        assert len(zoom_coordinates_x0y0) == 2 and zoom_out_factor >= 1
        if zoom_out_factor == 1:
            return
        h_orig, w_orig = self.input_size_hw
        self.scale_(scaling_multiplier=1 / zoom_out_factor)
        self.input_size_hw = (h_orig, w_orig)
        z_x0, z_y0 = zoom_coordinates_x0y0
        assert 0 <= z_x0 <= w_orig - 1 and 0 <= z_y0 <= h_orig - 1
        self._set('x', self.object_labels[:, 1] + z_x0)
        self._set('y', self.object_labels[:, 2] + z_y0)

    def reverse_zoom_in_and_rescale_(self, zoom_coordinates_x0y0, zoom_in_factor: float) -> None:
        """Inverse of ``zoom_in_and_rescale_`` for boxes that survived it (labels.py:410-434): back to the size of the zoom window, then to
        the window's position in the frame."""
        if len(self) == 0:
            return
        assert len(zoom_coordinates_x0y0) == 2 and zoom_in_factor >= 1
        if zoom_in_factor == 1:
            return
        frame_hw = self.input_size_hw
        self.scale_(scaling_multiplier=1 / zoom_in_factor)
        self._set('x', self.object_labels[:, 1] + zoom_coordinates_x0y0[0])
        self._set('y', self.object_labels[:, 2] + zoom_coordinates_x0y0[1])
        self.input_size_hw = frame_hw

    def reverse_zoom_out_and_rescale_(self, zoom_coordinates_x0y0, zoom_out_factor: float) -> None:
        """Inverse of ``zoom_out_and_rescale_`` (labels.py:459-484): off the paste position, then back up by the factor; the boxes must land
        inside the frame again."""
        if len(self) == 0:
            return
        assert len(zoom_coordinates_x0y0) == 2 and zoom_out_factor >= 1
        if zoom_out_factor == 1:
            return
        self._set('x', self.object_labels[:, 1] - zoom_coordinates_x0y0[0])
        self._set('y', self.object_labels[:, 2] - zoom_coordinates_x0y0[1])
        frame_hw = self.input_size_hw
        self.scale_(scaling_multiplier=zoom_out_factor)
        self.input_size_hw = frame_hw
        o, (ht, wd) = self.object_labels, frame_hw
        assert bool((o[:, 1] >= 0).all()) and bool((o[:, 1] + o[:, 3] <= wd - 1).all())
        assert bool((o[:, 2] >= 0).all()) and bool((o[:, 2] + o[:, 4] <= ht - 1).all())

    def clamp_to_frame_(self):
        ht, wd = self.input_size_hw
        o = self.object_labels
        x0, y0 = o[:, 1].clamp(0, wd - 1), o[:, 2].clamp(0, ht - 1)
        x1, y1 = (o[:, 1] + o[:, 3]).clamp(0, wd - 1), (o[:, 2] + o[:, 4]).clamp(0, ht - 1)
        o[:, 1], o[:, 2], o[:, 3], o[:, 4] = x0, y0, x1 - x0, y1 - y0

    def get_xywh(self, format_='center', add_class_id=False):
        o = self.object_labels
        x, y = (o[:, 1] + 0.5 * o[:, 3], o[:, 2] + 0.5 * o[:, 4]) if format_ == 'center' else (o[:, 1], o[:, 2])
        cols = [x, y, o[:, 3], o[:, 4]] + ([o[:, 5]] if add_class_id else [])
        return th.stack(cols, dim=-1)

    def get_xyxy(self, add_class_id=False):
        o = self.object_labels
        cols = [o[:, 1], o[:, 2], o[:, 1] + o[:, 3], o[:, 2] + o[:, 4]] + ([o[:, 5]] if add_class_id else [])
        return th.stack(cols, dim=-1)

    def get_labels_as_tensors(self, format_: str = 'yolox') -> th.Tensor:
        o = self.object_labels
        out = th.zeros((len(self), 7), dtype=th.float32, device=o.device)
        if len(self) == 0:
            return out
        if format_ == 'yolox':          # (cls, cx, cy, w, h, obj, cls_conf)
            out[:, 0], out[:, 1], out[:, 2] = o[:, 5], o[:, 1] + 0.5 * o[:, 3], o[:, 2] + 0.5 * o[:, 4]
            out[:, 3], out[:, 4], out[:, 5], out[:, 6] = o[:, 3], o[:, 4], o[:, 7], o[:, 6]
        elif format_ == 'prophesee':    # (x1, y1, x2, y2, obj, cls_conf, cls)
            out[:, 0], out[:, 1], out[:, 2], out[:, 3] = o[:, 1], o[:, 2], o[:, 1] + o[:, 3], o[:, 2] + o[:, 4]
            out[:, 4], out[:, 5], out[:, 6] = o[:, 7], o[:, 6], o[:, 5]
        else:
            raise NotImplementedError(format_)
        return out

    @staticmethod
    def pad_labels(obj_label_list: Sequence[Union['ObjectLabels', th.Tensor]], N: int, format_: str = 'yolox') -> th.Tensor:
        """[len(list), N, 7] zero-padded loss targets (labels.py:543-570).  All-``ObjectLabels`` input (the training step)
        is converted with one concatenation + one scatter instead of ~12 tiny tensor ops per frame."""
        assert format_ == 'yolox'
        first = obj_label_list[0]
        dev = first.device
        out = th.zeros((len(obj_label_list), N, 7), dtype=th.float32, device=dev)
        if all(isinstance(l, ObjectLabels) for l in obj_label_list):
            lens = [len(l) for l in obj_label_list]
            assert max(lens) <= N
            o = th.cat([l.object_labels for l in obj_label_list], dim=0)
            rows = th.stack([o[:, 5], o[:, 1] + 0.5 * o[:, 3], o[:, 2] + 0.5 * o[:, 4], o[:, 3], o[:, 4], o[:, 7], o[:, 6]], dim=1)
            frame = th.repeat_interleave(th.arange(len(lens)), th.tensor(lens)).to(dev)
            slot = th.cat([th.arange(n) for n in lens]).to(dev)
            out[frame, slot] = rows.to(th.float32)
            return out
        for i, l in enumerate(obj_label_list):
            t = l.get_labels_as_tensors('yolox') if isinstance(l, ObjectLabels) else l
            out[i, :len(t)] = t
        return out

    @staticmethod
    def get_labels_as_batched_tensor(obj_label_list: List['ObjectLabels'], format_: str = 'yolox') -> th.Tensor:
        assert len(obj_label_list) > 0
        N = max(len(x) for x in obj_label_list)
        assert N > 0
        return ObjectLabels.pad_labels(obj_label_list, N=N, format_=format_)

    @staticmethod
    def from_structured_array(labels: np.ndarray, input_size_hw: Tuple[int, int]) -> 'ObjectLabels':
        cols = []
        for k in FIELDS:
            src = k if k in labels.dtype.names else 'class_confidence'     # objectness defaults to class_confidence
            cols.append(labels[src].astype('float32'))
        return ObjectLabels(th.from_numpy(np.stack(cols, axis=1)), input_size_hw)

    def to_structured_array(self) -> np.ndarray:
        o = self.object_labels.detach().cpu().numpy()
        out = np.zeros((len(self),), dtype=BBOX_DTYPE)
        for k in FIELDS:
            out[k] = o[:, _IDX[k]]
        return out


class ObjectLabelFactory:
    """All boxes of one recording + the start offset of every labelled frame (reference: labels.py:188-246).
    ``factory[i]`` = the boxes of the i-th labelled frame as a fresh ``ObjectLabels`` (rescaled by 1 / downsample_factor
    when the recording is loaded at half resolution).  Boxes are clamped to the frame once, at construction."""

    def __init__(self, object_labels: th.Tensor, objframe_idx_2_label_idx: th.Tensor, input_size_hw: Tuple[int, int],
                 downsample_factor: Optional[float] = None):
        assert objframe_idx_2_label_idx.dtype == th.int64 and objframe_idx_2_label_idx.dim() == 1
        assert downsample_factor is None or downsample_factor > 1
        self._all = ObjectLabels(object_labels, tuple(input_size_hw))
        self._all.clamp_to_frame_()
        self.object_labels = self._all.object_labels
        self.input_size_hw = tuple(input_size_hw)
        self.starts = objframe_idx_2_label_idx.tolist()
        self.objframe_idx_2_label_idx = objframe_idx_2_label_idx
        self.downsample_factor = downsample_factor

    @staticmethod
    def from_structured_array(object_labels: np.ndarray, objframe_idx_2_label_idx: np.ndarray, input_size_hw: Tuple[int, int],
                              downsample_factor: Optional[float] = None) -> 'ObjectLabelFactory':
        rows = ObjectLabels.from_structured_array(object_labels, tuple(input_size_hw)).object_labels
        return ObjectLabelFactory(rows, th.from_numpy(objframe_idx_2_label_idx.astype('int64')), input_size_hw, downsample_factor)

    def __len__(self):
        return len(self.starts)

    def __getitem__(self, item: int) -> ObjectLabels:
        n = len(self)
        assert 0 <= item < n
        lo = self.starts[item]
        hi = self.object_labels.shape[0] if item == n - 1 else self.starts[item + 1]
        assert hi > lo
        out = ObjectLabels(self.object_labels[lo:hi].clone(), self.input_size_hw)
        if self.downsample_factor is not None:
            out.scale_(scaling_multiplier=1 / self.downsample_factor)
        return out


class SparselyBatchedObjectLabels:
    """One timestep of a batch: a list (len B) of ObjectLabels or None."""

    def __init__(self, sparse_object_labels_batch: List[Optional[ObjectLabels]]):
        self.sparse_object_labels_batch = [None if (l is not None and len(l) == 0) else l
                                           for l in sparse_object_labels_batch]

    def __len__(self):
        return len(self.sparse_object_labels_batch)

    def __iter__(self):
        return iter(self.sparse_object_labels_batch)

    def __getitem__(self, item: int):
        if item < 0 or item >= len(self):
            raise IndexError(item)
        return self.sparse_object_labels_batch[item]

    def __add__(self, other):
        return SparselyBatchedObjectLabels(self.sparse_object_labels_batch + other.sparse_object_labels_batch)

    def is_empty(self):
        return all(x is None for x in self.sparse_object_labels_batch)

    def set_non_gt_labels_to_none_(self):
        """Drop frames that only carry pseudo labels (never GT) -- modules/detection.py:141-147."""
        for i, l in enumerate(self.sparse_object_labels_batch):
            if l is not None and bool(l.is_pseudo_label().all()):
                self.sparse_object_labels_batch[i] = None

    def flip_lr_(self):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.flip_lr_()

    def reverse_flip_lr_(self):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.reverse_flip_lr_()

    def __eq__(self, other) -> bool:
        if not isinstance(other, SparselyBatchedObjectLabels) or len(self) != len(other):
            return False
        return all((a is None and b is None) or (a is not None and b is not None and a == b)
                   for a, b in zip(self.sparse_object_labels_batch, other.sparse_object_labels_batch))

    __hash__ = None

    def get_labels_padded(self, pad=None):
        """The labels with ``pad`` where a sample has none, and the indices of the samples that have labels (labels.py:731-734)."""
        return ([l if l is not None else pad for l in self.sparse_object_labels_batch],
                [i for i, l in enumerate(self.sparse_object_labels_batch) if l is not None])

    def set_empty_labels_to_none_(self):
        """A transform may have dropped every box of a frame (labels.py:650-654)."""
        for i, l in enumerate(self.sparse_object_labels_batch):
            if l is not None and len(l) == 0:
                self.sparse_object_labels_batch[i] = None

    def zoom_in_and_rescale_(self, *args, **kwargs):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.zoom_in_and_rescale_(*args, **kwargs)
        self.set_empty_labels_to_none_()

    def zoom_out_and_rescale_(self, *args, **kwargs):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.zoom_out_and_rescale_(*args, **kwargs)

    def reverse_zoom_in_and_rescale_(self, *args, **kwargs):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.reverse_zoom_in_and_rescale_(*args, **kwargs)
        self.set_empty_labels_to_none_()

    def reverse_zoom_out_and_rescale_(self, *args, **kwargs):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.reverse_zoom_out_and_rescale_(*args, **kwargs)

    def time_flip_(self):
        self.sparse_object_labels_batch.reverse()

    def to(self, *args, **kwargs):
        for l in self.sparse_object_labels_batch:
            if l is not None:
                l.to(*args, **kwargs)
        return self

    def get_valid_labels_and_batch_indices(self, ignore: bool = False, ignore_label: int = None):
        out, idx = [], []
        for i, l in enumerate(self.sparse_object_labels_batch):
            if l is None:
                continue
            if ignore and bool(l.is_ignore(ignore_label).all()):
                continue
            out.append(l)
            idx.append(i)
        return out, idx

    @staticmethod
    def transpose_list(lst: List['SparselyBatchedObjectLabels']) -> List['SparselyBatchedObjectLabels']:
        return [SparselyBatchedObjectLabels(list(t)) for t in zip(*lst)]
