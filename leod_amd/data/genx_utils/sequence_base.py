"""One recording on disk as an indexable source of loader samples (the interface of the reference's
data/genx_utils/sequence_base.py:28-236): which frames carry labels, label sub-sampling for the sparse-label regimes,
the time-reversed view, and the frame reads.

MI355X-first difference: a sample can be produced straight INTO a slice of a pinned batch buffer (``out=``), so that a batch
of B sequences x L frames is assembled by B parallel page-cache reads into one [L,B,20,H,W] uint8 tensor that goes to the
device in a single PCIe copy and is consumed by the stem kernel as is -- the reference materialises L x B tensors per batch,
collates them and casts to fp32 on the device.  Without ``out`` the sample is self-contained, like the reference's."""
import os
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from leod_amd.data.genx_utils.labels import ObjectLabelFactory, ObjectLabels, SparselyBatchedObjectLabels
from leod_amd.data.utils import misc
from leod_amd.data.utils.types import DatasetType, DataType

_ORIGINAL_HW = {DatasetType.GEN1: (240, 304), DatasetType.GEN4: (720, 1280)}


def get_original_hw(dataset_type: DatasetType) -> Tuple[int, int]:
    return _ORIGINAL_HW[dataset_type]


def get_event_representation_dir(path: Path, ev_representation_name: str) -> Path:
    ev_repr_dir = Path(path) / 'event_representations_v2' / ev_representation_name
    assert ev_repr_dir.is_dir(), f'{ev_repr_dir}'
    return ev_repr_dir


def get_objframe_idx_2_repr_idx(path: Path, ev_representation_name: str) -> np.ndarray:
    return np.load(str(get_event_representation_dir(path, ev_representation_name) / 'objframe_idx_2_repr_idx.npy'))


class SequenceBase:
    def __init__(self, path: Path, ev_representation_name: str, sequence_length: int, dataset_type: DatasetType,
                 downsample_by_factor_2: bool, only_load_end_labels: bool, objframe_idx: Optional[List[int]] = None,
                 data_ratio: float = -1.0, tflip_offset: int = -1):
        path = Path(path)
        assert sequence_length >= 1 and path.is_dir()
        assert dataset_type in (DatasetType.GEN1, DatasetType.GEN4), f'{dataset_type} not implemented'
        self.path = str(path)
        self.seq_len = sequence_length
        self.only_load_end_labels = only_load_end_labels
        # ---- frames: HDF5 'data' [N,20,H,W] uint8 or its raw .npy twin (leod_amd.data.utils.misc) -----------------------------
        ev_dir = get_event_representation_dir(path, ev_representation_name)
        stem = 'event_representations' + ('_ds2_nearest' if downsample_by_factor_2 else '')
        raw, h5 = misc.resolve_link(str(ev_dir / (stem + '.npy'))), misc.resolve_link(str(ev_dir / (stem + '.h5')))
        self.ev_repr_file = Path(raw if os.path.exists(raw) else h5)
        assert self.ev_repr_file.exists(), f'{self.ev_repr_file=}'
        # only the header is read here; the store is opened on the first frame read and shared / bounded per process
        self.num_ev_repr, self.frame_shape = misc.read_frame_header(str(self.ev_repr_file))
        # ---- labels --------------------------------------------------------------------------------------------------------
        labels, objframe_idx_2_label_idx = misc.read_npz_labels(str(path))
        self.label_factory = ObjectLabelFactory.from_structured_array(
            labels, objframe_idx_2_label_idx, get_original_hw(dataset_type), 2 if downsample_by_factor_2 else None)
        self.objframe_idx_2_repr_idx = get_objframe_idx_2_repr_idx(path, ev_representation_name)
        self.repr_idx_2_objframe_idx = {int(r): i for i, r in enumerate(self.objframe_idx_2_repr_idx)}
        self.real_all_objframe_idx, self.all_objframe_idx, self.skip_label = self._subsample_labels(data_ratio, objframe_idx)
        self._kept = set(self.all_objframe_idx)
        self._only_load_labels = False
        self.time_flip = False
        self.time_flip_label_offset = tflip_offset
        self._padding_representation = None

    # ---- frame store ----------------------------------------------------------------------------------------------------------
    @property
    def frames(self):
        """The recording's frame store from the per-process LRU (``misc.FRAME_STORES``): no sequence object owns an fd."""
        return misc.FRAME_STORES.get(str(self.ev_repr_file))

    def read_frames(self, start_idx: int, end_idx: int, out: Optional[np.ndarray] = None, reverse: bool = False) -> np.ndarray:
        """Frames [start_idx, end_idx) as [n,C,H,W] uint8 (into ``out`` if given).  ``reverse``: the time-reversed view --
        frames in reverse order with the channel axis reversed as well (polarity halves swapped, bins reversed; reference:
        time_flip_data :207-227)."""
        assert end_idx > start_idx
        if not reverse:
            return self.frames.read(start_idx, end_idx, out)
        block = self.frames.read(start_idx, end_idx)
        if out is None:
            return np.ascontiguousarray(block[::-1, ::-1])
        np.copyto(out, block[::-1, ::-1])
        return out

    @property
    def padding_representation(self) -> torch.Tensor:
        if self._padding_representation is None:
            self._padding_representation = torch.zeros(self.frame_shape, dtype=torch.uint8)
        return self._padding_representation

    # ---- labels ---------------------------------------------------------------------------------------------------------------
    def _subsample_labels(self, data_ratio: float, objframe_idx: Optional[List[int]]):
        """(every labelled frame, the labelled frames that stay visible, whether any are withheld) -- :117-147: either every
        round(1/ratio)-th labelled frame or an explicit list (an EMPTY list withholds all: pseudo-labelling a skipped recording)."""
        everything = sorted(self.repr_idx_2_objframe_idx.values())
        withhold = (0. < data_ratio < 1.) or (objframe_idx is not None)
        if not withhold:
            return tuple(everything), tuple(everything), False
        if not data_ratio <= 0.5:
            assert isinstance(objframe_idx, list) and len(objframe_idx) == 0, f'Invalid sparse {data_ratio=}'
        if objframe_idx is None:
            kept = everything[::round(1. / data_ratio)]
            if len(kept) == 0:
                kept = [everything[-1]]
        else:
            assert len(objframe_idx) > 0 or data_ratio == -1, 'No subsample label idx provided'
            kept = objframe_idx
        return tuple(everything), tuple(kept), True

    def _get_labels_from_repr_idx(self, repr_idx: int):
        """-> (boxes | None, visible?)"""
        idx = self.repr_idx_2_objframe_idx.get(repr_idx, None)
        if idx is None:
            return None, False
        return self.label_factory[idx], idx in self._kept

    def _load_range_labels(self, start_idx: int, end_idx: int, time_flip: Optional[bool] = None):
        """Labels of frames [start_idx, end_idx): (visible labels | None, withheld labels | None) per frame.  In the
        time-reversed view frame i shows what precedes it, so it takes the label of frame i + tflip_offset (:149-176)."""
        if self.time_flip if time_flip is None else time_flip:
            start_idx, end_idx = start_idx + self.time_flip_label_offset, end_idx + self.time_flip_label_offset
        labels, skipped = [], []
        for repr_idx in range(start_idx, end_idx):
            lab, visible = self._get_labels_from_repr_idx(repr_idx)
            labels.append(lab if visible else None)
            skipped.append(None if visible else lab)
        return labels, skipped

    # ---- sample assembly -----------------------------------------------------------------------------------------------------
    def _ev_repr_list(self, start_idx: int, end_idx: int, pad_front: int, pad_back: int, out: Optional[np.ndarray],
                      reverse: bool = False) -> List[torch.Tensor]:
        """L = pad_front + (end - start) + pad_back frame tensors [C,H,W]: views of ``out`` [L,C,H,W] when given."""
        n = end_idx - start_idx
        if out is None:
            buf = np.zeros((pad_front + n + pad_back,) + self.frame_shape, dtype=np.uint8)
        else:
            buf = out
            assert buf.shape == (pad_front + n + pad_back,) + self.frame_shape and buf.dtype == np.uint8
            if pad_front:
                buf[:pad_front] = 0
            if pad_back:
                buf[pad_front + n:] = 0
        self.read_frames(start_idx, end_idx, out=buf[pad_front:pad_front + n], reverse=reverse)
        return list(torch.from_numpy(buf).unbind(0)) if out is None else [torch.from_numpy(buf[t]) for t in range(buf.shape[0])]

    @staticmethod
    def time_flip_data(data: Dict[DataType, Any]) -> Dict[DataType, Any]:
        """Time-reverse a sample that was assembled in forward order (:207-227)."""
        assert data[DataType.IS_REVERSED]
        data[DataType.EV_IDX].reverse()
        data[DataType.EV_REPR] = [x.flip(0) for x in data[DataType.EV_REPR][::-1]]
        data[DataType.OBJLABELS_SEQ].time_flip_()
        data[DataType.IS_PADDED_MASK].reverse()
        if DataType.SKIPPED_OBJLABELS_SEQ in data:
            data[DataType.SKIPPED_OBJLABELS_SEQ].time_flip_()
        return data

    def _rand_another(self, idx=None) -> Any:
        if idx is None:
            idx = np.random.randint(0, len(self))
        return self[idx]

    def __len__(self) -> int:
        raise NotImplementedError

    def sample(self, index: int, out: Optional[np.ndarray] = None, time_flip: Optional[bool] = None) -> Any:
        """Sample ``index``; ``time_flip`` overrides the instance flag for this call only (several batch slots may stream the
        same recording object concurrently in different directions)."""
        raise NotImplementedError

    def __getitem__(self, index: int) -> Any:
        return self.sample(index)

    def is_only_loading_labels(self) -> bool:
        return self._only_load_labels

    def only_load_labels(self):
        self._only_load_labels = True

    def load_everything(self):
        self._only_load_labels = False
