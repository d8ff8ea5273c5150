"""A recording as a bag of independent samples, one per labelled frame: the labelled frame and the L - 1 frames before it
(the interface of the reference's data/genx_utils/sequence_rnd.py:11-148).  Every sample restarts the LSTM state."""
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

from leod_amd.data.genx_utils.labels import SparselyBatchedObjectLabels
from leod_amd.data.genx_utils.sequence_base import SequenceBase
from leod_amd.data.utils.types import DataType, DatasetType


class SequenceForRandomAccess(SequenceBase):
    def __init__(self, path: Path, ev_representation_name: str, sequence_length: int, dataset_type: DatasetType,
                 downsample_by_factor_2: bool, only_load_end_labels: bool, objframe_idx: Optional[List[int]] = None,
                 data_ratio: float = -1.0, tflip_offset: int = -1):
        shift_kept = objframe_idx is None
        super().__init__(path=path, ev_representation_name=ev_representation_name, sequence_length=sequence_length,
                         dataset_type=dataset_type, downsample_by_factor_2=downsample_by_factor_2,
                         only_load_end_labels=only_load_end_labels, objframe_idx=objframe_idx, data_ratio=data_ratio,
                         tflip_offset=tflip_offset)
        assert not self.only_load_end_labels
        # the first labelled frame with L - 1 frames of history in front of it
        enough = np.flatnonzero(np.asarray(self.objframe_idx_2_repr_idx) - self.seq_len + 1 >= 0)
        if len(enough) == 0:
            self.start_idx_offset, self.length = None, 0        # the recording is skipped
            return
        self.start_idx_offset = int(enough[0])
        if self.skip_label and shift_kept and self.start_idx_offset > 0:
            # the kept frames were sub-sampled from the first label on: move them behind the offset (:46-51)
            real = set(self.real_all_objframe_idx)
            self.all_objframe_idx = tuple(i + self.start_idx_offset for i in self.all_objframe_idx if i + self.start_idx_offset in real)
            self._kept = set(self.all_objframe_idx)
        self.same_last_idx = self.all_objframe_idx[-1] == self.real_all_objframe_idx[-1]
        self.length = len(self.label_factory) - self.start_idx_offset
        assert len(self.label_factory) == len(self.objframe_idx_2_repr_idx)

    def __len__(self):
        return self.length

    def sample(self, index: int, out: Optional[np.ndarray] = None, time_flip: Optional[bool] = None) -> Dict:
        L = self.seq_len
        time_flip = self.time_flip if time_flip is None else time_flip
        if time_flip:
            # reversed view: the labelled frame should come as LATE as possible, i.e. the window starts at it (in forward
            # indexing) and runs L frames ahead; the very last labelled frame has nothing after it -> draw another sample
            objframe = index
            if objframe == self.real_all_objframe_idx[-1]:
                return self._rand_another(idx=objframe, out=out, time_flip=time_flip)
            label_repr_idx = int(self.objframe_idx_2_repr_idx[objframe]) - self.time_flip_label_offset
            end_idx = min(self.num_ev_repr, label_repr_idx + L)
        else:
            objframe = index + self.start_idx_offset
            end_idx = int(self.objframe_idx_2_repr_idx[objframe]) + 1
        start_idx = end_idx - L
        assert start_idx >= 0, f'{self.ev_repr_file=}, {self.start_idx_offset=}, {start_idx=}, {end_idx=}'
        labels, skipped = self._load_range_labels(start_idx, end_idx, time_flip)
        if all(l is None for l in labels):                      # every label in the window is withheld
            return self._rand_another(out=out, time_flip=time_flip)
        ev_idx = list(range(start_idx, end_idx))
        if time_flip:
            ev_idx.reverse(); labels.reverse(); skipped.reverse()
        sample = {DataType.OBJLABELS_SEQ: SparselyBatchedObjectLabels(labels),
                  DataType.SKIPPED_OBJLABELS_SEQ: SparselyBatchedObjectLabels(skipped)}
        if self._only_load_labels:
            return sample
        sample.update({DataType.PATH: self.path, DataType.EV_IDX: ev_idx,
                       DataType.EV_REPR: self._ev_repr_list(start_idx, end_idx, 0, 0, out, reverse=time_flip),
                       DataType.IS_FIRST_SAMPLE: True, DataType.IS_LAST_SAMPLE: False, DataType.IS_REVERSED: time_flip,
                       DataType.IS_PADDED_MASK: [False] * L})
        return sample

    def _rand_another(self, idx=None, out: Optional[np.ndarray] = None, time_flip: Optional[bool] = None) -> Any:
        """Replacement draw (:119-148; numpy's global RNG, like the reference).  Without withheld labels this only happens
        for the last labelled frame in the reversed view: any other frame will do.  With withheld labels: one of the kept
        frames (not the last one in the reversed view when it is the recording's last label)."""
        time_flip = self.time_flip if time_flip is None else time_flip
        if not self.skip_label:
            assert time_flip, 'only happens when `time_flip` is True'
            assert idx == self.real_all_objframe_idx[-1], 'only happens when trying to load the last labeled frame'
            return self.sample(int(np.random.choice(len(self) - 1, 1)[0]), out=out, time_flip=time_flip)
        pool = self.all_objframe_idx[:-1] if (time_flip and self.same_last_idx) else self.all_objframe_idx
        idx = int(np.random.choice(pool, 1)[0])
        if not time_flip:
            idx -= self.start_idx_offset
        return self.sample(idx, out=out, time_flip=time_flip)
