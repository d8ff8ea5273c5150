"""A recording as a stream of consecutive L-frame samples (the interface of the reference's
data/genx_utils/sequence_streaming.py:54-277): ``is_first_sample`` tells the detector to reset its LSTM state, the following
samples continue where the previous one stopped, the last one is zero-padded to L frames (``is_padded_mask``)."""
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from leod_amd.data.genx_utils.labels import SparselyBatchedObjectLabels
from leod_amd.data.genx_utils.sequence_base import SequenceBase, get_objframe_idx_2_repr_idx
from leod_amd.data.utils.types import DataType, DatasetType


def _get_ev_repr_range_indices(indices: np.ndarray, max_len: int) -> List[Tuple[int, int]]:
    """Split the sorted labelled-frame indices wherever two neighbours are more than ``max_len`` frames apart; each run
    becomes a frame range [first - max_len + 1 (clipped at 0), last + 1) (:23-51)."""
    indices = np.asarray(indices)
    cuts = np.flatnonzero(np.diff(indices) > max_len)
    firsts = np.concatenate(([0], cuts + 1))
    lasts = np.concatenate((cuts, [len(indices) - 1]))
    return [(max(int(indices[a]) - max_len + 1, 0), int(indices[b]) + 1) for a, b in zip(firsts, lasts)]


class SequenceForIter(SequenceBase):
    def __init__(self, path: Path, ev_representation_name: str, sequence_length: int, dataset_type: DatasetType,
                 downsample_by_factor_2: bool, range_indices: Optional[Tuple[int, int]] = None,
                 objframe_idx: Optional[List[int]] = None, data_ratio: float = -1.0, tflip_offset: int = -1,
                 start_from_zero: bool = False):
        if 0. < data_ratio < 1.:
            assert len(objframe_idx) > 0, 'Should specify `objframe_idx` for streaming data'
        super().__init__(path=path, ev_representation_name=ev_representation_name, sequence_length=sequence_length,
                         dataset_type=dataset_type, downsample_by_factor_2=downsample_by_factor_2, only_load_end_labels=False,
                         objframe_idx=objframe_idx, data_ratio=data_ratio, tflip_offset=tflip_offset)
        if len(self.objframe_idx_2_repr_idx) == 0 and not start_from_zero:
            self.length = 0                                     # nothing labelled: the recording is skipped
            return
        L, n = sequence_length, self.num_ev_repr
        # the first sample ends no later than the first labelled frame (or starts at frame 0 when asked to)
        earliest = 0 if start_from_zero else max(int(self.objframe_idx_2_repr_idx[0]) - L + 1, 0)
        lo, hi = (earliest, n) if range_indices is None else range_indices
        assert 0 <= earliest <= lo < hi <= n, f'{earliest=}, {lo=}, {hi=}, {n=}, {path=}'
        self.start_indices = list(range(lo, hi, L))
        self.stop_indices = self.start_indices[1:] + [hi]
        self.length = len(self.start_indices)
        # time-reversed view: samples are cut from the END of the range backwards, e.g. lo 0, hi 21, L 10 -> [11,21) [1,11) [0,1)
        ends = list(range(hi, lo, -L))
        self.time_flip_stop_indices = ends
        self.time_flip_start_indices = ends[1:] + [lo]

    @staticmethod
    def get_sequences_with_guaranteed_labels(path: Path, ev_representation_name: str, sequence_length: int,
                                             dataset_type: DatasetType, downsample_by_factor_2: bool,
                                             tflip_offset: int = -1) -> List['SequenceForIter']:
        """Training streams: one ``SequenceForIter`` per run of labelled frames, so that every sample holds at least one
        label somewhere among its L frames (:127-161)."""
        objframe_idx_2_repr_idx = get_objframe_idx_2_repr_idx(path=path, ev_representation_name=ev_representation_name)
        if len(objframe_idx_2_repr_idx) == 0:
            return []
        return [SequenceForIter(path=path, ev_representation_name=ev_representation_name, sequence_length=sequence_length,
                                dataset_type=dataset_type, downsample_by_factor_2=downsample_by_factor_2, range_indices=r,
                                tflip_offset=tflip_offset)
                for r in _get_ev_repr_range_indices(objframe_idx_2_repr_idx, sequence_length)]

    def get_fully_padded_sample(self) -> Dict:
        """What an exhausted batch slot yields while other slots still stream (ShardedStreamingDataPipe fill value)."""
        none = SparselyBatchedObjectLabels([None] * self.seq_len)
        return {DataType.PATH: '', DataType.EV_IDX: [-1] * self.seq_len, DataType.EV_REPR: [self.padding_representation] * self.seq_len,
                DataType.OBJLABELS_SEQ: none, DataType.SKIPPED_OBJLABELS_SEQ: none, DataType.IS_FIRST_SAMPLE: False,
                DataType.IS_LAST_SAMPLE: False, DataType.IS_REVERSED: False, DataType.IS_PADDED_MASK: [True] * self.seq_len}

    def __len__(self):
        return self.length

    def sample(self, index: int, out: Optional[np.ndarray] = None, time_flip: Optional[bool] = None) -> Dict:
        """Sample ``index``: frames [start, stop) of the recording padded to L.  ``out`` [L,C,H,W] uint8 (e.g. a slot of a
        pinned batch buffer) receives the frames; EV_REPR are then views of it."""
        time_flip = self.time_flip if time_flip is None else time_flip
        if time_flip:
            start_idx, end_idx = self.time_flip_start_indices[index], self.time_flip_stop_indices[index]
        else:
            start_idx, end_idx = self.start_indices[index], self.stop_indices[index]
        n, L = end_idx - start_idx, self.seq_len
        assert L >= n > 0, f'{L=}, {n=}, {start_idx=}, {end_idx=}'
        pad = L - n
        labels, skipped = self._load_range_labels(start_idx, end_idx, time_flip)
        ev_idx = list(range(start_idx, end_idx))
        if time_flip:                           # reversed order; the padding still comes last
            ev_idx.reverse(); labels.reverse(); skipped.reverse()
        ev_idx += [-1] * pad
        labels += [None] * pad
        skipped += [None] * pad
        sample = {
            DataType.PATH: self.path, DataType.EV_IDX: ev_idx,
            DataType.OBJLABELS_SEQ: SparselyBatchedObjectLabels(labels),
            DataType.SKIPPED_OBJLABELS_SEQ: SparselyBatchedObjectLabels(skipped),
            DataType.IS_FIRST_SAMPLE: index == 0, DataType.IS_LAST_SAMPLE: index == self.length - 1,
            DataType.IS_REVERSED: time_flip, DataType.IS_PADDED_MASK: [False] * n + [True] * pad}
        if self._only_load_labels:
            sample[DataType.EV_REPR] = [self.padding_representation] * L
        else:
            sample[DataType.EV_REPR] = self._ev_repr_list(start_idx, end_idx, 0, pad, out, reverse=time_flip)
        return sample
