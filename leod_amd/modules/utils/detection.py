"""Loop bookkeeping around the detector (the interface of the reference's modules/utils/detection.py):
per-dataloader-worker LSTM state store with partial reset, labelled-frame feature gathering, mixed
(stream + random) batch merging."""
from enum import Enum, auto
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch as th

from leod_amd.data.genx_utils.labels import SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DatasetSamplingMode


class Mode(Enum):
    TRAIN = auto()
    VAL = auto()
    TEST = auto()


mode_2_string = {Mode.TRAIN: 'train', Mode.VAL: 'val', Mode.TEST: 'test'}
WORKER_ID_KEY = 'worker_id'
DATA_KEY = 'data'


def _map_tensors(obj, fn):
    if isinstance(obj, th.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    raise NotImplementedError(type(obj))


class BackboneFeatureSelector:
    """Collects, per stage, the feature maps of the frames that carry labels (detection.py:209-224 of the
    reference module) and concatenates them on the batch dim for one head pass."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.features: Dict[int, List[th.Tensor]] = {}

    def add_backbone_features(self, backbone_features: Dict[int, th.Tensor], selected_indices: Optional[List[int]] = None):
        if selected_indices is not None and len(selected_indices) == 0:
            return
        for k, v in backbone_features.items():
            sel = v if selected_indices is None else v[selected_indices]
            self.features.setdefault(k, []).append(sel)

    def get_batched_backbone_features(self) -> Optional[Dict[int, th.Tensor]]:
        if not self.features:
            return None
        return {k: th.cat(v, dim=0) for k, v in self.features.items()}

    def is_empty(self):
        return not self.features


class EventReprSelector:
    def __init__(self):
        self.repr_list: List[th.Tensor] = []

    def reset(self):
        self.repr_list = []

    def __len__(self):
        return len(self.repr_list)

    def add_ev_repr(self, ev_repr: th.Tensor, selected_indices: Optional[List[int]] = None):
        sel = ev_repr if selected_indices is None else ev_repr[selected_indices]
        self.repr_list.extend(sel.unbind(0))

    def get_ev_repr_as_list(self, start_idx: int = 0, end_idx: Optional[int] = None):
        if len(self) == 0:
            return None
        return self.repr_list[start_idx:len(self) if end_idx is None else end_idx]


class RNNStates:
    """{worker_id: [(h, c)] * stages}; ``reset`` zeroes the rows of samples that start a new sequence IN PLACE
    (the stored tensors are detached), ``save_states_and_detach`` truncates BPTT at the sample boundary."""

    def __init__(self):
        self.states: Dict[int, Any] = {}

    @classmethod
    def recursive_detach(cls, inp):
        return _map_tensors(inp, lambda t: t.detach())

    @classmethod
    def recursive_reset(cls, inp, indices_or_bool_tensor=None):
        def zero(t):
            assert t.requires_grad is False
            if indices_or_bool_tensor is None:
                t[:] = 0
            elif th.is_tensor(indices_or_bool_tensor) and indices_or_bool_tensor.dtype == th.bool:
                # same rows as ``t[mask] = 0``, written as a masked fill: no index list, no host synchronisation
                assert len(indices_or_bool_tensor) > 0
                mask = indices_or_bool_tensor.to(t.device, non_blocking=True)
                t.masked_fill_(mask.view((-1,) + (1,) * (t.dim() - 1)), 0)
            else:
                assert len(indices_or_bool_tensor) > 0
                t[indices_or_bool_tensor] = 0
            return t
        if th.is_tensor(indices_or_bool_tensor) and indices_or_bool_tensor.dtype == th.bool:
            # device states: the masked reset of ALL tensors (h and c of every stage) is one kernel launch
            ts = []
            _map_tensors(inp, lambda t: ts.append(t) or t)
            if ts and all(t.is_cuda and not t.requires_grad for t in ts):
                from leod_amd import ops
                mask = indices_or_bool_tensor.to(ts[0].device, non_blocking=True).contiguous()
                if ops.multi_ok(ts) and all(t.shape[0] == mask.numel() for t in ts):
                    assert mask.numel() > 0
                    ops.rows_masked_zero(ts, mask)
                    return inp
        return _map_tensors(inp, zero)

    def save_states_and_detach(self, worker_id: int, states) -> None:
        self.states[worker_id] = self.recursive_detach(states)

    def get_states(self, worker_id: int):
        return self.states.get(worker_id, None)

    def reset(self, worker_id: int, indices_or_bool_tensor=None):
        if worker_id in self.states:
            self.states[worker_id] = self.recursive_reset(self.states[worker_id], indices_or_bool_tensor)


class SeqLens:
    def __init__(self):
        self.lens: Dict[int, th.Tensor] = {}

    def update_lens(self, worker_id: int, lens: th.Tensor) -> None:
        self.lens[worker_id] = lens if worker_id not in self.lens else self.lens[worker_id] + lens

    def get_lens(self, worker_id: int) -> Optional[th.Tensor]:
        return self.lens.get(worker_id, None)

    def reset(self, worker_id: int, indices_or_bool_tensor=None):
        if worker_id not in self.lens:
            self.lens[worker_id] = th.zeros(len(indices_or_bool_tensor)).long()
        elif indices_or_bool_tensor is None:
            self.lens[worker_id] = th.zeros_like(self.lens[worker_id])
        else:
            self.lens[worker_id][indices_or_bool_tensor] = 0


def mixed_collate_fn(x1, x2):
    if isinstance(x1, th.Tensor):
        return th.cat((x1, x2))
    if isinstance(x1, SparselyBatchedObjectLabels):
        return x1 + x2
    if isinstance(x1, list):
        # per-timestep lists (frames, indices, masks, label containers) merge element-wise on the batch dim; per-sample
        # lists (paths, augmentation states) concatenate
        if len(x1) and isinstance(x1[0], (th.Tensor, SparselyBatchedObjectLabels, list)):
            assert len(x1) == len(x2)
            return [mixed_collate_fn(a, b) for a, b in zip(x1, x2)]
        return x1 + x2
    if isinstance(x1, dict):
        return {k: (mixed_collate_fn(x1[k], x2[k]) if isinstance(x1[k], dict) else x1[k] + x2[k]) for k in x1}
    raise NotImplementedError(type(x1))


def merge_mixed_batches(batch: Dict[str, Any]):
    """cat(stream half, random half); the worker id (state key) is the streaming loader's."""
    if DATA_KEY in batch:
        return batch
    rnd_data = batch[DatasetSamplingMode.RANDOM][DATA_KEY]
    stream_batch = batch[DatasetSamplingMode.STREAM]
    stream_data = stream_batch[DATA_KEY]
    assert rnd_data.keys() == stream_data.keys()
    return {WORKER_ID_KEY: stream_batch[WORKER_ID_KEY],
            DATA_KEY: {k: mixed_collate_fn(stream_data[k], rnd_data[k]) for k in rnd_data}}
