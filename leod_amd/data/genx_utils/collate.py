"""Batch assembly for the GenX loaders (what ``custom_collate_streaming`` / ``custom_collate_rnd`` + PyTorch's default collate do
for the reference, data/genx_utils/collate.py:53-80), MI355X-first: the B samples of a batch are produced straight into ONE
pinned ``[L,B,C,H,W]`` uint8 tensor (B parallel reads from the frame stores), so the batch reaches the device as a single
PCIe copy and ``EV_REPR`` is the list of its L frame views -- no per-sample tensors, no stacking, no fp32 cast."""
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from leod_amd.data.genx_utils.labels import SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType
from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY

_LIST_KEYS = (DataType.EV_IDX, DataType.IS_PADDED_MASK)
_FLAG_KEYS = (DataType.IS_FIRST_SAMPLE, DataType.IS_LAST_SAMPLE, DataType.IS_REVERSED)
_LABEL_KEYS = (DataType.OBJLABELS_SEQ, DataType.SKIPPED_OBJLABELS_SEQ)


def collate_samples(samples: Sequence[Dict], ev_buffer: Optional[torch.Tensor] = None) -> Dict:
    """B loader samples -> batch dictionary with the reference's layout: EV_REPR / EV_IDX / IS_PADDED_MASK are lists over
    the L timesteps of [B,...] tensors, label sequences are L ``SparselyBatchedObjectLabels`` of B entries, flags are [B]
    tensors, PATH a list of B strings, AUGM_STATE the list of B augmentation states.  With ``ev_buffer`` [L,B,C,H,W] (the
    samples' frames already live in it) EV_REPR is the list of its frame views."""
    first = samples[0]
    out: Dict[Any, Any] = {}
    for k in first:
        vals = [s[k] for s in samples]
        if k == DataType.EV_REPR:
            if ev_buffer is not None:
                out[k] = [ev_buffer[t] for t in range(ev_buffer.shape[0])]
            else:
                out[k] = [torch.stack([v[t] for v in vals]) for t in range(len(vals[0]))]
        elif k in _LABEL_KEYS:
            out[k] = SparselyBatchedObjectLabels.transpose_list(vals)
        elif k in _LIST_KEYS:
            out[k] = [torch.tensor([v[t] for v in vals]) for t in range(len(vals[0]))]
        elif k in _FLAG_KEYS:
            out[k] = torch.tensor(vals, dtype=torch.bool)
        else:                                                   # PATH, AUGM_STATE
            out[k] = list(vals)
    return out


class BatchAssembler:
    """Turns per-slot sample plans ``(sequence, index, time_flip[, augmentor])`` into a collated batch whose frames sit in
    one pinned buffer.  The B reads run on a thread pool (memory-mapped copies release the GIL)."""

    def __init__(self, seq_len: int, frame_shape: Tuple[int, int, int], pin_memory: bool = True, io_threads: int = 8):
        self.seq_len, self.frame_shape, self.pin = seq_len, tuple(frame_shape), pin_memory and torch.cuda.is_available()
        self.pool = ThreadPoolExecutor(max_workers=io_threads) if io_threads > 1 else None

    buffer_source = None     # process-wide override: callable(shape) -> uint8 tensor (modules/data/process_loader.py: ring slots)

    def new_buffer(self, batch_size: int) -> torch.Tensor:
        if BatchAssembler.buffer_source is not None:
            return BatchAssembler.buffer_source((self.seq_len, batch_size) + self.frame_shape)
        # a fresh pinned tensor per batch: torch's caching host allocator recycles the blocks and never hands one out while
        # an asynchronous copy that reads it is still in flight
        return torch.empty((self.seq_len, batch_size) + self.frame_shape, dtype=torch.uint8, pin_memory=self.pin)

    def assemble(self, plans: Sequence[Optional[tuple]], fill_value: Optional[Dict] = None) -> Dict:
        B = len(plans)
        buf = self.new_buffer(B)
        view = buf.numpy()

        def produce(b):
            plan = plans[b]
            if plan is None:                                    # exhausted slot: padding sample, zero frames
                view[:, b] = 0
                return dict(fill_value)
            seq, index, time_flip = plan[:3]
            sample = seq.sample(index, out=view[:, b], time_flip=time_flip)
            return sample

        samples = list(self.pool.map(produce, range(B))) if self.pool is not None else [produce(b) for b in range(B)]
        for b, plan in enumerate(plans):                        # label-side augmentation: sequential (draws from the global RNG)
            if plan is not None and len(plan) > 3 and plan[3] is not None:
                samples[b] = plan[3](samples[b])
        return collate_samples(samples, ev_buffer=buf)


def custom_collate_streaming(batch: Tuple[List[Dict], int]) -> Dict:
    samples, worker_id = batch
    assert isinstance(worker_id, int)
    return {DATA_KEY: collate_samples(samples), WORKER_ID_KEY: worker_id}


def custom_collate_rnd(batch: List[Dict], worker_id: int = 0) -> Dict:
    return {DATA_KEY: collate_samples(batch), WORKER_ID_KEY: worker_id}
