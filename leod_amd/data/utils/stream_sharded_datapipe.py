"""Evaluation / pseudo-labelling streams: whole recordings dealt to (rank, worker) pairs and, inside a worker, to the B batch
slots -- no recording is seen twice, exhausted slots are filled with padding samples (the interface of the reference's
data/utils/stream_sharded_datapipe.py:11-117, without torchdata: a worker's stream is a plain generator of per-slot sample
*plans* that the batch loader turns into frames inside a pinned buffer)."""
from typing import Any, Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist


def pyramid_indices(n: int) -> Iterator[int]:
    """0, 1, .., n-1, n-1, .., 1, 0, 0, 1, ...  -- dealing a length-sorted list this way balances the total length per bin."""
    while True:
        yield from range(n)
        yield from range(n - 1, -1, -1)


class ShardedStreamingDataPipe:
    def __init__(self, datapipe_list: List[Any], batch_size: int, fill_value: Optional[Any] = None):
        assert batch_size > 0
        self.datapipe_list = sorted(datapipe_list, key=len, reverse=True)          # long -> short (stable)
        self.batch_size = batch_size
        self.fill_value = fill_value

    yield_pyramid_indices = staticmethod(lambda start_idx, end_idx: (start_idx + i for i in pyramid_indices(end_idx - start_idx)))

    @classmethod
    def assign_datapipes_to_worker(cls, sorted_datapipe_list: List[Any], total_num_workers: int, global_worker_id: int) -> List[Any]:
        """The recordings of worker ``global_worker_id`` = rank * workers_per_rank + local id (:40-57)."""
        n = len(sorted_datapipe_list)
        assert n >= total_num_workers > global_worker_id, f'{n=}, {total_num_workers=}, {global_worker_id=}'
        owner = pyramid_indices(total_num_workers)
        return [dp for dp in sorted_datapipe_list if next(owner) == global_worker_id]

    def slot_streams(self, datapipe_list: List[Any], batch_size: int) -> List[List[Any]]:
        """The recordings each batch slot streams one after the other (:59-86)."""
        assert len(datapipe_list) >= batch_size > 0, \
            "Each worker must at least get 'batch_size' number of datapipes; decrease the number of workers."
        slots: List[List[Any]] = [[] for _ in range(batch_size)]
        slot = pyramid_indices(batch_size)
        for dp in sorted(datapipe_list, key=len, reverse=True):
            slots[next(slot)].append(dp)
        return slots

    @staticmethod
    def world() -> Tuple[int, int]:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def datapipes_of_worker(self, local_worker_id: int, local_num_workers: int) -> List[Any]:
        """The reference deals recordings to the global worker id rank * workers + worker (:88-105).  It only ever ran the
        time-flip TTA on one GPU (predict.py:167-169): with several ranks a recording and its time-reversed copy (two streams
        with the same ``path``) would land on different ranks and their detections could never be merged
        (``EventSeqData`` / ``EventSeqResult`` are per process).  When a path occurs more than once, ranks therefore own whole
        path groups (``leod_amd.parallel.shard_sequences``: length-balanced, no data-path collective) and only the dealing to
        the rank's own workers follows the reference."""
        rank, world = self.world()
        paths = [getattr(dp, 'path', id(dp)) for dp in self.datapipe_list]
        if world > 1 and len(set(paths)) < len(paths):
            from leod_amd.parallel import shard_sequences
            own = shard_sequences([len(dp) for dp in self.datapipe_list], world, rank, keys=paths)
            return self.assign_datapipes_to_worker([self.datapipe_list[i] for i in own], local_num_workers, local_worker_id)
        return self.assign_datapipes_to_worker(self.datapipe_list, local_num_workers * world, rank * local_num_workers + local_worker_id)

    def worker_plans(self, local_worker_id: int, local_num_workers: int) -> Iterator[List[Optional[Tuple[Any, int, Optional[bool]]]]]:
        """Batches of worker ``local_worker_id`` on this rank: per slot ``(recording, sample index, None)`` or ``None`` once
        the slot has run dry (ZipperLongest with the padding sample as fill value); ends when every slot is dry."""
        mine = self.datapipes_of_worker(local_worker_id, local_num_workers)
        slots = self.slot_streams(mine, self.batch_size)
        cursors = [((dp, i) for dp in stream for i in range(len(dp))) for stream in slots]
        while True:
            plans = [next(c, None) for c in cursors]
            if all(p is None for p in plans):
                return
            yield [None if p is None else (p[0], p[1], None) for p in plans]
