"""Training streams: every batch slot walks through ALL training sub-sequences in its own random order, so a batch always has
B live samples and every slot carries LSTM state from one sample to the next (the interface of the reference's
data/utils/stream_concat_datapipe.py:24-108, without torchdata).  An epoch ends when the first slot has seen everything."""
from typing import Any, Callable, Iterator, List, Optional, Tuple

import torch as th


class ConcatStreamingDataPipe:
    def __init__(self, datapipe_list: List[Any], batch_size: int, num_workers: int,
                 augmentation_pipeline: Optional[Callable[[], Any]] = None, print_seed_debug: bool = False):
        assert batch_size > 0
        self.datapipe_list = datapipe_list
        self.batch_size = batch_size
        self.new_augmentor = augmentation_pipeline           # () -> RandomSpatialAugmentorGenX (one per slot) or None
        self.print_seed_debug = print_seed_debug

    @staticmethod
    def random_torch_shuffle_list(data: List[Any]) -> Iterator[Any]:
        assert isinstance(data, list)
        return (data[i] for i in th.randperm(len(data)).tolist())

    def _slot(self, order: Iterator[Any], augmentor) -> Iterator[Tuple[Any, int, bool, Any]]:
        for dp in order:
            time_flip = False
            if augmentor is not None:
                # one draw per sub-sequence: the same flip / zoom for all of its samples; the time flip is a property of how
                # the sub-sequence is read, so it leaves the state here (sequence_streaming.py:296-307)
                augmentor.randomize_augmentation()
                time_flip = augmentor.augm_state.apply_t_flip
                augmentor.augm_state.apply_t_flip = False
            for i in range(len(dp)):
                yield dp, i, time_flip, augmentor

    def worker_plans(self, local_worker_id: int = 0, local_num_workers: int = 1):
        """Batches of one worker: per slot ``(sub-sequence, sample index, time_flip, augmentor)``.  The shuffles are drawn
        here, inside the worker, so that workers differ (:63-77)."""
        orders = [self.random_torch_shuffle_list(self.datapipe_list) for _ in range(self.batch_size)]
        slots = [self._slot(o, self.new_augmentor() if self.new_augmentor is not None else None) for o in orders]
        while True:
            plans = [next(s, None) for s in slots]
            if any(p is None for p in plans):               # Zipper: the shortest slot ends the epoch
                return
            yield plans
