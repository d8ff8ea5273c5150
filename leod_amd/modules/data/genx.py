"""Data module of the GenX datasets (the interface of the reference's modules/data/genx.py:16-232: ``DataModule(dataset_config,
num_workers_train, num_workers_eval, batch_size_train, batch_size_eval)`` with ``setup(stage)`` and the four
``*_dataloader()`` methods), without torch DataLoader worker processes.

What the reference delegates to DataLoader workers + pin-memory thread + default collate is one producer thread per loader here:
it walks the logical workers round-robin (each logical worker keeps its own slot streams and ``worker_id``, which keys the LSTM
state in the module exactly as a DataLoader worker id does), has a thread pool read the B x L frames of a batch straight into a
pinned ``[L,B,20,H,W]`` uint8 buffer, and keeps ``prefetch`` finished batches queued ahead of the training loop.  At
>= 4000 event-frames/s per GPU the loader has to deliver >= 6 GB/s of voxels: that is page-cache memcpy speed, not
Python-object speed."""
import math
import queue
import threading
from typing import Any, Dict, Iterator, List, Optional

import torch

from leod_amd.data.genx_utils.collate import BatchAssembler
from leod_amd.data.genx_utils.dataset_rnd import CustomConcatDataset, build_random_access_dataset, get_weighted_random_sampler
from leod_amd.data.genx_utils.dataset_streaming import build_streaming_dataset
from leod_amd.data.utils.types import DatasetMode, DatasetSamplingMode
from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY

_END = object()


def dist_rank_world(rank=None, world_size=None):
    """(rank, world_size) of this process in the data-parallel job: the arguments, else torch.distributed, else (0, 1)."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1
    return int(rank), int(world_size)


def shared_seed(world_size: int) -> int:
    """A seed every rank agrees on: drawn on rank 0, broadcast (0 without a process group) -- the role of
    ``DistributedSampler(seed=...)``, which Lightning puts in front of the reference's random-access loader."""
    if world_size <= 1:
        return 0
    import torch.distributed as dist
    box = [int(torch.empty((), dtype=torch.int64).random_().item()) % (1 << 62)]
    if dist.is_available() and dist.is_initialized():
        dist.broadcast_object_list(box, src=0)
    return int(box[0])


def get_dataloading_hw(dataset_config):
    hw = {'gen1': (240, 304), 'gen4': (720, 1280)}[dataset_config.name]
    return tuple(x // 2 for x in hw) if dataset_config.downsample_by_factor_2 else hw


class _PrefetchLoader:
    """Iterable of batch dictionaries produced ahead of time on a background thread."""

    def __init__(self, prefetch: int = 3):
        self.prefetch = prefetch

    def batches(self) -> Iterator[Dict]:                         # pragma: no cover
        raise NotImplementedError

    def __iter__(self):
        if self.prefetch <= 0:
            yield from self.batches()
            return
        q: 'queue.Queue[Any]' = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def produce():
            try:
                for b in self.batches():
                    while not stop.is_set():
                        try:
                            q.put(b, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(_END)
            except BaseException as e:                          # surface loader errors in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True, name='leod-loader')
        th.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:                                                # consumer left (epoch end, break, error): stop and wait for the producer
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)


class StreamLoader(_PrefetchLoader):
    """Batches of a streaming dataset (``ConcatStreamingDataPipe`` for training, ``ShardedStreamingDataPipe`` otherwise): the
    logical workers are served round-robin like DataLoader serves its worker processes; a worker that has run dry drops out."""

    def __init__(self, dataset, num_workers: int, pin_memory: bool = True, prefetch: int = 3, io_threads: int = 8):
        super().__init__(prefetch)
        self.dataset, self.num_workers = dataset, max(1, num_workers)
        first = dataset.datapipe_list[0]
        self.assembler = BatchAssembler(first.seq_len, first.frame_shape, pin_memory, io_threads)
        self.fill_value = getattr(dataset, 'fill_value', None)

    def batches(self):
        live = {w: iter(self.dataset.worker_plans(w, self.num_workers)) for w in range(self.num_workers)}
        while live:
            for w in list(live):
                plans = next(live[w], None)
                if plans is None:
                    del live[w]
                    continue
                yield {DATA_KEY: self.assembler.assemble(plans, self.fill_value), WORKER_ID_KEY: w}


class RandomLoader(_PrefetchLoader):
    """Shuffled (or class-weighted) batches of independent samples; ``drop_last`` as in the reference's training loader."""

    def __init__(self, dataset: CustomConcatDataset, batch_size: int, sampler=None, pin_memory: bool = True, prefetch: int = 3,
                 io_threads: int = 8, num_workers: int = 1, rank: int = 0, world_size: int = 1, seed: int = 0, epoch: int = 0):
        super().__init__(prefetch)
        self.dataset, self.batch_size, self.sampler = dataset, batch_size, sampler
        seq = dataset.datasets[0].sequence
        self.assembler = BatchAssembler(seq.seq_len, seq.frame_shape, pin_memory, io_threads)
        self.num_workers = max(1, num_workers)
        # N > 1 ranks: ONE order per epoch shared by all ranks (generator seeded seed + epoch, the same on every rank), rank r
        # takes order[r::world] -- what DistributedSampler / DistributedSamplerWrapper do for the reference under Lightning DDP
        # ``epoch``: where the shuffle sequence starts -- a loader that is re-created every epoch (a forked ProcessLoader worker, or a
        # trainer that calls train_dataloader() per epoch) is handed the count kept by its creator, a persistent one counts itself
        self.rank, self.world_size, self.seed, self.epoch = rank, max(1, world_size), seed, int(epoch)

    def _per_rank(self) -> int:
        return -(-len(self.dataset) // self.world_size)

    def __len__(self):
        return self._per_rank() // self.batch_size

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _order(self):
        if self.world_size == 1:                                # the global RNG, as the reference's single-GPU loader
            return list(iter(self.sampler)) if self.sampler is not None else torch.randperm(len(self.dataset)).tolist()
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        self.epoch += 1
        if self.sampler is not None:
            self.sampler.generator = g
            order = list(iter(self.sampler))
        else:
            order = torch.randperm(len(self.dataset), generator=g).tolist()
        total = self._per_rank() * self.world_size              # pad by wrapping so that every rank sees the same count
        order += order[:total - len(order)]
        return order[self.rank:total:self.world_size]

    def batches(self):
        order = self._order()
        B = self.batch_size
        for k in range(len(order) // B):
            idx = order[k * B:(k + 1) * B]
            # RNG draws here, one sample after the other; the pool only reads frames; the label transforms follow sequentially
            plans = [_ConcatSlot(*self.dataset.plan(i)).as_plan() for i in idx]
            # worker ids only label batches here (every random sample restarts the LSTM state); cycle them like DataLoader does
            yield {DATA_KEY: self.assembler.assemble(plans), WORKER_ID_KEY: k % self.num_workers}


class _ConcatSlot:
    """Adapter: one planned sample of the concatenated dataset behind the ``(source, index, time_flip, finisher)`` plans of
    the assembler: ``sample`` is the thread-pool part (frame reads), the finisher the sequential label-side augmentation with the
    state drawn for THIS sample."""

    def __init__(self, dataset, local: int, drawn):
        self.dataset, self.local, (self.time_flip, self.state) = dataset, local, drawn

    def sample(self, index, out=None, time_flip=None):
        return self.dataset.read(self.local, out, self.time_flip)

    def finish(self, item):
        return self.dataset.apply_augmentation(item, self.state)

    def as_plan(self):
        return (self, 0, self.time_flip, self.finish if self.state is not None else None)


class MixedLoader:
    """{RANDOM: batch, STREAM: batch} per step until the LONGER loader ends, the shorter one restarting as needed (Lightning's
    ``max_size_cycle`` handling of a dict of loaders, which is what the reference's ``train_dataloader`` returns);
    ``merge_mixed_batches`` in the module concatenates the halves."""

    def __init__(self, loaders: Dict[Any, Any]):
        self.loaders = loaders

    def __iter__(self):
        its = {k: iter(v) for k, v in self.loaders.items()}
        done = {k: False for k in its}
        while True:
            out = {}
            for k in its:
                b = next(its[k], None)
                if b is None:
                    done[k] = True
                    if all(done.values()):
                        return
                    its[k] = iter(self.loaders[k])
                    b = next(its[k])
                out[k] = b
            yield out


class DataModule:
    def __init__(self, dataset_config, num_workers_train: int, num_workers_eval: int, batch_size_train: int,
                 batch_size_eval: int, pin_memory: bool = True, prefetch: int = 3, io_threads: int = 8,
                 worker_process: bool = False, device=None, ring_slots: int = 8, rank: Optional[int] = None,
                 world_size: Optional[int] = None):
        assert num_workers_train >= 0 and num_workers_eval >= 0 and batch_size_train >= 1 and batch_size_eval >= 1
        self.dataset_config = dataset_config
        self.train_sampling_mode = DatasetSamplingMode(dataset_config.train.sampling)
        self.eval_sampling_mode = DatasetSamplingMode(dataset_config.eval.sampling)
        assert self.eval_sampling_mode == DatasetSamplingMode.STREAM
        self.overall_batch_size_train, self.overall_batch_size_eval = batch_size_train, batch_size_eval
        self.overall_num_workers_train, self.overall_num_workers_eval = num_workers_train, num_workers_eval
        self.loader_kw = dict(pin_memory=pin_memory, prefetch=prefetch, io_threads=io_threads)
        # worker_process: batch assembly in a forked process, frames through a HIP-registered shared ring (process_loader.py);
        # with `device` the loaders yield batches whose frames are already on that device (copied one batch ahead)
        self.worker_process, self.device, self.ring_slots = worker_process, device, ring_slots
        # Seeding contract under N > 1 ranks (batch sizes / worker counts here are PER RANK, as in the reference under Lightning DDP):
        #  * random-access loader: one shuffled order per epoch shared by all ranks, rank r takes order[r::world] (RandomLoader);
        #  * everything drawn from the process-global RNGs (stream shuffles, augmentation draws, replacement samples) must differ
        #    between ranks even if the user seeded all ranks identically: the loader's process is seeded base + rank -- the forked
        #    worker of ProcessLoader per epoch, the training process itself once (first train_dataloader()) for the in-process
        #    loaders.  Parameters are unaffected: FlatAdamW broadcasts rank 0's at construction.
        self.rank, self.world_size = dist_rank_world(rank, world_size)
        self._seed_shared: Optional[int] = None
        self._rank_seeded = False
        self._train_loaders_made = 0           # in-process train loaders created so far (their shuffle epoch, see RandomLoader)
        self.sampling_mode_2_dataset: Dict[Any, Any] = {}
        self.sampling_mode_2_train_workers: Dict[Any, int] = {}
        self.sampling_mode_2_train_batch_size: Dict[Any, int] = {}
        self.validation_dataset = self.test_dataset = self.predict_dataset = None

    def get_dataloading_hw(self):
        return get_dataloading_hw(self.dataset_config)

    def set_mixed_sampling_mode_variables_for_train(self):
        """Split batch and workers between the random and the streaming loader by ``train.mixed.w_*`` (:120-144)."""
        B, W = self.overall_batch_size_train, self.overall_num_workers_train
        assert B >= 2, 'Cannot use mixed mode with batch size smaller than 2'
        assert W >= 2, 'Cannot use mixed mode with num workers smaller than 2'
        w_rnd, w_str = self.dataset_config.train.mixed.w_random, self.dataset_config.train.mixed.w_stream
        assert w_rnd > 0 and w_str > 0
        bs_rnd = min(round(B * w_rnd / (w_str + w_rnd)), B - 1)
        workers_rnd = min(math.ceil(W * bs_rnd / B), W - 1)
        self.sampling_mode_2_train_batch_size = {DatasetSamplingMode.RANDOM: bs_rnd, DatasetSamplingMode.STREAM: B - bs_rnd}
        self.sampling_mode_2_train_workers = {DatasetSamplingMode.RANDOM: workers_rnd, DatasetSamplingMode.STREAM: W - workers_rnd}

    def _eval_dataset(self, mode: DatasetMode, pseudo_labeling: bool = False):
        return build_streaming_dataset(dataset_mode=mode, dataset_config=self.dataset_config,
                                       batch_size=self.overall_batch_size_eval, num_workers=self.overall_num_workers_eval,
                                       pseudo_labeling=pseudo_labeling)

    def setup(self, stage: Optional[str] = None) -> None:
        if stage == 'fit':
            mode = self.train_sampling_mode
            if mode == DatasetSamplingMode.MIXED:
                self.set_mixed_sampling_mode_variables_for_train()
            else:
                self.sampling_mode_2_train_workers[mode] = self.overall_num_workers_train
                self.sampling_mode_2_train_batch_size[mode] = self.overall_batch_size_train
            if mode in (DatasetSamplingMode.RANDOM, DatasetSamplingMode.MIXED):
                self.sampling_mode_2_dataset[DatasetSamplingMode.RANDOM] = build_random_access_dataset(
                    dataset_mode=DatasetMode.TRAIN, dataset_config=self.dataset_config)
            if mode in (DatasetSamplingMode.STREAM, DatasetSamplingMode.MIXED):
                self.sampling_mode_2_dataset[DatasetSamplingMode.STREAM] = build_streaming_dataset(
                    dataset_mode=DatasetMode.TRAIN, dataset_config=self.dataset_config,
                    batch_size=self.sampling_mode_2_train_batch_size[DatasetSamplingMode.STREAM],
                    num_workers=self.sampling_mode_2_train_workers[DatasetSamplingMode.STREAM])
            self.validation_dataset = self._eval_dataset(DatasetMode.TESTING)       # the reference validates on the test split (:166-170)
        elif stage == 'validate':
            self.validation_dataset = self._eval_dataset(DatasetMode.VALIDATION)
        elif stage == 'test':
            self.test_dataset = self._eval_dataset(DatasetMode.TESTING)
        elif stage == 'predict':                                # pseudo labels for the training split
            self.predict_dataset = self._eval_dataset(DatasetMode.TRAIN, pseudo_labeling=True)
        else:
            raise NotImplementedError(stage)

    def _in_process(self, build, batch_size: int, takes_epoch: bool = False):
        """``build(**loader_kw)`` as it is, or behind a ``ProcessLoader`` (the worker's loaders then keep one batch of
        look-ahead of their own: the ring is the prefetch queue)."""
        if not self.worker_process:
            return build(**self.loader_kw)
        from leod_amd.modules.data.process_loader import ProcessLoader
        kw = dict(self.loader_kw, pin_memory=False, prefetch=1)
        hw = self.get_dataloading_hw()
        L = self.dataset_config.sequence_length
        slot = L * batch_size * 20 * hw[0] * hw[1]
        fn = (lambda epoch=0: build(epoch=epoch, **kw)) if takes_epoch else (lambda: build(**kw))
        return ProcessLoader(fn, slot_bytes=slot, n_slots=self.ring_slots, device=self.device, rank=self.rank)

    def train_dataloader(self):
        B = max(self.sampling_mode_2_train_batch_size.values())
        if self.world_size > 1:
            if self._seed_shared is None:
                self._seed_shared = shared_seed(self.world_size)
            if not self.worker_process and not self._rank_seeded:
                from leod_amd.modules.data.process_loader import draw_base_seed, seed_worker
                # NOTE: this reseeds the PROCESS-GLOBAL torch / numpy / random generators of the training process (once): the in-process
                # loaders draw shuffles and augmentations from them and the ranks must not draw the same.  A seed_everything() done
                # before is superseded from here on -- said out loud, not silently; worker_process=True keeps the
                # training process's generators untouched.
                import warnings
                base = draw_base_seed()
                warnings.warn(f'leod_amd DataModule: reseeding the global RNGs of rank {self.rank} with {base} + rank for the in-process '
                              'training loaders (use worker_process=True to leave the training process\'s generators alone)')
                seed_worker(base, self.rank)
                self._rank_seeded = True
        if self.worker_process:
            return self._in_process(self._train_loaders, B, takes_epoch=True)        # the ProcessLoader counts its epochs and hands them to the worker
        epoch, self._train_loaders_made = self._train_loaders_made, self._train_loaders_made + 1
        return self._train_loaders(epoch=epoch, **self.loader_kw)

    def _train_loaders(self, epoch: int = 0, **loader_kw):
        loaders = {}
        for mode, dataset in self.sampling_mode_2_dataset.items():
            workers, bs = self.sampling_mode_2_train_workers[mode], self.sampling_mode_2_train_batch_size[mode]
            if mode == DatasetSamplingMode.STREAM:
                loaders[mode] = StreamLoader(dataset, num_workers=workers, **loader_kw)
            else:
                sampler = get_weighted_random_sampler(dataset) if self.dataset_config.train.random.weighted_sampling else None
                loaders[mode] = RandomLoader(dataset, batch_size=bs, sampler=sampler, num_workers=workers, rank=self.rank,
                                             world_size=self.world_size, seed=self._seed_shared or 0, epoch=epoch, **loader_kw)
        return next(iter(loaders.values())) if len(loaders) == 1 else MixedLoader(loaders)

    def _eval_loader(self, dataset):
        return self._in_process(lambda **kw: StreamLoader(dataset, num_workers=self.overall_num_workers_eval, **kw),
                                self.overall_batch_size_eval)

    def val_dataloader(self):
        return self._eval_loader(self.validation_dataset)

    def test_dataloader(self):
        return self._eval_loader(self.test_dataset)

    def predict_dataloader(self):
        return self._eval_loader(self.predict_dataset)
