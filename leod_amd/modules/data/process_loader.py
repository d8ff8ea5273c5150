"""Batch assembly in a worker PROCESS, frames handed over through a ring of shared, HIP-registered (pinned) slots.

Why: the in-process loaders (``genx.StreamLoader`` / ``RandomLoader`` / ``MixedLoader``) run their plan / label / collate code on a
thread of the training process.  At the bf16 step rate (24 ms per 245 MB batch, ~570 kernel launches per step, each a ctypes call
that drops the GIL) that thread and the launch thread trade the GIL hundreds of times per step: the loader-fed rate fell to 0.6-0.8x
of the HBM-resident rate although the loader alone keeps up (``profiles/r02_r_*``).  The reference gets the isolation from DataLoader
worker processes (modules/data/genx.py:172-232), paying a pickle + a pin-memory-thread copy per batch; here

* ONE forked worker runs the unchanged loader (so the batches are the ones the in-process loader yields), with
  ``BatchAssembler`` writing the frames straight into a slot of a shared-memory ring (``BatchAssembler.buffer_source``);
* the ring is registered with HIP in the training process (``hipHostRegister``), so a slot is DMA-able as it is: the frames cross
  the process boundary and PCIe without any further host copy;
* everything else of a batch (labels, flags, indices: a few KB) travels as numpy arrays over a pipe;
* with ``device`` given the training process issues the slot -> HBM copy on its own stream one batch ahead (what
  ``DevicePrefetcher`` does for the in-process loaders) and returns the slot to the worker once the copy's event has completed.

Without a device (CPU tests, host-side consumers) the frames of a yielded batch are views of its slot and stay valid until
``hold`` further batches have been drawn."""
import collections
import multiprocessing as mp
import pickle
import queue
import threading
import time
import traceback
from typing import Any, Callable, Dict, Iterable, List, Optional

import numpy as np
import torch

from leod_amd.data.genx_utils.collate import BatchAssembler
from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType

_END, _ERR, _BATCH = 0, 1, 2


# ---- transport encoding: tensors -> numpy, ring-resident frames -> (slot, shape) ---------------------------------------------------
class _Ring:
    def __init__(self, n_slots: int, slot_bytes: int):
        self.n_slots, self.slot_bytes = n_slots, int(slot_bytes)
        self.mem = torch.empty(n_slots * self.slot_bytes, dtype=torch.uint8).share_memory_()
        self.base = self.mem.data_ptr()

    def slot_of(self, t: torch.Tensor) -> Optional[int]:
        off = t.data_ptr() - self.base
        return off // self.slot_bytes if 0 <= off < self.n_slots * self.slot_bytes else None

    def view(self, slot: int, shape) -> torch.Tensor:
        """[L,B,C,H,W] tensor over the head of a slot; built through numpy so that it is a ROOT tensor (its frame views then have
        it as ``_base``, which is how ``Module._stack_frames`` recognises a batch that needs no stacking)."""
        n = int(np.prod(shape))
        assert 0 <= slot < self.n_slots and n <= self.slot_bytes, (slot, shape, self.slot_bytes)
        return torch.from_numpy(self.mem.numpy()[slot * self.slot_bytes: slot * self.slot_bytes + n].reshape(shape))


class _Stacked(list):
    """A list of L tensors that are the rows of ONE stacked tensor ``base`` (frames of a ring slot, per-timestep index / mask
    tensors): moved to the device with one copy."""
    base: Optional[torch.Tensor] = None


class _Packer:
    """Batch dictionary -> (structure of plain tuples, one byte blob holding every small array, ring slots used).  Lists of
    equally shaped tensors (EV_IDX, IS_PADDED_MASK: L tensors of [B]) travel stacked: one array, and later one device copy."""

    def __init__(self, ring: _Ring):
        self.ring, self.used, self.chunks, self.table, self.size = ring, [], [], [], 0

    def _array(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        pad = (-self.size) % 16
        if pad:
            self.chunks.append(b'\0' * pad); self.size += pad
        self.table.append((self.size, a.dtype.str, a.shape))
        self.chunks.append(a.tobytes()); self.size += a.nbytes
        return len(self.table) - 1

    def pack(self, o: Any, key=None):
        ring = self.ring
        if isinstance(o, list) and len(o) > 1 and all(torch.is_tensor(e) for e in o) and \
                all(e.shape == o[0].shape and e.dtype == o[0].dtype for e in o):
            step = o[0].numel() * o[0].element_size()
            slot = ring.slot_of(o[0])
            if slot is not None and o[0].data_ptr() == ring.base + slot * ring.slot_bytes and \
                    all(e.is_contiguous() and e.data_ptr() == o[0].data_ptr() + t * step for t, e in enumerate(o)):
                self.used.append(slot)
                return ('ring', slot, (len(o),) + tuple(o[0].shape))
            return ('stack', self._array(torch.stack(o).numpy()))
        if torch.is_tensor(o):
            return ('a', self._array(o.numpy()))
        if isinstance(o, np.ndarray) and o.dtype != object:
            return ('n', self._array(o))
        if isinstance(o, ObjectLabels):
            lab = o.object_labels
            return ('ol', self._array(lab.numpy() if torch.is_tensor(lab) else lab), o.input_size_hw)
        if isinstance(o, SparselyBatchedObjectLabels):
            return ('sb', [None if l is None else self.pack(l) for l in o.sparse_object_labels_batch])
        if isinstance(o, dict):
            return ('dict', [(k, self.pack(v, k)) for k, v in o.items()])
        if isinstance(o, list):
            return ('list', [self.pack(v) for v in o])
        if isinstance(o, tuple):
            return ('tuple', [self.pack(v) for v in o])
        return ('obj', o)

    def payload(self, batch) -> bytes:
        tree = self.pack(batch)
        return pickle.dumps((tree, self.table, b''.join(self.chunks), self.used), protocol=pickle.HIGHEST_PROTOCOL)


class _Unpacker:
    def __init__(self, ring: _Ring, table, blob: bytes):
        self.ring, self.table, self.buf = ring, table, bytearray(blob)      # writable: label tensors are augmented in place

    def _tensor(self, idx: int) -> torch.Tensor:
        off, dt, shape = self.table[idx]
        n = int(np.prod(shape))
        if n == 0:
            return torch.from_numpy(np.zeros(shape, dtype=np.dtype(dt)))
        return torch.from_numpy(np.frombuffer(self.buf, dtype=np.dtype(dt), count=n, offset=off).reshape(shape))

    def unpack(self, p):
        tag = p[0]
        if tag in ('ring', 'stack'):
            base = self.ring.view(p[1], p[2]) if tag == 'ring' else self._tensor(p[1])
            out = _Stacked(base[t] for t in range(base.shape[0]))
            out.base = base
            return out
        if tag == 'a':
            return self._tensor(p[1])
        if tag == 'n':
            return self._tensor(p[1]).numpy()
        if tag == 'ol':
            return ObjectLabels(self._tensor(p[1]), tuple(p[2]))
        if tag == 'sb':
            out = SparselyBatchedObjectLabels([])
            out.sparse_object_labels_batch = [None if l is None else self.unpack(l) for l in p[1]]
            return out
        if tag == 'dict':
            return {k: self.unpack(v) for k, v in p[1]}
        if tag == 'list':
            return [self.unpack(v) for v in p[1]]
        if tag == 'tuple':
            return tuple(self.unpack(v) for v in p[1])
        return p[1]


# ---- worker ---------------------------------------------------------------------------------------------------------------------------
def draw_base_seed() -> int:
    """One draw from the parent's torch generator (what torch's DataLoader does per epoch): advances the parent state."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def seed_worker(base_seed: int, rank: int = 0) -> int:
    """Seed torch / numpy / random of the calling process with base_seed + rank (DataLoader: base_seed + worker_id)."""
    import random as _random
    seed = (int(base_seed) + int(rank)) % (1 << 63)
    torch.manual_seed(seed)
    np.random.seed(seed % (1 << 32))
    _random.seed(seed)
    return seed


def _call_loader_fn(loader_fn, epoch: int):
    """``loader_fn(epoch=e)`` when it takes the epoch (the worker is a fresh fork every epoch, so the loaders it builds cannot count
    epochs themselves: without the parent's counter a RandomLoader would shuffle with seed + 0 every epoch), else ``loader_fn()``."""
    import inspect
    try:
        takes = 'epoch' in inspect.signature(loader_fn).parameters
    except (TypeError, ValueError):
        takes = False
    return loader_fn(epoch=epoch) if takes else loader_fn()


def _worker(loader_fn: Callable[[], Iterable], ring: _Ring, results, free_slots, base_seed: int = 0, rank: int = 0, epoch: int = 0):
    try:
        torch.set_num_threads(1)                                  # a forked child must not enter the parent's OpenMP pool
        # a forked child inherits the parent's RNG state unchanged: without a fresh seed every epoch replays the same shuffles and
        # augmentation draws.  torch's DataLoader hands its workers base_seed + worker_id (reference: DataLoader workers under
        # Lightning, modules/data/genx.py:172-232); here the worker of rank r in epoch e gets base_seed(e) + r.
        seed_worker(base_seed, rank)
        free: List[int] = list(range(ring.n_slots))
        lock = threading.Lock()

        def acquire(shape):                                       # called by the loaders' producer threads
            with lock:
                slot = free.pop(0) if free else free_slots.get()
            return ring.view(slot, shape)

        BatchAssembler.buffer_source = staticmethod(acquire)
        for batch in _call_loader_fn(loader_fn, epoch):
            results.put((_BATCH, _Packer(ring).payload(batch)))
        results.put((_END, None))
    except BaseException:                                         # surface loader errors in the consumer
        results.put((_ERR, traceback.format_exc()))


class ProcessLoader:
    """``for batch in ProcessLoader(loader_fn, slot_bytes, device=...)``: the batches of ``loader_fn()`` (any iterable of batch
    dictionaries built on ``BatchAssembler``), produced in a forked worker process.  One worker per iteration (epoch)."""

    def __init__(self, loader_fn: Callable[[], Iterable], slot_bytes: int, n_slots: int = 8, device=None, depth: int = 1,
                 hold: int = 2, timeout: float = 300., rank: Optional[int] = None):
        self.timeout = timeout
        if rank is None:                                          # data-parallel rank: folded into the worker's seed
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.rank = int(rank)
        self.loader_fn, self.slot_bytes, self.n_slots = loader_fn, int(slot_bytes), int(n_slots)
        self.device = torch.device(device) if device is not None else None
        self.depth, self.hold = max(1, depth), max(1, hold)
        self.ring: Optional[_Ring] = None
        self.registered = False
        self.copy_stream = None
        self.epoch = 0                                            # iterations started so far: handed to loader_fn(epoch=...) in the worker

    # -- shared ring, pinned once per loader -------------------------------------------------------------------------------------------
    def _ensure_ring(self):
        if self.ring is not None:
            return
        self.ring = _Ring(self.n_slots, self.slot_bytes)
        if self.device is not None and self.device.type == 'cuda':
            err = torch.cuda.cudart().cudaHostRegister(self.ring.base, self.ring.mem.numel(), 0)
            if int(err) != 0:
                raise RuntimeError(f'hipHostRegister of the {self.ring.mem.numel() >> 20} MiB batch ring failed ({err}); '
                                   'lower n_slots or raise the locked-memory limit')
            self.registered = True
            self.copy_stream = torch.cuda.Stream(device=self.device)

    def __del__(self):
        try:
            if self.registered:
                torch.cuda.cudart().cudaHostUnregister(self.ring.base)
        except Exception:                                         # noqa: BLE001  (interpreter shutdown)
            pass

    # -- iteration ----------------------------------------------------------------------------------------------------------------------
    def __iter__(self):
        self._ensure_ring()
        ctx = mp.get_context('fork')
        results, free_slots = ctx.Queue(maxsize=self.n_slots), ctx.Queue()
        # one draw from the parent's generator per epoch (as torch's DataLoader does): it advances the parent state, so consecutive
        # epochs get different worker seeds, and a seeded parent gives a reproducible sequence of epochs
        base_seed = draw_base_seed()
        epoch, self.epoch = self.epoch, self.epoch + 1
        proc = ctx.Process(target=_worker, args=(self.loader_fn, self.ring, results, free_slots, base_seed, self.rank, epoch), daemon=True,
                           name='leod-loader-proc')
        proc.start()
        pending: 'collections.deque' = collections.deque()       # (event | None, slots) waiting to go back to the worker
        staged: 'collections.deque' = collections.deque()        # (batch, event, slots) copies issued, not yet handed over
        held: 'collections.deque' = collections.deque()          # host mode: slots of the last `hold` yielded batches

        def reclaim(block: bool = False):
            while pending and (pending[0][0] is None or pending[0][0].query() or block):
                ev, slots = pending.popleft()
                if ev is not None and block:
                    ev.synchronize()
                for s in slots:
                    free_slots.put(s)

        def next_item():
            t0 = time.monotonic()
            while True:
                try:
                    return results.get(timeout=0.002)
                except queue.Empty:
                    reclaim()
                    if not proc.is_alive() and results.empty():
                        raise RuntimeError('loader worker process died')
                    if time.monotonic() - t0 > self.timeout:
                        out = sum(len(x[-1]) for x in list(staged) + list(pending)) + sum(len(x) for x in held)
                        raise RuntimeError(f'no batch from the loader worker for {self.timeout:.0f} s ({out} of {self.n_slots} ring slots '
                                           'are held on the consumer side: a batch that spans k slots needs n_slots > (hold + 1) * k)')

        try:
            while True:
                kind, blob = next_item()
                if kind == _END:
                    break
                if kind == _ERR:
                    raise RuntimeError('loader worker process failed:\n' + blob)
                tree, table, arrays, slots = pickle.loads(blob)
                batch = _Unpacker(self.ring, table, arrays).unpack(tree)
                if self.copy_stream is None:                      # host mode: views of the slots, valid for `hold` more batches
                    held.append(slots)
                    if len(held) > self.hold:
                        pending.append((None, held.popleft()))
                    reclaim()
                    yield batch
                    continue
                staged.append(self._stage(batch, slots))
                if len(staged) > self.depth:
                    yield self._hand_over(staged.popleft(), pending)
                reclaim()
            while staged:
                yield self._hand_over(staged.popleft(), pending)
        finally:
            if self.copy_stream is not None:
                self.copy_stream.synchronize()
            if proc.is_alive():
                proc.terminate()
            proc.join(timeout=5)
            for q in (results, free_slots):
                q.cancel_join_thread(); q.close()

    def _stage(self, batch, slots):
        with torch.cuda.stream(self.copy_stream):
            moved = self._to_device(batch)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return moved, ev, slots

    def _to_device(self, o):
        if isinstance(o, _Stacked):
            dev = o.base.to(self.device, non_blocking=True)       # frames: ONE DMA of the pinned slot; index lists: one small copy
            return [dev[t] for t in range(dev.shape[0])]
        if torch.is_tensor(o):
            return o.to(self.device, non_blocking=True)
        if isinstance(o, dict):
            return {k: self._to_device(v) for k, v in o.items()}
        if type(o) is list:
            return [self._to_device(v) for v in o]
        return o                                                  # labels, strings, augmentation states stay on the host

    def _hand_over(self, item, pending):
        from leod_amd.modules.data.prefetch import _record_stream
        moved, ev, slots = item
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        _record_stream(moved, main)
        pending.append((ev, slots))                               # the slot returns to the worker once its DMA has completed
        return moved
